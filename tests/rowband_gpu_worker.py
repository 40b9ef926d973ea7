"""Worker of tests/test_gpu_rowband.py: one rank of `python -m torch.distributed.run ... rowband_gpu_worker.py in.npz out.npz`.
All ranks share cuda:0 (the GPU box has one card), so the process group is gloo and the halo rows travel as host copies; on a
multi-GPU node the same code runs one rank per GPU over RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    from sr_mi355x import rowband
    d = np.load(sys.argv[1])
    f, n_iter, m = int(d["f"]), int(d["n_iter"]), int(d["m"])
    band, errs, bounds = rowband.ibp_row_bands(d["lr"], d["shifts"], d["psf"], d["hr0"], f, n_iter, 0.5, precision=str(d["prec"]),
                                               iters_per_exchange=m)
    full = rowband.gather_rows(band, bounds, d["hr0"].shape[0])
    if dist.get_rank() == 0:
        np.savez(sys.argv[2], hr=full, errors=np.asarray(errs), world=dist.get_world_size())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
