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
    stats = {}
    band, errs, bounds = rowband.ibp_row_bands(d["lr"], d["shifts"], d["psf"], d["hr0"], f, n_iter, 0.5, precision=str(d["prec"]),
                                               iters_per_exchange=m, stats=stats)
    full = rowband.gather_rows(band, bounds, d["hr0"].shape[0])
    if dist.get_rank() == 0:
        print(f"row bands, rank 0 of {dist.get_world_size()}: {stats}")  # the per-round split: compute (HIP events) / exchange (host time in the P2P calls)
        np.savez(sys.argv[2], hr=full, errors=np.asarray(errs), world=dist.get_world_size(), plan_path=stats.get("plan_path", ""), trace=stats.get("trace", ""),
                 rounds=stats.get("rounds", 0), compute_s=stats.get("compute_s", 0.0), exchange_host_s=stats.get("exchange_host_s", 0.0))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
