"""`python -m sr_mi355x.run_sr` -- the reference's `python run_sr.py` on the MI355X.

Same four flags as the reference's drivers (mono_cal_target/run_sr.py:320-331: --psf {gaussian,measured},
--psf-dir, --data-dir, --output-dir) plus --kind, because the reference has one script per experiment layout
while this is one entry point, and --precision.

Several GPUs: launch it under torchrun, one process per GPU --
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m sr_mi355x.run_sr --kind ... --data-dir ...
Session i goes to rank i mod G (the reference's outer loop order, mono_cal_target/run_sr.py:358-360); the ranks share nothing
but the output directory, so there is no collective on the data path, only a barrier before rank 0 reports the total time.
RANK / LOCAL_RANK / WORLD_SIZE are read, and the device chosen, BEFORE the first GPU call.
--row-bands (cal_target kinds) is the other mode: every rank walks all sessions and each image's IBP loop is split into row bands
whose halo rows travel point to point between neighbouring ranks (sr_mi355x/rowband.py; RCCL send/recv on the GPUs).
"""
import argparse
import os
import time

from . import api, session


def main(argv=None):
    ap = argparse.ArgumentParser(description="Multi-frame super-resolution (Native-2x, SAA, SAA+IBP) on an MI355X")
    ap.add_argument("--kind", required=True, choices=sorted(session.IBP_ITERATIONS),
                    help="experiment layout = which of the reference's run_sr.py drivers to mirror")
    ap.add_argument("--psf", choices=["gaussian", "measured"], default="gaussian")
    ap.add_argument("--psf-dir", default=None, help="pinhole calibration images (sweep*/pos4_(0,0).png)")
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32")
    ap.add_argument("--metrics", action="store_true",
                    help="mono_cal_target only: also write metrics.json (slanted-edge MTF50/MTF10, bar contrast: the "
                         "summary of the reference's analysis.ipynb) next to the PNGs")
    ap.add_argument("--row-bands", action="store_true",
                    help="several GPUs on ONE image: split every image's IBP loop into row bands (cal_target kinds; under torchrun)")
    args = ap.parse_args(argv)
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:  # one process per GPU: pick the device before anything touches it
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0 if os.environ.get("SRX_ONE_GPU") else local_rank)  # SRX_ONE_GPU: rehearsal of the N > 1 path on a one-GPU box
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # item sharding: a barrier and nothing else (gloo).  Row bands: halo rows as CUDA tensors over RCCL, host objects over gloo
        dist.init_process_group("cpu:gloo,cuda:nccl" if args.row_bands and not os.environ.get("SRX_ONE_GPU") else "gloo")
    api.set_precision(args.precision)
    if args.psf == "measured":
        if not args.psf_dir:
            ap.error("--psf measured needs --psf-dir")
        psf = session.load_measured_psf(args.psf_dir)
        print(f"  PSF: measured, kernel shape {psf.shape}")
    else:
        psf = api.make_gaussian_psf(session.PSF_SIZE, session.PSF_SIGMA)
        print(f"  PSF: Gaussian {session.PSF_SIZE}x{session.PSF_SIZE}, sigma={session.PSF_SIGMA}")
    sessions = session.discover_sessions(args.data_dir, args.kind)
    print(f"Found {len(sessions)} session(s):\n" + "\n".join(f"  {os.path.basename(s)}" for s in sessions))
    t0 = time.time()
    # --metrics: from the device tensors the PNGs were quantised from (session.write_metrics_device), not from the files
    metrics_cb = session.write_metrics_device if (args.metrics and args.kind == "mono_cal_target") else None
    session.process_sessions(sessions, psf, args.output_dir, args.kind, rank=rank, world=world, on_images=metrics_cb,
                             row_bands=args.row_bands and world > 1)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(f"\nAll sessions done in {(time.time() - t0) / 60:.1f} min" + (f" on {world} GPUs" if world > 1 else ""))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
