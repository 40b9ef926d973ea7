"""`python -m sr_mi355x.run_sr` -- the reference's `python run_sr.py` on the MI355X.

Same four flags as the reference's drivers (mono_cal_target/run_sr.py:320-331: --psf {gaussian,measured},
--psf-dir, --data-dir, --output-dir) plus --kind, because the reference has one script per experiment layout
while this is one entry point, and --precision.
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
    args = ap.parse_args(argv)
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
    for i, s in enumerate(sessions, 1):
        print(f"\n[{i}/{len(sessions)}] {os.path.basename(s)}")
        written = session.process_session(s, psf, args.output_dir, kind=args.kind)
        if args.metrics and args.kind == "mono_cal_target":
            for out_dir in written:
                session.write_metrics(out_dir)
    print(f"\nAll sessions done in {(time.time() - t0) / 60:.1f} min")


if __name__ == "__main__":
    main()
