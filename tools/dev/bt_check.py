"""dev: k_ibp_bfwd / k_ibp_bbwd against the oracle and the tile kernels; prints where they differ"""
import os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
from oracle import sr_oracle as O
O.set_threads(8)
S.set_precision("f32")
f = 2
psf = synth.gaussian_psf()
cases = [((64, 64), synth.MEASURED_4), ((90, 110), synth.MEASURED_4), ((150, 277), synth.MEASURED_4),
         ((70, 83), [(0.3, -1.2), (1.7, 0.45), (-1.9, 1.99), (0.0, 0.25), (-0.6, -0.6)]), ((131, 200), synth.NOMINAL_5), ((20, 33), synth.MEASURED_4)]
for (h, w), shifts in cases:
    truth = synth.truth_image(h * f, w * f, seed=77)
    lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=78)
    saa_o = O.shift_and_add(list(lr), shifts, f)
    for n in (1, 2, 6):
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, n, 0.5)
        hr, errs = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, n, 0.5, flags=S.FLAG_PER_FRAME)
        path = S.last_path()
        hr = hr[0].cpu().numpy(); errs = errs[0].cpu().numpy()
        d = np.abs(hr - hr_o)
        iy, ix = np.unravel_index(np.argmax(d), d.shape)
        bad = d > 1e-3
        print(f"{h}x{w} N={len(shifts)} it={n} path={path} max|d|={d.max():.3e} at ({iy},{ix}) bad={bad.sum()} trace rel={np.abs(errs/err_o-1).max():.2e}", flush=True)
        if bad.any():
            rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
            print("   bad rows", rows[:12], "...", rows[-6:], " bad cols", cols[:12], "...", cols[-6:])
        if n == 1 and path == "btile":
            hr_t, errs_t = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, n, 0.5, flags=S.FLAG_PER_FRAME | S.FLAG_TILES)
            print("   tiles path:", S.last_path(), f"max|d| vs oracle {np.abs(hr_t[0].cpu().numpy()-hr_o).max():.3e}")
