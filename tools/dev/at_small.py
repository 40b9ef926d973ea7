"""dev: the two-launch kernels on the smallest frames they take, odd sizes, N = 1, many iterations"""
import os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
from oracle import sr_oracle as O
O.set_threads(8)
S.set_precision("f32")
psf = synth.gaussian_psf()
PH4 = synth.phase_shifts(4)
cases = [(2, synth.phase_shifts(2), (16, 16)), (4, PH4, (8, 8)), (4, PH4[:1], (8, 9)), (2, [(0.25, 0.25)], (17, 31)), (3, [(0.1, 0.1), (0.1 + 1 / 3, 0.1 - 1 / 3)], (11, 13)),
         (4, PH4, (33, 47)), (2, [(1.75, -1.75), (-1.25, 0.25)], (40, 40)), (4, [(0.95, 0.95), (-0.8, -0.8)], (30, 30))]
for ci, (f, shifts, (h, w)) in enumerate(cases):
    truth = synth.truth_image(h * f, w * f, seed=7)
    lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=8)
    saa_o = O.shift_and_add(list(lr), shifts, f)
    for n in (1, 7, 40):
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, n, 0.5)
        hr, errs = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, n, 0.5)
        d = np.abs(hr[0].cpu().numpy() - hr_o)
        print(f"case {ci} f={f} {h}x{w} N={len(shifts)} it={n} path={S.last_path()} max|d|={d.max():.3e} bad={(d > 1e-3).sum()} trace rel={np.abs(errs[0].cpu().numpy()/err_o-1).max():.2e}", flush=True)
