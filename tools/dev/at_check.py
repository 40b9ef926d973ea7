"""dev: k_ibp_afwd / k_atile_near / k_ibp_abwd against the oracle; prints where they differ"""
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
cases = [(4, PH4, (40, 50)), (4, PH4, (80, 100)), (2, synth.phase_shifts(2), (90, 131)), (4, [s for s in PH4 if s[0] > 0 and s[1] > 0], (64, 70)),
         (4, [PH4[0], PH4[5], PH4[6], PH4[6], PH4[15]], (64, 112)), (2, [(0.5, 0.25), (-0.5, -0.25), (0.0, 0.75)], (90, 120)),
         (2, [(0.25, 0.25), (1.25, 0.25), (0.25, -0.75), (-0.75, 1.25)], (90, 120)), (3, synth.phase_shifts(3), (50, 66)), (2, synth.NOMINAL_5, (70, 90))]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for ci, (f, shifts, (h, w)) in enumerate(cases):
    if only >= 0 and ci != only:
        continue
    truth = synth.truth_image(h * f, w * f, seed=77)
    lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=78)
    saa_o = O.shift_and_add(list(lr), shifts, f)
    for n in (1, 2, 6):
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, n, 0.5)
        hr, errs = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, n, 0.5, flags=S.FLAG_AUTO)
        path = S.last_path()
        hr = hr[0].cpu().numpy(); errs = errs[0].cpu().numpy()
        d = np.abs(hr - hr_o)
        iy, ix = np.unravel_index(np.argmax(d), d.shape)
        bad = d > 1e-3
        print(f"case {ci} f={f} {h}x{w} N={len(shifts)} it={n} path={path} max|d|={d.max():.3e} at ({iy},{ix}) bad={bad.sum()} trace rel={np.abs(errs/err_o-1).max():.2e}", flush=True)
        if bad.any():
            rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
            print("   bad rows", rows[:12], "...", rows[-6:], "n", len(rows), " bad cols", cols[:12], "...", cols[-6:], "n", len(cols))
            break
