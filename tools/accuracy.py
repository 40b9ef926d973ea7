#!/usr/bin/env python3
"""f32 / f64 GPU result of the C2 patch workload against the float64 CPU oracle: max |diff|, PSNR(gpu, oracle) and the PSNR
of both against the synthetic truth (the reference's acceptance metric is a PSNR delta < 0.01 dB).
    python tools/accuracy.py            (SRX_LIB=... selects a diagnostic build)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
sys.path.insert(0, ROOT)
import sr_mi355x as S  # noqa: E402
from sr_mi355x import synth  # noqa: E402
from oracle import sr_oracle as O  # noqa: E402

f, shifts, psf = 4, synth.phase_shifts(4), synth.gaussian_psf()
truth = synth.truth_image(256, 256)
clean = np.stack([O.forward_model(truth, psf, s, f) for s in shifts])
lr = synth.sensor_frames(clean)
O.set_threads(8)
saa_o = O.shift_and_add(lr, shifts, f)
hr_o, err_o = O.ibp(lr, shifts, psf, saa_o.copy(), f, 80, 0.5)
for prec in ("f32", "f64"):
    S.set_precision(prec)
    saa_g = S.shift_and_add(lr, shifts, factor=f)
    hr_g, err_g = S.ibp(lr, shifts, psf, saa_g.copy(), factor=f, n_iter=80, step=0.5)
    print(f"{prec} path={S.last_path()}: saa max|d|={np.abs(saa_g - saa_o).max():.3e}  ibp max|d|={np.abs(hr_g - hr_o).max():.3e}  "
          f"PSNR(gpu,oracle)={synth.psnr(hr_g, hr_o):.2f} dB  PSNR vs truth gpu/oracle={synth.psnr(hr_g, truth):.6f}/{synth.psnr(hr_o, truth):.6f} dB  "
          f"mse-trace max rel={np.max(np.abs(np.asarray(err_g) - np.asarray(err_o)) / np.asarray(err_o)):.2e}")
