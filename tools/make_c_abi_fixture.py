#!/usr/bin/env python3
"""tests/golden/c_abi_c1.bin: the C1 golden case (tests/golden/synth_c1.npz, made by tools/make_golden.py from the imported
reference: mono_barcodes/run_sr.py nominal shifts, 32x32 LR, f = 2, Gaussian PSF) as one flat little-endian file that a plain
C++ host can read without numpy: int32 {N, h, w, f, n_iter}, then float64 shifts[N,2], psf[7,7], lr[N,h,w], saa[H,W],
ibp_after_n_iter[H,W], errors[n_iter]."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "synth_c1.npz"))
lr = g["lr_nom"].astype(np.float64)
N, h, w = lr.shape
f, n_iter = 2, 10
parts = [np.array([N, h, w, f, n_iter], dtype="<i4").tobytes()]
for a in (g["shifts_nom"], g["psf_g"], lr, g["saa_nom"], g[f"ibp_nom_{n_iter}"], g["ibp_nom_errors"][:n_iter]):
    parts.append(np.ascontiguousarray(a, dtype="<f8").tobytes())
dst = os.path.join(ROOT, "tests", "golden", "c_abi_c1.bin")
open(dst, "wb").write(b"".join(parts))
print(dst, os.path.getsize(dst), "bytes")
