#!/usr/bin/env python3
"""dev: the x4 16-phase 3072 x 4096 frame with a PSF that is not rank 1 -- window kernel (7 x 7 form) against the tile kernels."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import sr_mi355x as S
from sr_mi355x import synth
lr = torch.round(torch.rand((1, 16, 768, 1024), device="cuda") * 255)
sh = synth.phase_shifts(4)
saa = S.shift_and_add_batched(lr, sh, 4)
for name, psf in (("gauss", synth.gaussian_psf()), ("asym (5x5 core)", synth.asymmetric_psf()), ("full 7x7", synth.full_support_psf())):
    for fl, fn in ((S.FLAG_AUTO, "auto"), (S.FLAG_TILES, "tiles")):
        n = 40
        S.ibp_batched(lr, sh, psf, saa, 4, 2, 0.5, want_errors=False, flags=fl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S.ibp_batched(lr, sh, psf, saa, 4, n, 0.5, want_errors=False, flags=fl)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:18s} {fn:6s} path {S.last_path():8s} {dt / n * 1e6:8.1f} us per iteration (incl. per-call setup / {n})")
