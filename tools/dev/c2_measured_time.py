#!/usr/bin/env python3
"""dev: C2 (1024 x4 patches, 16 phases) with a PSF that is not rank 1 -- which path, how long per iteration."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import sr_mi355x as S
from sr_mi355x import synth
B = int(os.environ.get("B", "1024"))
lr = torch.round(torch.rand((B, 16, 64, 64), device="cuda") * 255)
sh = synth.phase_shifts(4)
saa = S.shift_and_add_batched(lr, sh, 4)
for name, psf in (("gauss", synth.gaussian_psf()), ("asym (5x5 core)", synth.asymmetric_psf()), ("full 7x7", synth.full_support_psf())):
    for n in (20,):
        S.ibp_batched(lr, sh, psf, saa, 4, 2, 0.5, want_errors=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S.ibp_batched(lr, sh, psf, saa, 4, n, 0.5, want_errors=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:18s} path {S.last_path():8s} {dt / n * 1e6:8.1f} us per iteration (incl. per-call setup / {n})")
