"""dev: per-iteration time of the implementations of a common-fraction frame k_ibp_dtile does not take"""
import os, sys, time
import numpy as np, torch
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
S.set_precision("f32")
psf = synth.gaussian_psf()
for f, shifts, (h, w) in ((4, synth.phase_shifts(4), (300, 377)), (2, synth.phase_shifts(2), (768, 1030)), (4, synth.phase_shifts(4), (768, 1024))):
    lr = torch.round(torch.rand((1, len(shifts), h, w), device="cuda") * 255)
    saa = S.shift_and_add_batched(lr, shifts, f)
    for name, flags in (("auto", S.FLAG_AUTO), ("two-launch", S.FLAG_DIAG_TWO_LAUNCH), ("tiles", S.FLAG_TILES)):
        for n in (10, 50):
            S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5, flags=flags)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5, flags=flags)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            if n == 10: t10 = dt
        print(f"x{f} {h*f}x{w*f} N={len(shifts)} {name:10s} path={S.last_path():7s} {(dt - t10) / 40 * 1e6:7.1f} us per iteration (setup {(t10 - (dt - t10) / 4) * 1e3:.2f} ms)", flush=True)
