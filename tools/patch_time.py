#!/usr/bin/env python3
"""Time the C2 iteration loop (B = 1024, 80 iterations) through whatever library SRX_LIB names.  Development tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch
from sr_mi355x import api as S, synth, _lib
f, shifts = 4, synth.phase_shifts(4)
Bt = int(os.environ.get("PT_B", "1024"))
g = torch.Generator(device="cuda").manual_seed(1)
lr = torch.rand((Bt, 16, 64, 64), device="cuda", generator=g) * 255
if not os.environ.get("PT_FLOAT"):  # PT_FLOAT=1: non-integer samples (the float form of the mosaic)
    lr = torch.round(lr)
hr0 = torch.rand((Bt, 256, 256), device="cuda", generator=g) * 255
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S.ibp_batched(lr, shifts, synth.gaussian_psf(), hr0, f, 80, 0.5, precision="f32", want_errors=True)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"{os.environ.get('SRX_LIB','libsrx.so'):60s} path={_lib.load().srx_last_path().decode()} {best*1e3:7.1f} ms  {best/80*1e6:6.0f} us/iteration", flush=True)
