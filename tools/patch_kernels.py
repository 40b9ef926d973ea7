#!/usr/bin/env python3
"""Per-kernel times of one C2-shaped ibp call (B = 1024, 80 iterations) from the library's launch profiler.  Development tool.
    PT_FLOAT=1: non-integer LR samples (the float form of the mosaic)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch
from sr_mi355x import api as S, synth, _lib
lib = _lib.load()
f, shifts = 4, synth.phase_shifts(4)
g = torch.Generator(device="cuda").manual_seed(1)
lr = torch.rand((1024, 16, 64, 64), device="cuda", generator=g) * 255
if not os.environ.get("PT_FLOAT"):
    lr = torch.round(lr)
hr0 = torch.rand((1024, 256, 256), device="cuda", generator=g) * 255
for rep in range(3):
    lib.srx_profile_enable(1)
    S.ibp_batched(lr, shifts, synth.gaussian_psf(), hr0, f, 80, 0.5, precision="f32", want_errors=True)
    torch.cuda.synchronize()
tot, cnt = ctypes.c_double(), ctypes.c_long()
for kid in range(lib.srx_profile_kernel_count()):
    if lib.srx_profile_get(kid, ctypes.byref(tot), ctypes.byref(cnt)) == 0 and cnt.value:
        print(f"{lib.srx_profile_kernel_name(kid).decode():24s} launches {cnt.value:4d}  total {tot.value:9.3f} ms")
print("path", lib.srx_last_path().decode())
