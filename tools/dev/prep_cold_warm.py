"""The LR-frame passes of a C2 step (prefilter, k_patch_build) with the frames warm in the Infinity Cache (the call repeated
back to back) and cold (1 GB written in between): library profile (HIP events per launch)."""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth, _lib
import bench
lib = _lib.load()
f, lr_hw, shifts, psf, B, n_iter, desc = bench.workload(synth, "c2", None, 1)
lr, _ = bench.make_inputs(S, synth, B, f, lr_hw, shifts, psf, n_unique=32, prec="f32", seed_base=1000)
big = torch.empty(256 * 1024 * 1024, device="cuda")
def step():
    saa = S.shift_and_add_batched(lr, shifts, f, precision="f32")
    return S.ibp_batched(lr, shifts, psf, saa, f, 1, 0.5, precision="f32", out=saa)
def prof(cold):
    for _ in range(2):
        step()
    lib.srx_profile_enable(1)
    for _ in range(5):
        if cold:
            big.fill_(1.0)
        step()
    torch.cuda.synchronize()
    out = {}
    for i in range(lib.srx_profile_kernel_count()):
        ms, n = ctypes.c_double(), ctypes.c_long()
        lib.srx_profile_get(i, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            out[lib.srx_profile_kernel_name(i).decode()] = round(ms.value / n.value * 1e3, 1)
    lib.srx_profile_enable(0)
    return out
print("warm:", prof(False))
print("cold:", prof(True))
