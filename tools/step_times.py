import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # development tool: the C2 step with the launch profiler on and off
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch, numpy as np
import bench
import sr_mi355x as S
from sr_mi355x import synth, _lib
lib = _lib.load()
wl = bench.workload(synth, "c2", None, None)
f, lr_hw, shifts, psf, B, n_iter, desc = wl
lr, _ = bench.make_inputs(S, synth, B, f, lr_hw, shifts, psf, n_unique=32, prec="f32", seed_base=1000)
def step():
    saa = S.shift_and_add_batched(lr, shifts, f, precision="f32")
    return S.ibp_batched(lr, shifts, psf, saa, f, n_iter, 0.5, precision="f32", out=saa)
tot, cnt = ctypes.c_double(), ctypes.c_long()
def ptime():
    for kid in range(lib.srx_profile_kernel_count()):
        if lib.srx_profile_kernel_name(kid).decode() == "k_ibp_patch":
            lib.srx_profile_get(kid, ctypes.byref(tot), ctypes.byref(cnt)); return tot.value, cnt.value
for mode in (0, 1, 0, 1):
    lib.srx_profile_enable(mode)
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        extra = ""
        if mode:
            extra = f" k_ibp_patch total {ptime()}"
            lib.srx_profile_enable(1)  # reset accumulators
        print(f"profile={mode} step {i}: {dt*1e3:.2f} ms{extra}", flush=True)
