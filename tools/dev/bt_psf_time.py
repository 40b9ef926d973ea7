"""dev: k_ibp_bfwd / k_ibp_bbwd per-iteration kernel times (the library's own HIP events) for the three PSF forms: rank 1, 5 x 5 core, full 7 x 7"""
import ctypes, os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch
import sr_mi355x as S
from sr_mi355x import _lib, synth
lib = _lib.load()
f, shifts = 2, synth.MEASURED_4
a = synth.asymmetric_psf()
psfs = {"gauss": synth.gaussian_psf(), "asym (5x5 core)": a, "full7": 0.6 * synth.gaussian_psf() + 0.4 * a}
for B in (1, 8):
    lr = torch.round(torch.rand((B, 4, 768, 1024), device="cuda") * 255)
    for name, psf in psfs.items():
        saa = S.shift_and_add_batched(lr, shifts, f)
        S.ibp_batched(lr, shifts, psf, saa, f, 10, 0.5)
        torch.cuda.synchronize()
        lib.srx_profile_enable(1)
        S.ibp_batched(lr, shifts, psf, saa, f, 50, 0.5)
        torch.cuda.synchronize()
        out = {}
        tot, cnt = ctypes.c_double(), ctypes.c_long()
        for kid in range(lib.srx_profile_kernel_count()):
            if lib.srx_profile_get(kid, ctypes.byref(tot), ctypes.byref(cnt)) == 0 and cnt.value:
                out[lib.srx_profile_kernel_name(kid).decode()] = round(tot.value / cnt.value * 1e3, 2)
        lib.srx_profile_enable(0)
        print(f"B={B} {name:18s} path={S.last_path()} {out} sum {sum(out.values()):.1f} us", flush=True)
