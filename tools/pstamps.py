#!/usr/bin/env python3
"""Phase stamps of k_ibp_patch from a DIAGNOSTIC build (hipcc -DSRX_STAMPS -> libsrx_stamps.so): s_memtime by lane 0 of every
wave at phase boundaries.  Shares only; the stamps themselves cost cycles.
    SRX_LIB=.../libsrx_stamps.so python tools/pstamps.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch  # noqa: E402
import sr_mi355x as S  # noqa: E402
from sr_mi355x import _lib, synth  # noqa: E402

f, shifts, psf, B = 4, synth.phase_shifts(4), synth.gaussian_psf(), int(os.environ.get("STAMPS_B", "1024"))
lr = torch.rand((B, 16, 64, 64), device="cuda") * 255
if not os.environ.get("PT_FLOAT"):  # PT_FLOAT=1: non-integer samples (the float form of the mosaic)
    lr = torch.round(lr)
saa = S.shift_and_add_batched(lr, shifts, f)
S.ibp_batched(lr, shifts, psf, saa, f, 3, 0.5)
NPH = 15
buf = np.zeros((24, 4096), dtype=np.uint64)
lib = _lib.load()
lib.srx_debug_pstamps.argtypes = [ctypes.c_void_p]
assert lib.srx_debug_pstamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
nb = min(B, 256)
t = buf[:NPH].astype(np.int64).reshape(NPH, 256, 16)[:, :nb]
names = ["park state + blur_v (1 barrier)", "V-fwd chain (2 barriers)", "rowbuf + barrier", "transpose 1", "prefetch + blur_h (1 barrier)",
         "H-fwd chain (2 barriers)", "strips + barrier", "near band + barrier", "G = M - C Y", "H-bwd (3 barriers)", "barrier",
         "transpose 2", "V-bwd recursions (3 barriers)", "V-blur' + state reload + update"]
tot = t[NPH - 1] - t[0]
print(f"k_ibp_patch, last iteration: {nb} blocks x 16 waves; cycles per wave, first -> last stamp: median {np.median(tot):.0f}  p10 {np.percentile(tot, 10):.0f}  p90 {np.percentile(tot, 90):.0f}")
for i in range(NPH - 1):
    d = t[i + 1] - t[i]
    print(f"  {names[i]:34s} median {np.median(d):8.0f}   mean {d.mean():8.0f}   min {d.min():7d}  max {d.max():7d}   share {100 * d.mean() / tot.mean():5.1f} %")

# Critical path: a phase that ends in a workgroup barrier lasts until its LAST wave arrives.  T_i = the time the last of the 16 waves
# passes stamp i; D_i = T_(i+1) - T_i is what the phase costs the iteration (the per-wave medians above smear barrier waits over
# neighbouring phases).  Also the spread of arrivals at each stamp (last - first wave).
T = t.max(axis=2)   # [NPH, blocks]
F = t.min(axis=2)
print("critical path per phase (last wave to last wave), median over blocks; spread = last - first wave at the phase's end")
tot_cp = np.median(T[NPH - 1] - T[0])
for i in range(NPH - 1):
    if i >= len(names) - 1 and i + 1 != NPH - 1:
        continue
    d = T[i + 1] - T[i]
    sp = T[i + 1] - F[i + 1]
    nm = names[i] if i < len(names) else f"phase {i}"
    print(f"  {nm:34s} {np.median(d):8.0f}   spread {np.median(sp):7.0f}   share {100 * np.median(d) / tot_cp:5.1f} %")
print(f"  total {tot_cp:.0f}")

# per wave (s = wave >> 2: block row, u = wave & 3: block column): when each wave passes a stamp, relative to the block's first wave
for ph in (12, 13, 14):
    rel = t[ph] - t[ph].min(axis=1, keepdims=True)
    print(f"stamp {ph}: median arrival per wave after the block's first wave (rows s = 0..3, columns u = 0..3)")
    print(np.median(rel, axis=0).reshape(4, 4).round(0))

# inside the LAST backward chain of the iteration (V-bwd; the H-bwd chain writes the same stamps earlier and is overwritten)
tt = buf[:24].astype(np.int64).reshape(24, 256, 16)[:, :nb]
Tm = tt.max(axis=2)
nm = {12: "T2 end -> chain entry", 15: "causal recursion + FIR'", 16: "carry store + barrier", 17: "fix-up + anticausal + 6 stores", 18: "barrier", 19: "halo reads + e[]", 20: "-> first quarter hook"}
seq = [12, 15, 16, 17, 18, 19, 20, 13]
print("V-bwd chain, critical path (last wave to last wave):")
for a_, b_ in zip(seq, seq[1:]):
    print(f"  {nm[a_]:34s} {np.median(Tm[b_] - Tm[a_]):8.0f}    per-wave median {np.median(tt[b_] - tt[a_]):8.0f}")
