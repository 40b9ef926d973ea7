#!/usr/bin/env python3
"""Per-phase cycle shares of the mosaic tile kernels from a DIAGNOSTIC build (hipcc -DSRX_STAMPS -> libsrx_stamps.so):
s_memtime stamps by thread 0 of every block at phase boundaries.  Read the shares, not the kernel's run time.
    SRX_LIB=.../libsrx_stamps.so python tools/stamps.py"""
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
saa = S.shift_and_add_batched(lr, shifts, f)
S.ibp_batched(lr, shifts, psf, saa, f, 3, 0.5, flags=S.FLAG_TILES)  # the mosaic tile kernels (the default is k_ibp_patch)
# one rgb_cal_target-shaped frame through the per-frame fused path (k_bwd_tile)
lr1 = torch.rand((1, 4, 768, 1024), device="cuda") * 255
saa1 = S.shift_and_add_batched(lr1, synth.MEASURED_4, 2)
S.ibp_batched(lr1, synth.MEASURED_4, synth.asymmetric_psf(), saa1, 2, 3, 0.5)
buf = np.zeros((5, 8, 40000), dtype=np.uint64)
lib = _lib.load()
lib.srx_debug_stamps.argtypes = [ctypes.c_void_p]
assert lib.srx_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = {0: ("k_fwd_mosaic", ["prefetch M,C", "load region + store", "column pass", "row pass", "pixel phase", "reduce+atomic"]),
         1: ("k_bwd_mosaic", ["prefetch hr", "load region + store", "column pass", "row pass", "zero border", "blur+update"]),
         4: ("k_fwd_tile (one 1536x2048 frame)", ["start", "load region", "prefilter 2-D", "taps + residuals", "reduce"]),
         3: ("k_bwd_tile (one 1536x2048 frame)", ["start", "gather", "prefilter 2-D", "zero strips", "7x7 + update"]),
         2: ("k_saa_tile", ["start", "frame 0 fetch+stash", "frame 0 row pass", "frame 0 column pass", "frame 0 stash next", "frames 1..N-1", "region + walks", "output"])}
for k, (kn, ph) in names.items():
    t = buf[k].astype(np.int64)
    nb = 25 * B if k == 0 else (768 if k >= 3 else 16 * B)
    last = len(ph) - 1
    ok = (t[0, :nb] > 0) & (t[last, :nb] > t[0, :nb])
    tot = (t[last, :nb] - t[0, :nb])[ok]
    if not ok.sum():
        print(f"{kn}: not run")
        continue
    print(f"{kn}: blocks {ok.sum()}, median cycles/block {np.median(tot):.0f} (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
    for i in range(last):
        d = (t[i + 1, :nb] - t[i, :nb])[ok]
        print(f"    {ph[i + 1]:22s} median {np.median(d):8.0f}  mean {d.mean():8.0f}  share of mean {100 * d.mean() / tot.mean():5.1f} %")
    if k == 3:  # gather sub-phases (stamps 5, 6, 7 lie between 0 and 1)
        for name, i0, i1 in (("  first patch fetch + stash", 0, 5), ("  frame 0 rows", 5, 6), ("  frames 1..N-1", 6, 7), ("  sums -> LDS + barrier", 7, 1)):
            d = (t[i1, :nb] - t[i0, :nb])[ok]
            print(f"    {name:28s} median {np.median(d):8.0f}  mean {d.mean():8.0f}")
    if k == 0:  # fwd: 5 x 5 tiles per item, x fastest -- edge classes (assumes the C2 shape and no XCD remap effect on class sizes)
        d = (t[4, :nb] - t[3, :nb])[ok]
        print(f"    pixel phase percentiles 10/50/75/90/99: {np.percentile(d, [10, 50, 75, 90, 99]).round(0)}")
