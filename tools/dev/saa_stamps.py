#!/usr/bin/env python3
"""Phase stamps of the accumulate pass of shift_and_add (k_saa_tile<ACC>, diagnostic build -DSRX_STAMPS): C2, thread 0 of every block."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch
import sr_mi355x as S
from sr_mi355x import _lib, synth
B = int(os.environ.get("STAMPS_B", "1024"))
lr = torch.round(torch.rand((B, 16, 64, 64), device="cuda") * 255)
S.shift_and_add_batched(lr, synth.phase_shifts(4), 4)
buf = np.zeros((5, 8, 40000), dtype=np.uint64)
lib = _lib.load(); lib.srx_debug_stamps.argtypes = [ctypes.c_void_p]
assert lib.srx_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf[2].astype(np.int64)
nb = 9 * B
names = ["geometry + frame 0 fetch + stash", "frame 0 row pass (+ barrier)", "frame 0 column pass", "frame 0 stash next (+ barrier)", "frames 1..N-1"]
ok = (t[0, :nb] > 0) & (t[5, :nb] > t[0, :nb])
tot = (t[5, :nb] - t[0, :nb])[ok]
print(f"blocks {ok.sum()}, median cycles/block {np.median(tot):.0f} (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f})")
for i in range(5):
    d = (t[i + 1, :nb] - t[i, :nb])[ok]
    print(f"    {names[i]:34s} median {np.median(d):8.0f}  mean {d.mean():8.0f}  share {100 * d.mean() / tot.mean():5.1f} %")
