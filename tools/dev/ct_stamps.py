#!/usr/bin/env python3
"""Phase stamps of k_ibp_ctile<double> (diagnostic build, -DSRX_STAMPS): one 3072x4096 frame, lane 0 of every wave of the first tiles."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch
import sr_mi355x as S
from sr_mi355x import _lib, synth
S.set_precision("f64")
lr = torch.round(torch.rand((1, 5, 1536, 2048), device="cuda", dtype=torch.float64) * 255)
saa = S.shift_and_add_batched(lr, synth.NOMINAL_5, 2)
S.ibp_batched(lr, synth.NOMINAL_5, synth.gaussian_psf(), saa, 2, 3, 0.5)
print("path", S.last_path())
buf = np.zeros((24, 4096), dtype=np.uint64)
lib = _lib.load(); lib.srx_debug_pstamps.argtypes = [ctypes.c_void_p]
assert lib.srx_debug_pstamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
NPH = 9
t = buf[:NPH].astype(np.int64).reshape(NPH, 1024, 4)
names = ["load + blur_rows", "blur_lanes (barrier)", "near band (edge tiles)", "G step (CM loads)", "strips + blur'_lanes", "blur'_rows", "update", "stores issued"]
ok = t[0, :, 0] > 0
t = t[:, ok]
tot = t[NPH - 1] - t[0]
print(f"{ok.sum()} tiles x 4 waves; cycles first -> last stamp: median {np.median(tot):.0f}  p10 {np.percentile(tot, 10):.0f}  p90 {np.percentile(tot, 90):.0f}")
for i in range(NPH - 1):
    d = t[i + 1] - t[i]
    print(f"  {names[i]:26s} median {np.median(d):8.0f}  mean {d.mean():8.0f}  max {d.max():8d}  share {100 * d.mean() / tot.mean():5.1f} %")
st, en = t[0].min(axis=1), t[NPH - 1].max(axis=1)
for x in range(8):
    a, b = st[x::8], en[x::8]
    a0 = a.min()
    starts = np.sort(a - a0)
    print(f"XCD {x}: {len(a)} tiles, span {b.max() - a0:7d} cycles, tile duration median {np.median(b - a):6.0f};  starts: quartiles {np.percentile(starts, 25):.0f} / {np.percentile(starts, 50):.0f} / {np.percentile(starts, 75):.0f} / {starts.max():.0f}")
