"""dev: phase stamps of k_ibp_afwd from a -DSRX_STAMPS -DSRX_STAMPS_INNER build"""
import ctypes, os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch
import sr_mi355x as S
from sr_mi355x import _lib, synth
f, shifts, psf = 4, synth.phase_shifts(4), synth.gaussian_psf()
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (300, 377)
lr = torch.round(torch.rand((1, 16, h, w), device="cuda") * 255)
saa = S.shift_and_add_batched(lr, shifts, f)
S.ibp_batched(lr, shifts, psf, saa, f, 3, 0.5, flags=S.FLAG_DIAG_TWO_LAUNCH)
assert S.last_path() == "atile"
buf = np.zeros((24, 4096), dtype=np.uint64)
lib = _lib.load()
lib.srx_debug_pstamps.argtypes = [ctypes.c_void_p]
assert lib.srx_debug_pstamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
nb = min(((h * f + 25) // 96 + 1) * ((w * f + 25) // 96 + 1), 1024)
t = buf.astype(np.int64).reshape(24, 1024, 4)[:, :nb]
ok = (t[0] > 0).all(axis=1) & (t[11] > 0).all(axis=1)
t = t[:, ok]
names = {1: "S loads issued", 2: "H-blur (1 barrier)", 3: "H-pad + prefilter (2 barriers)", 4: "H-FIR", 5: "transpose", 6: "operand loads issued", 7: "V-blur (1 barrier)",
         8: "V-pad + prefilter (2 barriers)", 9: "V-FIR", 10: "G step + stores", 11: "Y band stores"}
tot = t[11] - t[0]
print(f"k_ibp_afwd {h*f}x{w*f}: {t.shape[1]} windows; cycles per wave first -> last stamp: median {np.median(tot):.0f} p10 {np.percentile(tot,10):.0f} p90 {np.percentile(tot,90):.0f}")
T = t.max(axis=2)
for b in range(1, 12):
    d = t[b] - t[b - 1]
    print(f"   {names[b]:34s} per wave median {np.median(d):7.0f} p90 {np.percentile(d,90):7.0f}   critical path {np.median(T[b] - T[b-1]):7.0f}")
inner = {17: "chain 1 + carry store", 18: "barrier 1", 19: "fix-up + chain 2 + 3 stores", 20: "barrier 2"}
print("   inside the V-prefilter:")
for b in (17, 18, 19, 20):
    d = t[b] - t[b - 1]
    print(f"      {inner[b]:30s} per wave median {np.median(d):7.0f} p90 {np.percentile(d,90):7.0f}")
