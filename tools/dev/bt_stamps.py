"""dev: phase stamps of k_ibp_bfwd / k_ibp_bbwd from a -DSRX_STAMPS build: SRX_LIB=.../libsrx_stamps.so python tools/dev/bt_stamps.py [B]"""
import ctypes, os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import torch
import sr_mi355x as S
from sr_mi355x import _lib, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
f, shifts, psf = 2, synth.MEASURED_4, (synth.asymmetric_psf() if len(sys.argv) > 2 and sys.argv[2] == "asym" else synth.gaussian_psf())
lr = torch.round(torch.rand((B, 4, 768, 1024), device="cuda") * 255)
saa = S.shift_and_add_batched(lr, shifts, f)
S.ibp_batched(lr, shifts, psf, saa, f, 3, 0.5)
assert S.last_path() == "btile"
buf = np.zeros((24, 4096), dtype=np.uint64)
lib = _lib.load()
lib.srx_debug_pstamps.argtypes = [ctypes.c_void_p]
assert lib.srx_debug_pstamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
nb = 374
t = buf.astype(np.int64).reshape(24, 1024, 4)[:, :nb]
names = {0: "start", 1: "hr loads issued", 2: "V-blur", 3: "V-replicate + prefilter", 4: "barrier + transpose", 5: "H-blur", 6: "H-repl + prefilter + barrier",
         7: "(last pair) hfir", 8: "transpose", 9: "setup + lr loads + exchange barrier", 10: "vfir + stores",
         12: "start", 13: "(last pair) E loads + vfir'", 14: "transpose", 15: "exchange barrier", 16: "hfir'", 17: "barrier + H-prefilter", 18: "zero + H-blur' + barrier",
         19: "transpose", 20: "hv loads + V-prefilter", 21: "zero + V-blur'", 22: "update stores"}
for seq, nm in (([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10], "k_ibp_bfwd"), ([12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22], "k_ibp_bbwd")):
    tot = t[seq[-1]] - t[seq[0]]
    print(f"{nm}: cycles per wave first -> last stamp: median {np.median(tot):.0f} p10 {np.percentile(tot,10):.0f} p90 {np.percentile(tot,90):.0f}; (100 MHz counter x ~24 = core cycles?)")
    for a, b in zip(seq, seq[1:]):
        d = t[b] - t[a]
        print(f"   {names[b]:38s} median {np.median(d):8.0f}  p90 {np.percentile(d,90):8.0f}")
    T, F = t.max(axis=2), t.min(axis=2)
    print("   critical path (last wave to last wave; spread = last - first wave at the phase's end), median over windows:")
    for a, b in zip(seq, seq[1:]):
        print(f"   {names[b]:38s} {np.median(T[b] - T[a]):8.0f}   spread {np.median(T[b] - F[b]):7.0f}")
    print(f"   total {np.median(T[seq[-1]] - F[seq[0]]):.0f}; first-wave start to last-wave start {np.median(T[seq[0]] - F[seq[0]]):.0f}")
    for st in (seq[0], seq[len(seq) // 2], seq[-1]):
        rel = t[st] - t[st].min(axis=1, keepdims=True)
        print(f"   stamp {st}: median arrival of waves 0..3 after the window's first wave: {np.median(rel, axis=0).round(0)}")
    start = t[seq[0]]
    print("   window start spread (first..last wave start over all windows):", int(start.max() - start.min()), " end spread:", int(t[seq[-1]].max() - t[seq[-1]].min()), " whole kernel:", int(t[seq[-1]].max() - start.min()))
