import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/enph459-super-resolution_amd")
import torch, sr_mi355x as S
g = dict(np.load("/root/repo/tests/golden/synth_c2_small.npz"))
S.set_precision("f32")
for name, lr, sh, psf, init in (("n16", g["lr16"], g["shifts16"], g["psf_g"], g["saa16"]), ("n4", g["lr4"], g["shifts4"], g["psf_m"], g["saa4"])):
    hr, e = S.ibp(list(lr), sh, psf, init, 4, 1, 0.5, verbose=False)
    ref = g["ibp16_1"] if name == "n16" else g["ibp4_1"]
    d = np.abs(hr - ref)
    print(name, os.environ.get("SRX_BWD_NOSTAGE"), "max", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "n>1e-3:", (d > 1e-3).sum())
    ys, xs = np.where(d > 1e-3)
    if len(ys): print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
