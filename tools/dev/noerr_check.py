"""dev: the window kernels without the MSE trace (errors == NULL) give the same state"""
import os, sys
import numpy as np, torch
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
S.set_precision("f32")
psf = synth.gaussian_psf()
for name, f, shifts, (h, w) in (("btile", 2, synth.MEASURED_4, (150, 277)), ("atile", 4, synth.phase_shifts(4), (60, 77))):
    lr = torch.round(torch.rand((3, len(shifts), h, w), device="cuda") * 255)
    saa = S.shift_and_add_batched(lr, shifts, f)
    a, e = S.ibp_batched(lr, shifts, psf, saa, f, 5, 0.5)
    p = S.last_path()
    b = S.ibp_batched(lr, shifts, psf, saa, f, 5, 0.5, want_errors=False)
    b = b[0] if isinstance(b, tuple) else b
    z = S.ibp_batched(lr, shifts, psf, saa, f, 0, 0.5)
    z = z[0] if isinstance(z, tuple) else z
    print(name, p, "no-trace equal:", bool(torch.equal(a, b)), " zero iterations = init:", bool(torch.equal(z, saa)))
