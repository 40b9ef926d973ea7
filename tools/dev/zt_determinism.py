#!/usr/bin/env python3
"""dev: which combination makes the frame kernel's 7 x 7 forms deviate from call to call?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
reps = 20
rng = np.random.default_rng(4)
S.set_precision("f32")
base = np.rint(rng.uniform(0, 255, (1, 5, 600, 800)))
for psfn, psf in (("gauss", synth.gaussian_psf()), ("asym5", synth.asymmetric_psf()), ("full7", synth.full_support_psf())):
    for fl in (False, True):
        for n in (1, 2, 3):
            lr = torch.from_numpy(base * 0.75 + 0.3 if fl else base).float().cuda()
            saa = S.shift_and_add_batched(lr, synth.NOMINAL_5, 2)
            outs = [tuple(x.clone() for x in S.ibp_batched(lr, synth.NOMINAL_5, psf, saa, 2, n, 0.5)) for _ in range(reps)]
            bh = sum(1 for o in outs[1:] if not torch.equal(o[0], outs[0][0]))
            be = sum(1 for o in outs[1:] if not torch.equal(o[1], outs[0][1]))
            md = max(float((o[0] - outs[0][0]).abs().max()) for o in outs[1:])
            print(f"{psfn:6s} float={int(fl)} n={n} path {S.last_path()} deviating: hr {bh} trace {be} of {reps - 1}  max|hr diff| {md:.3e}", flush=True)
