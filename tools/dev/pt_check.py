#!/usr/bin/env python3
"""dev: the patch kernel's 7 x 7 form on count maps that are not 0/1 products, against the oracle (one iteration)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
from oracle import sr_oracle as O
O.set_threads(16)
P4 = synth.phase_shifts(4)
cases = {"dup": P4 + [P4[5]], "lattice4": [P4[0], P4[5], P4[6], P4[15]], "grid16": P4, "sub12": [s for s in P4 if s[0] > -0.3]}
for psfn, psf in (("full7", synth.full_support_psf()),):
    for cn, sh in cases.items():
        for integer in (True, False):
            truth = synth.truth_image(256, 256, seed=5)
            lr = np.stack([O.forward_model(truth, psf, s, 4) for s in sh])
            lr = synth.sensor_frames(lr, seed=9) if integer else np.clip(lr + 0.37, 0, 255)
            saa = O.shift_and_add(list(lr), sh, 4)
            hr_o, _ = O.ibp(list(lr), sh, psf, saa, 4, 2, 0.5)
            hr, _ = S.ibp_batched(lr[None], sh, psf, saa[None], 4, 2, 0.5)
            d = np.abs(hr[0].double().cpu().numpy() - hr_o)
            print(f"{psfn:6s} {cn:9s} int={int(integer)} path={S.last_path()} max|d|={d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)}  mean {d.mean():.2e}", flush=True)
