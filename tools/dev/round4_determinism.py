#!/usr/bin/env python3
"""Run-to-run determinism of this round's kernels on rough data (where a stale value is a large error): the float64 strip kernels, the
patch kernel's 7 x 7 forms (phase grid and count plane, byte and float mosaic), k_ibp_ctile<double> on 40-row regions, k_ibp_ztile's
5 x 5-core form, the two-pass shift_and_add.  python tools/dev/round4_determinism.py [reps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(4)
P4 = synth.phase_shifts(4)
bad_total = 0


def check(name, prec, lr, shifts, psf, f, n_iter, want):
    global bad_total
    S.set_precision(prec)
    dt = torch.float64 if prec == "f64" else torch.float32
    lr_t = torch.from_numpy(lr).to(dt).cuda()
    saa = [S.shift_and_add_batched(lr_t, shifts, f).clone() for _ in range(reps)]
    outs = [tuple(x.clone() for x in S.ibp_batched(lr_t, shifts, psf, saa[0], f, n_iter, 0.5)) for _ in range(reps)]
    path = S.last_path()
    nbad = sum(1 for o in outs[1:] if not (torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])))
    nbad_saa = sum(1 for s in saa[1:] if not torch.equal(s, saa[0]))
    bad_total += nbad + nbad_saa + (path != want)
    print(f"{name:34s} {prec} path {path:6s} (want {want}) deviating calls: ibp {nbad} / saa {nbad_saa} of {reps}", flush=True)


rough = lambda *shape: np.rint(rng.uniform(0, 255, shape))
check("patch x4 grid, asym PSF, bytes", "f32", rough(6, 16, 64, 64), P4, synth.asymmetric_psf(), 4, 3, "patch")
check("patch x4 grid, full 7x7, float", "f32", rough(6, 16, 64, 64) * 0.75 + 0.3, P4, synth.full_support_psf(), 4, 3, "patch")
check("patch x4 dup (count plane), full 7x7", "f32", rough(6, 17, 64, 64), P4 + [P4[5]], synth.full_support_psf(), 4, 3, "patch")
check("patch x4 lattice, asym, float", "f32", rough(6, 4, 64, 64) * 0.75 + 0.3, [P4[i] for i in (0, 5, 6, 15)], synth.asymmetric_psf(), 4, 3, "patch")
check("strips x4 grid, bytes", "f64", rough(130, 16, 64, 64), P4, synth.gaussian_psf(), 4, 3, "stile")
check("strips x4 dup, float", "f64", rough(6, 17, 64, 64) * 0.75 + 0.3, P4 + [P4[5]], synth.gaussian_psf(), 4, 3, "stile")
check("frame, float64 (40-row regions)", "f64", rough(1, 5, 300, 400) * 0.75 + 0.3, synth.NOMINAL_5, synth.gaussian_psf(), 2, 3, "ctile")
check("frame, 5x5 core of a 7x7", "f32", rough(1, 5, 600, 800) * 0.75 + 0.3, synth.NOMINAL_5, synth.asymmetric_psf(), 2, 3, "ztile")
check("frame, full 7x7", "f32", rough(1, 5, 600, 800), synth.NOMINAL_5, synth.full_support_psf(), 2, 3, "ztile")
# the older register-resident kernels with frames that are not integers (state AND trace)
check("x4 frame, 16 phases, float (windows)", "f32", rough(1, 16, 80, 100) * 0.75 + 0.3, P4, synth.gaussian_psf(), 4, 3, "dtile")
check("x4 small frame, float (two-launch)", "f32", rough(2, 16, 40, 50) * 0.75 + 0.3, P4, synth.gaussian_psf(), 4, 3, "atile")
check("x2 measured shifts, float", "f32", rough(2, 4, 150, 277) * 0.75 + 0.3, synth.MEASURED_4, synth.gaussian_psf(), 2, 3, "btile")
check("x2 measured shifts, asym PSF, float", "f32", rough(2, 4, 150, 277) * 0.75 + 0.3, synth.MEASURED_4, synth.asymmetric_psf(), 2, 3, "btile")
check("x4 patch, gaussian, float", "f32", rough(6, 16, 64, 64) * 0.75 + 0.3, P4, synth.gaussian_psf(), 4, 3, "patch")
check("x3 free shifts, float (tile kernels)", "f32", rough(1, 3, 60, 70) * 0.75 + 0.3, [(0.1, -0.2), (-0.3, 0.25), (0.4, 0.05)], synth.gaussian_psf(), 3, 3, "fused")
check("x2 free shifts, float64 (tile kernels)", "f64", rough(1, 3, 60, 70) * 0.75 + 0.3, [(0.1, -0.2), (-0.3, 0.25), (0.4, 0.05)], synth.gaussian_psf(), 2, 3, "fused")
print("TOTAL deviating:", bad_total)
sys.exit(1 if bad_total else 0)
