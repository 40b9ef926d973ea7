"""dev: run-to-run determinism of the window kernels on rough frames (the test that found the two gfx950 hazards of DESIGN section 5)"""
import os, sys
import numpy as np, torch
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
S.set_precision("f32")
psf = synth.gaussian_psf()
rng = np.random.default_rng(1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, f, shifts, (h, w), flags in (("btile", 2, synth.MEASURED_4, (600, 800), S.FLAG_AUTO), ("atile", 4, synth.phase_shifts(4), (300, 401), S.FLAG_AUTO),
                                       ("atile-large", 4, synth.phase_shifts(4), (512, 768), S.FLAG_DIAG_TWO_LAUNCH)):
    lr = torch.from_numpy(np.clip(np.rint(rng.uniform(0, 255, (2, len(shifts), h, w))), 0, 255) * 0.75 + 0.3).float().cuda()
    init = torch.from_numpy(rng.uniform(0, 255, (2, h * f, w * f))).float().cuda()
    outs, traces = [], []
    for rep in range(reps):
        hr, errs = S.ibp_batched(lr, shifts, psf, init, f, 3, 0.5, flags=flags)
        outs.append(hr.clone()); traces.append(errs.clone())
    torch.cuda.synchronize()
    nbad = sum(1 for o, t in zip(outs, traces) if not (torch.equal(o, outs[0]) and torch.equal(t, traces[0])))
    print(name, S.last_path(), f"{h}x{w} x{f}: deviating repetitions {nbad} of {reps}", flush=True)
