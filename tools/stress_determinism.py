"""Run-to-run determinism of one ibp call on a rough frame: python tools/stress_determinism.py [psf] [reps] [h] [w]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
psf_name = sys.argv[1] if len(sys.argv) > 1 else "asym"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h, w = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (600, 800)
frac = (sys.argv[5] == "frac") if len(sys.argv) > 5 else True
f, shifts = 2, synth.NOMINAL_5
psf = synth.asymmetric_psf() if psf_name == "asym" else synth.gaussian_psf()
rng = np.random.default_rng(1)
lr = np.clip(np.rint(rng.uniform(0, 255, (len(shifts), h, w))), 0, 255)
if frac:
    lr = lr * 0.75 + 0.3
S.set_precision("f32")
lr_t = torch.from_numpy(lr)[None].float().cuda()
init = torch.from_numpy(rng.uniform(0, 255, (1, h * f, w * f))).float().cuda()
outs = []
for rep in range(reps):
    hr, errs = S.ibp_batched(lr_t, shifts, psf, init, f, 2, 0.5)
    outs.append(hr[0].clone())
torch.cuda.synchronize()
ref = torch.median(torch.stack(outs[:5]), dim=0).values
nbad = 0
for rep, o in enumerate(outs):
    d = (o - ref).abs()
    if float(d.max()) > 1e-6:
        nbad += 1
        bad = torch.nonzero(d > 1e-6).cpu().numpy()
        rows, cols = sorted(set(bad[:, 0].tolist())), (int(bad[:, 1].min()), int(bad[:, 1].max()))
        print("  rep", rep, "max", float(d.max()), "n", len(bad), "rows", rows[:12], "(mod 52:", sorted(set(r % 52 for r in rows))[:12], ") cols", cols, "(mod 244:", cols[0] % 244, cols[1] % 244, ")")
print(os.environ.get("SRX_LIB", "default"), psf_name, S.last_path(), f"{h}x{w}", "frac" if frac else "u8", "deviating reps:", nbad, "of", reps)
