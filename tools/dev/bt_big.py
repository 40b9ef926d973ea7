"""dev: btile vs the tile kernels on large frames; prints where they differ"""
import os, sys
import numpy as np, torch
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
S.set_precision("f32")
f, shifts, psf = 2, synth.MEASURED_4, synth.gaussian_psf()
for (h, w) in ((300, 500), (768, 1024)):
    x = torch.from_numpy(synth.truth_image(h * f // 4, w * f // 4, seed=26)).cuda().float().repeat(4, 4)[None].contiguous()
    lr = torch.stack([S.forward_model_batched(x, psf, s, f) for s in shifts], dim=1).contiguous()
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    lr = torch.clamp(torch.round(lr + 2.0 * torch.randn(lr.shape, generator=gen, device="cuda")), 0, 255)
    saa = S.shift_and_add_batched(lr, shifts, f)
    for n in (1, 3):
        hr, e = S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5); p = S.last_path()
        hr_t, e_t = S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5, flags=S.FLAG_TILES)
        d = (hr - hr_t).abs()[0].cpu().numpy()
        bad = d > 1e-3
        print(f"{h}x{w} it={n} {p} vs {S.last_path()}: max|d|={d.max():.3e} bad={bad.sum()} finite={bool(torch.isfinite(hr).all())} trace rel {float((e / e_t - 1).abs().max()):.2e}", flush=True)
        if bad.any():
            rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
            print("   bad rows", rows[:10], "..", rows[-5:], "n", len(rows), " bad cols", cols[:10], "..", cols[-5:], "n", len(cols))
