#!/usr/bin/env python3
"""dev: does the 7 x 7 patch kernel on a count plane read memory nobody wrote?  Output buffer and caching-allocator blocks pre-filled with NaN."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth
from oracle import sr_oracle as O
O.set_threads(16)
P4 = synth.phase_shifts(4)
sh = P4 + [P4[5]]
psf = synth.full_support_psf()
truth = synth.truth_image(256, 256, seed=5)
lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, 4) for s in sh]), seed=9)
saa = O.shift_and_add(list(lr), sh, 4)
hr_o, e_o = O.ibp(list(lr), sh, psf, saa, 4, 1, 0.5)
lr_d, saa_d = torch.from_numpy(lr)[None].float().cuda(), torch.from_numpy(saa)[None].float().cuda()
for fill_ws, fill_out in ((False, False), (True, False), (False, True), (True, True)):
    torch.cuda.empty_cache()
    if fill_ws:  # poison what the allocator will hand out as workspace: allocate big blocks, fill with NaN, free
        junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(4)]
        del junk
    out = torch.full((1, 256, 256), float("nan") if fill_out else 0.0, device="cuda")
    hr, e = S.ibp_batched(lr_d, sh, psf, saa_d, 4, 1, 0.5, out=out)
    d = (hr[0].double().cpu().numpy() - hr_o)
    print(f"poison ws={fill_ws} out={fill_out}: path {S.last_path()} nan in result {int(np.isnan(d).sum())}  max|d| {np.nanmax(np.abs(d)):.3e}  trace {e[0].cpu().numpy()} vs {e_o}")
