"""Worker of tests/test_gpu_graph.py (a process of its own: the HIP runtime reads DEBUG_CLR_GRAPH_PACKET_CAPTURE once, when it loads).
For every implementation the dispatcher picks: capture shift_and_add + ibp of a small batch into a HIP graph (torch.cuda.CUDAGraph), replay it
on frames the capture has not seen -- three sets, the third not integer-valued -- and compare with the plain calls, bit for bit.
Prints one JSON line: {case: {"ok": bool, "path": str, "detail": str}}."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
import sr_mi355x as S  # noqa: E402
from sr_mi355x import synth  # noqa: E402

CASES = {  # name: (precision, f, shifts, (h, w), psf, n_iter, path expected, composed?)
    "patch_f32": ("f32", 4, "phase4", (64, 64), "gauss", 3, "patch", False),
    "patch_7x7": ("f32", 4, "phase4", (64, 64), "asym", 2, "patch", False),
    "strips_f64": ("f64", 4, "phase4", (64, 64), "gauss", 2, "stile", False),
    "frame_f32": ("f32", 2, "nominal5", (80, 150), "gauss", 3, "ztile", False),
    "frame_7x7": ("f32", 2, "nominal5", (80, 150), "asym", 2, "ztile", False),
    "frame_f64": ("f64", 2, "nominal5", (80, 150), "gauss", 2, "ctile", False),
    "windows_x4": ("f32", 4, "phase4", (72, 80), "gauss", 2, "dtile", False),
    "small_phase_grid": ("f32", 4, "phase4", (40, 50), "gauss", 2, "atile", False),
    "shifted_frames": ("f32", 2, "measured4", (70, 90), "gauss", 3, "btile", False),
    "shifted_7x7": ("f32", 2, "measured4", (70, 90), "asym", 2, "btile", False),
    "mosaic_tiles_f64": ("f64", 2, "phase2", (40, 60), "asym", 2, "mosaic", False),
    "per_frame_tiles_f64": ("f64", 2, "measured4", (40, 60), "gauss", 2, "fused", False),
    "composed": ("f32", 2, "measured4", (30, 44), "gauss", 2, "composed", True),
}
SHIFTS = {"phase4": synth.phase_shifts(4), "phase2": synth.phase_shifts(2), "nominal5": synth.NOMINAL_5, "measured4": synth.MEASURED_4}


def run_case(name):
    prec, f, shname, (h, w), psfname, n_iter, want, composed = CASES[name]
    shifts = SHIFTS[shname]
    psf = synth.gaussian_psf() if psfname == "gauss" else synth.asymmetric_psf()
    fl = S.FLAG_COMPOSED if composed else S.FLAG_AUTO
    rng = np.random.default_rng(31)
    dt = torch.float32 if prec == "f32" else torch.float64
    B = 3
    frames = [torch.from_numpy(np.rint(rng.uniform(0, 255, (B, len(shifts), h, w)))).to(dt).cuda() for _ in range(3)]
    frames[2] = frames[2] * 0.75 + 0.3   # not integer-valued: the float forms of the operand planes, chosen on the device
    lr = frames[0].clone()

    def step():
        saa = S.shift_and_add_batched(lr, shifts, f, precision=prec, flags=fl)
        hr, err = S.ibp_batched(lr, shifts, psf, saa, f, n_iter, 0.5, precision=prec, flags=fl)
        return saa, hr, err

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    path = S.last_path()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = step()
    detail = []
    for k in (1, 2, 0, 2):   # frames the capture has not seen, the captured ones, and a repeat
        lr.copy_(frames[k])
        g.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in out]
        ref = step()
        torch.cuda.synchronize()
        for nm, a, b in zip(("shift_and_add", "hr", "errors"), got, ref):
            if not torch.equal(a, b):
                detail.append(f"set {k}: {nm} differs by {float((a.double() - b.double()).abs().max()):.3e}")
        if not (bool(torch.isfinite(got[1]).all()) and bool(torch.isfinite(got[2]).all())):
            detail.append(f"set {k}: not finite")
    if path != want:
        detail.append(f"path {path}, expected {want}")
    return {"ok": not detail, "path": path, "detail": "; ".join(detail)}


if __name__ == "__main__":
    names = sys.argv[1:] or sorted(CASES)
    print(json.dumps({n: run_case(n) for n in names}), flush=True)
