#!/usr/bin/env python3
"""patch_check.py -- development check of the patch-resident IBP path (csrc/srx_patch.hpp) on the GPU box:
k_ibp_patch against the CPU oracle and against the tile path (FLAG_TILES) on C2-shaped patches, plus a timing.

    python tools/patch_check.py [B]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "enph459-super-resolution_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import sr_oracle as O  # noqa: E402
from sr_mi355x import api as S, synth  # noqa: E402
from sr_mi355x import _lib  # noqa: E402


def run(lr, shifts, psf, hr0, f, n_iter, patch):
    hr, err = S.ibp_batched(lr, shifts, psf, hr0, f, n_iter, 0.5, precision="f32", flags=S.FLAG_AUTO if patch else S.FLAG_TILES)
    path = _lib.load().srx_last_path().decode()
    torch.cuda.synchronize()
    return hr.double().cpu().numpy(), err.cpu().numpy(), path


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ok = True
    for f, shifts, name in ((4, synth.phase_shifts(4), "x4 phase grid"), (2, synth.phase_shifts(2), "x2 phase grid")):
        h = 256 // f
        H = 256
        psf = synth.gaussian_psf()
        truths = [synth.truth_image(H, H, seed=synth.SEED_TRUTH + i) for i in range(B)]
        O.set_threads(8)
        lr = np.stack([synth.sensor_frames(np.stack([O.forward_model(t, psf, s, f) for s in shifts]), seed=460 + i)
                       for i, t in enumerate(truths)])
        hr0 = np.stack([O.shift_and_add(list(x), shifts, f) for x in lr])
        for n_iter in (1, 2, 10):
            ref = [O.ibp(list(lr[i]), shifts, psf, hr0[i], f, n_iter, 0.5) for i in range(B)]
            rh = np.stack([r[0] for r in ref])
            re = np.stack([np.array(r[1]) for r in ref])
            for patch in (True, False):
                hr, err, path = run(lr, shifts, psf, hr0, f, n_iter, patch)
                d = np.abs(hr - rh).max()
                de = np.abs(err / re - 1).max()
                print(f"{name} n_iter={n_iter:2d} path={path:7s} max|hr-oracle|={d:.3e}  max rel err trace={de:.2e}", flush=True)
                if patch and (path != "patch" or d > 1e-3 or de > 2e-5):
                    ok = False
    # timing, C2 workload
    f, shifts = 4, synth.phase_shifts(4)
    Bt = 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    lr = torch.round(torch.rand((Bt, 16, 64, 64), device="cuda", generator=g) * 255)
    hr0 = torch.rand((Bt, 256, 256), device="cuda", generator=g) * 255
    for patch in (True, False):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            S.ibp_batched(lr, shifts, synth.gaussian_psf(), hr0, f, 80, 0.5, precision="f32", want_errors=True,
                          flags=S.FLAG_AUTO if patch else S.FLAG_TILES)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"B={Bt} 80 iterations path={_lib.load().srx_last_path().decode()}: {dt * 1e3:.1f} ms  ({dt / 80 * 1e6:.0f} us / iteration)", flush=True)
    print("PATCH_CHECK", "OK" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
