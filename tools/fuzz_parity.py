#!/usr/bin/env python3
"""Randomised parity sweep: shift_and_add + ibp on random shapes / factors / frame sets / PSFs against the CPU oracle
(float64 tolerance 1e-8, float32 1e-3), every case through the auto-selected path and the composed path.
    python tools/fuzz_parity.py [n_cases] [seed]          (needs an MI355X; the oracle is only the checker)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
sys.path.insert(0, ROOT)
import sr_mi355x as S  # noqa: E402
from sr_mi355x import synth  # noqa: E402
from oracle import sr_oracle as O  # noqa: E402


def _psf_variety(rng, full=True):
    """Gaussian (rank 1) half of the time, else the 5 x 5-core asymmetric PSF (the reference's measured one looks like it) or, when the
    kernel has that form, one with full 7 x 7 support."""
    u = rng.uniform()
    if u < 0.5:
        return synth.gaussian_psf()
    return synth.full_support_psf() if (full and u > 0.8) else synth.asymmetric_psf()


def random_case(rng):
    f = int(rng.choice([2, 2, 3, 4, 4]))
    kind = rng.choice(["phase", "lattice", "free", "far", "patch", "frame0", "shifted", "window"], p=[0.2, 0.2, 0.14, 0.1, 0.1, 0.1, 0.1, 0.06])
    N = int(rng.integers(1, 7))
    if kind == "patch":      # k_ibp_patch's domain: a 256 x 256 HR patch, Gaussian PSF, a subset of a phase grid with delta = 1/2
        f = int(rng.choice([2, 4]))
        grid = synth.phase_shifts(f)
        N = int(rng.integers(2, len(grid) + 1))
        idx = rng.choice(len(grid), size=N, replace=False)
        psf = [synth.gaussian_psf(), synth.gaussian_psf(), synth.asymmetric_psf(), synth.full_support_psf()][int(rng.integers(0, 4))]  # (round 4: its 7 x 7 forms too)
        return f, [(float(grid[i][0]), float(grid[i][1])) for i in idx], 256 // f, 256 // f, psf, int(rng.integers(1, 6)), kind
    if kind == "window":     # k_ibp_dtile's domain: a common fraction > 0 on a frame of at least 256 rows x 192 columns (HR), rows in quads, columns
        f = int(rng.choice([2, 4]))  # in groups of 16; a subset of a phase grid (byte mosaic + 0/1 masks) or, one time in three, non-integer frames
        grid = synth.phase_shifts(f)
        N = int(rng.integers(2, len(grid) + 1))
        idx = rng.choice(len(grid), size=N, replace=False)
        H, W = 4 * int(rng.integers(64, 100)), 16 * int(rng.integers(12, 26))
        return f, [(float(grid[i][0]), float(grid[i][1])) for i in idx], -(-H // f), -(-W // f), _psf_variety(rng), int(rng.integers(1, 4)), kind
    if kind == "shifted":    # k_ibp_bfwd / k_ibp_bbwd's domain: x2, per-frame fractions, |2 s| <= 4, Gaussian PSF, frames of one to several windows
        N = int(rng.integers(2, 9))
        shifts = [(float(rng.uniform(-1.99, 1.99)), float(rng.uniform(-1.99, 1.99))) for _ in range(N)]
        if rng.uniform() < 0.3:  # some frames on integer or half-pixel positions among them
            shifts[0] = (float(rng.integers(-3, 4)) / 2, float(rng.integers(-3, 4)) / 2)
        return 2, shifts, int(rng.integers(16, 180)), int(rng.integers(16, 220)), _psf_variety(rng), int(rng.integers(1, 6)), kind
    if kind == "frame0":     # k_ibp_ztile's domain: delta = 0 (integer HR shifts), at least 128 x 128 HR pixels, Gaussian PSF
        f = int(rng.choice([2, 3, 4]))
        lo = -min(3, f - 1)
        shifts = [(int(rng.integers(lo, 2)) / f, int(rng.integers(lo, 2)) / f) for _ in range(N)]
        h, w = int(rng.integers(-(-128 // f), 400 // f)), int(rng.integers(-(-128 // f), 400 // f))
        return f, shifts, h, w, _psf_variety(rng), int(rng.integers(1, 6)), kind
    if kind == "phase":      # a subset of the full f x f phase grid (one common sub-pixel fraction)
        grid = synth.phase_shifts(f)
        idx = rng.choice(len(grid), size=min(N, len(grid)), replace=False)
        shifts = [grid[i] for i in idx]
    elif kind == "lattice":  # integer HR offsets + one common fraction per axis, with repeats allowed
        dy, dx = rng.choice([0.0, 0.25, 0.5, 0.2]), rng.choice([0.0, 0.5, 0.75, 0.1])
        shifts = [((int(rng.integers(-4, 4)) + dy) / f, (int(rng.integers(-4, 4)) + dx) / f) for _ in range(N)]
    elif kind == "free":
        shifts = [tuple(rng.uniform(-0.9, 0.9, 2)) for _ in range(N)]
    else:                    # beyond the fused paths' reach: composed
        shifts = [tuple(rng.uniform(-4.0, 4.0, 2)) for _ in range(N)]
    h, w = int(rng.integers(8, 110)), int(rng.integers(8, 110))
    psf = [synth.gaussian_psf(), synth.asymmetric_psf(), synth.asymmetric_psf()[1:6, 2:5] / synth.asymmetric_psf()[1:6, 2:5].sum()][int(rng.integers(0, 3))]
    n_iter = int(rng.integers(1, 5))
    return f, [(float(a), float(b)) for a, b in shifts], h, w, psf, n_iter, kind


def run(n_cases=100, seed=2026):
    rng = np.random.default_rng(seed)
    O.set_threads(8)
    worst = {"f64": 0.0, "f32": 0.0}
    paths = {}
    t0 = time.time()
    for ci in range(n_cases):
        f, shifts, h, w, psf, n_iter, kind = random_case(rng)
        lr = np.clip(np.rint(rng.uniform(0, 255, (len(shifts), h, w))), 0, 255)
        if rng.uniform() < 0.25:  # a quarter of the cases with fractional samples: the float forms of the mosaic / operand planes
            lr = lr * 0.75 + 0.3
        saa_o = O.shift_and_add(list(lr), shifts, f)
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, n_iter, 0.5)
        for prec, tol in (("f64", 1e-8), ("f32", 1e-3)):
            S.set_precision(prec)
            for flags in (S.FLAG_AUTO, S.FLAG_COMPOSED):
                saa = S.shift_and_add_batched(lr[None], shifts, f, flags=flags)[0].double().cpu().numpy()
                p_saa = S.last_path()
                import torch
                hr, errs = S.ibp_batched(torch.from_numpy(lr)[None], shifts, psf, torch.from_numpy(saa_o)[None], f, n_iter, 0.5, flags=flags)
                p_ibp = S.last_path()
                d = max(float(np.abs(saa - saa_o).max()), float(np.abs(hr[0].double().cpu().numpy() - hr_o).max()))
                e = float(np.max(np.abs(errs[0].cpu().numpy() - np.asarray(err_o)) / np.maximum(np.asarray(err_o), 1e-30)))
                worst[prec] = max(worst[prec], d)
                paths[(p_saa, p_ibp)] = paths.get((p_saa, p_ibp), 0) + 1
                if not (d <= tol and e <= (1e-9 if prec == "f64" else 1e-4)):
                    print(f"FAIL case {ci} {prec} flags={flags} kind={kind} f={f} N={len(shifts)} h={h} w={w} psf={psf.shape} it={n_iter} "
                          f"paths={p_saa}/{p_ibp} max|d|={d:.3e} trace rel={e:.3e}\n  shifts={shifts}")
                    return 1
        if ci % 10 == 9:
            print(f"{ci + 1} cases ok, {time.time() - t0:.0f} s, worst f64 {worst['f64']:.2e} f32 {worst['f32']:.2e}", flush=True)
    O.set_threads(1)
    print("paths taken (saa, ibp):", paths)
    print(f"all {n_cases} cases within tolerance; worst max|gpu - oracle|: f64 {worst['f64']:.2e}, f32 {worst['f32']:.2e}")
    return 0


if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 2026))
