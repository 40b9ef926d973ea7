"""Seeded synthetic inputs for tests and bench.py.

The reference ships no synthetic data; its real inputs are uint8 sensor frames
(mono_cal_target/run_sr.py:73-75).  This module makes images with the same statistics a
calibration chart / barcode sheet has -- smooth texture + a zone plate (all spatial
frequencies) + flat rectangles with hard edges -- and the shift tables the reference
uses (mono_barcodes/run_sr.py:71-76, mono_cal_target/run_sr.py:59-66,
rgb_cal_target/results/*/shifts.json).  numpy only: it runs on the GPU box without the
reference and without the oracle.
"""
import numpy as np

SEED_TRUTH, SEED_NOISE, SEED_CROP = 459, 460, 461

# nominal +-0.5 px corners (mono_barcodes/run_sr.py:71-76)
NOMINAL_4 = [(+0.5, -0.5), (+0.5, +0.5), (-0.5, -0.5), (-0.5, +0.5)]
# centre + 4 corners (mono_cal_target/run_sr.py:59-66)
NOMINAL_5 = [(0.0, 0.0)] + NOMINAL_4
# measured shifts of the rgb_cal_target session, red-LR pixels (results/*/shifts.json:3-18)
MEASURED_4 = [(0.47225, -0.43385), (0.4897, 0.4641), (-0.4798, -0.4553), (-0.4856, 0.4369)]


def phase_shifts(f):
    """All f*f sub-pixel phases, s = ((a - (f-1)/2)/f, (b - (f-1)/2)/f): config C2's N=16 at f=4."""
    c = (f - 1) / 2.0
    return [((a - c) / f, (b - c) / f) for a in range(f) for b in range(f)]


def full_support_psf():
    """asymmetric_psf with weight on its outer ring too (the asymmetric one is clipped to its 5 x 5 core, as the reference's measured PSF
    is): what exercises the full 7 x 7 form of the register-resident kernels."""
    k = asymmetric_psf() + 0.004 * (1.0 + 0.5 * np.cos(np.arange(49.0)).reshape(7, 7))
    return k / k.sum()


def _gauss_smooth(u, sigma):
    """Separable Gaussian smoothing with edge replication (plain numpy)."""
    r = int(4 * sigma + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * sigma * sigma))
    g /= g.sum()
    p = np.pad(u, r, mode="edge")
    p = sum(g[i] * p[i:i + u.shape[0], :] for i in range(2 * r + 1))
    return sum(g[i] * p[:, i:i + u.shape[1]] for i in range(2 * r + 1))


def truth_image(H, W, seed=SEED_TRUTH):
    """float64 [H, W] in [0, 255]: 0.5 smooth noise + 0.3 zone plate + 0.2 rectangles."""
    rng = np.random.default_rng(seed)
    u = _gauss_smooth(rng.uniform(0.0, 1.0, (H, W)), 2.0)
    u = (u - u.min()) / max(u.max() - u.min(), 1e-30) * 255.0
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    r2 = (yy - H / 2.0) ** 2 + (xx - W / 2.0) ** 2
    zone = 127.5 * (1.0 + np.cos(np.pi * r2 / (2.0 * max(H, W))))
    rect = np.full((H, W), 127.5)
    for _ in range(20):
        y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
        hh, ww = int(rng.integers(2, max(3, H // 3))), int(rng.integers(2, max(3, W // 3)))
        rect[y0:y0 + hh, x0:x0 + ww] = rng.uniform(0.0, 255.0)
    return np.clip(0.5 * u + 0.3 * zone + 0.2 * rect, 0.0, 255.0)


def sensor_frames(clean_lr, seed=SEED_NOISE, sigma=1.0):
    """Noise sigma DN, round, clip to uint8 range, back to float64 (uint8-valued floats, as load_gray gives)."""
    rng = np.random.default_rng(seed)
    clean_lr = np.asarray(clean_lr, dtype=np.float64)
    return np.clip(np.rint(clean_lr + rng.normal(0.0, sigma, clean_lr.shape)), 0.0, 255.0)


def gaussian_psf(size=7, sigma=1.0):
    """Normalised 2-D Gaussian, as make_gaussian_psf (mono_cal_target/run_sr.py:104-111)."""
    hw = size // 2
    y, x = np.mgrid[-hw:hw + 1, -hw:hw + 1].astype(np.float64)
    k = np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def asymmetric_psf(size=7, seed=7):
    """A non-separable, asymmetric 7x7 PSF standing in for load_measured_psf's output
    (mono_cal_target/run_sr.py:114-152) when the pinhole images are not at hand."""
    rng = np.random.default_rng(seed)
    hw = size // 2
    y, x = np.mgrid[-hw:hw + 1, -hw:hw + 1].astype(np.float64)
    k = np.exp(-((x - 0.3) ** 2 / 1.1 + (y + 0.2) ** 2 / 0.7 + 0.35 * x * y))
    k *= 1.0 + 0.1 * rng.uniform(-1, 1, k.shape)
    k = np.clip(k - 0.02 * k.max(), 0.0, None)
    return k / k.sum()


def psnr(a, b, peak=255.0):
    mse = float(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2))
    return float("inf") if mse == 0.0 else 10.0 * np.log10(peak * peak / mse)
