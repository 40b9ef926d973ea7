"""Quality metrics that consume the reconstructions: host-side numpy restatements of the
reference's analysis functions, same names, arguments and return values.

  local_contrast, slanted_edge_esf, esf_to_mtf, mtf_at_fraction   mono_cal_target/analysis.ipynb cells 4, 7, 10
  subpixel_centre, radial_average, gauss2d, fit_gaussian_psf, compute_mtf   data_collection/psf_mtf_utils.py:67-175
  psnr                                                            SURVEY.md 8d (10 log10(255^2 / MSE), float outputs)
  psnr_affine                                                     the vendor GUI's PSNR after an affine intensity fit,
                                                                  opt_materials/software/XPR_Software.py:735-745,1215-1256
                                                                  (skimage is absent here: parity unpinned)

These are tiny 1-D / small 2-D computations on the host (the reference runs them in a notebook); they take numpy arrays
(e.g. the float64 arrays `sr_mi355x.ibp` returns) and do not touch the GPU.  Pinned by tests/golden/metrics.npz, produced
by tools/make_golden_metrics.py from the reference's own functions on its committed result PNGs.
"""
import sys

import numpy as np

from .synth import psnr  # noqa: F401  (re-exported)


# ---------------------------------------------------------------------------------------------------------------
# small filters with scipy.ndimage's 'reflect' (half-sample symmetric) boundary = numpy 'symmetric' padding
# ---------------------------------------------------------------------------------------------------------------
def _correlate1d(a, k, axis):
    k = np.asarray(k, dtype=np.float64)
    r = len(k) // 2
    pad = [(0, 0)] * a.ndim
    pad[axis] = (r, r)
    ap = np.pad(a, pad, mode="symmetric")
    out = np.zeros_like(a, dtype=np.float64)
    n = a.shape[axis]
    for j, kj in enumerate(k):
        sl = [slice(None)] * a.ndim
        sl[axis] = slice(j, j + n)
        out += kj * ap[tuple(sl)]
    return out


def _gaussian_filter(a, sigma, truncate=4.0):
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    k /= k.sum()
    out = np.asarray(a, dtype=np.float64)
    for ax in range(out.ndim):
        out = _correlate1d(out, k, ax)
    return out


def _sobel(a, axis):
    out = _correlate1d(a, [-1.0, 0.0, 1.0], axis)
    for ax in range(a.ndim):
        if ax != axis:
            out = _correlate1d(out, [1.0, 2.0, 1.0], ax)
    return out


# ---------------------------------------------------------------------------------------------------------------
# analysis.ipynb
# ---------------------------------------------------------------------------------------------------------------
def local_contrast(profile, window=20):
    """Michelson contrast (max - min) / (max + min) of profile[i - w/2 : i + w/2]; 0 within w/2 of either end
    (analysis.ipynb cell 4)."""
    profile = np.asarray(profile, dtype=np.float64)
    n, hw = len(profile), window // 2
    out = np.zeros(n)
    if hw == 0 or n < 2 * hw + 1:
        return out
    win = np.lib.stride_tricks.sliding_window_view(profile, 2 * hw)[: n - 2 * hw]  # window i starts at i - hw
    mn, mx = win.min(axis=1), win.max(axis=1)
    out[hw:n - hw] = (mx - mn) / (mx + mn + 1e-9)
    return out


def slanted_edge_esf(roi, side="left", verbose=False):
    """4x oversampled edge-spread function of ONE edge of a thick line at any angle (analysis.ipynb cell 7):
    Sobel magnitude of the sigma-1.5 smoothed ROI, top 15 % as edge pixels, a centre line through all of them, the
    pixels on the chosen side re-fitted, every ROI pixel projected on that edge's normal, distances in (-8, 10) binned
    at 1/4 px (empty bins interpolated), flipped if needed so that the ESF rises.  Returns (esf_x, esf_y, angle_deg)."""
    roi = np.asarray(roi)
    smooth = _gaussian_filter(roi.astype(np.float64), 1.5)
    mag = np.sqrt(_sobel(smooth, 1) ** 2 + _sobel(smooth, 0) ** 2)
    rs, cs = np.where(mag > np.percentile(mag, 85))
    if len(rs) < 20:
        raise RuntimeError("Too few edge pixels detected")
    rows_are_x = (rs.max() - rs.min()) >= (cs.max() - cs.min())
    u, v = (rs, cs) if rows_are_x else (cs, rs)  # fit v = m u + b
    m_c, b_c = np.polyfit(u, v, 1)
    edge_dist = (v - m_c * u - b_c) / np.sqrt(1 + m_c ** 2)
    sel = edge_dist < 0 if side == "left" else edge_dist > 0
    if sel.sum() < 10:
        raise RuntimeError(f"Too few edge pixels on {side} side")
    m, b = np.polyfit(u[sel], v[sel], 1)
    norm = np.sqrt(1 + m ** 2)
    rr, cc = np.mgrid[:roi.shape[0], :roi.shape[1]]
    if rows_are_x:
        angle = np.degrees(np.arctan2(1, m))
        dist = (cc - m * rr - b) / norm
    else:
        angle = np.degrees(np.arctan2(m, 1))
        dist = (rr - m * cc - b) / norm
    if verbose:
        print(f"  Edge angle: {angle:.1f} deg, {int(sel.sum())}/{len(rs)} edge pixels ({side} side)")
    fd, fv = dist.ravel(), roi.ravel().astype(np.float64)
    keep = (fd > -8) & (fd < 10)
    fd, fv = fd[keep], fv[keep]
    bw = 0.25
    bins = np.arange(fd.min(), fd.max() + bw, bw)
    esf_x = 0.5 * (bins[:-1] + bins[1:])
    # bin i holds bins[i] <= d < bins[i+1]: searchsorted on the edges themselves reproduces that comparison exactly
    which = np.searchsorted(bins, fd, side="right") - 1
    ok = (which >= 0) & (which < len(esf_x))
    cnt = np.bincount(which[ok], minlength=len(esf_x)).astype(np.float64)
    esf_y = np.full(len(esf_x), np.nan)
    for i in np.nonzero(cnt)[0]:  # per-bin np.mean (pairwise summation), as the reference computes it
        esf_y[i] = fv[ok][which[ok] == i].mean()
    valid = ~np.isnan(esf_y)
    if valid.sum() > 2:
        esf_y = np.interp(esf_x, esf_x[valid], esf_y[valid])
    if esf_y[-1] < esf_y[0]:
        esf_x, esf_y = -esf_x[::-1], esf_y[::-1]
    return esf_x, esf_y, angle


def esf_to_mtf(esf_x, esf_y):
    """ESF -> LSF (np.gradient) -> Hann window -> |FFT|, first half, normalised to its DC value.
    Returns (freq [cycles / px], mtf, lsf) (analysis.ipynb cell 7)."""
    esf_x, esf_y = np.asarray(esf_x, dtype=np.float64), np.asarray(esf_y, dtype=np.float64)
    lsf = np.gradient(esf_y, esf_x)
    n = len(lsf)
    mtf = np.abs(np.fft.fft(lsf * np.hanning(n)))[: n // 2]
    if mtf[0] > 0:
        mtf = mtf / mtf[0]
    freq = np.fft.fftfreq(n, d=np.mean(np.diff(esf_x)))[: n // 2]
    return freq, mtf, lsf


def mtf_at_fraction(freq, mtf, fraction=0.5):
    """Frequency at which the MTF first falls through `fraction` (linear interpolation between the two samples that bracket
    the crossing); nan when the curve starts below it or never gets there.  Same values as analysis.ipynb cell 10 /
    psf_mtf_utils.py:165-178."""
    freq, mtf = np.asarray(freq, dtype=np.float64), np.asarray(mtf, dtype=np.float64)
    below = mtf < fraction
    # a downward crossing is an at-or-above sample directly followed by a below sample
    falls = np.flatnonzero(~below[:-1] & below[1:])
    if falls.size == 0:
        return np.nan
    lo = falls[0]
    rise, run = mtf[lo + 1] - mtf[lo], freq[lo + 1] - freq[lo]
    return freq[lo] if abs(rise) < 1e-12 else freq[lo] + (fraction - mtf[lo]) * run / rise


# ---------------------------------------------------------------------------------------------------------------
# psf_mtf_utils.py
# ---------------------------------------------------------------------------------------------------------------
def subpixel_centre(psf):
    """(row, col) first moments of the PSF where it exceeds a tenth of its peak (psf_mtf_utils.py:67-71).  The 2-D moments are
    taken from the two marginal sums of the thresholded spot."""
    psf = np.asarray(psf, dtype=np.float64)
    spot = psf * (psf > 0.1 * psf.max())
    mass = spot.sum()
    return float(np.arange(psf.shape[0]) @ spot.sum(axis=1)) / mass, float(np.arange(psf.shape[1]) @ spot.sum(axis=0)) / mass


def radial_average(data_2d, center=None, max_radius=None):
    """Ring means: ring k collects the pixels whose distance to `center` (row, col) truncates to k.  Returns (radii, profile);
    an empty ring gives 0 (psf_mtf_utils.py:74-95).  One histogram pass (np.bincount) for the sums, one for the counts."""
    data_2d = np.asarray(data_2d, dtype=np.float64)
    h, w = data_2d.shape
    cy, cx = (0.5 * h, 0.5 * w) if center is None else center
    if max_radius is None:
        max_radius = int(min(cy, cx, h - cy, w - cx))
    ring = np.hypot(np.arange(w)[None, :] - cx, np.arange(h)[:, None] - cy).astype(np.intp).ravel()
    nbin = max(int(max_radius), 0)
    inside = ring < nbin
    tot = np.bincount(ring[inside], weights=data_2d.ravel()[inside], minlength=nbin)[:nbin]
    cnt = np.bincount(ring[inside], minlength=nbin)[:nbin]
    return np.arange(nbin), np.divide(tot, cnt, out=np.zeros(nbin), where=cnt > 0)


def _precision_matrix(sigma_x, sigma_y, theta):
    """Half the inverse covariance of an elliptical Gaussian whose x-axis is turned by theta: R diag(1/2s^2) R^T."""
    c, s = np.cos(theta), np.sin(theta)
    rot = np.array([[c, -s], [s, c]])
    return rot.T @ np.diag([0.5 / sigma_x ** 2, 0.5 / sigma_y ** 2]) @ rot


def gauss2d(xy, amp, x0, y0, sigma_x, sigma_y, theta, offset):
    """offset + amp exp(-d^T Q d), d = (x - x0, y - y0), flattened: the model of psf_mtf_utils.py:98-106 as a quadratic form
    on the precision matrix Q."""
    d = np.stack([np.asarray(xy[0], dtype=np.float64) - x0, np.asarray(xy[1], dtype=np.float64) - y0])
    q = np.einsum("i...,ij,j...->...", d, _precision_matrix(sigma_x, sigma_y, theta), d)
    return (offset + amp * np.exp(-q)).ravel()


def fit_gaussian_psf(psf):
    """Bounded least-squares fit of gauss2d to a PSF image, started at its centre of mass with sigma 2 px
    (psf_mtf_utils.py:109-127; scipy.optimize as there).  Returns (params, model image), or (None, None) if the fit fails."""
    from scipy.optimize import curve_fit

    psf = np.asarray(psf, dtype=np.float64)
    h, w = psf.shape
    grid = np.meshgrid(np.arange(w), np.arange(h))  # (x, y)
    cy, cx = subpixel_centre(psf)
    peak = psf.max()
    #            amp       x0  y0  sigma_x  sigma_y  theta    offset
    bounds = {"lo": (0.0, 0.0, 0.0, 0.3, 0.3, -np.pi, -np.inf), "hi": (2.0 * peak, w, h, 0.5 * w, 0.5 * h, np.pi, 0.5 * peak)}
    try:
        popt, _ = curve_fit(gauss2d, grid, psf.ravel(), p0=(peak, cx, cy, 2.0, 2.0, 0.0, 0.0), bounds=(bounds["lo"], bounds["hi"]),
                            maxfev=20000)
    except RuntimeError as exc:
        print(f"fit_gaussian_psf: no convergence ({exc})", file=sys.stderr)
        return None, None
    return popt, gauss2d(grid, *popt).reshape(h, w)


def compute_mtf(psf, pixel_pitch_um=None):
    """Radial MTF of a PSF (psf_mtf_utils.py:130-162): the unit-sum PSF centred in a max(256, shape)^2 field, |FFT2| with the
    zero frequency in the middle, scaled to peak 1, then ring means.  Returns (freq, mtf_radial, mtf_2d, freq_label, nyquist);
    frequencies in cycles/pixel, or cycles/mm when the pixel pitch is given."""
    psf = np.asarray(psf, dtype=np.float64)
    n = max(256, *psf.shape)
    before = [(n - e) // 2 for e in psf.shape]
    field = np.pad(psf, [(b, n - e - b) for b, e in zip(before, psf.shape)])
    total = field.sum()
    if total > 0:
        field = field / total
    mtf_2d = np.abs(np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(field))))
    top = mtf_2d.max()
    if top > 0:
        mtf_2d = mtf_2d / top
    rings, mtf_radial = radial_average(mtf_2d, (0.5 * n, 0.5 * n), n // 2)
    per_px = rings / float(n)
    if pixel_pitch_um is None:
        return per_px, mtf_radial, mtf_2d, "cycles/pixel", 0.5
    pitch_mm = pixel_pitch_um * 1e-3
    return per_px / pitch_mm, mtf_radial, mtf_2d, "cycles/mm", 0.5 / pitch_mm


# ---------------------------------------------------------------------------------------------------------------
# vendor GUI
# ---------------------------------------------------------------------------------------------------------------
def psnr_affine(ref, test, border=10, data_range=1.0):
    """PSNR of `test` against `ref` after the least-squares affine intensity fit test -> a test + b, both scaled to
    [0, 1] from 8-bit, a `border`-px frame excluded (the vendor GUI's comparison, XPR_Software.py:735-745,1215-1256;
    its skimage call cannot be run here, so this restatement is parity-unpinned)."""
    r = np.asarray(ref, dtype=np.float64)[border:-border or None, border:-border or None] / 255.0
    t = np.asarray(test, dtype=np.float64)[border:-border or None, border:-border or None] / 255.0
    a, b = np.polyfit(t.ravel(), r.ravel(), 1)
    mse = np.mean((r - (a * t + b)) ** 2)
    return np.inf if mse == 0 else 10.0 * np.log10(data_range ** 2 / mse)


# ---------------------------------------------------------------------------------------------------------------
# the notebook's summary for a mono_cal_target session (analysis.ipynb cells 3-10)
# ---------------------------------------------------------------------------------------------------------------
SENSOR_PITCH_MM = 3.45e-3            # analysis.ipynb cell 9
ROI1_COL_LR, ROI1_ROWS_LR = 1350, (620, 780)          # cell 3: vertical cut through the horizontal bar groups
ROI2_LR = ((950, 1050), (1280, 1380))                 # cell 6: the thick diagonal line, lower right


def cal_target_report(images, factor=2, side="left"):
    """MTF50 / MTF10 [cycles/mm] of the slanted edge in ROI 2 and the mean local Michelson contrast (window 16) of the
    bar cross-section in ROI 1, per image, for full-frame mono_cal_target reconstructions (HR = factor x 1536 x 2048).
    `images`: {title: 2-D array}.  Returns {title: {"mtf50": ., "mtf10": ., "edge_angle_deg": ., "mean_contrast": .}}."""
    hr_pitch = SENSOR_PITCH_MM / factor
    out = {}
    for title, img in images.items():
        img = np.asarray(img, dtype=np.float64)
        if img.shape != (1536 * factor, 2048 * factor):
            raise ValueError(f"{title}: the notebook's ROIs are defined on the {1536 * factor} x {2048 * factor} cal-target frame")
        roi = img[ROI2_LR[0][0] * factor:ROI2_LR[0][1] * factor, ROI2_LR[1][0] * factor:ROI2_LR[1][1] * factor]
        ex, ey, ang = slanted_edge_esf(roi, side=side)
        fr, mtf, _ = esf_to_mtf(ex, ey)
        fc = fr / hr_pitch
        v = fc > 0
        prof = img[ROI1_ROWS_LR[0] * factor:ROI1_ROWS_LR[1] * factor, ROI1_COL_LR * factor]
        ct = local_contrast(prof, window=16)
        out[title] = {"mtf50": float(mtf_at_fraction(fc[v], mtf[v], 0.5)), "mtf10": float(mtf_at_fraction(fc[v], mtf[v], 0.1)),
                      "edge_angle_deg": float(ang), "mean_contrast": float(ct[8:-8].mean())}
    return out
