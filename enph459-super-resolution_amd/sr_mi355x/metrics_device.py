"""Device forms of the quality metrics on top of libsrx.so's `srx_pair_moments_*`, `srx_local_contrast_*`,
`srx_ring_sums_*`, `srx_spot_moments_*`, `srx_edge_magnitude_f64`, `srx_edge_dist_range`, `srx_edge_bins_*` (include/srx.h).

Same names, arguments and return values as `sr_mi355x.metrics` (which mirrors the reference's notebook / calibration functions:
mono_cal_target/analysis.ipynb cells 4, 7, 10; data_collection/psf_mtf_utils.py:67-178; the vendor GUI's affine-fit PSNR,
opt_materials/software/XPR_Software.py:735-745, 1215-1256), but the images stay on the GPU: what runs on the device is every pass over
a frame or an ROI (the 12.6 MP PSNR reductions, the ROI's Gaussian + Sobel, the projection and 1/4-px binning, the ring means, the
sliding-window contrast); what stays on the host are the few-thousand-operation tails -- the 85th percentile and the two line fits of
the edge pixels, the ESF interpolation, the 72-sample gradient / Hann / FFT (metrics.esf_to_mtf), the 7-parameter Gaussian fit and the
256^2 FFT of compute_mtf.  There is no CPU fallback for the device parts: without libsrx.so / a GPU these functions raise.
"""
import ctypes

import numpy as np
import torch

from . import _lib, api, metrics

_c = ctypes


def _dev64(x):
    """-> contiguous float64 CUDA tensor (uint8 / float inputs alike: the notebook works on the PNGs' values as float64)"""
    if isinstance(x, torch.Tensor):
        return x.to(device=api._device(), dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64))).to(api._device())


def _dev(x):
    """-> (contiguous float32 / float64 CUDA tensor, 'f32' | 'f64'): tensors keep their precision, everything else becomes float64"""
    if isinstance(x, torch.Tensor) and x.dtype == torch.float32:
        return x.to(device=api._device()).contiguous(), "f32"
    return _dev64(x), "f64"


def _ws(B, H, W, nbin):
    return api._ws(_lib.load().srx_metrics_workspace_bytes(int(B), int(H), int(W), int(nbin)))


def pair_moments(ref, test, border=0):
    """[B, 7] float64 (host): n, sum t, sum r, sum t^2, sum t r, sum r^2, sum (r - t)^2 over ref / test [B, H, W] (or [H, W]) minus a
    `border`-pixel frame, one fused pass over the two device images."""
    r, prec = _dev(ref)
    t = test.to(device=r.device, dtype=r.dtype).contiguous() if isinstance(test, torch.Tensor) else _dev64(test).to(r.dtype)
    if r.dim() == 2:
        r, t = r[None], t[None]
    if r.shape != t.shape:
        raise ValueError("ref and test differ in shape")
    B, H, W = r.shape
    out = torch.empty((B, 7), dtype=torch.float64, device=r.device)
    wt, wp, wn = _ws(B, H, W, 1)
    _lib.check(api._fn("srx_pair_moments", prec)(api._p(r), api._p(t), B, H, W, int(border), api._p(out), wp, wn, api._stream()), "srx_pair_moments")
    return out.cpu().numpy()


def psnr(a, b, peak=255.0):
    """10 log10(peak^2 / MSE) of two device images, per item for a batch; float (or a list for a batch)."""
    m = pair_moments(a, b)
    v = [float("inf") if row[6] == 0.0 else float(10.0 * np.log10(peak * peak * row[0] / row[6])) for row in m]
    return v[0] if len(v) == 1 else v


def psnr_affine(ref, test, border=10, data_range=1.0):
    """metrics.psnr_affine on device images: the least-squares line test -> a test + b from the five second-order moments, then
    MSE = (S_rr - S_tr^2 / S_tt) / n in centred moments, everything scaled to [0, 1] from 8-bit."""
    n, st, sr, stt, strr, srr, _ = pair_moments(ref, test, border=border)[0]
    s_tt, s_tr, s_rr = stt - st * st / n, strr - st * sr / n, srr - sr * sr / n
    mse = max((s_rr - (s_tr * s_tr / s_tt if s_tt > 0 else 0.0)) / n, 0.0) / (255.0 * 255.0)
    return np.inf if mse == 0 else 10.0 * np.log10(data_range ** 2 / mse)


def local_contrast(profile, window=20):
    """metrics.local_contrast of a device profile [n] (or [B, n]); float64 numpy out."""
    p = _dev64(profile)
    one = p.dim() == 1
    p = p[None] if one else p
    out = torch.empty_like(p)
    _lib.check(_lib.load().srx_local_contrast_f64(api._p(p), p.shape[0], p.shape[1], int(window), api._p(out), api._stream()), "srx_local_contrast")
    res = out.cpu().numpy()
    return res[0] if one else res


def radial_average(data_2d, center=None, max_radius=None):
    """metrics.radial_average on a device image: (radii, ring means)."""
    d, prec = _dev(data_2d)
    h, w = d.shape
    cy, cx = (0.5 * h, 0.5 * w) if center is None else center
    if max_radius is None:
        max_radius = int(min(cy, cx, h - cy, w - cx))
    nbin = max(int(max_radius), 0)
    if nbin == 0:
        return np.arange(0), np.zeros(0)
    out = torch.empty(2 * nbin, dtype=torch.float64, device=d.device)
    wt, wp, wn = _ws(1, h, w, nbin)
    _lib.check(api._fn("srx_ring_sums", prec)(api._p(d), h, w, float(cy), float(cx), nbin, api._p(out), wp, wn, api._stream()), "srx_ring_sums")
    o = out.cpu().numpy()
    tot, cnt = o[:nbin], o[nbin:]
    return np.arange(nbin), np.divide(tot, cnt, out=np.zeros(nbin), where=cnt > 0)


def subpixel_centre(psf):
    """metrics.subpixel_centre on a device PSF image: (row, col) first moments where it exceeds a tenth of its peak."""
    p, prec = _dev(psf)
    out = torch.empty(4, dtype=torch.float64, device=p.device)
    _lib.check(api._fn("srx_spot_moments", prec)(api._p(p), p.shape[0], p.shape[1], api._p(out), api._stream()), "srx_spot_moments")
    _, mass, sy, sx = out.cpu().numpy()
    return float(sy / mass), float(sx / mass)


def compute_mtf(psf, pixel_pitch_um=None):
    """metrics.compute_mtf with the ring means of the 2-D MTF taken on the device (the 256^2 FFT stays on the host)."""
    psf = np.asarray(psf.cpu().numpy() if isinstance(psf, torch.Tensor) else psf, dtype=np.float64)
    _, _, mtf_2d, label, nyq = metrics.compute_mtf(psf, pixel_pitch_um)
    n = mtf_2d.shape[0]
    rings, mtf_radial = radial_average(mtf_2d, (0.5 * n, 0.5 * n), n // 2)
    per_px = rings / float(n)
    return (per_px if pixel_pitch_um is None else per_px / (pixel_pitch_um * 1e-3)), mtf_radial, mtf_2d, label, nyq


def edge_magnitude(roi, sigma=1.5):
    """Sobel magnitude of the Gaussian-smoothed ROI (scipy.ndimage 'reflect' boundaries), device float64 tensor."""
    r = _dev64(roi)
    h, w = r.shape
    mag = torch.empty_like(r)
    wt, wp, wn = _ws(1, h, w, 1)
    _lib.check(_lib.load().srx_edge_magnitude_f64(api._p(r), h, w, float(sigma), api._p(mag), wp, wn, api._stream()), "srx_edge_magnitude")
    return mag


def slanted_edge_esf(roi, side="left", verbose=False):
    """metrics.slanted_edge_esf with the ROI on the device: the Gaussian + Sobel magnitude, the projection of every pixel on the edge's
    normal and the 1/4-px binning run in libsrx; the percentile and the two line fits over the edge pixels run on the host from the
    magnitude image (one small device-to-host copy).  Returns (esf_x, esf_y, angle_deg)."""
    lib = _lib.load()
    r = _dev64(roi)
    h, w = r.shape
    mag = edge_magnitude(r).cpu().numpy()
    rs, cs = np.where(mag > np.percentile(mag, 85))
    if len(rs) < 20:
        raise RuntimeError("Too few edge pixels detected")
    rows_are_x = bool((rs.max() - rs.min()) >= (cs.max() - cs.min()))
    u, v = (rs, cs) if rows_are_x else (cs, rs)
    m_c, b_c = np.polyfit(u, v, 1)
    edge_dist = (v - m_c * u - b_c) / np.sqrt(1 + m_c ** 2)
    sel = edge_dist < 0 if side == "left" else edge_dist > 0
    if sel.sum() < 10:
        raise RuntimeError(f"Too few edge pixels on {side} side")
    m, b = np.polyfit(u[sel], v[sel], 1)
    norm = float(np.sqrt(1 + m ** 2))
    angle = np.degrees(np.arctan2(1, m)) if rows_are_x else np.degrees(np.arctan2(m, 1))
    if verbose:
        print(f"  Edge angle: {angle:.1f} deg, {int(sel.sum())}/{len(rs)} edge pixels ({side} side)")
    rng = torch.empty(2, dtype=torch.float64, device=r.device)
    _lib.check(lib.srx_edge_dist_range(h, w, float(m), float(b), norm, int(rows_are_x), api._p(rng), api._stream()), "srx_edge_dist_range")
    lo, hi = (float(x) for x in rng.cpu().numpy())
    bw = 0.25
    bins = np.arange(lo, hi + bw, bw)  # the edges the device forms as lo + i * bw
    esf_x = 0.5 * (bins[:-1] + bins[1:])
    nbin = len(esf_x)
    out = torch.empty(2 * nbin, dtype=torch.float64, device=r.device)
    wt, wp, wn = _ws(1, h, w, nbin)
    _lib.check(lib.srx_edge_bins_f64(api._p(r), h, w, float(m), float(b), norm, int(rows_are_x), lo, bw, nbin, api._p(out), wp, wn, api._stream()),
               "srx_edge_bins")
    o = out.cpu().numpy()
    tot, cnt = o[:nbin], o[nbin:]
    esf_y = np.full(nbin, np.nan)
    np.divide(tot, cnt, out=esf_y, where=cnt > 0)
    valid = ~np.isnan(esf_y)
    if valid.sum() > 2:
        esf_y = np.interp(esf_x, esf_x[valid], esf_y[valid])
    if esf_y[-1] < esf_y[0]:
        esf_x, esf_y = -esf_x[::-1], esf_y[::-1]
    return esf_x, esf_y, angle


def cal_target_report(images, factor=2, side="left"):
    """metrics.cal_target_report on DEVICE images ({title: CUDA tensor [1536 f, 2048 f] holding the values the PNG gets, i.e. already
    clamped and truncated to 0..255}): MTF50 / MTF10 of the slanted edge in ROI 2, the mean local Michelson contrast of the bar
    cross-section in ROI 1 -- the notebook's summary (analysis.ipynb cells 3-10) -- without reading the PNGs back."""
    hr_pitch = metrics.SENSOR_PITCH_MM / factor
    out = {}
    for title, img in images.items():
        if tuple(img.shape) != (1536 * factor, 2048 * factor):
            raise ValueError(f"{title}: the notebook's ROIs are defined on the {1536 * factor} x {2048 * factor} cal-target frame")
        (r0, r1), (c0, c1) = metrics.ROI2_LR
        roi = img[r0 * factor:r1 * factor, c0 * factor:c1 * factor]
        ex, ey, ang = slanted_edge_esf(roi, side=side)
        fr, mtf, _ = metrics.esf_to_mtf(ex, ey)
        fc = fr / hr_pitch
        v = fc > 0
        prof = img[metrics.ROI1_ROWS_LR[0] * factor:metrics.ROI1_ROWS_LR[1] * factor, metrics.ROI1_COL_LR * factor]
        ct = local_contrast(prof, window=16)
        out[title] = {"mtf50": float(metrics.mtf_at_fraction(fc[v], mtf[v], 0.5)), "mtf10": float(metrics.mtf_at_fraction(fc[v], mtf[v], 0.1)),
                      "edge_angle_deg": float(ang), "mean_contrast": float(ct[8:-8].mean())}
    return out
