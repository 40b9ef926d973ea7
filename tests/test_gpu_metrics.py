"""Device forms of the quality metrics (sr_mi355x/metrics_device.py over libsrx's srx_pair_moments / srx_local_contrast /
srx_ring_sums / srx_spot_moments / srx_edge_* entry points) against the reference's own functions run on its committed result PNGs
(tests/golden/metrics.npz, tools/make_golden_metrics.py) -- the same vectors and tolerances tests/test_metrics.py holds the host forms to --
and against the host forms on frame-sized inputs."""
import os

import numpy as np
import pytest
import torch

import sr_mi355x as S  # noqa: F401
from sr_mi355x import metrics as M
from sr_mi355x import metrics_device as D
from sr_mi355x import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(HERE, "golden", "metrics.npz"))


@pytest.mark.parametrize("name", ["native_2x", "SAA"])
def test_local_contrast_golden(g, name):
    prof = torch.from_numpy(g[f"{name}_profile"].astype(np.float64)).cuda()
    np.testing.assert_allclose(D.local_contrast(prof, window=16), g[f"{name}_contrast16"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(D.local_contrast(prof), g[f"{name}_contrast20"], rtol=0, atol=1e-15)
    both = torch.stack([prof, torch.flip(prof, dims=(0,))])
    out = D.local_contrast(both, window=16)
    assert np.array_equal(out[0], D.local_contrast(prof, window=16))
    assert np.array_equal(out[1], M.local_contrast(g[f"{name}_profile"][::-1].astype(np.float64), window=16))  # (the window is [i - w/2, i + w/2): not symmetric)
    assert not D.local_contrast(prof[:10], window=20).any()  # shorter than the window: zeros, as the reference


@pytest.mark.parametrize("name", ["native_2x", "SAA"])
@pytest.mark.parametrize("side", ["left", "right"])
def test_slanted_edge_mtf_golden(g, name, side):
    roi = torch.from_numpy(g[f"{name}_roi"]).cuda()  # uint8 on the device, as the quantiser leaves it
    ex, ey, ang = D.slanted_edge_esf(roi, side=side)
    np.testing.assert_allclose(ang, g[f"{name}_{side}_angle"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ex, g[f"{name}_{side}_esf_x"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ey, g[f"{name}_{side}_esf_y"], rtol=0, atol=1e-8)
    fr, mtf, lsf = M.esf_to_mtf(ex, ey)
    np.testing.assert_allclose(mtf, g[f"{name}_{side}_mtf"], rtol=0, atol=1e-8)
    fc = fr / (3.45e-3 / 2)
    v = fc > 0
    assert abs(M.mtf_at_fraction(fc[v], mtf[v], 0.5) - g[f"{name}_{side}_mtf50"]) < 1e-6
    assert abs(M.mtf_at_fraction(fc[v], mtf[v], 0.1) - g[f"{name}_{side}_mtf10"]) < 1e-6


def test_edge_magnitude_matches_the_host_filters(g):
    roi = g["SAA_roi"].astype(np.float64)
    smooth = M._gaussian_filter(roi, 1.5)
    want = np.sqrt(M._sobel(smooth, 1) ** 2 + M._sobel(smooth, 0) ** 2)
    got = D.edge_magnitude(torch.from_numpy(roi).cuda()).cpu().numpy()
    assert np.abs(got - want).max() < 1e-11
    odd = roi[:37, :53]  # ragged shape: every border pixel goes through the 'reflect' index map
    smooth = M._gaussian_filter(odd, 1.5)
    want = np.sqrt(M._sobel(smooth, 1) ** 2 + M._sobel(smooth, 0) ** 2)
    assert np.abs(D.edge_magnitude(torch.from_numpy(np.ascontiguousarray(odd)).cuda()).cpu().numpy() - want).max() < 1e-11


def test_ring_means_and_centre_golden(g):
    spot = torch.from_numpy(g["spot"]).cuda()
    cy, cx = D.subpixel_centre(spot)
    np.testing.assert_allclose([cy, cx], g["spot_centre"], rtol=0, atol=1e-12)
    r, prof = D.radial_average(spot)  # (default centre and radius, as the golden was made)
    assert np.array_equal(r, g["spot_radial_r"])
    np.testing.assert_allclose(prof, g["spot_radial"], rtol=0, atol=1e-12)
    r, prof = D.radial_average(spot, (cy, cx), 18)  # a centre that is not on the pixel grid
    np.testing.assert_allclose(prof, M.radial_average(g["spot"], (cy, cx), 18)[1], rtol=0, atol=1e-12)
    psf_m = np.load(os.path.join(HERE, "golden", "synth_c1.npz"))["psf_m"]
    for name, p, pitch in (("psfm", psf_m, 3.45), ("spot", g["spot"], None)):
        fr, prof, m2d, label, nyq = D.compute_mtf(p, pixel_pitch_um=pitch)
        np.testing.assert_allclose(fr, g[f"{name}_mtf_freq"], rtol=1e-14)
        np.testing.assert_allclose(prof, g[f"{name}_mtf_radial"], rtol=0, atol=1e-13)
        assert abs(M.mtf_at_fraction(fr, prof, 0.5) - g[f"{name}_mtf50"]) < 1e-9
    # default centre / radius, a non-square image, float32 input
    img = np.random.default_rng(3).uniform(0, 1, (70, 96))
    r0, p0 = M.radial_average(img)
    r1, p1 = D.radial_average(torch.from_numpy(img).cuda())
    assert np.array_equal(r0, r1) and np.abs(p0 - p1).max() < 1e-13
    r2, p2 = D.radial_average(torch.from_numpy(img).cuda().float())
    assert np.abs(p2 - M.radial_average(img.astype(np.float32))[1]).max() < 1e-12


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_pair_moments_psnr_and_affine_fit(prec):
    """One fused reduction over two device frames: moments against numpy, PSNR against synth.psnr, the affine-fit PSNR against the host form
    (metrics.psnr_affine: np.polyfit + mean), on a batch, with a border, bit-identical run to run."""
    rng = np.random.default_rng(5)
    dt = torch.float32 if prec == "f32" else torch.float64
    a = np.clip(synth.truth_image(301, 517, seed=8), 0, 255)
    b = np.clip(0.93 * a + 4.0 + rng.normal(0, 2.0, a.shape), 0, 255)
    ta, tb = torch.from_numpy(a).cuda().to(dt), torch.from_numpy(b).cuda().to(dt)
    a, b = ta.double().cpu().numpy(), tb.double().cpu().numpy()  # what the device holds
    m = D.pair_moments(ta, tb, border=10)[0]
    ai, bi = a[10:-10, 10:-10], b[10:-10, 10:-10]
    want = [ai.size, bi.sum(), ai.sum(), (bi * bi).sum(), (ai * bi).sum(), (ai * ai).sum(), ((ai - bi) ** 2).sum()]
    np.testing.assert_allclose(m, want, rtol=1e-12)
    assert abs(D.psnr(ta, tb) - synth.psnr(a, b)) < 1e-9
    assert abs(D.psnr_affine(ta, tb) - M.psnr_affine(a, b)) < 1e-6
    assert D.psnr(ta, ta) == float("inf")
    two = D.pair_moments(torch.stack([ta, tb]), torch.stack([tb, ta]))
    np.testing.assert_allclose(two[0, 6], two[1, 6], rtol=1e-15)
    assert np.array_equal(two, D.pair_moments(torch.stack([ta, tb]), torch.stack([tb, ta])))


def test_full_frame_psnr_and_report_from_device_images():
    """The cal-target frame size (3072 x 4096): PSNR of two device images equals the host figure; cal_target_report on device tensors equals the
    host report on the same values (a synthetic chart with a slanted bar where the notebook's ROI 2 sits)."""
    f = 2
    H, W = 1536 * f, 2048 * f
    yy, xx = np.mgrid[0:H, 0:W]
    (r0, r1), (c0, c1) = M.ROI2_LR
    cyr, cxr = 0.5 * (r0 + r1) * f, 0.5 * (c0 + c1) * f
    d = ((xx - cxr) * np.cos(0.35) + (yy - cyr) * np.sin(0.35))
    img = 40.0 + 170.0 / (1.0 + np.exp(-(np.abs(d) - 22.0) / 1.6))        # a dark diagonal line, two soft edges
    bars = 128.0 + 100.0 * np.sign(np.sin(yy / 3.1))                          # horizontal bars for ROI 1
    c = M.ROI1_COL_LR * f
    img[:, c - 40:c + 40] = bars[:, c - 40:c + 40]
    rng = np.random.default_rng(9)
    q = np.clip(img + rng.normal(0, 1.5, img.shape), 0, 255).astype(np.uint8)
    q2 = np.clip(0.97 * img + 3 + rng.normal(0, 1.5, img.shape), 0, 255).astype(np.uint8)
    tq, tq2 = torch.from_numpy(q).cuda().float(), torch.from_numpy(q2).cuda().float()
    assert abs(D.psnr(tq, tq2) - synth.psnr(q, q2)) < 1e-9
    assert abs(D.psnr_affine(tq, tq2) - M.psnr_affine(q, q2)) < 1e-6
    host = M.cal_target_report({"a": q.astype(np.float64)}, factor=f)["a"]
    dev = D.cal_target_report({"a": tq}, factor=f)["a"]
    for k in host:
        assert abs(host[k] - dev[k]) <= 1e-6 * max(1.0, abs(host[k])), (k, host[k], dev[k])
