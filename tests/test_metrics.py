"""Quality metrics against the reference's own functions run on its committed result PNGs
(tests/golden/metrics.npz, tools/make_golden_metrics.py).  Host-side numpy: runs without a GPU."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load(name):
    # the package's __init__ is lazy and these modules do not touch the GPU: load them by path
    pkg = os.path.join(ROOT, "enph459-super-resolution_amd", "sr_mi355x")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
    import sr_mi355x.metrics as m
    return m


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(HERE, "golden", "metrics.npz"))


@pytest.fixture(scope="module")
def M():
    return _load("metrics")


@pytest.mark.parametrize("name", ["native_2x", "SAA"])
def test_local_contrast(M, g, name):
    prof = g[f"{name}_profile"].astype(np.float64)
    np.testing.assert_allclose(M.local_contrast(prof, window=16), g[f"{name}_contrast16"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(M.local_contrast(prof), g[f"{name}_contrast20"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", ["native_2x", "SAA"])
@pytest.mark.parametrize("side", ["left", "right"])
def test_slanted_edge_mtf(M, g, name, side):
    roi = g[f"{name}_roi"].astype(np.float64)
    ex, ey, ang = M.slanted_edge_esf(roi, side=side)
    np.testing.assert_allclose(ang, g[f"{name}_{side}_angle"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ex, g[f"{name}_{side}_esf_x"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ey, g[f"{name}_{side}_esf_y"], rtol=0, atol=1e-8)
    fr, mtf, lsf = M.esf_to_mtf(ex, ey)
    np.testing.assert_allclose(fr, g[f"{name}_{side}_freq"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(mtf, g[f"{name}_{side}_mtf"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(lsf, g[f"{name}_{side}_lsf"], rtol=0, atol=1e-7)
    fc = fr / (3.45e-3 / 2)
    v = fc > 0
    assert abs(M.mtf_at_fraction(fc[v], mtf[v], 0.5) - g[f"{name}_{side}_mtf50"]) < 1e-6   # cycles / mm
    assert abs(M.mtf_at_fraction(fc[v], mtf[v], 0.1) - g[f"{name}_{side}_mtf10"]) < 1e-6


def test_esf_to_mtf_on_golden_esf(M, g):
    fr, mtf, lsf = M.esf_to_mtf(g["SAA_left_esf_x"], g["SAA_left_esf_y"])
    np.testing.assert_allclose(mtf, g["SAA_left_mtf"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(lsf, g["SAA_left_lsf"], rtol=0, atol=1e-11)


def test_mtf_at_fraction_edge_cases(M):
    f = np.linspace(0, 1, 11)
    assert np.isnan(M.mtf_at_fraction(f, np.ones(11), 0.5))        # never drops
    assert np.isnan(M.mtf_at_fraction(f, np.zeros(11), 0.5))       # never above
    assert abs(M.mtf_at_fraction(f, 1 - f, 0.5) - 0.5) < 1e-12


def test_compute_mtf_and_centre(M, g):
    psf_m = np.load(os.path.join(HERE, "golden", "synth_c1.npz"))["psf_m"]
    for name, p, pitch in (("psfm", psf_m, 3.45), ("spot", g["spot"], None)):
        fr, prof, m2d, label, nyq = M.compute_mtf(p, pixel_pitch_um=pitch)
        np.testing.assert_allclose(fr, g[f"{name}_mtf_freq"], rtol=1e-14)
        np.testing.assert_allclose(prof, g[f"{name}_mtf_radial"], rtol=0, atol=1e-13)
        c = m2d.shape[0] // 2
        np.testing.assert_allclose(m2d[c - 16:c + 17, c - 16:c + 17], g[f"{name}_mtf_2d_centre"], rtol=0, atol=1e-13)
        assert nyq == pytest.approx(float(g[f"{name}_nyquist"]))
        assert label == ("cycles/mm" if pitch else "cycles/pixel")
        assert abs(M.mtf_at_fraction(fr, prof, 0.5) - g[f"{name}_mtf50"]) < 1e-9
        np.testing.assert_allclose(M.subpixel_centre(p), g[f"{name}_centre"], rtol=0, atol=1e-12)
    r, prof = M.radial_average(g["spot"])
    np.testing.assert_array_equal(r, g["spot_radial_r"])
    np.testing.assert_allclose(prof, g["spot_radial"], rtol=0, atol=1e-12)


def test_fit_gaussian_psf(M, g):
    pytest.importorskip("scipy.optimize")
    popt, fit = M.fit_gaussian_psf(g["spot"])
    np.testing.assert_allclose(popt, g["spot_fit_params"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fit, g["spot_fit_image"], rtol=0, atol=1e-5)
    # the spot was synthesised with amp 200, centre (20.6, 19.7), sigmas 2.4 / 3.3, theta 0.4, offset 3
    assert abs(popt[1] - 20.6) < 0.05 and abs(popt[2] - 19.7) < 0.05


def test_psnr_affine(M):
    rng = np.random.default_rng(0)
    ref = rng.uniform(0, 255, (64, 64))
    assert M.psnr_affine(ref, 0.5 * ref + 20.0) > 250.0                 # an affine intensity change is fitted away
    noisy = ref + rng.normal(0, 2.55, ref.shape)                         # sigma = 0.01 of full scale -> 40 dB
    assert abs(M.psnr_affine(ref, noisy) - 40.0) < 0.5


def test_cal_target_report_on_committed_pngs(M, g):
    """The notebook's summary on full frames assembled from the golden ROI crops (everything else flat): same MTF50 /
    MTF10 as the reference's functions gave on the real PNGs."""
    imgs = {}
    for name in ("native_2x", "SAA"):
        full = np.full((3072, 4096), 128.0)
        full[1900:2100, 2560:2760] = g[f"{name}_roi"].astype(np.float64)
        full[1240:1560, 2700] = g[f"{name}_profile"].astype(np.float64)
        imgs[name] = full
    rep = M.cal_target_report(imgs, factor=2)
    for name in ("native_2x", "SAA"):
        assert abs(rep[name]["mtf50"] - g[f"{name}_left_mtf50"]) < 1e-6
        assert abs(rep[name]["mtf10"] - g[f"{name}_left_mtf10"]) < 1e-6
        assert abs(rep[name]["edge_angle_deg"] - g[f"{name}_left_angle"]) < 1e-9
        assert abs(rep[name]["mean_contrast"] - g[f"{name}_contrast16"][8:-8].mean()) < 1e-12
    with pytest.raises(ValueError):
        M.cal_target_report({"x": np.zeros((100, 100))})
