"""Pins the CPU oracle (oracle/sr_oracle.c) to the REFERENCE: every oracle function is
compared with golden vectors that tools/make_golden.py produced by running the
reference's own functions (mono_cal_target/run_sr.py:157-209) in the build container.
float64 both sides; tolerance 1e-9 DN absolute on 0..255 data (FFT-vs-direct convolution
and summation-order round-off are ~1e-12)."""
import numpy as np
import pytest

from oracle import sr_oracle as O

TOL = 1e-9


def close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = float(np.abs(a - b).max()) if a.size else 0.0
    assert d <= tol, d


def test_blur(g_c1, g_rag):
    close(O.blur(g_c1["truth"], g_c1["psf_g"]), g_c1["blur_g"])
    close(O.blur(g_c1["truth"], g_c1["psf_m"]), g_c1["blur_m"])
    close(O.blur(g_rag["truth"], g_rag["k53"]), g_rag["blur53"])


def test_shift(g_c1, g_rag):
    close(O.ndi_shift(g_c1["truth"], g_c1["shift_frac_arg"]), g_c1["shift_frac"])
    close(O.ndi_shift(g_c1["truth"], g_c1["shift_int_arg"]), g_c1["shift_int"])
    close(O.ndi_shift(g_rag["truth"], g_rag["shift_big_arg"]), g_rag["shift_big"])


def test_integer_shift_is_translation(g_c1):
    """ndi_shift by whole pixels == translation with edge replication (the index map of SURVEY 8a row S)."""
    x = g_c1["truth"]
    out = O.ndi_shift(x, (1.0, -1.0))
    H, W = x.shape
    ii = np.clip(np.arange(H) - 1, 0, H - 1)
    jj = np.clip(np.arange(W) + 1, 0, W - 1)
    close(out, x[np.ix_(ii, jj)], 1e-11)


def test_zoom(g_c1, g_c2s, g_rag):
    close(O.ndi_zoom(g_c1["lr_nom"][0], 2), g_c1["zoom2"])
    close(O.ndi_zoom(np.mean(g_c1["lr_nom"].astype(np.float64), axis=0), 2), g_c1["native2"])
    close(O.ndi_zoom(g_c2s["lr16"][3], 4), g_c2s["zoom4"])
    close(O.ndi_zoom(g_rag["truth"][:21, :17], 3), g_rag["zoom3"])


def test_forward_back(g_c1, g_c2s, g_rag):
    for k, s in enumerate(g_c1["shifts_nom"]):
        close(O.forward_model(g_c1["truth"], g_c1["psf_g"], s, 2), g_c1["fwd_nom"][k])
    for k, s in enumerate(g_c1["shifts_meas"]):
        fwd = O.forward_model(g_c1["truth"], g_c1["psf_m"], s, 2)
        close(fwd, g_c1["fwd_meas"][k])
        e = g_c1["lr_meas"][k].astype(np.float64) - g_c1["fwd_meas"][k]
        close(O.back_project(e, g_c1["psf_m"], s, 2, g_c1["truth"].shape), g_c1["bp_meas"][k])
    for k, s in enumerate(g_c1["shifts_nom"]):
        e = g_c1["lr_nom"][k].astype(np.float64) - g_c1["fwd_nom"][k]
        close(O.back_project(e, g_c1["psf_g"], s, 2, g_c1["truth"].shape), g_c1["bp_nom"][k])
    for k, s in enumerate(g_c2s["shifts16"]):
        close(O.forward_model(g_c2s["truth"], g_c2s["psf_g"], s, 4), g_c2s["fwd16"][k])
    # ragged: sim is ceil(H/f) x ceil(W/f); back_project pads the up-sampled error to hr_shape
    for k, s in enumerate(g_rag["shifts"]):
        close(O.forward_model(g_rag["truth"], g_rag["psf_m"], s, 2), g_rag["fwd"][k])
    e = g_rag["lr"][0].astype(np.float64) - g_rag["fwd"][0][:32, :33]
    close(O.back_project(e, g_rag["psf_m"], g_rag["shifts"][0], 2, g_rag["truth"].shape), g_rag["bp0"])


def test_shift_and_add(g_c1, g_c2s):
    close(O.shift_and_add(list(g_c1["lr_nom"]), g_c1["shifts_nom"], 2), g_c1["saa_nom"])
    lr_avg = g_c1["lr_reps"].astype(np.float64).mean(axis=0)
    close(O.shift_and_add(list(lr_avg), g_c1["shifts_meas"], 2), g_c1["saa_meas"])
    close(O.shift_and_add(list(g_c2s["lr16"]), g_c2s["shifts16"], 4), g_c2s["saa16"])
    close(O.shift_and_add(list(g_c2s["lr4"]), g_c2s["shifts4"], 4), g_c2s["saa4"])


@pytest.mark.parametrize("n", [1, 2, 10, 80])
def test_ibp_c1_nominal(g_c1, n):
    hr, errs = O.ibp(list(g_c1["lr_nom"]), g_c1["shifts_nom"], g_c1["psf_g"], g_c1["saa_nom"], 2, n, 0.5)
    close(hr, g_c1[f"ibp_nom_{n}"])
    close(errs, g_c1["ibp_nom_errors"][:n])


@pytest.mark.parametrize("n", [1, 2, 10, 50])
def test_ibp_c1_measured(g_c1, n):
    lr_avg = g_c1["lr_reps"].astype(np.float64).mean(axis=0)
    hr, errs = O.ibp(list(lr_avg), g_c1["shifts_meas"], g_c1["psf_m"], g_c1["saa_meas"], 2, n, 0.5)
    close(hr, g_c1[f"ibp_meas_{n}"])
    close(errs, g_c1["ibp_meas_errors"][:n])


@pytest.mark.parametrize("n", [1, 10, 80])
def test_ibp_c2_small(g_c2s, n):
    hr, errs = O.ibp(list(g_c2s["lr16"]), g_c2s["shifts16"], g_c2s["psf_g"], g_c2s["saa16"], 4, n, 0.5)
    close(hr, g_c2s[f"ibp16_{n}"])
    close(errs, g_c2s["ibp16_errors"][:n])
    hr, errs = O.ibp(list(g_c2s["lr4"]), g_c2s["shifts4"], g_c2s["psf_m"], g_c2s["saa4"], 4, n, 0.5)
    close(hr, g_c2s[f"ibp4_{n}"])
    close(errs, g_c2s["ibp4_errors"][:n])


def test_ibp_c2_full(g_c2f):
    O.set_threads(8)
    try:
        saa = O.shift_and_add(list(g_c2f["lr16"]), g_c2f["shifts16"], 4)
        close(saa, g_c2f["saa16"])
        hr, errs = O.ibp(list(g_c2f["lr16"]), g_c2f["shifts16"], g_c2f["psf_g"], saa, 4, 80, 0.5)
    finally:
        O.set_threads(1)
    close(hr, g_c2f["ibp16_80"])
    close(errs, g_c2f["ibp16_errors"])


def test_ibp_frame_80_iterations(g_frame):
    """delta = 0 frames of 144 x 280 HR pixels (large enough for the one-launch frame kernel), 80 iterations of the reference's ibp
    (mono_cal_target/run_sr.py:190-209): synthetic N = 5 nominal with the Gaussian and the measured PSF, and a crop of the
    committed mono_cal_target frames."""
    g = g_frame
    lr = list(g["lr5"].astype(np.float64))
    close(O.shift_and_add(lr, g["shifts5"], 2), g["saa5"])
    hr, errs = O.ibp(lr, g["shifts5"], g["psf_g"], g["saa5"], 2, 80, 0.5)
    close(hr, g["ibp5_80"])
    np.testing.assert_allclose(errs, g["ibp5_errors"], rtol=1e-10)
    hr10, _ = O.ibp(lr, g["shifts5"], g["psf_g"], g["saa5"], 2, 10, 0.5)
    close(hr10, g["ibp5_10"], 1e-4)  # stored as float32
    hr, errs = O.ibp(lr, g["shifts5"], g["psf_m"], g["saa5"], 2, 80, 0.5)
    close(hr, g["ibp5m_80"])
    np.testing.assert_allclose(errs, g["ibp5m_errors"], rtol=1e-10)
    lr = list(g["real_lr"].astype(np.float64))
    close(O.shift_and_add(lr, g["real_shifts"], 2), g["real_saa"])
    hr, errs = O.ibp(lr, g["real_shifts"], g["psf_g"], g["real_saa"], 2, 80, 0.5)
    close(hr, g["real_ibp80"])
    np.testing.assert_allclose(errs, g["real_errors"], rtol=1e-10)


def test_ibp_ragged(g_rag):
    hr, errs = O.ibp(list(g_rag["lr"]), g_rag["shifts"], g_rag["psf_m"], g_rag["hr_init"], 2, 10, 0.5)
    close(hr, g_rag["ibp_10"])
    close(errs, g_rag["ibp_errors"])


@pytest.mark.parametrize("name", ["mono_tl", "mono_br", "mono_mid", "rgb_tr", "rgb_mid"])
def test_real_crops(g_real, name):
    fam = name.split("_")[0]
    lr = g_real[f"{name}_lr"].astype(np.float64)
    if fam == "rgb":
        # redo extract_red (rgb_cal_target/run_sr.py:73-75) + rep mean (:107-108) from raw Bayer crops
        raw = g_real[f"{name}_raw"].astype(np.float64)  # [4, R, 96, 96]
        lr2 = np.stack([O.mean0(np.stack([O.extract_red(r) for r in reps])) for reps in raw])
        close(lr2, lr, 0.0)
    shifts = g_real[f"{fam}_shifts"]
    psf = g_real["psf_g"] if fam == "mono" else g_real["psf_m"]
    close(O.ndi_zoom(O.mean0(lr), 2), g_real[f"{name}_native"])
    saa = O.shift_and_add(list(lr), shifts, 2)
    close(saa, g_real[f"{name}_saa"])
    hr, errs = O.ibp(list(lr), shifts, psf, saa, 2, 10, 0.5)
    close(hr, g_real[f"{name}_ibp10"])
    close(errs, g_real[f"{name}_errors"])


def test_quantize_truncates():
    x = np.array([-3.0, 0.0, 0.999, 1.0, 127.5, 254.9999, 255.0, 300.0])
    assert O.quantize_u8(x).tolist() == [0, 0, 0, 1, 127, 254, 255, 255]
    assert O.quantize_u8(x).tolist() == np.clip(x, 0, 255).astype(np.uint8).tolist()


def test_psf_matches_reference(g_c1):
    close(O.make_gaussian_psf(), g_c1["psf_g"], 1e-17)


def test_interleave4_known_answer():
    """The vendor live view's 4-frame interleave (XPR_Software.py:196-205, 388-410): pinned by DOCUMENTED semantics, not by a
    reference run (cv2 is absent here) -- tests/golden/interleave4_3x3.npz is written out by hand from the definition of
    cv2.warpAffine for a pure translation and of BORDER_REFLECT_101 (tools/make_interleave_fixture.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interleave4_3x3.npz"))
    assert np.array_equal(O.interleave4(g["frames"]), g["expected"])


# ---- round 4: reference-generated goldens sized for the window kernels (tools/make_golden.py --only-windows) ----
F32_STORE = 2e-5  # big outputs are stored as float32 (half an ulp at 255 is 7.6e-6)


def _win_rgb_frames(g):
    """extract_red (rgb_cal_target/run_sr.py:73-75) + rep mean (:107-108) from the raw Bayer crops of every rep"""
    raw = g["raw"].astype(np.float64)  # [4 corners, R reps, 2h, 2w]
    return [np.mean(np.stack([r[0::2, 0::2] for r in reps]), axis=0) for reps in raw]


@pytest.mark.parametrize("psf_tag", ["g", "m"])
def test_window_golden_rgb_crop(psf_tag):
    """rgb_cal_target's own inputs (rep-averaged, non-integer red frames; measured shifts) for its 50 iterations with the default
    Gaussian PSF and with --psf measured (rgb_cal_target/run_sr.py:59, :128-166, :204-223)."""
    from conftest import load_golden
    g = load_golden("win_btile.npz")
    lr, sh = _win_rgb_frames(g), g["shifts"]
    close(O.shift_and_add(lr, sh, 2), g["saa"], F32_STORE)
    init = g["saa"].astype(np.float64)
    O.set_threads(8)
    try:
        hr, errs = O.ibp(lr, sh, g[f"psf_{psf_tag}"], init, 2, 50, 0.5)
    finally:
        O.set_threads(1)
    close(hr, g[f"ibp50_{psf_tag}"], F32_STORE)
    np.testing.assert_allclose(errs, g[f"errors_{psf_tag}"], rtol=1e-10)


@pytest.mark.parametrize("name", ["win_dtile", "win_dtile_float", "win_atile"])
def test_window_golden_phase_grids(name):
    """x4, all 16 phases, 80 iterations of the reference's ibp (mono_cal_target/run_sr.py:190-209) on 288 x 320 HR (integer and
    half-integer frames) and 160 x 200 HR."""
    from conftest import load_golden
    g = load_golden(name + ".npz")
    lr = g["lr16"].astype(np.float64) if "lr16" in g else 0.5 * (g["lr16_a"].astype(np.float64) + g["lr16_b"].astype(np.float64))
    sh, init = g["shifts16"], g["saa16"].astype(np.float64)
    O.set_threads(8)
    try:
        close(O.shift_and_add(list(lr), sh, 4), g["saa16"], F32_STORE)
        hr, errs = O.ibp(list(lr), sh, g["psf_g"], init, 4, 80, 0.5)
    finally:
        O.set_threads(1)
    close(hr, g["ibp16_80"], F32_STORE)
    np.testing.assert_allclose(errs, g["ibp16_errors"], rtol=1e-10)
