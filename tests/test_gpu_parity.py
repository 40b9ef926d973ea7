"""GPU parity tests: the HIP path (libsrx.so through the C ABI, via the sr_mi355x shim) against
(1) golden vectors made by the reference's own functions (tests/golden, tools/make_golden.py)
and (2) the CPU oracle on seeded inputs.  Tolerances (DN on 0..255 data):
    f64 : 1e-9 primitives, 1e-8 after 80 IBP iterations           (the reference's precision)
    f32 : 5e-4 primitives, 1e-3 after 80 IBP iterations (measured: 1.2e-4 ... 2.2e-4), PSNR(build, ref) > 90 dB and
          |PSNR(build, truth) - PSNR(ref, truth)| < 0.01 dB         (the north-star bar)
    index maps (decimate / zero-insert / Bayer red / quantiser): bit-exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

torch = pytest.importorskip("torch")
import sr_mi355x as S  # noqa: E402
from sr_mi355x import synth  # noqa: E402

PRIM_TOL = {"f64": 1e-9, "f32": 5e-4}
IBP_TOL = {"f64": 1e-8, "f32": 1e-3}
ERR_RTOL = {"f64": 1e-10, "f32": 2e-5}


@pytest.fixture(params=["f64", "f32"])
def prec(request):
    S.set_precision(request.param)
    yield request.param
    S.set_precision("f32")


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = float(np.abs(a - b).max())
    assert d <= tol, f"max |delta| = {d:.3e} > {tol:.1e}"


def test_native_library_is_loaded():
    from sr_mi355x import _lib
    assert _lib.load().srx_version() >= 100
    maps = open("/proc/self/maps").read()
    assert "libsrx.so" in maps


def test_blur(prec, g_c1, g_rag):
    t = PRIM_TOL[prec]
    close(S.blur(g_c1["truth"], g_c1["psf_g"]), g_c1["blur_g"], t)
    close(S.blur(g_c1["truth"], g_c1["psf_m"]), g_c1["blur_m"], t)
    close(S.blur(g_rag["truth"], g_rag["k53"]), g_rag["blur53"], t)


def test_shift(prec, g_c1, g_rag):
    t = PRIM_TOL[prec]
    close(S.ndi_shift(g_c1["truth"], g_c1["shift_frac_arg"]), g_c1["shift_frac"], t)
    close(S.ndi_shift(g_c1["truth"], g_c1["shift_int_arg"]), g_c1["shift_int"], t)
    close(S.ndi_shift(g_rag["truth"], g_rag["shift_big_arg"]), g_rag["shift_big"], t)


def test_zoom(prec, g_c1, g_c2s, g_rag):
    t = PRIM_TOL[prec]
    close(S.ndi_zoom(g_c1["lr_nom"][0], 2), g_c1["zoom2"], t)
    close(S.ndi_zoom(g_c2s["lr16"][3], 4), g_c2s["zoom4"], t)
    close(S.ndi_zoom(g_rag["truth"][:21, :17], 3), g_rag["zoom3"], t)


def test_forward_back(prec, g_c1, g_c2s, g_rag):
    t = PRIM_TOL[prec]
    for k, s in enumerate(g_c1["shifts_meas"]):
        close(S.forward_model(g_c1["truth"], g_c1["psf_m"], s, 2), g_c1["fwd_meas"][k], t)
        e = g_c1["lr_meas"][k].astype(np.float64) - g_c1["fwd_meas"][k]
        close(S.back_project(e, g_c1["psf_m"], s, 2, g_c1["truth"].shape), g_c1["bp_meas"][k], t)
    for k, s in enumerate(g_c1["shifts_nom"]):
        close(S.forward_model(g_c1["truth"], g_c1["psf_g"], s, 2), g_c1["fwd_nom"][k], t)
    for k in (0, 5, 15):
        close(S.forward_model(g_c2s["truth"], g_c2s["psf_g"], g_c2s["shifts16"][k], 4), g_c2s["fwd16"][k], t)
    for k, s in enumerate(g_rag["shifts"]):
        close(S.forward_model(g_rag["truth"], g_rag["psf_m"], s, 2), g_rag["fwd"][k], t)
    e = g_rag["lr"][0].astype(np.float64) - g_rag["fwd"][0][:32, :33]
    close(S.back_project(e, g_rag["psf_m"], g_rag["shifts"][0], 2, g_rag["truth"].shape), g_rag["bp0"], t)


def test_shift_and_add(prec, g_c1, g_c2s):
    t = PRIM_TOL[prec]
    close(S.shift_and_add(list(g_c1["lr_nom"]), g_c1["shifts_nom"], 2), g_c1["saa_nom"], t)
    lr_avg = g_c1["lr_reps"].astype(np.float64).mean(axis=0)
    close(S.shift_and_add(list(lr_avg), g_c1["shifts_meas"], 2), g_c1["saa_meas"], t)
    close(S.shift_and_add(list(g_c2s["lr16"]), g_c2s["shifts16"], 4), g_c2s["saa16"], t)
    close(S.shift_and_add(list(g_c2s["lr4"]), g_c2s["shifts4"], 4), g_c2s["saa4"], t)


@pytest.mark.parametrize("flags", ["composed", "auto"])
@pytest.mark.parametrize("n", [1, 2, 10, 80])
def test_ibp_c1_nominal(prec, g_c1, n, flags):
    fl = S.FLAG_COMPOSED if flags == "composed" else S.FLAG_AUTO
    lr = torch.from_numpy(g_c1["lr_nom"].astype(np.float64))[None]
    hr, errs = S.ibp_batched(lr, g_c1["shifts_nom"], g_c1["psf_g"], g_c1["saa_nom"][None], 2, n, 0.5, flags=fl)
    close(hr[0].cpu().numpy(), g_c1[f"ibp_nom_{n}"], IBP_TOL[prec])
    np.testing.assert_allclose(errs[0].cpu().numpy(), g_c1["ibp_nom_errors"][:n], rtol=ERR_RTOL[prec])


@pytest.mark.parametrize("flags", ["composed", "auto"])
@pytest.mark.parametrize("n", [1, 10, 50])
def test_ibp_c1_measured(prec, g_c1, n, flags):
    fl = S.FLAG_COMPOSED if flags == "composed" else S.FLAG_AUTO
    lr_avg = g_c1["lr_reps"].astype(np.float64).mean(axis=0)
    hr, errs = S.ibp_batched(lr_avg[None], g_c1["shifts_meas"], g_c1["psf_m"], g_c1["saa_meas"][None], 2, n, 0.5,
                             flags=fl)
    close(hr[0].cpu().numpy(), g_c1[f"ibp_meas_{n}"], IBP_TOL[prec])
    np.testing.assert_allclose(errs[0].cpu().numpy(), g_c1["ibp_meas_errors"][:n], rtol=ERR_RTOL[prec])


@pytest.mark.parametrize("n", [1, 10, 80])
def test_ibp_c2_small(prec, g_c2s, n):
    hr, errs = S.ibp(list(g_c2s["lr16"]), g_c2s["shifts16"], g_c2s["psf_g"], g_c2s["saa16"], 4, n, 0.5, verbose=False)
    close(hr, g_c2s[f"ibp16_{n}"], IBP_TOL[prec])
    np.testing.assert_allclose(errs, g_c2s["ibp16_errors"][:n], rtol=ERR_RTOL[prec])
    hr, errs = S.ibp(list(g_c2s["lr4"]), g_c2s["shifts4"], g_c2s["psf_m"], g_c2s["saa4"], 4, n, 0.5, verbose=False)
    close(hr, g_c2s[f"ibp4_{n}"], IBP_TOL[prec])
    np.testing.assert_allclose(errs, g_c2s["ibp4_errors"][:n], rtol=ERR_RTOL[prec])


def u8_close(hr, ref):
    """uint8 outputs (the reference's truncating quantiser, run_sr.py:303): <= 1 LSB, >= 99.9 % identical"""
    q, qr = S.quantize_u8(hr).astype(np.int16), np.clip(ref, 0, 255).astype(np.uint8).astype(np.int16)
    assert np.abs(q - qr).max() <= 1 and (q == qr).mean() >= 0.999, (np.abs(q - qr).max(), (q == qr).mean())


@pytest.mark.parametrize("case", ["synth_gauss", "synth_measured_psf", "real_crop"])
def test_ibp_frame_80_iterations_golden(prec, g_frame, case):
    """The kernel behind the reference's shipped defaults (delta = 0 on frames: k_ibp_ztile in f32) against REFERENCE-generated
    goldens at its full 80 iterations (mono_cal_target/run_sr.py:190-209): 144 x 280 HR = 3 x 2 ragged tiles."""
    g = g_frame
    if case == "real_crop":
        lr, sh, psf, saa_ref, ref, eref = g["real_lr"], g["real_shifts"], g["psf_g"], g["real_saa"], g["real_ibp80"], g["real_errors"]
    else:
        lr, sh, saa_ref = g["lr5"], g["shifts5"], g["saa5"]
        psf, ref, eref = (g["psf_g"], g["ibp5_80"], g["ibp5_errors"]) if case == "synth_gauss" else (g["psf_m"], g["ibp5m_80"], g["ibp5m_errors"])
    lr = list(lr.astype(np.float64))
    saa = S.shift_and_add(lr, sh, 2)
    close(saa, saa_ref, PRIM_TOL[prec])
    hr, errs = S.ibp(lr, sh, psf, saa_ref, 2, 80, 0.5, verbose=False)
    # float32: k_ibp_ztile (either PSF form); float64: the transpose-free k_ibp_ctile for a rank-1 PSF, the tile kernels otherwise
    assert S.last_path() == ("ztile" if prec == "f32" else "mosaic" if case == "synth_measured_psf" else "ctile")
    close(hr, ref, IBP_TOL[prec])
    np.testing.assert_allclose(errs, eref, rtol=ERR_RTOL[prec])
    u8_close(hr, ref)
    if case != "synth_measured_psf":  # the float32 form of the transpose-free kernel (on request: k_ibp_ztile is faster in float32)
        S.set_precision("f32")
        hr_c, errs_c = S.ibp_batched(np.stack(lr)[None], sh, psf, saa_ref[None], 2, 80, 0.5, flags=S.FLAG_DIAG_COLUMN_TILES)
        assert S.last_path() == "ctile"
        close(hr_c[0].cpu().numpy(), ref, IBP_TOL["f32"])
        np.testing.assert_allclose(errs_c[0].cpu().numpy(), eref, rtol=ERR_RTOL["f32"])
        S.set_precision(prec)
    if case == "synth_gauss":
        for n in (1, 10):
            hr_n, errs_n = S.ibp(lr, sh, psf, saa_ref, 2, n, 0.5, verbose=False)
            close(hr_n, g[f"ibp5_{n}"], IBP_TOL["f32"])  # stored as float32
            np.testing.assert_allclose(errs_n, eref[:n], rtol=ERR_RTOL[prec])
        truth = g["truth"].astype(np.float64)
        assert synth.psnr(hr, ref) > 90.0 and abs(synth.psnr(hr, truth) - synth.psnr(ref, truth)) < 0.01


def test_ibp_c2_full_psnr(prec, g_c2f):
    """Config C2 patch (f=4, N=16, 64x64 LR -> 256x256, 80 iterations): the north-star PSNR bar."""
    saa = S.shift_and_add(list(g_c2f["lr16"]), g_c2f["shifts16"], 4)
    close(saa, g_c2f["saa16"], PRIM_TOL[prec])
    hr, errs = S.ibp(list(g_c2f["lr16"]), g_c2f["shifts16"], g_c2f["psf_g"], saa, 4, 80, 0.5, verbose=False)
    if prec == "f32":
        assert S.last_path() == "patch"  # an eligibility regression would otherwise test the tile kernels silently
    ref, truth = g_c2f["ibp16_80"], g_c2f["truth"].astype(np.float64)
    close(hr, ref, IBP_TOL[prec])
    assert synth.psnr(hr, ref) > 90.0
    assert abs(synth.psnr(hr, truth) - synth.psnr(ref, truth)) < 0.01
    np.testing.assert_allclose(errs, g_c2f["ibp16_errors"], rtol=ERR_RTOL[prec])
    # uint8 outputs (truncating quantiser): <= 1 LSB, >= 99.9 % identical
    q, qr = S.quantize_u8(hr).astype(np.int16), np.clip(ref, 0, 255).astype(np.uint8).astype(np.int16)
    assert np.abs(q - qr).max() <= 1 and (q == qr).mean() >= 0.999


def test_ibp_ragged(prec, g_rag):
    hr, errs = S.ibp(list(g_rag["lr"]), g_rag["shifts"], g_rag["psf_m"], g_rag["hr_init"], 2, 10, 0.5, verbose=False)
    close(hr, g_rag["ibp_10"], IBP_TOL[prec])
    np.testing.assert_allclose(errs, g_rag["ibp_errors"], rtol=ERR_RTOL[prec])


@pytest.mark.parametrize("name", ["mono_tl", "mono_br", "mono_mid", "rgb_tr", "rgb_mid"])
def test_real_crops(prec, g_real, name):
    fam = name.split("_")[0]
    lr = g_real[f"{name}_lr"].astype(np.float64)
    if fam == "rgb":
        raw = g_real[f"{name}_raw"]  # uint8 [4 corners, R reps, 96, 96] Bayer crops
        S.set_precision("f64")       # loader arithmetic is checked bit-exactly in float64
        lr2 = np.stack([S.mean_frames(np.stack([S.extract_red(r.astype(np.float64)) for r in reps])) for reps in raw])
        S.set_precision(prec)
        assert np.array_equal(lr2, lr)
    shifts = g_real[f"{fam}_shifts"]
    psf = g_real["psf_g"] if fam == "mono" else g_real["psf_m"]
    close(S.ndi_zoom(S.mean_frames(lr), 2), g_real[f"{name}_native"], PRIM_TOL[prec])
    saa = S.shift_and_add(list(lr), shifts, 2)
    close(saa, g_real[f"{name}_saa"], PRIM_TOL[prec])
    hr, errs = S.ibp(list(lr), shifts, psf, saa, 2, 10, 0.5, verbose=False)
    close(hr, g_real[f"{name}_ibp10"], IBP_TOL[prec])
    np.testing.assert_allclose(errs, g_real[f"{name}_errors"], rtol=ERR_RTOL[prec])


def test_index_maps_bit_exact(prec):
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 255, (37, 50))
    if prec == "f32":
        x = x.astype(np.float32).astype(np.float64)
    for f in (2, 3, 4):
        assert np.array_equal(S.decimate(x, f), x[::f, ::f])
    assert np.array_equal(S.decimate(x, 2, 1, 1), x[1::2, 1::2])
    assert np.array_equal(S.extract_red(x), x[0::2, 0::2])
    e = x[:12, :17]
    for f, shape in ((2, (24, 34)), (2, (27, 31)), (4, (48, 68)), (3, (30, 60))):
        up = np.zeros((e.shape[0] * f, e.shape[1] * f))
        up[::f, ::f] = e
        up = np.pad(up, ((0, max(0, shape[0] - up.shape[0])), (0, max(0, shape[1] - up.shape[1]))))[:shape[0], :shape[1]]
        assert np.array_equal(S.zero_insert(e, f, shape), up)
    y = np.concatenate([x.ravel(), [-3.0, 0.0, 0.999, 1.0, 254.9999, 255.0, 300.0, 127.5]])
    if prec == "f32":
        y = y.astype(np.float32).astype(np.float64)
    assert np.array_equal(S.quantize_u8(y), np.clip(y, 0, 255).astype(np.uint8))
    u8 = rng.integers(0, 256, (5, 9, 11), dtype=np.uint8)
    assert np.array_equal(S.u8_to_float(u8).cpu().numpy(), u8.astype(np.float64))
    st = rng.integers(0, 256, (5, 9, 11)).astype(np.float64)
    if prec == "f64":
        assert np.array_equal(S.mean_frames(st), st.mean(axis=0))


def test_batch_equals_loop(prec, g_c1):
    """B independent items in one call == B single calls."""
    rng = np.random.default_rng(11)
    lr = np.stack([np.clip(g_c1["lr_nom"].astype(np.float64) + rng.normal(0, 3, g_c1["lr_nom"].shape), 0, 255)
                   for _ in range(3)])
    saa = S.shift_and_add_batched(lr, g_c1["shifts_nom"], 2)
    hr, errs = S.ibp_batched(lr, g_c1["shifts_nom"], g_c1["psf_g"], saa, 2, 5, 0.5)
    for b in range(3):
        s1 = S.shift_and_add_batched(lr[b:b + 1], g_c1["shifts_nom"], 2)
        h1, e1 = S.ibp_batched(lr[b:b + 1], g_c1["shifts_nom"], g_c1["psf_g"], s1, 2, 5, 0.5)
        assert torch.equal(s1[0], saa[b])
        assert torch.equal(h1[0], hr[b])
        np.testing.assert_allclose(e1[0].cpu().numpy(), errs[b].cpu().numpy(), rtol=1e-12)


def test_profiler_from_two_threads():
    """srx_profile_enable / srx_profile_get while another thread launches: the log is cleared under a running launch without a crash, a
    lost record or a record landing in a recycled slot (generation-tagged handles); every launch still computes the same result."""
    import ctypes
    import threading
    from sr_mi355x import _lib
    lib = _lib.load()
    S.set_precision("f32")
    x = torch.from_numpy(synth.truth_image(96, 96, seed=2)).cuda().float()[None].contiguous()
    psf = synth.gaussian_psf()
    ref = S.blur_batched(x, psf).clone()
    stop, bad = threading.Event(), []

    def worker():
        torch.cuda.set_device(0)
        lr = torch.stack([S.forward_model_batched(x, psf, s, 2) for s in synth.NOMINAL_4], dim=1).contiguous()
        while not stop.is_set():
            hr, _ = S.ibp_batched(lr, synth.NOMINAL_4, psf, x, 2, 2, 0.5)
            if not torch.isfinite(hr).all():
                bad.append("non-finite")

    t = threading.Thread(target=worker)
    t.start()
    try:
        tot, cnt = ctypes.c_double(), ctypes.c_long()
        for i in range(300):
            lib.srx_profile_enable(i & 1)
            out = S.blur_batched(x, psf)
            assert torch.equal(out, ref)
            for kid in range(lib.srx_profile_kernel_count()):
                assert lib.srx_profile_get(kid, ctypes.byref(tot), ctypes.byref(cnt)) == 0 and cnt.value >= 0 and tot.value >= 0.0
    finally:
        stop.set()
        t.join()
        lib.srx_profile_enable(0)
    assert not bad


def test_errors_are_reported():
    from sr_mi355x import _lib
    with pytest.raises(_lib.SrxError):
        S.blur(np.zeros((8, 8)), np.ones((16, 16)))  # 256 taps > SRX_MAX_KERNEL_TAPS


def test_large_image_chunked_prefilter(prec):
    """Lines longer than one prefilter chunk (256 rows / 512 columns): shift and zoom vs the oracle."""
    from oracle import sr_oracle as O
    O.set_threads(8)
    rng = np.random.default_rng(5)
    x = np.rint(rng.uniform(0, 255, (700, 1100)))
    t = PRIM_TOL[prec]
    close(S.ndi_shift(x, (0.9445, -0.8677)), O.ndi_shift(x, (0.9445, -0.8677)), t)
    close(S.ndi_zoom(x[:300, :577], 2), O.ndi_zoom(x[:300, :577], 2), t)
    O.set_threads(1)


FUSED_CFGS = {
    # name: (factor, shifts (LR px), psf, (h, w), expected path)
    "f2_meas": (2, synth.MEASURED_4, "asym", (150, 277), "fused", "btile"),  # distinct sub-pixel fractions: per-frame; x2 in float32 on register-resident windows (7 x 7 form)
    "f3_k5": (3, [(0.2, -0.4), (-1.0 / 3, 1.0 / 3), (0.9, 0.1)], "asym5", (60, 75), "fused"),
    "f2_nom5": (2, synth.NOMINAL_5, "gauss", (131, 200), "mosaic", "ztile"),  # integer HR shifts: pure depth-to-space; IBP on CU-resident tiles
    "f2_nom4_big": (2, synth.NOMINAL_4, "gauss", (150, 277), "mosaic", "ztile"),  # several 244-pixel tiles per axis, ragged last tiles
    "f3_int_odd": (3, [(0.0, 0.0), (1.0 / 3, -2.0 / 3), (-2.0 / 3, 1.0 / 3), (1.0 / 3, 1.0 / 3)], "gauss", (45, 61), "mosaic", "ztile"),  # x3, odd HR height 135:
    #                 the last row pair of the one-launch kernel's state planes is half image, half zero border
    "f4_nom4": (4, synth.NOMINAL_4, "asym", (70, 90), "mosaic"),
    "f4_ph16": (4, synth.phase_shifts(4), "gauss", (70, 90), "mosaic"),     # the bench workload: all fractions 0.5
    "f3_ph9": (3, synth.phase_shifts(3), "asym", (50, 66), "mosaic", "ztile"),  # fractions 0 (3 phases centred on 0); a PSF that is not rank 1: the 7 x 7 form of the one-launch kernel
    "f2_nom5_asym": (2, synth.NOMINAL_5, "asym", (131, 200), "mosaic", "ztile"),  # the reference's --psf measured on its nominal shifts (5 x 5 core of a 7 x 7)
    "f2_nom5_full7": (2, synth.NOMINAL_5, "full7", (131, 200), "mosaic", "ztile"),  # a PSF with weight on its outer ring: the full 7 x 7 form
    "f2_mixed": (2, [(0.5, 0.25), (-0.5, -0.25), (0.0, 0.75)], "asym", (90, 120), "mosaic"),  # y integer, x fraction 0.5
    "f2_multi": (2, [(0.25, 0.25), (1.25, 0.25), (0.25, -0.75), (-0.75, 1.25)], "gauss", (90, 120), "mosaic"),  # C = 4 on one phase
    "f4_frac": (4, [(0.05, 0.3), (0.3, 0.05), (-0.2, -0.45), (0.55, -0.2), (-0.45, 0.55)], "asym", (40, 50), "mosaic"),  # fractions 0.2
}


@pytest.mark.parametrize("cfg", sorted(FUSED_CFGS))
def test_fused_path_vs_oracle(prec, cfg):
    """Fused paths ("mosaic" when all shifts share one sub-pixel fraction, per-frame "fused" tiles otherwise) on
    images spanning several tiles, against the oracle (the literal restatement of the reference's loop)."""
    from oracle import sr_oracle as O
    O.set_threads(8)
    try:
        f, shifts, psf_name, (h, w), want = FUSED_CFGS[cfg][:5]
        want_ibp = FUSED_CFGS[cfg][5] if len(FUSED_CFGS[cfg]) > 5 else want
        psf = {"asym": synth.asymmetric_psf(), "gauss": synth.gaussian_psf(),
               "asym5": synth.asymmetric_psf()[1:6, 1:6] / synth.asymmetric_psf()[1:6, 1:6].sum(), "full7": synth.full_support_psf()}[psf_name]
        truth = synth.truth_image(h * f, w * f, seed=77)
        lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=78)
        saa_o = O.shift_and_add(list(lr), shifts, f)
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, 6, 0.5)
    finally:
        O.set_threads(1)
    saa = S.shift_and_add(list(lr), shifts, f)
    assert S.last_path() == want
    close(saa, saa_o, PRIM_TOL[prec])
    saa_p = S.shift_and_add_batched(lr[None], shifts, f, flags=S.FLAG_PER_FRAME)
    assert S.last_path() == "fused"
    close(saa_p[0].cpu().numpy(), saa_o, PRIM_TOL[prec])
    hr, errs = S.ibp(list(lr), shifts, psf, saa_o, f, 6, 0.5, verbose=False)
    want64 = "ctile" if (want_ibp == "ztile" and psf_name == "gauss") else want  # float64: the transpose-free frame kernel (rank-1 PSFs)
    flags_t = S.FLAG_TILES | (S.FLAG_PER_FRAME if want == "fused" else 0)
    if want_ibp == "mosaic" and psf_name == "gauss":
        want_ibp = "atile"  # float32, rank-1 PSF, a common fraction > 0 on a frame k_ibp_dtile does not take: the two-launch window kernels
    assert S.last_path() == (want_ibp if prec == "f32" else want64)
    close(hr, hr_o, IBP_TOL[prec])
    np.testing.assert_allclose(errs, err_o, rtol=ERR_RTOL[prec])
    if want_ibp != want and prec == "f32":  # the tile kernels of srx_mosaic.hpp on the same input
        hr_t, errs_t = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, 6, 0.5, flags=flags_t)
        assert S.last_path() == want
        close(hr_t[0].cpu().numpy(), hr_o, IBP_TOL[prec])
        np.testing.assert_allclose(errs_t[0].cpu().numpy(), err_o, rtol=ERR_RTOL[prec])
    # the composed (literal) and the per-frame fused HIP paths agree with it
    hr_c, errs_c = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, 6, 0.5, flags=S.FLAG_COMPOSED)
    assert S.last_path() == "composed"
    close(hr_c[0].cpu().numpy(), hr, 2 * IBP_TOL[prec])
    hr_p, errs_p = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, 6, 0.5, flags=S.FLAG_PER_FRAME)
    # (x2, float32, a PSF within 7 x 7: the per-frame formulation runs on register-resident windows, srx_btile.hpp)
    assert S.last_path() == ("btile" if prec == "f32" and f == 2 else "fused")
    close(hr_p[0].cpu().numpy(), hr, 2 * IBP_TOL[prec])
    np.testing.assert_allclose(errs_p[0].cpu().numpy(), err_o, rtol=ERR_RTOL[prec])


def test_one_launch_kernel_float_frames():
    """k_ibp_ztile on frames that are not 8-bit integers (rep means, calibrated frames): the float operand planes instead of the
    packed 16-bit ones, against the oracle and the tile kernels."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    f, shifts, psf = 2, synth.NOMINAL_5, synth.gaussian_psf()
    truth = synth.truth_image(300, 262, seed=5)
    O.set_threads(8)
    try:
        lr = np.stack([O.forward_model(truth, psf, s, f) for s in shifts]) + 0.37  # fractional samples
        saa_o = O.shift_and_add(list(lr), shifts, f)
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, 7, 0.5)
    finally:
        O.set_threads(1)
    hr, errs = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, 7, 0.5)
    assert S.last_path() == "ztile"
    close(hr[0].cpu().numpy(), hr_o, IBP_TOL["f32"])
    np.testing.assert_allclose(errs[0].cpu().numpy(), err_o, rtol=ERR_RTOL["f32"])
    hr_t, _ = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, 7, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "mosaic" and float((hr - hr_t).abs().max()) < 5e-4


def test_frame_kernel_is_deterministic():
    """Repeated calls of the one-launch frame kernel give bit-identical results, in both PSF forms, on rough data (where a stale
    value is a large error).  Round 3 found the 7 x 7 form deviating in 7 of 40 calls: the first ds_write_addtid_b32 behind a
    scalar write of M0 needs a wait state (srx_patch.hpp, SRX_M0_NOP); tools/stress_determinism.py is the long form of this test."""
    S.set_precision("f32")
    f, shifts = 2, synth.NOMINAL_5
    rng = np.random.default_rng(1)
    lr = torch.from_numpy(np.rint(rng.uniform(0, 255, (1, 5, 600, 800))) * 0.75 + 0.3).float().cuda()
    init = torch.from_numpy(rng.uniform(0, 255, (1, 1200, 1600))).float().cuda()
    for psf in (synth.asymmetric_psf(), synth.gaussian_psf(), synth.full_support_psf()):
        outs = [tuple(x.clone() for x in S.ibp_batched(lr, shifts, psf, init, f, 2, 0.5)) for _ in range(30)]
        assert S.last_path() == "ztile"
        assert all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
        # round 4: the MSE trace too -- its constant part (the scatter of the frames that share a phase) was one atomicAdd per block of
        # k_mosaic_build and changed in its last bits from call to call on frames that are not integers (tools/dev/zt_determinism.py)
        assert all(torch.equal(outs[0][1], o[1]) for o in outs[1:])
    small = lr[:, :, :90, :120].contiguous()
    outs = [tuple(x.clone() for x in S.ibp_batched(small, shifts, synth.gaussian_psf(), init[:, :180, :240].contiguous(), f, 2, 0.5, flags=S.FLAG_COMPOSED))
            for _ in range(8)]
    assert S.last_path() == "composed"  # ... and the composed path's (one atomicAdd per block of k_residual before)
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    # the transpose-free frame kernel in both precisions
    lr6 = torch.from_numpy(np.rint(rng.uniform(0, 255, (1, 5, 300, 400))) * 0.75 + 0.3).cuda()
    init6 = torch.from_numpy(rng.uniform(0, 255, (1, 600, 800))).cuda()
    for prec6, fl6 in (("f64", S.FLAG_AUTO), ("f32", S.FLAG_DIAG_COLUMN_TILES)):
        S.set_precision(prec6)
        outs = [S.ibp_batched(lr6, synth.NOMINAL_5, synth.gaussian_psf(), init6, 2, 2, 0.5, flags=fl6)[0].clone() for _ in range(12)]
        assert S.last_path() == "ctile"
        assert all(torch.equal(outs[0], o) for o in outs[1:])
    S.set_precision("f32")
    # the delta != 0 frame kernel, both window shapes (round 3: the 128-bit stores behind which a vector instruction overwrote the data
    # registers -- tools/microbench/store_data_war.hip -- deviated in 12 of 12 calls of the 4 x 4 shape)
    f4, sh16 = 4, synth.phase_shifts(4)
    lr4 = torch.from_numpy(np.rint(rng.uniform(0, 255, (1, 16, 80, 100)))).float().cuda()
    init4 = torch.from_numpy(rng.uniform(0, 255, (1, 320, 400))).float().cuda()
    for fl in (S.FLAG_AUTO, S.FLAG_DIAG_WIDE_WINDOWS):
        outs = [S.ibp_batched(lr4, sh16, synth.gaussian_psf(), init4, f4, 2, 0.5, flags=fl)[0].clone() for _ in range(12)]
        assert S.last_path() == "dtile"
        assert all(torch.equal(outs[0], o) for o in outs[1:])
    # and the patch kernel (the same in-wave transposes)
    f, shifts, psf = 4, synth.phase_shifts(4), synth.gaussian_psf()
    lr = torch.from_numpy(np.rint(rng.uniform(0, 255, (64, 16, 64, 64)))).float().cuda()
    init = torch.from_numpy(rng.uniform(0, 255, (64, 256, 256))).float().cuda()
    outs = [S.ibp_batched(lr, shifts, psf, init, f, 3, 0.5)[0].clone() for _ in range(10)]
    assert S.last_path() == "patch"
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_tiny_and_odd_inputs(prec):
    """Shapes below the fused paths' minimum (composed path), a 1-pixel-wide frame, N = 1, n_iter = 0."""
    from oracle import sr_oracle as O
    rng = np.random.default_rng(21)
    psf = synth.asymmetric_psf()
    for (h, w, f, shifts) in [(6, 7, 2, synth.NOMINAL_4), (5, 1, 2, [(0.25, 0.0), (-0.25, 0.0)]), (9, 8, 3, [(0.1, 0.2)])]:
        lr = np.rint(rng.uniform(0, 255, (len(shifts), h, w)))
        saa_o = O.shift_and_add(list(lr), shifts, f)
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, 4, 0.5)
        saa = S.shift_and_add(list(lr), shifts, f)
        close(saa, saa_o, PRIM_TOL[prec])
        hr, errs = S.ibp(list(lr), shifts, psf, saa_o, f, 4, 0.5, verbose=False)
        close(hr, hr_o, IBP_TOL[prec])
        np.testing.assert_allclose(errs, err_o, rtol=ERR_RTOL[prec])
        hr0, errs0 = S.ibp(list(lr), shifts, psf, saa_o, f, 0, 0.5, verbose=False)  # n_iter = 0: hr_init copy, no errors
        assert np.array_equal(hr0, saa_o if prec == "f64" else saa_o.astype(np.float32).astype(np.float64)) and errs0 == []


def test_unsupported_configurations_are_reported():
    from sr_mi355x import _lib
    lr = np.zeros((33, 8, 8))
    with pytest.raises(_lib.SrxError) as ei:  # N = 33 > SRX_MAX_FRAMES
        S.shift_and_add(list(lr), [(0.0, 0.0)] * 33, 2)
    assert ei.value.status == _lib.E_UNSUPPORTED
    with pytest.raises(_lib.SrxError):        # fused path demanded for a ragged shape
        S.ibp_batched(np.zeros((1, 2, 8, 8)), [(0, 0), (0.5, 0.5)], synth.gaussian_psf(), np.zeros((1, 17, 16)), 2, 1, 0.5,
                      flags=S.FLAG_FUSED)


def test_interleave4_depth_to_space():
    """Vendor live-view interleave (XPR_Software.py:388-410): bit-exact vs the numpy restatement, and in the interior
    the pure PixelShuffle index map out[2i + py_k, 2j + px_k] = frame_k[i, j]."""
    from oracle import sr_oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", "interleave4_3x3.npz"))  # hand-derived known answer (documented semantics)
    assert np.array_equal(S.interleave4(g["frames"]), g["expected"])
    rng = np.random.default_rng(8)
    for (h, w) in [(5, 7), (32, 48), (1, 1)]:
        fr = rng.integers(0, 256, (4, h, w), dtype=np.uint8)
        out = S.interleave4(fr)
        assert out.dtype == np.uint8 and out.shape == (2 * h, 2 * w)
        assert np.array_equal(out, O.interleave4(fr))
    fr = rng.integers(0, 256, (4, 16, 16), dtype=np.uint8)
    out = S.interleave4(fr)
    for k, (tx, ty) in enumerate([(0, 0), (0, 1), (-1, 1), (-1, 0)]):
        # frame k's samples land on HR phase (ty mod 2, tx mod 2); away from the reflected border nothing else does
        y = np.arange(2, 30)
        yy, xx = np.meshgrid(y, y, indexing="ij")
        m = ((yy - ty) % 2 == 0) & ((xx - tx) % 2 == 0)
        assert np.array_equal(out[2:30, 2:30][m[:, :]], fr[k][((yy - ty) // 2)[m], ((xx - tx) // 2)[m]])


# ---------------------------------------------------------------------------------------------------------
# Full-size properties (BASELINE.json sizes, too large for the CPU oracle): size-independent invariants
# ---------------------------------------------------------------------------------------------------------
def _full_c2_batch(B):
    """B patches of the headline workload: 64x64 LR, f=4, N=16 phases, frames from the product's own forward model."""
    f, shifts, psf = 4, synth.phase_shifts(4), synth.gaussian_psf()
    truths = np.stack([synth.truth_image(256, 256, seed=4000 + i) for i in range(8)])
    x = torch.from_numpy(truths).cuda().float().repeat((B + 7) // 8, 1, 1)[:B].contiguous()
    x = x + torch.arange(B, device="cuda", dtype=torch.float32)[:, None, None] * 0.01  # make every item distinct
    lr = torch.stack([S.forward_model_batched(x, psf, s, f, precision="f32") for s in shifts], dim=1).contiguous()
    return f, shifts, psf, x, lr


def test_full_size_ibp_fixed_point():
    """Noise-free frames of x: every residual is zero, so IBP started at x must stay at x and report MSE ~ 0
    (headline batch, B = 1024 patches of 256x256; and one 3072x4096 mono_cal_target-shaped frame)."""
    S.set_precision("f32")
    f, shifts, psf, x, lr = _full_c2_batch(1024)
    hr, errs = S.ibp_batched(lr, shifts, psf, x, f, 3, 0.5)
    assert S.last_path() == "patch"
    assert float((hr - x).abs().max()) < 2e-3 and float(errs.max()) < 1e-6
    hr, errs = S.ibp_batched(lr, shifts, psf, x, f, 3, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "mosaic"
    assert float((hr - x).abs().max()) < 2e-3 and float(errs.max()) < 1e-6
    # the reference's largest shape: N = 5 nominal shifts, f = 2, 1536x2048 -> 3072x4096, one item
    f2, sh5, psf_a = 2, synth.NOMINAL_5, synth.asymmetric_psf()
    big = torch.from_numpy(synth.truth_image(384, 512, seed=5)).cuda().float().repeat(8, 8)[None].contiguous()
    assert big.shape == (1, 3072, 4096)
    lr5 = torch.stack([S.forward_model_batched(big, psf_a, s, f2) for s in sh5], dim=1).contiguous()
    hr5, e5 = S.ibp_batched(lr5, sh5, psf_a, big, f2, 2, 0.5)
    assert float((hr5 - big).abs().max()) < 2e-3 and float(e5.max()) < 1e-6


def test_full_size_paths_agree_and_items_are_independent():
    """Headline patch size, noisy frames: the patch-resident kernel, the mosaic tile kernels, the per-frame fused path and (on a
    few items) the composed path give the same result; an item's result does not depend on its batch."""
    S.set_precision("f32")
    f, shifts, psf, x, lr = _full_c2_batch(96)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    lr = torch.clamp(torch.round(lr + 2.0 * torch.randn(lr.shape, generator=gen, device="cuda")), 0, 255)
    saa = S.shift_and_add_batched(lr, shifts, f)
    assert S.last_path() == "mosaic"
    saa_p = S.shift_and_add_batched(lr, shifts, f, flags=S.FLAG_PER_FRAME)
    assert float((saa - saa_p).abs().max()) < 2e-3
    hr_m, e_m = S.ibp_batched(lr, shifts, psf, saa, f, 8, 0.5)
    assert S.last_path() == "patch"
    hr_t, e_t = S.ibp_batched(lr, shifts, psf, saa, f, 8, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "mosaic"
    assert float((hr_m - hr_t).abs().max()) < 5e-4
    np.testing.assert_allclose(e_m.cpu().numpy(), e_t.cpu().numpy(), rtol=2e-6)
    hr_p, e_p = S.ibp_batched(lr, shifts, psf, saa, f, 8, 0.5, flags=S.FLAG_PER_FRAME)
    assert S.last_path() == "fused"
    assert float((hr_m - hr_p).abs().max()) < 5e-3
    np.testing.assert_allclose(e_m.cpu().numpy(), e_p.cpu().numpy(), rtol=2e-5)
    hr_c, e_c = S.ibp_batched(lr[:3], shifts, psf, saa[:3], f, 8, 0.5, flags=S.FLAG_COMPOSED)
    assert float((hr_m[:3] - hr_c).abs().max()) < 5e-3
    one, e1 = S.ibp_batched(lr[41:42], shifts, psf, saa[41:42], f, 8, 0.5)
    assert torch.equal(one[0], hr_m[41])


def test_full_size_linearity():
    """blur, forward_model and back_project are linear maps: f(a x + b y) = a f(x) + b f(y) at the headline size."""
    S.set_precision("f32")
    rng = torch.Generator(device="cuda")
    rng.manual_seed(9)
    x = torch.rand((64, 256, 256), generator=rng, device="cuda") * 255
    y = torch.rand((64, 256, 256), generator=rng, device="cuda") * 255
    psf, s = synth.asymmetric_psf(), (0.375, -0.125)
    for fn in (lambda t: S.blur_batched(t, psf), lambda t: S.forward_model_batched(t, psf, s, 4),
               lambda t: S.back_project_batched(S.forward_model_batched(t, psf, s, 4), psf, s, 4, (256, 256))):
        lhs = fn(0.25 * x + 0.5 * y)
        rhs = 0.25 * fn(x) + 0.5 * fn(y)
        assert float((lhs - rhs).abs().max()) < 2e-3


def test_full_frame_paths_agree_and_trace_is_deterministic():
    """The reference's own full-frame shapes, noisy frames, one item.  mono_cal_target (N = 5 nominal, f = 2, 3072x4096:
    the delta = 0 kernel on CU-resident tiles, then the two-launch mosaic tile kernels) against the per-frame fused path;
    rgb_cal_target (N = 4 measured shifts, 1536x2048: per-frame fused) against the composed path.  The MSE trace is a
    fixed-order sum of partial sums (no atomics): two runs are bit-identical."""
    S.set_precision("f32")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    for f, shifts, psf, tile, reps, want, main, other in (
            (2, synth.NOMINAL_5, synth.gaussian_psf(), (384, 512), (8, 8), "ztile", S.FLAG_AUTO, S.FLAG_PER_FRAME),
            (2, synth.NOMINAL_5, synth.gaussian_psf(), (384, 512), (8, 8), "mosaic", S.FLAG_TILES, S.FLAG_PER_FRAME),
            (2, synth.MEASURED_4, synth.asymmetric_psf(), (384, 512), (4, 4), "btile", S.FLAG_AUTO, S.FLAG_TILES),
            (2, synth.MEASURED_4, synth.asymmetric_psf(), (384, 512), (4, 4), "fused", S.FLAG_TILES, S.FLAG_COMPOSED)):
        big = torch.from_numpy(synth.truth_image(*tile, seed=6)).cuda().float().repeat(*reps)[None].contiguous()
        lr = torch.stack([S.forward_model_batched(big, psf, s, f) for s in shifts], dim=1)
        lr = torch.clamp(torch.round(lr + 2.0 * torch.randn(lr.shape, generator=gen, device="cuda")), 0, 255).contiguous()
        saa = S.shift_and_add_batched(lr, shifts, f)
        hr_a, e_a = S.ibp_batched(lr, shifts, psf, saa, f, 4, 0.5, flags=main)
        assert S.last_path() == want
        hr_b, e_b = S.ibp_batched(lr, shifts, psf, saa, f, 4, 0.5, flags=main)
        assert torch.equal(hr_a, hr_b) and torch.equal(e_a, e_b)
        hr_o, e_o = S.ibp_batched(lr, shifts, psf, saa, f, 4, 0.5, flags=other)
        assert S.last_path() != want
        assert float((hr_a - hr_o).abs().max()) < 5e-3
        np.testing.assert_allclose(e_a.cpu().numpy(), e_o.cpu().numpy(), rtol=2e-5)
        assert float(e_a[0, -1]) < float(e_a[0, 0])


def test_randomised_parity_sweep():
    """80 random (factor, frame set, shape, PSF, iteration count) cases through the auto-selected and the composed path
    in both precisions against the oracle (tools/fuzz_parity.py; 4000 cases of the same generator pass in ~95 s)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    try:
        assert fz.run(80, seed=459) == 0
    finally:
        S.set_precision("f32")


def test_batches_beyond_one_launch():
    """Batches larger than one launch's gridDim.z are split inside the library (srx_api.hip: chunks of 32768 items, or of
    32768 // N for shift_and_add's frame stacks; the workspace is sized for one chunk).  Items on both sides of a chunk
    boundary equal the same item computed alone."""
    S.set_precision("f32")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(21)
    # shift_and_add: B * N = 2100 * 16 > 32768 -> chunks of 2048 items
    f, shifts = 4, synth.phase_shifts(4)
    lr = torch.round(torch.rand((2100, 16, 8, 8), generator=gen, device="cuda") * 255)
    saa = S.shift_and_add_batched(lr, shifts, f)
    for i in (0, 2047, 2048, 2099):
        assert torch.equal(saa[i], S.shift_and_add_batched(lr[i:i + 1], shifts, f)[0])
    # ibp: B = 33000 > 32768
    f2, sh2, psf = 2, [(0.5, -0.5), (-0.5, 0.5)], synth.gaussian_psf()
    lr2 = torch.round(torch.rand((33000, 2, 16, 16), generator=gen, device="cuda") * 255)
    init = S.shift_and_add_batched(lr2, sh2, f2)
    hr, errs = S.ibp_batched(lr2, sh2, psf, init, f2, 3, 0.5)
    path = S.last_path()
    for i in (0, 32767, 32768, 32999):
        h1, e1 = S.ibp_batched(lr2[i:i + 1], sh2, psf, init[i:i + 1], f2, 3, 0.5)
        assert S.last_path() == path
        assert torch.equal(hr[i], h1[0])
        np.testing.assert_allclose(errs[i].cpu().numpy(), e1[0].cpu().numpy(), rtol=1e-12)


def test_delta_zero_fused_forward_matches_two_kernel_form(prec):
    """delta = 0 (the reference's nominal +-0.5 px at f = 2): the forward kernel that blurs its own image tiles
    (k_blurfwd_zero) against blur + index-map kernels (SRX_FLAG_DIAG_NO_ZERO_FUSE), on an image with ragged edge tiles."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    dt = torch.float64 if prec == "f64" else torch.float32
    lr = torch.round(torch.rand((3, 5, 75, 131), generator=gen, device="cuda", dtype=dt) * 255)
    init = S.shift_and_add_batched(lr, synth.NOMINAL_5, 2)
    for psf in (synth.gaussian_psf(), synth.asymmetric_psf()):
        hr_a, e_a = S.ibp_batched(lr, synth.NOMINAL_5, psf, init, 2, 5, 0.5, flags=S.FLAG_TILES)
        assert S.last_path() == "mosaic"
        hr_b, e_b = S.ibp_batched(lr, synth.NOMINAL_5, psf, init, 2, 5, 0.5, flags=S.FLAG_TILES | S.FLAG_DIAG_NO_ZERO_FUSE)
        assert S.last_path() == "mosaic"
        assert float((hr_a - hr_b).abs().max()) <= (1e-10 if prec == "f64" else 2e-4)
        np.testing.assert_allclose(e_a.cpu().numpy(), e_b.cpu().numpy(), rtol=1e-12 if prec == "f64" else 1e-6)


def test_shift_and_add_two_pass_equals_one_pass(prec):
    """Round 4: shift_and_add on a common fraction runs as accumulate (halo-free tiles of the W plane) + shift; the one-kernel form is
    kept on request.  The same sums in the same order, so the same bits: x2 / x3 / x4, shapes that are not multiples of any tile edge,
    planes smaller than one tile, a batch."""
    S.set_precision(prec)
    rng = np.random.default_rng(77)
    for f, shifts, (h, w), B in ((4, synth.phase_shifts(4), (64, 64), 3), (2, synth.NOMINAL_5, (150, 233), 1), (3, synth.phase_shifts(3), (41, 57), 2),
                                 (2, synth.phase_shifts(2), (9, 11), 1), (4, synth.phase_shifts(4)[:7], (100, 30), 1)):
        lr = torch.from_numpy(rng.uniform(0, 255, (B, len(shifts), h, w))).cuda()
        a = S.shift_and_add_batched(lr, shifts, f)
        assert S.last_path() == "mosaic"
        b = S.shift_and_add_batched(lr, shifts, f, flags=S.FLAG_DIAG_SAA_ONE_PASS)
        assert S.last_path() == "mosaic"
        assert torch.equal(a, b), (f, h, w, float((a - b).abs().max()))


# ---------------------------------------------------------------------------------------------------------
# The patch-resident kernel (csrc/srx_patch.hpp): one workgroup per 256 x 256 HR patch, all iterations in one launch
# ---------------------------------------------------------------------------------------------------------
def _patch_case(f, shifts, n_items, integer_lr=True, seed=300, psf_name="gauss"):
    from oracle import sr_oracle as O
    psf = {"gauss": synth.gaussian_psf(), "asym": synth.asymmetric_psf(), "full7": synth.full_support_psf()}[psf_name]
    O.set_threads(8)
    try:
        truths = [synth.truth_image(256, 256, seed=seed + i) for i in range(n_items)]
        lr = np.stack([np.stack([O.forward_model(t, psf, s, f) for s in shifts]) for t in truths])
        lr = np.stack([synth.sensor_frames(x, seed=seed + 50 + i) for i, x in enumerate(lr)]) if integer_lr else \
            np.clip(lr + np.random.default_rng(seed).normal(0, 1, lr.shape), 0, 255)  # rep-averaged style: not integers
        saa = np.stack([O.shift_and_add(list(x), shifts, f) for x in lr])
    finally:
        O.set_threads(1)
    return psf, lr, saa


PATCH_CFGS = {
    # name: (factor, shifts, integer-valued LR).  "dup": two frames on one phase -> the count map is not a 0/1 product
    "x4_grid": (4, synth.phase_shifts(4), True),
    "x4_grid_float": (4, synth.phase_shifts(4), False),
    "x2_grid": (2, synth.phase_shifts(2), True),
    "x4_dup": (4, synth.phase_shifts(4) + [synth.phase_shifts(4)[5]], True),
    "x2_half_row": (2, [(0.25, 0.25), (0.25, -0.25)], True),
}
# round 4: a PSF that is not rank 1 on the patch kernel (the 7 x 7 form of its two blurs; "asym": outer ring zero like the reference's measured PSF)
PATCH_CFGS_7X7 = {
    "x4_grid_asym": (4, synth.phase_shifts(4), True, "asym"),
    "x4_grid_full7": (4, synth.phase_shifts(4), True, "full7"),
    "x4_grid_float_asym": (4, synth.phase_shifts(4), False, "asym"),
    "x4_sub12_full7": (4, [s for s in synth.phase_shifts(4) if s[0] > -0.3], False, "full7"),  # a 3 x 4 product grid, non-integer frames
    "x4_dup_full7": (4, synth.phase_shifts(4) + [synth.phase_shifts(4)[5]], True, "full7"),    # a count plane (two frames on one phase)
    "x4_lattice_float_full7": (4, [synth.phase_shifts(4)[i] for i in (0, 5, 6, 15)], False, "full7"),  # a lattice that is no product, float mosaic
    "x4_dup_float_asym": (4, synth.phase_shifts(4) + [synth.phase_shifts(4)[5]], False, "asym"),
    "x2_grid_asym": (2, synth.phase_shifts(2), True, "asym"),
    "x2_half_row_full7": (2, [(0.25, 0.25), (0.25, -0.25)], False, "full7"),
}


@pytest.mark.parametrize("cfg", sorted(PATCH_CFGS) + sorted(PATCH_CFGS_7X7))
def test_patch_kernel_vs_oracle(cfg):
    """k_ibp_patch against the oracle after 1, 2, 10 and 80 iterations (HR state and MSE trace), and against the tile kernels:
    full phase grids (uint8 mosaic + 0/1 count masks), non-integer frames (float mosaic), frames sharing a phase (count plane); round 4:
    the same with PSFs that are not rank 1 (a 5 x 5 core like the reference's measured PSF, and full 7 x 7 support)."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    f, shifts, integer_lr = (PATCH_CFGS.get(cfg) or PATCH_CFGS_7X7[cfg])[:3]
    psf, lr, saa = _patch_case(f, shifts, 2, integer_lr, psf_name=PATCH_CFGS_7X7[cfg][3] if cfg in PATCH_CFGS_7X7 else "gauss")
    O.set_threads(8)
    try:
        for n in (1, 2, 10, 80):
            hr, errs = S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5)
            assert S.last_path() == "patch"
            for i in range(2 if n < 80 else 1):
                hr_o, err_o = O.ibp(list(lr[i]), shifts, psf, saa[i], f, n, 0.5)
                close(hr[i].cpu().numpy(), hr_o, IBP_TOL["f32"])
                np.testing.assert_allclose(errs[i].cpu().numpy(), err_o, rtol=ERR_RTOL["f32"])
    finally:
        O.set_threads(1)
    hr_t, e_t = S.ibp_batched(lr, shifts, psf, saa, f, 80, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "mosaic"
    assert float((hr - hr_t).abs().max()) < 5e-4
    np.testing.assert_allclose(errs.cpu().numpy(), e_t.cpu().numpy(), rtol=2e-6)
    for _ in range(4):  # the same bits from call to call
        hr_r, errs_r = S.ibp_batched(lr, shifts, psf, saa, f, 80, 0.5)
        assert torch.equal(hr_r, hr) and torch.equal(errs_r, errs)


def test_patch_kernel_in_place_batches_and_fallbacks():
    """hr_out aliasing hr_init; a batch equals its items; shapes / PSFs the patch kernel does not take fall back to the tiles."""
    S.set_precision("f32")
    f, shifts = 4, synth.phase_shifts(4)
    psf, lr, saa = _patch_case(f, shifts, 5, seed=340)
    lr_d, saa_d = torch.from_numpy(lr).cuda().float(), torch.from_numpy(saa).cuda().float()
    hr, errs = S.ibp_batched(lr_d, shifts, psf, saa_d, f, 12, 0.5)
    assert S.last_path() == "patch"
    buf = saa_d.clone()
    hr2, errs2 = S.ibp_batched(lr_d, shifts, psf, buf, f, 12, 0.5, out=buf)
    assert hr2.data_ptr() == buf.data_ptr() and torch.equal(hr, hr2) and torch.equal(errs, errs2)
    one, e1 = S.ibp_batched(lr_d[3:4], shifts, psf, saa_d[3:4], f, 12, 0.5)
    assert torch.equal(one[0], hr[3]) and torch.equal(e1[0], errs[3])
    S.ibp_batched(lr_d[:1], shifts, synth.asymmetric_psf(), saa_d[:1], f, 2, 0.5)          # a PSF that is not rank 1: the kernel's 7 x 7 form
    assert S.last_path() == "patch"
    h7, e7 = S.ibp_batched(lr_d[:2], shifts, synth.full_support_psf(), saa_d[:2], f, 3, 0.5)
    h7b, e7b = S.ibp_batched(lr_d[:2], shifts, synth.full_support_psf(), saa_d[:2], f, 3, 0.5, exact_workspace=False)  # shape-only arena
    assert S.last_path() == "patch" and torch.equal(h7, h7b) and torch.equal(e7, e7b)
    dup = shifts + [shifts[5]]                                                              # ... also with a count plane (two frames on one phase)
    S.ibp_batched(torch.cat([lr_d[:1], lr_d[:1, 5:6]], dim=1), dup, synth.asymmetric_psf(), saa_d[:1], f, 2, 0.5)
    assert S.last_path() == "patch"
    S.ibp_batched(lr_d[:1, :, :32, :32], shifts, psf, saa_d[:1, :128, :128], f, 2, 0.5)    # 128 x 128 HR: the two-launch window kernels
    assert S.last_path() == "atile"
    S.ibp_batched(lr_d[:1].double(), shifts, psf, saa_d[:1].double(), f, 2, 0.5, precision="f64")
    assert S.last_path() == "stile"                                                         # float64: the strip kernels (srx_stile.hpp)
    S.ibp_batched(lr_d[:1].double(), shifts, synth.asymmetric_psf(), saa_d[:1].double(), f, 2, 0.5, precision="f64")
    assert S.last_path() == "mosaic"


@pytest.mark.parametrize("cfg", sorted(PATCH_CFGS))
def test_strip_kernels_f64_vs_oracle(cfg):
    """Round 4: float64 patches with a common fraction > 0 (C2 in the reference's own precision, mono_cal_target/run_sr.py:74) run as two
    launches per iteration on register-resident strips (k_ibp_sv / k_ibp_sh, srx_stile.hpp) instead of the tile kernels.  Against the oracle
    after 1, 2, 10 and 80 iterations at the float64 tolerances (HR state 1e-8 DN, MSE trace 1e-10), against the tile kernels, in place, and
    a batch against its items: full phase grids (byte mosaic + 0/1 count masks), non-integer frames (float64 mosaic), frames sharing a
    phase (count plane), a half grid (nothing above the image)."""
    from oracle import sr_oracle as O
    S.set_precision("f64")
    try:
        f, shifts, integer_lr = PATCH_CFGS[cfg]
        psf, lr, saa = _patch_case(f, shifts, 2, integer_lr)
        O.set_threads(8)
        try:
            for n in (1, 2, 10, 80):
                hr, errs = S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5)
                assert S.last_path() == "stile"
                for i in range(2 if n < 80 else 1):
                    hr_o, err_o = O.ibp(list(lr[i]), shifts, psf, saa[i], f, n, 0.5)
                    close(hr[i].cpu().numpy(), hr_o, IBP_TOL["f64"])
                    np.testing.assert_allclose(errs[i].cpu().numpy(), err_o, rtol=ERR_RTOL["f64"])
        finally:
            O.set_threads(1)
        hr_t, e_t = S.ibp_batched(lr, shifts, psf, saa, f, 80, 0.5, flags=S.FLAG_TILES)
        assert S.last_path() == "mosaic"
        assert float((hr - hr_t).abs().max()) < 1e-8
        np.testing.assert_allclose(errs.cpu().numpy(), e_t.cpu().numpy(), rtol=1e-10)
        lr_d, buf = torch.from_numpy(lr).cuda(), torch.from_numpy(saa).cuda()
        hr2, errs2 = S.ibp_batched(lr_d, shifts, psf, buf, f, 80, 0.5, out=buf)
        assert hr2.data_ptr() == buf.data_ptr() and torch.equal(hr, hr2) and torch.equal(errs, errs2)
        one, e1 = S.ibp_batched(lr_d[1:2], shifts, psf, torch.from_numpy(saa[1:2]).cuda(), f, 80, 0.5)
        assert torch.equal(one[0], hr[1]) and torch.equal(e1[0], errs[1])
    finally:
        S.set_precision("f32")


def test_strip_kernels_f64_chunks_and_traceless_calls():
    """The float64 strip path runs a batch in chunks of 128 patches: 130 patches (a full chunk and a short one) against single-item calls of
    items of both chunks, bit for bit (state and MSE trace); want_errors=False gives the same state; zero iterations copy the input."""
    S.set_precision("f64")
    try:
        f, shifts = 4, synth.phase_shifts(4)
        psf = synth.gaussian_psf()
        rng = np.random.default_rng(991)
        lr = torch.from_numpy(np.rint(rng.uniform(0, 255, (130, 16, 64, 64)))).cuda()
        lr[7] = lr[7] * 0.5 + 0.25  # one patch with non-integer samples: the float64 mosaic among byte mosaics, in the first chunk
        init = S.shift_and_add_batched(lr, shifts, f)
        hr, errs = S.ibp_batched(lr, shifts, psf, init, f, 3, 0.5)
        assert S.last_path() == "stile"
        for i in (0, 7, 127, 128, 129):
            one, e1 = S.ibp_batched(lr[i:i + 1], shifts, psf, init[i:i + 1], f, 3, 0.5)
            assert torch.equal(one[0], hr[i]) and torch.equal(e1[0], errs[i]), i
        hr_n, none = S.ibp_batched(lr, shifts, psf, init, f, 3, 0.5, want_errors=False)
        assert none is None and torch.equal(hr_n, hr)
        hr_b, errs_b = S.ibp_batched(lr[:3], shifts, psf, init[:3], f, 3, 0.5, exact_workspace=False)  # the arena sized by the shape-only bound
        assert S.last_path() == "stile" and torch.equal(hr_b, hr[:3]) and torch.equal(errs_b, errs[:3])
        hr_0, _ = S.ibp_batched(lr[:2], shifts, psf, init[:2], f, 0, 0.5)
        assert torch.equal(hr_0, init[:2])
    finally:
        S.set_precision("f32")


def test_patch_tables_from_the_lr_frames_equal_the_plane_route():
    """Round 4: on a full phase grid the patch path builds its operand planes straight from the LR frames (k_patch_build -- the byte plane of every patch, then the
    float plane of those with a sample that is not an 8-bit integer --, k_patch_near_build) instead of through the batch's M / C / Mu planes (k_mosaic_build, k_patch_prep, k_patch_near_m: kept, on request,
    as the cross-check).  Same tables, so the same bits: integer frames (byte mosaic), non-integer frames (float mosaic), a mixed batch,
    a 3 x 4 sub-grid, x2."""
    S.set_precision("f32")
    psf = synth.gaussian_psf()
    for f, shifts in ((4, synth.phase_shifts(4)), (4, [s for s in synth.phase_shifts(4) if s[0] > -0.3]), (2, synth.phase_shifts(2))):
        _, lr, saa = _patch_case(f, shifts, 3)
        lr_d, saa_d = torch.from_numpy(lr).cuda().float(), torch.from_numpy(saa).cuda().float()
        lr_d[1] = lr_d[1] * 0.75 + 0.3  # item 1: non-integer samples
        lr_d[2, 3, 5, 7] += 0.5         # item 2: ONE non-integer sample
        a, ea = S.ibp_batched(lr_d, shifts, psf, saa_d, f, 7, 0.5)
        assert S.last_path() == "patch"
        b, eb = S.ibp_batched(lr_d, shifts, psf, saa_d, f, 7, 0.5, flags=S.FLAG_DIAG_NO_ZERO_FUSE)
        assert S.last_path() == "patch"
        assert torch.equal(a, b) and torch.equal(ea, eb)


# ---------------------------------------------------------------------------------------------------------
# k_ibp_dtile: frames with a common sub-pixel fraction > 0 on overlapping register-resident windows (srx_dtile.hpp)
# ---------------------------------------------------------------------------------------------------------
_PH4 = synth.phase_shifts(4)
DTILE_CFGS = {
    # name: (factor, shifts, (h, w) LR, flags, integer frames)
    "x4_ph16_narrow": (4, _PH4, (80, 100), 0, True),                      # 2 x 4 windows of 4 x 3 waves, byte mosaic, 0/1 count masks
    "x4_ph16_wide": (4, _PH4, (80, 100), "wide", True),                   # 2 x 2 windows of 4 x 4 waves
    "x4_ph16_float": (4, _PH4, (80, 100), "wide", False),                 # fractional samples: the float mosaic
    "x4_ph16_3x3win": (4, _PH4, (160, 176), 0, True),                     # interior windows (no image edge on any side)
    "x2_ph4": (2, synth.phase_shifts(2), (150, 232), 0, True),            # n in {-1, 0}: nothing above / left of the image
    "x2_ph4_w240": (2, synth.phase_shifts(2), (128, 120), 0, True),       # HR 256 x 240: ONE 4 x 3-wave window; the arena sized by the shape-only bound
    "x4_n01": (4, [s for s in _PH4 if s[0] > 0 and s[1] > 0], (80, 100), 0, True),       # n in {0, 1}: samples above the image, no near band inside
    "x4_sub12": (4, [s for s in _PH4 if s[0] > -0.3], (80, 100), "wide", True),           # 3 x 4 product grid
    "x4_lattice": (4, [_PH4[0], _PH4[5], _PH4[6], _PH4[6], _PH4[15]], (64, 112), 0, False),  # two frames on one phase: the count plane
    # round 4: PSFs that are not rank 1 (srx_patch.hpp's 7 x 7 form on the windows)
    "x4_ph16_asym": (4, _PH4, (80, 100), 0, True, "asym"),                # 5 x 5 core (the reference's measured PSF has that support)
    "x4_ph16_3x3win_asym": (4, _PH4, (160, 176), 0, False, "asym"),       # interior windows, float mosaic
    "x2_ph4_asym": (2, synth.phase_shifts(2), (150, 232), 0, True, "asym"),
    "x4_lattice_asym": (4, [_PH4[0], _PH4[5], _PH4[6], _PH4[6], _PH4[15]], (64, 112), 0, False, "asym"),  # count plane
}


@pytest.mark.parametrize("cfg", sorted(DTILE_CFGS))
def test_frame_fraction_kernel_vs_oracle(cfg):
    """The one-launch delta != 0 frame kernel against the oracle after 1, 2 and 6 iterations (state and MSE trace), against the
    tile kernels it replaces, in place, and as a batch."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    f, shifts, (h, w), fl, integer = DTILE_CFGS[cfg][:5]
    flags = S.FLAG_DIAG_WIDE_WINDOWS if fl == "wide" else S.FLAG_AUTO
    psf = {"gauss": synth.gaussian_psf(), "asym": synth.asymmetric_psf(), "full7": synth.full_support_psf()}[DTILE_CFGS[cfg][5] if len(DTILE_CFGS[cfg]) > 5 else "gauss"]
    O.set_threads(16)
    try:
        lrs, saas = [], []
        for i in range(2):
            truth = synth.truth_image(h * f, w * f, seed=500 + i)
            lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=600 + i)
            lr = lr if integer else lr * 0.75 + 0.3
            lrs.append(lr), saas.append(O.shift_and_add(list(lr), shifts, f))
        lr, saa = np.stack(lrs), np.stack(saas)
        for n in (1, 2, 6):
            hr, errs = S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5, flags=flags, exact_workspace=(n != 2))
            assert S.last_path() == "dtile"
            for i in range(2 if n == 6 else 1):
                hr_o, err_o = O.ibp(list(lr[i]), shifts, psf, saa[i], f, n, 0.5)
                close(hr[i].cpu().numpy(), hr_o, IBP_TOL["f32"])
                np.testing.assert_allclose(errs[i].cpu().numpy(), err_o, rtol=ERR_RTOL["f32"])
    finally:
        O.set_threads(1)
    hr_t, e_t = S.ibp_batched(lr, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "mosaic"
    assert float((hr - hr_t).abs().max()) < 5e-4
    np.testing.assert_allclose(errs.cpu().numpy(), e_t.cpu().numpy(), rtol=2e-6)
    buf = torch.from_numpy(saa).cuda().float()
    hr2, errs2 = S.ibp_batched(lr, shifts, psf, buf, f, 6, 0.5, flags=flags, out=buf)
    assert hr2.data_ptr() == buf.data_ptr() and torch.equal(hr, hr2) and torch.equal(errs, errs2)
    one, e1 = S.ibp_batched(lr[1:2], shifts, psf, saa[1:2], f, 6, 0.5, flags=flags)
    assert torch.equal(one[0], hr[1]) and torch.equal(e1[0], errs[1])


@pytest.mark.parametrize("shape", [(80, 100), (160, 176)])
def test_frame_fraction_kernel_80_iterations(shape):
    """80 iterations of the x4 / 16-phase workload on a frame of several windows -- 2 x 4 edge windows, and 3 x 5 with windows that
    touch no image edge -- against the oracle (the north-star tolerances)."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    f, shifts, psf, (h, w) = 4, _PH4, synth.gaussian_psf(), shape
    O.set_threads(16)
    try:
        truth = synth.truth_image(h * f, w * f, seed=71)
        lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=72)
        saa_o = O.shift_and_add(list(lr), shifts, f)
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, 80, 0.5)
    finally:
        O.set_threads(1)
    saa = S.shift_and_add(list(lr), shifts, f)
    close(saa, saa_o, PRIM_TOL["f32"])
    hr, errs = S.ibp(list(lr), shifts, psf, saa_o, f, 80, 0.5, verbose=False)
    assert S.last_path() == "dtile"
    close(hr, hr_o, IBP_TOL["f32"])
    np.testing.assert_allclose(errs, err_o, rtol=ERR_RTOL["f32"])
    assert synth.psnr(hr, hr_o) > 90.0 and abs(synth.psnr(hr, truth) - synth.psnr(hr_o, truth)) < 0.01
    u8_close(hr, hr_o)


def test_full_size_x4_frame_paths_agree():
    """SURVEY 8d's C3-f4 at full size (768 x 1024 -> 3072 x 4096, all 16 phases): the window kernel and the tile kernels agree, noise-free
    frames of x keep x, repeated calls are bit-identical, and a batch of two equals its items."""
    S.set_precision("f32")
    f, shifts, psf = 4, _PH4, synth.gaussian_psf()
    x = torch.from_numpy(synth.truth_image(384, 512, seed=25)).cuda().float().repeat(8, 8)[None].contiguous()
    lr = torch.stack([S.forward_model_batched(x, psf, s, f) for s in shifts], dim=1).contiguous()
    assert lr.shape == (1, 16, 768, 1024)
    hr, errs = S.ibp_batched(lr, shifts, psf, x, f, 3, 0.5)
    assert S.last_path() == "dtile"
    assert float((hr - x).abs().max()) < 2e-3 and float(errs.max()) < 1e-6
    gen = torch.Generator(device="cuda")
    gen.manual_seed(8)
    lrn = torch.clamp(torch.round(lr + 2.0 * torch.randn(lr.shape, generator=gen, device="cuda")), 0, 255)
    saa = S.shift_and_add_batched(lrn, shifts, f)
    hr_d, e_d = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5)
    assert S.last_path() == "dtile"
    hr_t, e_t = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "mosaic"
    assert float((hr_d - hr_t).abs().max()) < 5e-4
    np.testing.assert_allclose(e_d.cpu().numpy(), e_t.cpu().numpy(), rtol=2e-6)
    assert float(e_d[0, -1]) < float(e_d[0, 0])
    for _ in range(3):
        hr_r, e_r = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5)
        assert torch.equal(hr_r, hr_d) and torch.equal(e_r, e_d)
    two_lr, two_saa = torch.cat([lrn, torch.flip(lrn, dims=(2,))]), None
    two_saa = S.shift_and_add_batched(two_lr, shifts, f)
    hr2, e2 = S.ibp_batched(two_lr, shifts, psf, two_saa, f, 6, 0.5)
    assert torch.equal(hr2[0], hr_d[0]) and torch.equal(e2[0], e_d[0])  # (the window plan does not depend on the batch)
    hr_w, e_w = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_DIAG_WIDE_WINDOWS)
    assert float((hr_w - hr_d).abs().max()) < 5e-4  # the two window shapes agree to the warm-up truncation
    hr2w, e2w = S.ibp_batched(two_lr, shifts, psf, two_saa, f, 6, 0.5, flags=S.FLAG_DIAG_WIDE_WINDOWS)
    assert torch.equal(hr2w[0], hr_w[0]) and torch.equal(e2w[0], e_w[0])


_FIVE_WIDE = [(0.3, -1.2), (1.7, 0.45), (-1.9, 1.99), (0.0, 0.25), (-0.6, -0.6)]
BTILE_CFGS = {
    # name: (shifts (LR px), (h, w) LR, PSF)
    "meas4": (synth.MEASURED_4, (150, 277), "gauss"),   # the reference's rgb_cal_target shifts: 4 x 6 windows, ragged last windows
    "five_wide": (_FIVE_WIDE, (70, 83), "gauss"),  # odd N (a half-empty pair), |2 s| up to 4, an integer one
    "tiny": (synth.MEASURED_4, (20, 33), "gauss"),      # one window; the image ends inside its first block row
    "smallest": (synth.MEASURED_4, (16, 16), "gauss"),  # 32 x 32 HR, the smallest frame the kernels take
    "two_frames": ([(0.37, -0.21), (-0.12, 0.45)], (64, 96), "gauss"),  # one pair
    # a PSF that is not rank 1 (rgb_cal_target --psf measured, rgb_cal_target/run_sr.py:128-166): the 7 x 7 along registers and lanes
    "meas4_asym": (synth.MEASURED_4, (150, 277), "asym"),
    "five_wide_asym": (_FIVE_WIDE, (70, 83), "asym"),
    "tiny_asym": (synth.MEASURED_4, (20, 33), "asym"),
    "smallest_asym5": (synth.MEASURED_4, (16, 16), "asym5"),  # a 5 x 5 kernel embedded in the 7 x 7
    "meas4_gauss_as_7x7": (synth.MEASURED_4, (90, 140), "gauss7x7"),  # the Gaussian through the 7 x 7 form (SRX_FLAG_DIAG_NO_SEPARABLE)
    "meas4_full7": (synth.MEASURED_4, (90, 140), "full7"),  # not rank 1 AND non-zero out to the 7 x 7's corners (the asymmetric ones have a 5 x 5 core)
}


def _btile_psf(kind):
    a = synth.asymmetric_psf()
    return {"gauss": synth.gaussian_psf(), "gauss7x7": synth.gaussian_psf(), "asym": a, "asym5": a[1:6, 1:6] / a[1:6, 1:6].sum(),
            "full7": 0.6 * synth.gaussian_psf() + 0.4 * a}[kind]


@pytest.mark.parametrize("cfg", sorted(BTILE_CFGS))
def test_frame_shift_kernel_vs_oracle(cfg):
    """k_ibp_bfwd / k_ibp_bbwd (per-frame fractional shifts at x2, the reference's rgb_cal_target, on register-resident windows)
    against the oracle after 1, 2 and 6 iterations (state and MSE trace), against the tile kernels they replace, in place, as a
    batch, and run to run."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    shifts, (h, w), kind = BTILE_CFGS[cfg]
    f, psf = 2, _btile_psf(kind)
    fl = S.FLAG_DIAG_NO_SEPARABLE if kind == "gauss7x7" else S.FLAG_AUTO
    O.set_threads(16)
    try:
        lrs, saas = [], []
        for i in range(2):
            truth = synth.truth_image(h * f, w * f, seed=800 + i)
            lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=900 + i)
            lrs.append(lr), saas.append(O.shift_and_add(list(lr), shifts, f))
        lr, saa = np.stack(lrs), np.stack(saas)
        for n in (1, 2, 6):
            hr, errs = S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5, flags=fl)
            assert S.last_path() == "btile"
            for i in range(2 if n == 6 else 1):
                hr_o, err_o = O.ibp(list(lr[i]), shifts, psf, saa[i], f, n, 0.5)
                close(hr[i].cpu().numpy(), hr_o, IBP_TOL["f32"])
                np.testing.assert_allclose(errs[i].cpu().numpy(), err_o, rtol=ERR_RTOL["f32"])
    finally:
        O.set_threads(1)
    hr_t, e_t = S.ibp_batched(lr, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "fused"
    assert float((hr - hr_t).abs().max()) < 5e-4
    np.testing.assert_allclose(errs.cpu().numpy(), e_t.cpu().numpy(), rtol=2e-6)
    if kind == "gauss7x7":  # the same PSF through the 7 + 7 form of the same kernels
        hr_s, e_s = S.ibp_batched(lr, shifts, psf, saa, f, 6, 0.5)
        assert S.last_path() == "btile" and float((hr - hr_s).abs().max()) < 2e-4
        np.testing.assert_allclose(errs.cpu().numpy(), e_s.cpu().numpy(), rtol=2e-6)
    buf = torch.from_numpy(saa).cuda().float()
    hr2, errs2 = S.ibp_batched(lr, shifts, psf, buf, f, 6, 0.5, out=buf, flags=fl)
    assert hr2.data_ptr() == buf.data_ptr() and torch.equal(hr, hr2) and torch.equal(errs, errs2)
    one, e1 = S.ibp_batched(lr[1:2], shifts, psf, saa[1:2], f, 6, 0.5, flags=fl)
    assert torch.equal(one[0], hr[1]) and torch.equal(e1[0], errs[1])
    for _ in range(3):  # run to run
        hr3, errs3 = S.ibp_batched(lr, shifts, psf, saa, f, 6, 0.5, flags=fl)
        assert torch.equal(hr, hr3) and torch.equal(errs, errs3)


@pytest.mark.parametrize("kind", ["gauss", "asym"])
def test_frame_shift_kernel_80_iterations(kind):
    """The reference's rgb_cal_target (four measured shifts; its default Gaussian PSF and a PSF that is not rank 1, --psf measured;
    rgb_cal_target/run_sr.py:128-166, 171-192, 340-373) for its 50 and for 80 iterations on a frame of several windows, against the oracle
    at the north-star tolerances."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    f, shifts, psf, (h, w) = 2, synth.MEASURED_4, _btile_psf(kind), (110, 150)
    O.set_threads(16)
    try:
        truth = synth.truth_image(h * f, w * f, seed=81)
        lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=82)
        saa_o = O.shift_and_add(list(lr), shifts, f)
        refs = {n: O.ibp(list(lr), shifts, psf, saa_o, f, n, 0.5) for n in (50, 80)}
    finally:
        O.set_threads(1)
    for n, (hr_o, err_o) in refs.items():
        hr, errs = S.ibp(list(lr), shifts, psf, saa_o, f, n, 0.5, verbose=False)
        assert S.last_path() == "btile"
        close(hr, hr_o, IBP_TOL["f32"])
        np.testing.assert_allclose(errs, err_o, rtol=ERR_RTOL["f32"])
        assert synth.psnr(hr, hr_o) > 90.0 and abs(synth.psnr(hr, truth) - synth.psnr(hr_o, truth)) < 0.01
        u8_close(hr, hr_o)


def test_full_size_rgb_frame_paths_agree():
    """The reference's rgb_cal_target shape at full size (768 x 1024 -> 1536 x 2048, N = 4 measured shifts): the window kernels and the
    tile kernels agree, noise-free frames of x keep x, repeated calls are bit-identical, and a batch equals its items."""
    S.set_precision("f32")
    f, shifts, psf = 2, synth.MEASURED_4, synth.gaussian_psf()
    x = torch.from_numpy(synth.truth_image(384, 512, seed=26)).cuda().float().repeat(4, 4)[None].contiguous()
    lr = torch.stack([S.forward_model_batched(x, psf, s, f) for s in shifts], dim=1).contiguous()
    assert lr.shape == (1, 4, 768, 1024)
    hr, errs = S.ibp_batched(lr, shifts, psf, x, f, 3, 0.5)
    assert S.last_path() == "btile"
    assert float((hr - x).abs().max()) < 2e-3 and float(errs.max()) < 1e-6
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    lrn = torch.clamp(torch.round(lr + 2.0 * torch.randn(lr.shape, generator=gen, device="cuda")), 0, 255)
    saa = S.shift_and_add_batched(lrn, shifts, f)
    hr_d, e_d = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5)
    assert S.last_path() == "btile"
    hr_t, e_t = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "fused"
    assert float((hr_d - hr_t).abs().max()) < 5e-4
    np.testing.assert_allclose(e_d.cpu().numpy(), e_t.cpu().numpy(), rtol=2e-6)
    assert float(e_d[0, -1]) < float(e_d[0, 0])
    for _ in range(3):
        hr_r, e_r = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5)
        assert torch.equal(hr_r, hr_d) and torch.equal(e_r, e_d)
    two_lr = torch.cat([lrn, torch.flip(lrn, dims=(2,))])
    two_saa = S.shift_and_add_batched(two_lr, shifts, f)
    hr2, e2 = S.ibp_batched(two_lr, shifts, psf, two_saa, f, 6, 0.5)
    assert torch.equal(hr2[0], hr_d[0]) and torch.equal(e2[0], e_d[0])


ATILE_CFGS = {
    # name: (factor, shifts, (h, w) LR, flags)
    "x4_ph16_small": (4, _PH4, (40, 50), 0),                     # 160 x 200 HR: a frame k_ibp_dtile does not take (two-launch kernels by default)
    "x4_ph16_forced": (4, _PH4, (80, 100), "two"),                # ... and one it does, on request
    "x2_ph4_ragged": (2, synth.phase_shifts(2), (90, 131), 0),   # W = 262: not a multiple of 16
    "x4_lattice": (4, [_PH4[0], _PH4[5], _PH4[6], _PH4[6], _PH4[15]], (64, 77), 0),   # two frames on one phase (counts of 2), W = 308
    "x2_mixed": (2, [(0.5, 0.25), (-0.5, -0.25), (0.0, 0.75)], (90, 120), 0),           # y integer, x fraction 0.5
    "x3_frac": (3, [(0.1, 0.4), (0.1 + 1.0 / 3, 0.4 - 2.0 / 3), (0.1 - 1.0 / 3, 0.4 + 1.0 / 3)], (50, 66), 0),  # x3, fractions 0.3 / 0.2
}


@pytest.mark.parametrize("cfg", sorted(ATILE_CFGS))
def test_two_launch_window_kernels_vs_oracle(cfg):
    """k_ibp_afwd / k_atile_near / k_ibp_abwd (frames with a common fraction, two launches per iteration on 2 x 2-wave windows) against the
    oracle after 1, 2 and 6 iterations (state and MSE trace), against the tile kernels, in place, as a batch."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    f, shifts, (h, w), fl = ATILE_CFGS[cfg]
    flags = S.FLAG_DIAG_TWO_LAUNCH if fl == "two" else S.FLAG_AUTO
    psf = synth.gaussian_psf()
    O.set_threads(16)
    try:
        lrs, saas = [], []
        for i in range(2):
            truth = synth.truth_image(h * f, w * f, seed=520 + i)
            lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=620 + i)
            lrs.append(lr), saas.append(O.shift_and_add(list(lr), shifts, f))
        lr, saa = np.stack(lrs), np.stack(saas)
        for n in (1, 2, 6):
            hr, errs = S.ibp_batched(lr, shifts, psf, saa, f, n, 0.5, flags=flags)
            assert S.last_path() == "atile"
            for i in range(2 if n == 6 else 1):
                hr_o, err_o = O.ibp(list(lr[i]), shifts, psf, saa[i], f, n, 0.5)
                close(hr[i].cpu().numpy(), hr_o, IBP_TOL["f32"])
                np.testing.assert_allclose(errs[i].cpu().numpy(), err_o, rtol=ERR_RTOL["f32"])
    finally:
        O.set_threads(1)
    hr_t, e_t = S.ibp_batched(lr, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_TILES)
    assert S.last_path() == "mosaic"
    assert float((hr - hr_t).abs().max()) < 5e-4
    np.testing.assert_allclose(errs.cpu().numpy(), e_t.cpu().numpy(), rtol=2e-6)
    buf = torch.from_numpy(saa).cuda().float()
    hr2, errs2 = S.ibp_batched(lr, shifts, psf, buf, f, 6, 0.5, flags=flags, out=buf)
    assert hr2.data_ptr() == buf.data_ptr() and torch.equal(hr, hr2) and torch.equal(errs, errs2)
    one, e1 = S.ibp_batched(lr[1:2], shifts, psf, saa[1:2], f, 6, 0.5, flags=flags)
    assert torch.equal(one[0], hr[1]) and torch.equal(e1[0], errs[1])


def test_two_launch_window_kernels_80_iterations_and_full_size():
    """80 iterations on a frame of several windows against the oracle; SURVEY 8d's C3-f4 at full size against k_ibp_dtile."""
    from oracle import sr_oracle as O
    S.set_precision("f32")
    f, shifts, psf, (h, w) = 4, _PH4, synth.gaussian_psf(), (60, 77)
    O.set_threads(16)
    try:
        truth = synth.truth_image(h * f, w * f, seed=73)
        lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=74)
        saa_o = O.shift_and_add(list(lr), shifts, f)
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, 80, 0.5)
    finally:
        O.set_threads(1)
    hr, errs = S.ibp(list(lr), shifts, psf, saa_o, f, 80, 0.5, verbose=False)
    assert S.last_path() == "atile"
    close(hr, hr_o, IBP_TOL["f32"])
    np.testing.assert_allclose(errs, err_o, rtol=ERR_RTOL["f32"])
    assert synth.psnr(hr, hr_o) > 90.0 and abs(synth.psnr(hr, truth) - synth.psnr(hr_o, truth)) < 0.01
    u8_close(hr, hr_o)
    x = torch.from_numpy(synth.truth_image(384, 512, seed=25)).cuda().float().repeat(8, 8)[None].contiguous()
    lrf = torch.stack([S.forward_model_batched(x, psf, s, f) for s in shifts], dim=1).contiguous()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(8)
    lrn = torch.clamp(torch.round(lrf + 2.0 * torch.randn(lrf.shape, generator=gen, device="cuda")), 0, 255)
    saa = S.shift_and_add_batched(lrn, shifts, f)
    hr_d, e_d = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5)
    assert S.last_path() == "dtile"
    hr_a, e_a = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_DIAG_TWO_LAUNCH)
    assert S.last_path() == "atile"
    assert float((hr_d - hr_a).abs().max()) < 5e-4
    np.testing.assert_allclose(e_d.cpu().numpy(), e_a.cpu().numpy(), rtol=2e-6)
    for _ in range(2):
        hr_r, e_r = S.ibp_batched(lrn, shifts, psf, saa, f, 6, 0.5, flags=S.FLAG_DIAG_TWO_LAUNCH)
        assert torch.equal(hr_r, hr_a) and torch.equal(e_r, e_a)


# ---------------------------------------------------------------------------------------------------------
# Round 4: REFERENCE-generated goldens for the window kernels (tools/make_golden.py --only-windows)
# ---------------------------------------------------------------------------------------------------------
WIN_RGB_PATH = {"g": "btile", "m": "btile"}  # the implementation each PSF of rgb_cal_target must reach in float32


@pytest.mark.parametrize("psf_tag", ["g", "m"])
def test_window_golden_rgb_crop(prec, psf_tag):
    """What rgb_cal_target actually feeds the path-B window kernels: rep-averaged (non-integer) red frames of its committed session
    through extract_red + rep mean, its measured shifts, its ibp for its 50 iterations with the default Gaussian PSF and with
    --psf measured (rgb_cal_target/run_sr.py:59, :78-113, :128-166, :204-223), generated by the reference itself."""
    from conftest import load_golden
    g = load_golden("win_btile.npz")
    S.set_precision("f64")  # loader arithmetic bit-exactly in float64
    lr = np.stack([S.mean_frames(np.stack([S.extract_red(r.astype(np.float64)) for r in reps])) for reps in g["raw"]])
    S.set_precision(prec)
    assert lr.shape == (4, 64, 96) and np.abs(lr - np.rint(lr)).max() > 0.1  # non-integer frames
    sh, psf, init = g["shifts"], g[f"psf_{psf_tag}"], g["saa"].astype(np.float64)
    close(S.shift_and_add(list(lr), sh, 2), init, PRIM_TOL[prec] + 2e-5)  # stored as float32
    hr, errs = S.ibp(list(lr), sh, psf, init, 2, 50, 0.5, verbose=False)
    assert S.last_path() == (WIN_RGB_PATH[psf_tag] if prec == "f32" else "fused")
    close(hr, g[f"ibp50_{psf_tag}"], IBP_TOL["f32"])
    np.testing.assert_allclose(errs, g[f"errors_{psf_tag}"], rtol=ERR_RTOL[prec])
    u8_close(hr, g[f"ibp50_{psf_tag}"].astype(np.float64))
    if psf_tag == "g":
        hr1, e1 = S.ibp(list(lr), sh, psf, init, 2, 1, 0.5, verbose=False)
        close(hr1, g["ibp1_g"], IBP_TOL["f32"])


@pytest.mark.parametrize("name", ["win_dtile", "win_dtile_float", "win_atile"])
def test_window_golden_phase_grids(name):
    """x4 / 16 phases through the reference's ibp for 80 iterations: 288 x 320 HR (2 x 2 windows of k_ibp_dtile, byte mosaic;
    half-integer frames: its float mosaic) and 160 x 200 HR (the two-launch window kernels)."""
    from conftest import load_golden
    g = load_golden(name + ".npz")
    S.set_precision("f32")
    want = "atile" if name == "win_atile" else "dtile"
    lr = g["lr16"].astype(np.float64) if "lr16" in g else 0.5 * (g["lr16_a"].astype(np.float64) + g["lr16_b"].astype(np.float64))
    sh, psf, init = g["shifts16"], g["psf_g"], g["saa16"].astype(np.float64)
    close(S.shift_and_add(list(lr), sh, 4), init, PRIM_TOL["f32"])
    for n in (1, 10, 80):
        if f"ibp16_{n}" not in g:
            continue
        hr, errs = S.ibp(list(lr), sh, psf, init, 4, n, 0.5, verbose=False)
        assert S.last_path() == want
        close(hr, g[f"ibp16_{n}"], IBP_TOL["f32"])
        np.testing.assert_allclose(errs, g["ibp16_errors"][:n], rtol=ERR_RTOL["f32"])
    u8_close(hr, g["ibp16_80"].astype(np.float64))
    truth = synth.truth_image(init.shape[0], init.shape[1], seed=int(g["truth_seed"]))
    ref = g["ibp16_80"].astype(np.float64)
    assert synth.psnr(hr, ref) > 90.0 and abs(synth.psnr(hr, truth) - synth.psnr(ref, truth)) < 0.01


# ---------------------------------------------------------------------------------------------------------
# Parity holes named by the round-1 review
# ---------------------------------------------------------------------------------------------------------
def test_make_gaussian_psf_matches_reference(g_c1):
    """sr_mi355x.make_gaussian_psf (what run_sr uses; mono_cal_target/run_sr.py:104-111) against the reference's own PSF."""
    k = np.asarray(S.make_gaussian_psf())
    assert k.dtype == np.float64 and k.shape == (7, 7) and np.abs(k - g_c1["psf_g"]).max() < 1e-17


@pytest.mark.parametrize("cfg", ["mosaic_half", "per_frame"])
def test_80_iterations_multi_tile(prec, cfg):
    """80 iterations on a non-square image spanning several tiles, against the oracle: the mosaic tile kernels with a common
    fraction of 1/2 (in-tile prefilter with truncated warm-ups), and the per-frame fused path on measured shifts."""
    from oracle import sr_oracle as O
    O.set_threads(16)
    try:
        if cfg == "mosaic_half":
            f, shifts, psf, (h, w), want, flags = 2, synth.phase_shifts(2), synth.gaussian_psf(), (75, 139), "mosaic", S.FLAG_AUTO
        else:
            f, shifts, psf, (h, w), want, flags = 2, synth.MEASURED_4, synth.asymmetric_psf(), (83, 131), "fused", S.FLAG_PER_FRAME
        truth = synth.truth_image(h * f, w * f, seed=91)
        lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=92)
        saa_o = O.shift_and_add(list(lr), shifts, f)
        hr_o, err_o = O.ibp(list(lr), shifts, psf, saa_o, f, 80, 0.5)
    finally:
        O.set_threads(1)
    hr, errs = S.ibp_batched(lr[None], shifts, psf, saa_o[None], f, 80, 0.5, flags=flags)
    # (float32: the two-launch window kernels for the rank-1 PSF at a common fraction, the path-B window kernels' 7 x 7 form on measured shifts)
    assert S.last_path() == (("atile" if want == "mosaic" else "btile") if prec == "f32" else want)
    close(hr[0].cpu().numpy(), hr_o, IBP_TOL[prec])
    np.testing.assert_allclose(errs[0].cpu().numpy(), err_o, rtol=ERR_RTOL[prec])


def test_full_frame_batch_c4_share():
    """C4's single-GPU share: a batch of B = 4 full 3072 x 4096 frames (mono_cal_target's shape) in one call equals the frames
    reconstructed one by one, bit for bit; noise-free frames of x keep x (fixed point)."""
    S.set_precision("f32")
    f, shifts, psf = 2, synth.NOMINAL_5, synth.gaussian_psf()
    base = torch.from_numpy(synth.truth_image(384, 512, seed=15)).cuda().float().repeat(8, 8)
    x = torch.stack([torch.roll(base, shifts=(37 * i, 91 * i), dims=(0, 1)) for i in range(4)]).contiguous()
    lr = torch.stack([S.forward_model_batched(x, psf, s, f) for s in shifts], dim=1).contiguous()
    assert lr.shape == (4, 5, 1536, 2048)
    hr, errs = S.ibp_batched(lr, shifts, psf, x, f, 3, 0.5)
    assert S.last_path() == "ztile"
    assert float((hr - x).abs().max()) < 2e-3 and float(errs.max()) < 1e-6
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    lrn = torch.clamp(torch.round(lr + 2.0 * torch.randn(lr.shape, generator=gen, device="cuda")), 0, 255)
    saa = S.shift_and_add_batched(lrn, shifts, f)
    hr4, e4 = S.ibp_batched(lrn, shifts, psf, saa, f, 4, 0.5)
    for i in (0, 3):
        s1 = S.shift_and_add_batched(lrn[i:i + 1], shifts, f)
        assert torch.equal(s1[0], saa[i])
        h1, e1 = S.ibp_batched(lrn[i:i + 1], shifts, psf, s1, f, 4, 0.5)
        assert torch.equal(h1[0], hr4[i]) and torch.equal(e1[0], e4[i])
