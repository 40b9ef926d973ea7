"""CPU-side checks of the drop-in boundary: libsrx.so loads without a GPU and exports every
symbol include/srx.h declares; the ctypes table and the header agree."""
import os
import re

from sr_mi355x import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "srx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(srx_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_loads():
    _lib.build()
    lib = _lib.load()
    assert lib.srx_version() >= 100
    assert lib.srx_strerror(0) == b"ok"
    assert lib.srx_strerror(-3).startswith(b"workspace")


def test_every_declared_symbol_is_exported():
    _lib.build()
    lib = _lib.load()
    hdr = header_symbols()
    assert len(hdr) >= 30
    for name in hdr:
        assert hasattr(lib, name), f"{name} declared in include/srx.h but not exported by libsrx.so"
    assert sorted(_lib.symbols()) == hdr, "ctypes table and header disagree"


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    a = lib.srx_ibp_workspace_bytes(4, 1, 4, 32, 32, 64, 64, 2, _lib.FLAG_COMPOSED)
    b = lib.srx_ibp_workspace_bytes(8, 1, 4, 32, 32, 64, 64, 2, _lib.FLAG_COMPOSED)
    assert 0 < a < b
    assert lib.srx_shift_workspace_bytes(4, 2, 64, 64) > 2 * 88 * 88 * 4
    assert lib.srx_saa_workspace_bytes(4, 1, 4, 32, 32, 2) > 0


def test_argument_validation_without_gpu():
    """Invalid arguments are rejected before any HIP call."""
    lib = _lib.load()
    assert lib.srx_blur_f32(None, 1, 8, 8, None, 7, 7, None, None) == _lib.E_INVALID
    assert lib.srx_u8_to_f32(None, 0, None, None) == _lib.E_INVALID
    assert lib.srx_ibp_f64(None, 1, 4, 8, 8, None, None, 7, 7, None, 16, 16, 2, 1, 0.5, None, None, None, 0, None,
                           0) == _lib.E_INVALID


def test_oversized_planes_are_refused_without_gpu():
    """One image plane of 2 GiB or more (32-bit offsets inside a plane, include/srx.h) is SRX_E_UNSUPPORTED before any HIP call;
    the pointers are never dereferenced."""
    import ctypes
    import numpy as np
    lib = _lib.load()
    dp = ctypes.POINTER(ctypes.c_double)
    fake = ctypes.c_void_p(4096)
    k = np.full((7, 7), 1.0 / 49)
    sh = np.zeros((4, 2))
    kp, sp = k.ctypes.data_as(dp), sh.ctypes.data_as(dp)
    H = W = 1 << 15   # 32768 x 32768 float32 = 4 GiB, float64 = 8 GiB
    assert lib.srx_blur_f32(fake, 1, H, W, kp, 7, 7, fake, None) == _lib.E_UNSUPPORTED
    assert lib.srx_shift_cubic_f64(fake, 1, H, W, 0.5, 0.5, fake, fake, 1 << 20, None) == _lib.E_UNSUPPORTED
    assert lib.srx_zoom_cubic_f32(fake, 1, H // 2, W // 2, 2, fake, fake, 1 << 20, None) == _lib.E_UNSUPPORTED
    assert lib.srx_forward_f32(fake, 1, H, W, kp, 7, 7, 0.5, 0.5, 2, fake, fake, 1 << 20, None) == _lib.E_UNSUPPORTED
    assert lib.srx_backproject_f32(fake, 1, H // 2, W // 2, kp, 7, 7, 0.5, 0.5, 2, H, W, fake, fake, 1 << 20, None) == _lib.E_UNSUPPORTED
    assert lib.srx_saa_f32(fake, 1, 4, H // 2, W // 2, sp, 2, fake, fake, 1 << 20, None, 0) == _lib.E_UNSUPPORTED
    for fn in (lib.srx_ibp_f32, lib.srx_ibp_f64):
        assert fn(fake, 1, 4, H // 2, W // 2, sp, kp, 7, 7, fake, H, W, 2, 1, 0.5, fake, None, fake, 1 << 20, None, 0) == _lib.E_UNSUPPORTED
    # just past the limit (23200^2 x 4 B = 2.15 GB); the float64 planes of the reference's largest image are far below it
    assert lib.srx_blur_f32(fake, 1, 23200, 23200, kp, 7, 7, fake, None) == _lib.E_UNSUPPORTED
    assert lib.srx_ibp_workspace_bytes(8, 1, 5, 1536, 2048, 3072, 4096, 2, 0) > 0


def test_exact_workspace_never_exceeds_the_shape_bound():
    """srx_ibp_workspace_bytes (shape only) must cover srx_ibp_workspace_bytes_for (shifts + PSF at hand) whatever
    implementation the call picks: a caller that sizes its arena once by the bound may not see SRX_E_WORKSPACE.
    (Round 3 shipped a bound that forgot k_ibp_dtile's 4 x 3-wave windows: HR widths 192..255.)"""
    import ctypes
    import numpy as np
    from sr_mi355x import synth
    lib = _lib.load()
    dp = ctypes.POINTER(ctypes.c_double)
    psfs = [synth.gaussian_psf(), synth.asymmetric_psf(), synth.full_support_psf()]   # rank 1 / 5 x 5 core / full 7 x 7: three kernel forms
    worst = 0.0
    for f, shift_sets in ((2, (synth.NOMINAL_4, synth.NOMINAL_5, synth.MEASURED_4, synth.phase_shifts(2))),
                          (3, (synth.phase_shifts(3),)), (4, (synth.phase_shifts(4), synth.NOMINAL_4))):
        for shifts in shift_sets:
            sh = np.ascontiguousarray(np.asarray(shifts, dtype=np.float64))
            N = len(shifts)
            for H in (32, 64, 96, 128, 132, 192, 252, 256, 260, 320, 512, 1024):
                for W in (32, 64, 128, 176, 192, 208, 224, 240, 256, 272, 512, 1040):
                    if H % f or W % f:
                        continue
                    for psf in psfs:
                        k = np.ascontiguousarray(psf)
                        for eb in (4, 8):
                            for B in (1, 3, 130):   # 130: past one chunk of the float64 strip kernels (128 patches)
                                bound = lib.srx_ibp_workspace_bytes(eb, B, N, H // f, W // f, H, W, f, 0)
                                need = lib.srx_ibp_workspace_bytes_for(eb, B, N, H // f, W // f, H, W, f, sh.ctypes.data_as(dp),
                                                                       k.ctypes.data_as(dp), 7, 7, 0)
                                assert 0 < need <= bound, (f, N, H, W, eb, B, need, bound)
                                worst = max(worst, need / bound)
    assert worst <= 1.0
