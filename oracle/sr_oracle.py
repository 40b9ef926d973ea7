"""ctypes front-end of the CPU oracle (oracle/sr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  Function names and
argument meaning follow the reference's SR core, mono_cal_target/run_sr.py:157-209
(blur, forward_model, back_project, shift_and_add, ibp) plus the two SciPy calls it
makes directly (ndi_zoom :279, ndi_shift :163).  All arrays float64, C-contiguous.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsr_oracle.so")
_lib = None

_D = ctypes.POINTER(ctypes.c_double)
_L = ctypes.c_long


def build(force=False):
    src = os.path.join(_HERE, "sr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def set_threads(n):
    """Number of OpenMP threads the oracle uses (1 = the reference's behaviour)."""
    omp = ctypes.CDLL("libgomp.so.1")
    omp.omp_set_num_threads(int(n))


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _p(x):
    return x.ctypes.data_as(_D)


def spline_filter(a, mode="mirror"):
    a = _a(a).copy()
    lib().orc_spline_filter2d(_p(a), _L(a.shape[0]), _L(a.shape[1]), 0 if mode == "mirror" else 1)
    return a


def ndi_shift(a, shift_yx):
    a = _a(a)
    out = np.empty_like(a)
    rc = lib().orc_shift(_p(a), _L(a.shape[0]), _L(a.shape[1]), ctypes.c_double(shift_yx[0]),
                         ctypes.c_double(shift_yx[1]), _p(out))
    assert rc == 0
    return out


def ndi_zoom(a, factor):
    a = _a(a)
    Ho, Wo = int(round(a.shape[0] * factor)), int(round(a.shape[1] * factor))
    out = np.empty((Ho, Wo))
    rc = lib().orc_zoom(_p(a), _L(a.shape[0]), _L(a.shape[1]), _L(Ho), _L(Wo), _p(out))
    assert rc == 0
    return out


def blur(img, kernel):
    img, kernel = _a(img), _a(kernel)
    out = np.empty_like(img)
    lib().orc_blur(_p(img), _L(img.shape[0]), _L(img.shape[1]), _p(kernel), _L(kernel.shape[0]),
                   _L(kernel.shape[1]), _p(out))
    return out


def forward_model(hr, kernel, shift_yx, factor):
    hr, kernel = _a(hr), _a(kernel)
    H, W = hr.shape
    out = np.empty((-(-H // factor), -(-W // factor)))
    rc = lib().orc_forward_model(_p(hr), _L(H), _L(W), _p(kernel), _L(kernel.shape[0]), _L(kernel.shape[1]),
                                 ctypes.c_double(shift_yx[0]), ctypes.c_double(shift_yx[1]), _L(factor), _p(out))
    assert rc == 0
    return out


def back_project(error_lr, kernel, shift_yx, factor, hr_shape):
    e, kernel = _a(error_lr), _a(kernel)
    H, W = hr_shape
    out = np.empty((H, W))
    rc = lib().orc_back_project(_p(e), _L(e.shape[0]), _L(e.shape[1]), _p(kernel), _L(kernel.shape[0]),
                                _L(kernel.shape[1]), ctypes.c_double(shift_yx[0]), ctypes.c_double(shift_yx[1]),
                                _L(factor), _L(H), _L(W), _p(out))
    assert rc == 0
    return out


def shift_and_add(lr_list, shifts_yx, factor=2, order=3):
    assert order == 3
    lr = _a(np.stack([np.asarray(x, dtype=np.float64) for x in lr_list]))
    sh = _a(np.asarray(shifts_yx, dtype=np.float64).reshape(-1, 2))
    N, h, w = lr.shape
    out = np.empty((h * factor, w * factor))
    rc = lib().orc_shift_and_add(_p(lr), _L(N), _L(h), _L(w), _p(sh), _L(factor), _p(out))
    assert rc == 0
    return out


def ibp(lr_list, shifts_yx, kernel, hr_init, factor=2, n_iter=80, step=0.5):
    lr = _a(np.stack([np.asarray(x, dtype=np.float64) for x in lr_list]))
    sh = _a(np.asarray(shifts_yx, dtype=np.float64).reshape(-1, 2))
    kernel, hr_init = _a(kernel), _a(hr_init)
    N, h, w = lr.shape
    H, W = hr_init.shape
    hr = np.empty((H, W))
    errors = np.empty(n_iter)
    rc = lib().orc_ibp(_p(lr), _L(N), _L(h), _L(w), _p(sh), _p(kernel), _L(kernel.shape[0]), _L(kernel.shape[1]),
                       _p(hr_init), _L(H), _L(W), _L(factor), _L(n_iter), ctypes.c_double(step), _p(hr), _p(errors))
    assert rc == 0
    return hr, list(errors)


def quantize_u8(x):
    x = _a(x)
    out = np.empty(x.shape, dtype=np.uint8)
    lib().orc_quantize_u8(_p(x), _L(x.size), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out


def decimate(img, f, py=0, px=0):
    img = _a(img)
    H, W = img.shape
    out = np.empty((-(-(H - py) // f), -(-(W - px) // f)))
    lib().orc_decimate(_p(img), _L(H), _L(W), _L(f), _L(py), _L(px), _p(out))
    return out


def extract_red(img):
    return decimate(img, 2, 0, 0)


def mean0(stack):
    stack = _a(stack)
    R = stack.shape[0]
    out = np.empty(stack.shape[1:])
    lib().orc_mean0(_p(stack), _L(R), _L(out.size), _p(out))
    return out


def make_gaussian_psf(size=7, sigma=1.0):
    """mono_cal_target/run_sr.py:104-111."""
    hw = size // 2
    y, x = np.mgrid[-hw:hw + 1, -hw:hw + 1].astype(np.float64)
    k = np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def interleave4(frames_u8):
    """numpy restatement of the vendor GUI's 4-frame interleave (opt_materials/software/XPR_Software.py:196-205,
    388-410).  OpenCV is not installed in the build container, so this op is pinned by documented semantics, not by a
    reference run: it follows cv2.warpAffine for an integer translation (dst(x, y) = src(x - tx, y - ty)) with
    BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba) and np.sum(..., dtype=np.uint8) (modulo 256), and is checked against the
    hand-derived known answer tests/golden/interleave4_3x3.npz (tools/make_interleave_fixture.py)."""
    fr = np.asarray(frames_u8, dtype=np.uint8)
    _, h, w = fr.shape
    H, W = 2 * h, 2 * w
    shifts = [(0, 0), (0, 1), (-1, 1), (-1, 0)]  # (tx, ty) of M0..M3

    def r101(i, n):
        i = np.asarray(i)
        if n == 1:
            return np.zeros_like(i)
        for _ in range(4):
            i = np.where(i < 0, -i, i)
            i = np.where(i >= n, 2 * (n - 1) - i, i)
        return i

    acc = np.zeros((H, W), dtype=np.uint32)
    for k, (tx, ty) in enumerate(shifts):
        plane = np.zeros((H, W), dtype=np.uint8)
        plane[::2, ::2] = fr[k]
        yy = r101(np.arange(H) - ty, H)
        xx = r101(np.arange(W) - tx, W)
        acc += plane[np.ix_(yy, xx)]
    return (acc % 256).astype(np.uint8)
