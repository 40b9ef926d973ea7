"""Host-side mirror of the reference's SR core on top of libsrx.so (HIP, gfx950).

Same names, argument order and meaning as the module-level functions of the reference's
drivers (mono_cal_target/run_sr.py:157-209; identical in the other three run_sr.py):

    blur(img, kernel)                                              :157-158
    forward_model(hr, kernel, shift_yx, factor)                    :161-165
    back_project(error_lr, kernel, shift_yx, factor, hr_shape)     :168-178
    shift_and_add(lr_list, shifts_yx, factor=2, order=3)           :181-187
    ibp(lr_list, shifts_yx, kernel, hr_init, factor=2, n_iter=80, step=0.5) -> (hr, errors)   :190-209
    ndi_zoom(img, factor, order=3) / ndi_shift(img, shift, order=3, mode='nearest')  (the two SciPy
        calls the drivers make directly, :279 and :163)

so `from sr_mi355x import ibp, shift_and_add, ...` (or rebinding those names in a run_sr
module) makes the reference's own process_session run on the MI355X.  numpy in -> float64
numpy out (like the reference); torch CUDA tensors in -> torch tensors out, no host copy.
`*_batched` variants take a leading batch of independent work items (patches, sessions x reps).

Compute precision: 'f32' (default; HBM-bound fast path, |delta| ~ 2e-4 DN vs float64) or
'f64' (the reference's own precision, ~1e-10 DN).  torch is only plumbing here (device
memory + streams); every FLOP runs in libsrx.so, and there is no CPU fallback.
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib
from ._lib import (FLAG_AUTO, FLAG_COMPOSED, FLAG_FUSED, FLAG_PER_FRAME, FLAG_TILES, FLAG_DIAG_NO_ZERO_FUSE,  # noqa: F401  (re-exported)
                   FLAG_DIAG_NO_SEPARABLE, FLAG_DIAG_NO_PREFILTER_TILE, FLAG_DIAG_V1, FLAG_DIAG_WIDE_WINDOWS, FLAG_DIAG_COLUMN_TILES,
                   FLAG_DIAG_TWO_LAUNCH, FLAG_DIAG_SAA_ONE_PASS)

_TORCH_DT = {"f32": torch.float32, "f64": torch.float64}
_ELEM = {"f32": 4, "f64": 8}
# The working precision is a per-THREAD setting with a process-wide default: the session driver's loader thread works in float64
# (precision_override) while the main thread reconstructs in float32, and neither may see the other's choice.
_DEFAULT_PRECISION = "f32"
_tls = threading.local()


def set_precision(p):
    """'f32' or 'f64': element type of every device buffer and of the arithmetic.  Sets the process default (what every thread
    uses unless it is inside a precision_override) and drops the calling thread's override, if any."""
    global _DEFAULT_PRECISION
    if p not in _TORCH_DT:
        raise ValueError("precision must be 'f32' or 'f64'")
    _DEFAULT_PRECISION = p
    _tls.p = None


def get_precision():
    return getattr(_tls, "p", None) or _DEFAULT_PRECISION


class precision_override:
    """with precision_override('f64'): ... -- the calling thread's precision inside the block, other threads untouched."""

    def __init__(self, p):
        if p not in _TORCH_DT:
            raise ValueError("precision must be 'f32' or 'f64'")
        self.p = p

    def __enter__(self):
        self.prev = getattr(_tls, "p", None)
        _tls.p = self.p

    def __exit__(self, *exc):
        _tls.p = self.prev


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("sr_mi355x needs an MI355X (HIP device); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _to_dev(x, prec):
    """-> (contiguous CUDA tensor of the compute dtype, was_numpy)"""
    dt = _TORCH_DT[prec]
    if isinstance(x, torch.Tensor):
        return x.to(device=_device(), dtype=dt).contiguous(), False
    a = np.ascontiguousarray(np.asarray(x))
    if a.dtype != np.uint8:
        a = a.astype(np.float64, copy=False)
    return torch.from_numpy(a).to(device=_device()).to(dt).contiguous(), True


def _stack_dev(lst, prec):
    if isinstance(lst, torch.Tensor):
        return _to_dev(lst, prec)
    if len(lst) and isinstance(lst[0], torch.Tensor):
        return torch.stack([t.to(device=_device(), dtype=_TORCH_DT[prec]) for t in lst]).contiguous(), False
    return _to_dev(np.stack([np.asarray(a) for a in lst]), prec)


def _out(t, was_numpy):
    return t.to(torch.float64).cpu().numpy() if was_numpy else t


def _host_f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a, a.ctypes.data_as(_lib._HD)


def _ws(nbytes):
    t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=_device())
    return t, ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(t.numel())


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _fn(name, prec):
    return getattr(_lib.load(), f"{name}_{prec}")


def last_path():
    """Which code path the last shift_and_add/ibp call took: 'fused' or 'composed'."""
    return _lib.load().srx_last_path().decode()


# ------------------------------------------------------------------------------------------
# batched primitives: tensors [B, ...] on the device
# ------------------------------------------------------------------------------------------
def blur_batched(img, kernel, precision=None):
    prec = precision or get_precision()
    x, _ = _to_dev(img, prec)
    B, H, W = x.shape
    k, kp = _host_f64(kernel)
    out = torch.empty_like(x)
    _lib.check(_fn("srx_blur", prec)(_p(x), B, H, W, kp, k.shape[0], k.shape[1], _p(out), _stream()), "srx_blur")
    return out


def shift_batched(img, shift_yx, precision=None):
    prec = precision or get_precision()
    x, _ = _to_dev(img, prec)
    B, H, W = x.shape
    out = torch.empty_like(x)
    wt, wp, wn = _ws(_lib.load().srx_shift_workspace_bytes(_ELEM[prec], B, H, W))
    _lib.check(_fn("srx_shift_cubic", prec)(_p(x), B, H, W, float(shift_yx[0]), float(shift_yx[1]), _p(out), wp, wn,
                                            _stream()), "srx_shift_cubic")
    return out


def zoom_batched(img, factor, precision=None):
    prec = precision or get_precision()
    x, _ = _to_dev(img, prec)
    B, h, w = x.shape
    f = int(factor)
    if f != factor or f < 1:
        raise ValueError("zoom factor must be a positive integer")
    out = torch.empty((B, h * f, w * f), dtype=x.dtype, device=x.device)
    wt, wp, wn = _ws(_lib.load().srx_zoom_workspace_bytes(_ELEM[prec], B, h, w, f))
    _lib.check(_fn("srx_zoom_cubic", prec)(_p(x), B, h, w, f, _p(out), wp, wn, _stream()), "srx_zoom_cubic")
    return out


def forward_model_batched(hr, kernel, shift_yx, factor, precision=None):
    prec = precision or get_precision()
    x, _ = _to_dev(hr, prec)
    B, H, W = x.shape
    f = int(factor)
    k, kp = _host_f64(kernel)
    out = torch.empty((B, -(-H // f), -(-W // f)), dtype=x.dtype, device=x.device)
    wt, wp, wn = _ws(_lib.load().srx_forward_workspace_bytes(_ELEM[prec], B, H, W))
    _lib.check(_fn("srx_forward", prec)(_p(x), B, H, W, kp, k.shape[0], k.shape[1], float(shift_yx[0]),
                                        float(shift_yx[1]), f, _p(out), wp, wn, _stream()), "srx_forward")
    return out


def back_project_batched(error_lr, kernel, shift_yx, factor, hr_shape, precision=None):
    prec = precision or get_precision()
    e, _ = _to_dev(error_lr, prec)
    B, eh, ew = e.shape
    H, W = int(hr_shape[0]), int(hr_shape[1])
    k, kp = _host_f64(kernel)
    out = torch.empty((B, H, W), dtype=e.dtype, device=e.device)
    wt, wp, wn = _ws(_lib.load().srx_backproject_workspace_bytes(_ELEM[prec], B, H, W))
    _lib.check(_fn("srx_backproject", prec)(_p(e), B, eh, ew, kp, k.shape[0], k.shape[1], float(shift_yx[0]),
                                            float(shift_yx[1]), int(factor), H, W, _p(out), wp, wn, _stream()),
               "srx_backproject")
    return out


def shift_and_add_batched(lr, shifts_yx, factor=2, precision=None, flags=FLAG_AUTO):
    """lr [B, N, h, w] -> [B, h*f, w*f]."""
    prec = precision or get_precision()
    x, _ = _to_dev(lr, prec)
    B, N, h, w = x.shape
    f = int(factor)
    sh, shp = _host_f64(shifts_yx, (N, 2))
    out = torch.empty((B, h * f, w * f), dtype=x.dtype, device=x.device)
    wt, wp, wn = _ws(_lib.load().srx_saa_workspace_bytes(_ELEM[prec], B, N, h, w, f))
    _lib.check(_fn("srx_saa", prec)(_p(x), B, N, h, w, shp, f, _p(out), wp, wn, _stream(), flags), "srx_saa")
    return out


def ibp_batched(lr, shifts_yx, kernel, hr_init, factor=2, n_iter=80, step=0.5, precision=None, flags=FLAG_AUTO,
                want_errors=True, out=None, exact_workspace=True):
    """lr [B, N, h, w], hr_init [B, H, W] -> (hr [B, H, W], errors float64 [B, n_iter] or None).
    `exact_workspace=False` sizes the arena by the shape-only bound (srx_ibp_workspace_bytes), as a caller without the shift table
    at hand would."""
    prec = precision or get_precision()
    x, _ = _to_dev(lr, prec)
    h0, _ = _to_dev(hr_init, prec)
    B, N, h, w = x.shape
    Bh, H, W = h0.shape
    if Bh != B:
        raise ValueError("lr and hr_init disagree on the batch size")
    f = int(factor)
    sh, shp = _host_f64(shifts_yx, (N, 2))
    k, kp = _host_f64(kernel)
    if out is not None:  # the library writes B*H*W elements of the call's precision straight through this pointer
        if (not isinstance(out, torch.Tensor) or not out.is_cuda or out.dtype != _TORCH_DT[prec] or tuple(out.shape) != (B, H, W)
                or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous CUDA tensor of dtype {_TORCH_DT[prec]} and shape {(B, H, W)}")
    hr = torch.empty_like(h0) if out is None else out
    errors = torch.empty((B, int(n_iter)), dtype=torch.float64, device=x.device) if want_errors else None
    lib = _lib.load()
    wt, wp, wn = _ws(lib.srx_ibp_workspace_bytes_for(_ELEM[prec], B, N, h, w, H, W, f, shp, kp, k.shape[0], k.shape[1], flags)
                     if exact_workspace else lib.srx_ibp_workspace_bytes(_ELEM[prec], B, N, h, w, H, W, f, flags))
    _lib.check(_fn("srx_ibp", prec)(_p(x), B, N, h, w, shp, kp, k.shape[0], k.shape[1], _p(h0), H, W, f, int(n_iter),
                                    float(step), _p(hr), _p(errors) if want_errors else None, wp, wn, _stream(), flags),
               "srx_ibp")
    return hr, errors


class IbpPlan:
    """srx_ibp_plan_* (include/srx.h): the IBP loop of a batch in instalments -- the per-call tables built once, `run(n)` for n more
    iterations, rows of the state readable / replaceable in between.  `trace_rows=(lo, hi)`: the HR rows whose LR samples the MSE trace
    counts (a row band of a larger image counts its own rows only; `supports_trace_rows` says whether this plan can).  The tensors
    passed in and the workspace live as long as the plan."""

    def __init__(self, lr, shifts_yx, kernel, hr_init, factor=2, step=0.5, precision=None, flags=FLAG_AUTO, trace_rows=None):
        self.prec = precision or get_precision()
        self.lr, _ = _to_dev(lr, self.prec)
        h0, _ = _to_dev(hr_init, self.prec)
        self.B, self.N, self.h, self.w = self.lr.shape
        _, self.H, self.W = h0.shape
        self._sh = _host_f64(shifts_yx, (self.N, 2))
        self._k = _host_f64(kernel)
        lo, hi = trace_rows if trace_rows is not None else (0, self.H)
        lib = _lib.load()
        self._ws = _ws(lib.srx_ibp_plan_workspace_bytes(_ELEM[self.prec], self.B, self.N, self.h, self.w, self.H, self.W, int(factor), flags))
        self._h = ctypes.c_void_p()
        k = self._k[0]
        _lib.check(_fn("srx_ibp_plan_create", self.prec)(_p(self.lr), self.B, self.N, self.h, self.w, self._sh[1], self._k[1], k.shape[0], k.shape[1],
                                                         _p(h0), self.H, self.W, int(factor), float(step), int(lo), int(hi), self._ws[1], self._ws[2],
                                                         _stream(), flags, ctypes.byref(self._h)), "srx_ibp_plan_create")
        self.path = lib.srx_ibp_plan_path(self._h).decode()
        self.supports_trace_rows = bool(lib.srx_ibp_plan_supports_trace_rows(self._h))

    def run(self, n_iter, want_errors=True):
        """n more iterations -> errors float64 [B, n] on the device (this run's slice of the trace) or None"""
        errors = torch.empty((self.B, int(n_iter)), dtype=torch.float64, device=self.lr.device) if want_errors and n_iter > 0 else None
        _lib.check(_lib.load().srx_ibp_plan_run(self._h, int(n_iter), _p(errors) if errors is not None else None, _stream()), "srx_ibp_plan_run")
        return errors

    def get_rows(self, lo, hi, out=None):
        out = torch.empty((self.B, hi - lo, self.W), dtype=_TORCH_DT[self.prec], device=self.lr.device) if out is None else out
        _lib.check(_fn("srx_ibp_plan_get_rows", self.prec)(self._h, int(lo), int(hi), _p(out), _stream()), "srx_ibp_plan_get_rows")
        return out

    def set_rows(self, lo, hi, rows):
        rows = rows.to(device=self.lr.device, dtype=_TORCH_DT[self.prec]).contiguous()
        if tuple(rows.shape) != (self.B, hi - lo, self.W):
            raise ValueError(f"rows must have shape {(self.B, hi - lo, self.W)}")
        _lib.check(_fn("srx_ibp_plan_set_rows", self.prec)(self._h, int(lo), int(hi), _p(rows), _stream()), "srx_ibp_plan_set_rows")

    def result(self):
        return self.get_rows(0, self.H)

    def close(self):
        if self._h:
            _lib.load().srx_ibp_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# the reference's call surface (single image; numpy or torch)
# ------------------------------------------------------------------------------------------
def blur(img, kernel):
    x, was_np = _to_dev(img, get_precision())
    return _out(blur_batched(x[None], kernel)[0], was_np)


def ndi_shift(img, shift, order=3, mode="nearest"):
    if order != 3 or mode != "nearest":
        raise NotImplementedError("only order=3, mode='nearest' (the reference's call) is provided")
    x, was_np = _to_dev(img, get_precision())
    return _out(shift_batched(x[None], shift)[0], was_np)


def ndi_zoom(img, zoom, order=3):
    if order != 3:
        raise NotImplementedError("only order=3 (the reference's call) is provided")
    x, was_np = _to_dev(img, get_precision())
    return _out(zoom_batched(x[None], zoom)[0], was_np)


def forward_model(hr, kernel, shift_yx, factor):
    x, was_np = _to_dev(hr, get_precision())
    return _out(forward_model_batched(x[None], kernel, shift_yx, factor)[0], was_np)


def back_project(error_lr, kernel, shift_yx, factor, hr_shape):
    x, was_np = _to_dev(error_lr, get_precision())
    return _out(back_project_batched(x[None], kernel, shift_yx, factor, hr_shape)[0], was_np)


def shift_and_add(lr_list, shifts_yx, factor=2, order=3):
    if order != 3:
        raise NotImplementedError("only order=3 (the reference's call) is provided")
    x, was_np = _stack_dev(lr_list, get_precision())
    return _out(shift_and_add_batched(x[None], shifts_yx, factor)[0], was_np)


def ibp(lr_list, shifts_yx, kernel, hr_init, factor=2, n_iter=80, step=0.5, verbose=True):
    """Returns (hr, errors) like run_sr.py:190-209; `errors` is a list of n_iter floats."""
    x, was_np = _stack_dev(lr_list, get_precision())
    h0, was_np_h = _to_dev(hr_init, get_precision())
    hr, errs = ibp_batched(x[None], shifts_yx, kernel, h0[None], factor, n_iter, step)
    errors = [float(e) for e in errs[0].cpu().numpy()]
    if verbose:  # the reference prints the running MSE every 10 iterations (:207-208)
        for it in range(9, len(errors), 10):
            print(f"    iter {it + 1:3d}/{n_iter}  MSE = {errors[it]:.4f}")
    return _out(hr[0], was_np or was_np_h), errors


# ------------------------------------------------------------------------------------------
# driver glue (index maps, loaders' arithmetic, quantiser)
# ------------------------------------------------------------------------------------------
def decimate(img, f, py=0, px=0):
    """img[py::f, px::f] (run_sr.py:165)."""
    x, was_np = _to_dev(img, get_precision())
    H, W = x.shape
    out = torch.empty((-(-(H - py) // f), -(-(W - px) // f)), dtype=x.dtype, device=x.device)
    _lib.check(_fn("srx_decimate", get_precision())(_p(x), 1, H, W, int(f), int(py), int(px), _p(out), _stream()),
               "srx_decimate")
    return _out(out, was_np)


def extract_red(img):
    """Bayer RGGB red plane img[0::2, 0::2] (rgb_cal_target/run_sr.py:73-75)."""
    return decimate(img, 2, 0, 0)


def zero_insert(err, f, hr_shape):
    """up = zeros(hr_shape); up[::f, ::f] = err (pad/crop) (run_sr.py:170-175)."""
    x, was_np = _to_dev(err, get_precision())
    eh, ew = x.shape
    H, W = int(hr_shape[0]), int(hr_shape[1])
    out = torch.empty((H, W), dtype=x.dtype, device=x.device)
    _lib.check(_fn("srx_zero_insert", get_precision())(_p(x), 1, eh, ew, int(f), H, W, _p(out), _stream()),
               "srx_zero_insert")
    return _out(out, was_np)


def mean_frames(stack):
    """np.mean(stack, axis=0) (run_sr.py:274; rgb_cal_target/run_sr.py:107-108)."""
    x, was_np = _stack_dev(stack, get_precision())
    R = x.shape[0]
    n = x[0].numel()
    out = torch.empty(x.shape[1:], dtype=x.dtype, device=x.device)
    _lib.check(_fn("srx_mean_frames", get_precision())(_p(x), 1, R, n, _p(out), _stream()), "srx_mean_frames")
    return _out(out, was_np)


def mean_frames_batched(stacks):
    """[B, R, ...] -> [B, ...]: np.mean(axis=0) of every item's stack in one call."""
    x, was_np = _to_dev(stacks, get_precision())
    B, R = x.shape[0], x.shape[1]
    n = x[0, 0].numel()
    out = torch.empty((B,) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    _lib.check(_fn("srx_mean_frames", get_precision())(_p(x), B, R, n, _p(out), _stream()), "srx_mean_frames")
    return _out(out, was_np)


def quantize_u8(img):
    """np.clip(img, 0, 255).astype(np.uint8): clamp, then truncate (run_sr.py:303)."""
    x, was_np = _to_dev(img, get_precision())
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(_fn("srx_quantize_u8", get_precision())(_p(x), x.numel(), _p(out), _stream()), "srx_quantize_u8")
    return out.cpu().numpy() if was_np else out


def u8_to_float(img_u8, precision=None):
    """uint8 frame -> float on the device (load_gray's cast, run_sr.py:73-75)."""
    prec = precision or get_precision()
    t = img_u8 if isinstance(img_u8, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img_u8))
    t = t.to(device=_device(), dtype=torch.uint8).contiguous()
    out = torch.empty(t.shape, dtype=_TORCH_DT[prec], device=t.device)
    _lib.check(_fn("srx_u8_to", prec)(_p(t), t.numel(), _p(out), _stream()), "srx_u8_to")
    return out


def make_gaussian_psf(size=7, sigma=1.0):
    """Normalised 2-D Gaussian PSF (run_sr.py:104-111); host-side, float64."""
    hw = size // 2
    y, x = np.mgrid[-hw:hw + 1, -hw:hw + 1].astype(np.float64)
    k = np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def interleave4(frames_u8):
    """The vendor live view's 4-frame pixel interleave (XPR_Software.py:388-410): uint8 [4, h, w] (or [B, 4, h, w])
    -> uint8 [2h, 2w]; frame k lands on the HR lattice phase its half-pixel shift corresponds to."""
    t = frames_u8 if isinstance(frames_u8, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames_u8))
    was_np = not isinstance(frames_u8, torch.Tensor)
    t = t.to(device=_device(), dtype=torch.uint8).contiguous()
    single = t.dim() == 3
    if single:
        t = t[None]
    B, four, h, w = t.shape
    if four != 4:
        raise ValueError("interleave4 needs exactly 4 frames")
    out = torch.empty((B, 2 * h, 2 * w), dtype=torch.uint8, device=t.device)
    _lib.check(_lib.load().srx_interleave4_u8(_p(t), B, h, w, _p(out), _stream()), "srx_interleave4_u8")
    out = out[0] if single else out
    return out.cpu().numpy() if was_np else out
