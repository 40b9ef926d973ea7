"""ctypes binding of libsrx.so (include/srx.h).  No torch types cross this boundary:
device pointers go through as integers, small parameter arrays as host float64 buffers.

The product path has no CPU fallback: if libsrx.so is missing, load() raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SRX_LIB", os.path.join(_HERE, "libsrx.so"))  # SRX_LIB: diagnostic builds only
_lib = None

OK, E_INVALID, E_UNSUPPORTED, E_WORKSPACE, E_HIP = 0, -1, -2, -3, -4
FLAG_AUTO, FLAG_COMPOSED, FLAG_FUSED, FLAG_PER_FRAME, FLAG_TILES = 0, 1, 2, 4, 8
# diagnostic path switches (include/srx.h): implementations the tests hold to the same results
FLAG_DIAG_NO_ZERO_FUSE, FLAG_DIAG_NO_SEPARABLE, FLAG_DIAG_NO_PREFILTER_TILE, FLAG_DIAG_V1 = 0x100, 0x200, 0x400, 0x800
FLAG_DIAG_WIDE_WINDOWS, FLAG_DIAG_COLUMN_TILES, FLAG_DIAG_TWO_LAUNCH, FLAG_DIAG_SAA_ONE_PASS = 0x1000, 0x2000, 0x4000, 0x8000

_c = ctypes
_P, _I, _D, _Z, _U = _c.c_void_p, _c.c_int, _c.c_double, _c.c_size_t, _c.c_uint
_HD = _c.POINTER(_c.c_double)  # host float64 array

# name -> (restype, argtypes) for every symbol include/srx.h declares; {T} expands to f32 and f64
_TYPED = {
    "srx_blur_{T}": (_I, [_P, _I, _I, _I, _HD, _I, _I, _P, _P]),
    "srx_shift_cubic_{T}": (_I, [_P, _I, _I, _I, _D, _D, _P, _P, _Z, _P]),
    "srx_zoom_cubic_{T}": (_I, [_P, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "srx_forward_{T}": (_I, [_P, _I, _I, _I, _HD, _I, _I, _D, _D, _I, _P, _P, _Z, _P]),
    "srx_backproject_{T}": (_I, [_P, _I, _I, _I, _HD, _I, _I, _D, _D, _I, _I, _I, _P, _P, _Z, _P]),
    "srx_saa_{T}": (_I, [_P, _I, _I, _I, _I, _HD, _I, _P, _P, _Z, _P, _U]),
    "srx_ibp_{T}": (_I, [_P, _I, _I, _I, _I, _HD, _HD, _I, _I, _P, _I, _I, _I, _I, _D, _P, _P, _P, _Z, _P, _U]),
    "srx_decimate_{T}": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "srx_zero_insert_{T}": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "srx_mean_frames_{T}": (_I, [_P, _I, _I, _Z, _P, _P]),
    "srx_u8_to_{T}": (_I, [_P, _Z, _P, _P]),
    "srx_quantize_u8_{T}": (_I, [_P, _Z, _P, _P]),
    "srx_ibp_plan_create_{T}": (_I, [_P, _I, _I, _I, _I, _HD, _HD, _I, _I, _P, _I, _I, _I, _D, _I, _I, _P, _Z, _P, _U, _c.POINTER(_P)]),
    "srx_ibp_plan_get_rows_{T}": (_I, [_P, _I, _I, _P, _P]),
    "srx_ibp_plan_set_rows_{T}": (_I, [_P, _I, _I, _P, _P]),
    "srx_pair_moments_{T}": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "srx_local_contrast_{T}": (_I, [_P, _I, _I, _I, _P, _P]),
    "srx_ring_sums_{T}": (_I, [_P, _I, _I, _D, _D, _I, _P, _P, _Z, _P]),
    "srx_spot_moments_{T}": (_I, [_P, _I, _I, _P, _P]),
    "srx_edge_bins_{T}": (_I, [_P, _I, _I, _D, _D, _D, _I, _D, _D, _I, _P, _P, _Z, _P]),
}
_PLAIN = {
    "srx_version": (_I, []),
    "srx_strerror": (_c.c_char_p, [_I]),
    "srx_last_path": (_c.c_char_p, []),
    "srx_profile_enable": (None, [_I]),
    "srx_profile_kernel_count": (_I, []),
    "srx_profile_kernel_name": (_c.c_char_p, [_I]),
    "srx_profile_get": (_I, [_I, _c.POINTER(_c.c_double), _c.POINTER(_c.c_long)]),
    "srx_interleave4_u8": (_I, [_P, _I, _I, _I, _P, _P]),
    "srx_shift_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "srx_zoom_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "srx_forward_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "srx_backproject_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "srx_saa_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "srx_ibp_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _U]),
    "srx_ibp_workspace_bytes_for": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _HD, _HD, _I, _I, _U]),
    "srx_ibp_plan_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _U]),
    "srx_ibp_plan_run": (_I, [_P, _I, _P, _P]),
    "srx_ibp_plan_path": (_c.c_char_p, [_P]),
    "srx_ibp_plan_supports_trace_rows": (_I, [_P]),
    "srx_ibp_plan_destroy": (None, [_P]),
    "srx_metrics_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "srx_edge_magnitude_f64": (_I, [_P, _I, _I, _D, _P, _P, _Z, _P]),
    "srx_edge_dist_range": (_I, [_I, _I, _D, _D, _D, _I, _P, _P]),
}


def symbols():
    """Every symbol name include/srx.h declares."""
    names = list(_PLAIN)
    for pat in _TYPED:
        names += [pat.format(T="f32"), pat.format(T="f64")]
    return names


def build(force=False):
    """hipcc --offload-arch=gfx950 ... -> sr_mi355x/libsrx.so (cross-compiles without a GPU)."""
    pkg_root = os.path.dirname(_HERE)
    if force and os.path.exists(SO_PATH):
        os.remove(SO_PATH)
    subprocess.check_call(["make", "-s", "-C", pkg_root])
    return SO_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} not found: build it with `make -C {os.path.dirname(_HERE)}` "
                           "(sr_mi355x has no CPU fallback)")
    # torch first: PyTorch-ROCm ships its own libamdhip64 and libsrx.so must bind to THAT runtime (the one that owns the
    # caller's device pointers and streams).  Loaded the other way round, libsrx.so pulls in /opt/rocm's copy, the process
    # holds two HIP runtimes and every launch on a torch pointer fails with hipErrorInvalidValue.
    import torch  # noqa: F401
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in _PLAIN.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for pat, (res, args) in _TYPED.items():
        for t in ("f32", "f64"):
            fn = getattr(lib, pat.format(T=t))
            fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class SrxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = load().srx_strerror(status).decode()
        super().__init__(f"{where}: {msg} (status {status})")


def check(status, where):
    if status != OK:
        raise SrxError(status, where)
