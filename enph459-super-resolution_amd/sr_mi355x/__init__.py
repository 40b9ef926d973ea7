"""sr_mi355x -- MI355X-native multi-frame super-resolution core (libsrx.so + host shim).

Mirrors the call surface of the reference's SR core (mono_cal_target/run_sr.py:157-209):
blur, forward_model, back_project, shift_and_add, ibp (+ ndi_zoom / ndi_shift), see api.py.
Attributes are resolved lazily so that `sr_mi355x.synth` (numpy only) imports without torch.
"""
import importlib

_API = ("blur", "forward_model", "back_project", "shift_and_add", "ibp", "ndi_zoom", "ndi_shift",
        "blur_batched", "shift_batched", "zoom_batched", "forward_model_batched", "back_project_batched",
        "shift_and_add_batched", "ibp_batched", "decimate", "extract_red", "zero_insert", "mean_frames", "mean_frames_batched",
        "quantize_u8", "u8_to_float", "interleave4", "make_gaussian_psf", "set_precision", "get_precision", "precision_override", "last_path",
        "FLAG_AUTO", "FLAG_COMPOSED", "FLAG_FUSED", "FLAG_PER_FRAME", "FLAG_TILES", "FLAG_DIAG_NO_ZERO_FUSE",
        "FLAG_DIAG_NO_SEPARABLE", "FLAG_DIAG_NO_PREFILTER_TILE", "FLAG_DIAG_V1", "FLAG_DIAG_WIDE_WINDOWS", "FLAG_DIAG_COLUMN_TILES", "FLAG_DIAG_TWO_LAUNCH", "FLAG_DIAG_SAA_ONE_PASS")

__all__ = list(_API)


def __getattr__(name):
    if name in _API:
        return getattr(importlib.import_module(".api", __name__), name)
    if name in ("api", "synth", "_lib", "session", "parallel", "rowband", "metrics"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
