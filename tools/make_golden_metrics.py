#!/usr/bin/env python3
"""Generate tests/golden/metrics.npz by running the REFERENCE's own quality-metric functions.

Build container only (needs /root/reference).  Nothing is copied: the function definitions of
mono_cal_target/analysis.ipynb (cells 4, 7, 10) are compiled from the notebook at run time, and
data_collection/psf_mtf_utils.py is imported by path.  Inputs are crops of the reference's committed result PNGs
(the notebook's own ROIs) and the measured PSF already held in tests/golden/synth_c1.npz; only arrays are written.

    python tools/make_golden_metrics.py
"""
import ast
import contextlib
import importlib.util
import io
import json
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SR_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
SESSION = os.path.join(REF, "mono_cal_target", "results", "cal_target_mono_tilt0.14128_settletime50ms")


def notebook_functions(path):
    """Namespace holding every top-level `def` of the notebook's code cells (no other statement is executed)."""
    nb = json.load(open(path))
    ns = {"np": np}
    for cell in nb["cells"]:
        if cell["cell_type"] != "code":
            continue
        src = "".join(line for line in cell["source"] if not line.lstrip().startswith("%"))
        tree = ast.parse(src)
        defs = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
        if defs:
            exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    nb = notebook_functions(os.path.join(REF, "mono_cal_target", "analysis.ipynb"))
    spec = importlib.util.spec_from_file_location("ref_psf_mtf_utils", os.path.join(REF, "data_collection", "psf_mtf_utils.py"))
    pm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pm)
    d = {}
    # the notebook's ROIs (analysis.ipynb cells 3 and 6)
    roi2 = (slice(1900, 2100), slice(2560, 2760))
    for name in ("native_2x", "SAA"):
        img = np.array(Image.open(os.path.join(SESSION, name + ".png")), dtype=np.float64)
        roi = img[roi2]
        prof = img[1240:1560, 2700]
        d[f"{name}_roi"] = roi.astype(np.uint8)
        d[f"{name}_profile"] = prof.astype(np.uint8)
        d[f"{name}_contrast16"] = nb["local_contrast"](prof, window=16)
        d[f"{name}_contrast20"] = nb["local_contrast"](prof)
        for side in ("left", "right"):
            with contextlib.redirect_stdout(io.StringIO()):
                ex, ey, ang = nb["slanted_edge_esf"](roi, side=side)
            fr, mtf, lsf = nb["esf_to_mtf"](ex, ey)
            pitch = 3.45e-3 / 2
            fc = fr / pitch
            v = fc > 0
            d[f"{name}_{side}_esf_x"], d[f"{name}_{side}_esf_y"], d[f"{name}_{side}_angle"] = ex, ey, np.float64(ang)
            d[f"{name}_{side}_freq"], d[f"{name}_{side}_mtf"], d[f"{name}_{side}_lsf"] = fr, mtf, lsf
            d[f"{name}_{side}_mtf50"] = np.float64(nb["mtf_at_fraction"](fc[v], mtf[v], 0.5))
            d[f"{name}_{side}_mtf10"] = np.float64(nb["mtf_at_fraction"](fc[v], mtf[v], 0.1))
    # PSF analytics (psf_mtf_utils.py:74-175) on the measured 7x7 PSF and on a 41x41 rotated elliptical spot
    psf_m = np.load(os.path.join(OUT, "synth_c1.npz"))["psf_m"]
    yy, xx = np.mgrid[:41, :41].astype(np.float64)
    rng = np.random.default_rng(462)
    spot = pm.gauss2d((xx, yy), 200.0, 20.6, 19.7, 2.4, 3.3, 0.4, 3.0).reshape(41, 41) + rng.normal(0, 0.5, (41, 41))
    spot = np.clip(spot, 0, None)
    d["spot"] = spot
    for name, p, pitch in (("psfm", psf_m, 3.45), ("spot", spot, None)):
        fr, prof, m2d, label, nyq = pm.compute_mtf(p, pixel_pitch_um=pitch)
        c = m2d.shape[0] // 2
        d[f"{name}_mtf_freq"], d[f"{name}_mtf_radial"], d[f"{name}_nyquist"] = fr, prof, np.float64(nyq)
        d[f"{name}_mtf_2d_centre"] = m2d[c - 16:c + 17, c - 16:c + 17]  # the full 256 x 256 map is 0.5 MB
        d[f"{name}_mtf50"] = np.float64(pm.mtf_at_fraction(fr, prof, 0.5))
        d[f"{name}_centre"] = np.array(pm.subpixel_centre(p))
    rad, prof = pm.radial_average(spot)
    d["spot_radial_r"], d["spot_radial"] = rad, prof
    popt, fit = pm.fit_gaussian_psf(spot)
    d["spot_fit_params"], d["spot_fit_image"] = popt, fit
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **d)
    print("wrote metrics.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items() if "mtf50" in k or "angle" in k})
    for k in sorted(d):
        if "mtf50" in k or "mtf10" in k or "angle" in k:
            print(f"  {k} = {float(d[k]):.6f}")


if __name__ == "__main__":
    main()
