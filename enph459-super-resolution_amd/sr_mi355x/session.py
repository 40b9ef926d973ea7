"""Session driver: the counterpart of the reference's `process_session` / `process_combo` on the device path.

Reproduces, for the four experiment layouts of the reference, the file discovery rules, frame order, shift
tables, outputs and `done.flag` skip logic:

  kind            reference                                 input files                         LR frames
  --------------  ----------------------------------------  ----------------------------------  -----------------------------
  mono_cal_target mono_cal_target/run_sr.py:59-99,262-315   center.png, shift_0..3.png          5, nominal (0,0), (+-.5,+-.5)
  rgb_cal_target  rgb_cal_target/run_sr.py:63-113,276-333   corner{c}_rep{rr}.png+metadata.json 4 = red plane, mean over reps,
                                                                                                shifts = expected px / 2
  mono_barcodes   mono_barcodes/run_sr.py:71-130,293-351    corner{c}_rep{rr}.png               4 per rep, nominal +-0.5
  rgb_barcodes    rgb_barcodes/run_sr.py:78-143,306-364     corner{c}_rep{rr}.png               4 per rep (red), nominal +-0.5

PNG decode/encode stays on the host (PIL), everything between -- uint8 -> float, Bayer red extraction, rep
averaging, frame mean, Native-2x zoom, shift_and_add, ibp, clip+truncate to uint8 -- runs in libsrx on the GPU.
The reference's matplotlib figures (comparison.png, convergence.png) are not produced; the IBP MSE trace is
written as `convergence.json` instead.
"""
import json
import os
import re

import numpy as np

from . import api

UPSAMPLE_FACTOR = 2
PSF_SIZE, PSF_SIGMA, PSF_HALFWIDTH = 7, 1.0, 3
IBP_STEP_SIZE = 0.5
IBP_ITERATIONS = {"mono_cal_target": 80, "rgb_cal_target": 50, "mono_barcodes": 80, "rgb_barcodes": 80}

# mono_cal_target/run_sr.py:59-66
IMAGE_SHIFTS = [("center.png", (0.0, 0.0)), ("shift_0.png", (+0.5, -0.5)), ("shift_1.png", (+0.5, +0.5)),
                ("shift_2.png", (-0.5, -0.5)), ("shift_3.png", (-0.5, +0.5))]
# mono_barcodes/run_sr.py:71-76, rgb_barcodes/run_sr.py:78-83 (corner0..3)
CORNER_SHIFTS = [(+0.5, -0.5), (+0.5, +0.5), (-0.5, -0.5), (-0.5, +0.5)]
# rgb_cal_target/run_sr.py:63
CORNER_ORDER = ["(-x,+y)", "(+x,+y)", "(-x,-y)", "(+x,-y)"]


def _png_u8(path):
    from PIL import Image
    a = np.array(Image.open(path))
    if a.ndim == 3:  # load_gray: channel mean (run_sr.py:73-75)
        return a.astype(np.float64).mean(axis=2)
    return a


LOADER_PRECISION = "f64"  # uint8 -> float, Bayer extraction and the rep / frame means are done in the reference's
#                           own float64 (exact for uint8 data), whatever precision the SR kernels then run in: a mean
#                           of k/5 values rounded to float32 flips the truncating uint8 quantiser on ~0.5 % of pixels.


class _loader_precision:
    def __enter__(self):
        self.prev = api.get_precision()
        api.set_precision(LOADER_PRECISION)

    def __exit__(self, *exc):
        api.set_precision(self.prev)


def load_gray_dev(path):
    """PNG -> float64 image on the device (load_gray, mono_cal_target/run_sr.py:73-75)."""
    a = _png_u8(path)
    if a.dtype == np.uint8:
        return api.u8_to_float(a, precision=LOADER_PRECISION)
    return api._to_dev(a, LOADER_PRECISION)[0]


def detect_kind(session_dir):
    names = os.listdir(session_dir)
    if "center.png" in names:
        return "mono_cal_target"
    if any(re.match(r"corner\d+_rep\d+\.png", n) for n in names):
        return None  # caller must say which of the three corner layouts it is
    raise FileNotFoundError(f"no SR input images in {session_dir}")


def load_mono_cal_session(session_dir):
    """mono_cal_target/run_sr.py:78-99 -> (frames on device, shifts)."""
    frames, shifts = [], []
    for fname, s in IMAGE_SHIFTS:
        path = os.path.join(session_dir, fname)
        if not os.path.exists(path):
            continue
        frames.append(load_gray_dev(path))
        shifts.append(s)
    if len(frames) < 2:
        raise FileNotFoundError(f"Need at least 2 images in {session_dir}")
    return frames, shifts


def load_rgb_cal_combo(combo_dir):
    """rgb_cal_target/run_sr.py:78-113: red plane of every rep, mean over reps, shifts = metadata px / 2."""
    with open(os.path.join(combo_dir, "metadata.json")) as fp:
        meta = json.load(fp)

    def get_shift(label):
        if "expected_shifts" in meta:
            s = meta["expected_shifts"][label]
            return s["dy_px"] / 2.0, s["dx_px"] / 2.0
        if "corners" in meta:
            c = meta["corners"][label]
            return c["expected_dy_px"] / 2.0, c["expected_dx_px"] / 2.0
        raise KeyError(f"Cannot find shift for {label} in metadata")

    frames, shifts = [], []
    for idx, label in enumerate(CORNER_ORDER):
        reps = sorted(f for f in os.listdir(combo_dir) if f.startswith(f"corner{idx}_rep") and f.endswith(".png"))
        if not reps:
            raise FileNotFoundError(f"No images for corner{idx} in {combo_dir}")
        with _loader_precision():
            reds = [api.extract_red(load_gray_dev(os.path.join(combo_dir, r))) for r in reps]
            frames.append(api.mean_frames(reds))
        shifts.append(get_shift(label))
    return frames, shifts


def load_corner_reps(session_dir, red):
    """mono_barcodes/run_sr.py:89-130 / rgb_barcodes/run_sr.py:102-143 -> (list over reps of 4 frames, shifts)."""
    rep_indices = sorted({int(m.group(1)) for m in (re.match(r"corner\d+_rep(\d+)\.png", n) for n in os.listdir(session_dir))
                          if m})
    if not rep_indices:
        raise FileNotFoundError(f"No corner*_rep*.png files in {session_dir}")
    all_reps = []
    for ri in rep_indices:
        frames = []
        for ci in range(4):
            path = os.path.join(session_dir, f"corner{ci}_rep{ri:02d}.png")
            if not os.path.exists(path):
                raise FileNotFoundError(f"Missing {path}")
            img = load_gray_dev(path)
            with _loader_precision():
                frames.append(api.extract_red(img) if red else img)
        all_reps.append(frames)
    return all_reps, list(CORNER_SHIFTS)


def reconstruct(frames, shifts, psf_kernel, n_iter, factor=UPSAMPLE_FACTOR, step=IBP_STEP_SIZE):
    """Native-2x, SAA and SAA+IBP of one frame set, all on the device (run_sr.py:274-292).
    -> dict of float device tensors + the MSE trace."""
    import torch
    lr64 = torch.stack(frames)
    with _loader_precision():
        mean_lr = api.mean_frames(lr64)  # float64; also what LR_(red_)mean.png is quantised from
    lr = lr64.to(api._TORCH_DT[api.get_precision()])
    native = api.zoom_batched(mean_lr[None], factor)[0]
    saa = api.shift_and_add_batched(lr[None], shifts, factor)
    hr, errs = api.ibp_batched(lr[None], shifts, psf_kernel, saa.clone(), factor, n_iter, step)
    return {"native_2x": native, "SAA": saa[0], "SAA_IBP": hr[0], "LR_mean": mean_lr}, [float(e) for e in errs[0].cpu()]


def _save_outputs(out_dir, images, errors, lr_name, extra=None):
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    for name in ("native_2x", "SAA", "SAA_IBP"):
        Image.fromarray(api.quantize_u8(images[name]).cpu().numpy()).save(os.path.join(out_dir, f"{name}.png"))
    with _loader_precision():
        Image.fromarray(api.quantize_u8(images["LR_mean"]).cpu().numpy()).save(os.path.join(out_dir, lr_name))
    with open(os.path.join(out_dir, "convergence.json"), "w") as fp:
        json.dump({"ibp_mse": errors}, fp)
    if extra:
        for fname, obj in extra.items():
            with open(os.path.join(out_dir, fname), "w") as fp:
                json.dump(obj, fp, indent=2)
    open(os.path.join(out_dir, "done.flag"), "w").close()


def process_session(session_dir, psf_kernel, output_base, kind=None, n_iter=None, verbose=True):
    """Counterpart of process_session / process_combo.  Returns the list of output directories written
    (empty if everything was already done)."""
    kind = kind or detect_kind(session_dir)
    if kind not in IBP_ITERATIONS:
        raise ValueError("kind must be one of " + ", ".join(IBP_ITERATIONS))
    n_iter = IBP_ITERATIONS[kind] if n_iter is None else n_iter
    name = os.path.basename(os.path.normpath(session_dir))
    say = print if verbose else (lambda *a, **k: None)
    written = []
    if kind in ("mono_cal_target", "rgb_cal_target"):
        out_dir = os.path.join(output_base, name)
        if os.path.exists(os.path.join(out_dir, "done.flag")):
            say(f"  [skip] {name} - already done")
            return written
        if kind == "mono_cal_target":
            frames, shifts = load_mono_cal_session(session_dir)
            lr_name, extra = "LR_mean.png", None
        else:
            frames, shifts = load_rgb_cal_combo(session_dir)
            lr_name = "LR_red_mean.png"
            extra = {"shifts.json": {"shifts_lr_yx": [list(s) for s in shifts], "corner_labels": CORNER_ORDER}}
        images, errors = reconstruct(frames, shifts, psf_kernel, n_iter)
        _save_outputs(out_dir, images, errors, lr_name, extra)
        say(f"  Output: {out_dir}")
        written.append(out_dir)
        return written
    red = kind == "rgb_barcodes"
    all_reps, shifts = load_corner_reps(session_dir, red)
    for rep_idx, frames in enumerate(all_reps):
        out_dir = os.path.join(output_base, name, f"rep{rep_idx}")
        if os.path.exists(os.path.join(out_dir, "done.flag")):
            say(f"  [skip] rep {rep_idx} - already done")
            continue
        images, errors = reconstruct(frames, shifts, psf_kernel, n_iter)
        _save_outputs(out_dir, images, errors, "LR_red_mean.png" if red else "LR_mean.png")
        say(f"    Output: {out_dir}")
        written.append(out_dir)
    return written


def write_metrics(out_dir, factor=UPSAMPLE_FACTOR):
    """metrics.json for one mono_cal_target output directory: the reference's notebook summary (analysis.ipynb cells
    3-10) computed from the uint8 PNGs just written, as the notebook does."""
    import numpy as np
    from PIL import Image
    from . import metrics
    imgs = {n: np.array(Image.open(os.path.join(out_dir, n + ".png")), dtype=np.float64) for n in ("native_2x", "SAA", "SAA_IBP")}
    rep = metrics.cal_target_report(imgs, factor=factor)
    with open(os.path.join(out_dir, "metrics.json"), "w") as fp:
        json.dump(rep, fp, indent=2)
    return rep


def discover_sessions(data_dir, kind):
    """main()'s session discovery (mono_cal_target/run_sr.py:338-343 and the corner-layout variants)."""
    out = []
    for d in sorted(os.listdir(data_dir)):
        full = os.path.join(data_dir, d)
        if not os.path.isdir(full):
            continue
        names = os.listdir(full)
        if kind == "mono_cal_target":
            ok = "center.png" in names
        elif kind == "rgb_cal_target":
            ok = "metadata.json" in names
        else:
            ok = any(re.match(r"corner\d+_rep\d+\.png", n) for n in names)
        if ok:
            out.append(full)
    if not out:
        raise FileNotFoundError(f"No session folders found in {data_dir}")
    return out


# ---------------------------------------------------------------------------------------------------------
# measured PSF (load_measured_psf, mono_cal_target/run_sr.py:114-152): host side, once per run, tiny
# ---------------------------------------------------------------------------------------------------------
def psf_from_pinhole_images(images, halfwidth=PSF_HALFWIDTH):
    """images: iterable of 2-D arrays (pinhole frames).  Peak-aligned mean of +-(halfwidth+6) crops, central
    (2*halfwidth+1)^2 window, minus the mean of its four 3x3 corner blocks, clipped at 0, normalised to sum 1."""
    margin = halfwidth + 6
    patches = []
    for img in images:
        img = np.asarray(img, dtype=np.float64)
        pr, pc = np.unravel_index(img.argmax(), img.shape)
        R = margin
        if pr < R or pr + R + 1 > img.shape[0] or pc < R or pc + R + 1 > img.shape[1]:
            continue  # peak too close to the edge (the reference skips these)
        patches.append(img[pr - R:pr + R + 1, pc - R:pc + R + 1].copy())
    if not patches:
        raise FileNotFoundError("no usable pinhole image")
    avg = np.mean(patches, axis=0)
    R = margin
    k = avg[R - halfwidth:R + halfwidth + 1, R - halfwidth:R + halfwidth + 1].copy()
    corners = np.concatenate([k[:3, :3].ravel(), k[:3, -3:].ravel(), k[-3:, :3].ravel(), k[-3:, -3:].ravel()])
    k -= np.mean(corners)
    k = np.clip(k, 0, None)
    return k / k.sum()


def load_measured_psf(psf_dir):
    """Average the pos4_(0,0).png pinhole frames of every sweep directory (run_sr.py:114-152)."""
    imgs = []
    for sweep in sorted(os.listdir(psf_dir)):
        path = os.path.join(psf_dir, sweep, "pos4_(0,0).png")
        if os.path.isdir(os.path.join(psf_dir, sweep)) and os.path.exists(path):
            a = _png_u8(path)
            imgs.append(a.astype(np.float64))
    if not imgs:
        raise FileNotFoundError(f"No pos4_(0,0).png found under {psf_dir}")
    return psf_from_pinhole_images(imgs)
