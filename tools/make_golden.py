#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own functions.

Runs only in the build container (needs /root/reference).  The reference's run_sr.py
scripts are loaded by path with importlib -- nothing is copied -- and fed seeded
synthetic inputs (sr_mi355x.synth) and crops of the reference's committed real inputs.
Only arrays (inputs + expected outputs) are written; no reference source travels.

    python tools/make_golden.py                  # writes tests/golden/
    python tools/make_golden.py --only-frames    # only frame_zero.npz (round 3)
    python tools/make_golden.py --only-windows   # only win_btile / win_dtile / win_atile.npz (round 4)
"""
import contextlib
import glob
import importlib.util
import io
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SR_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
from sr_mi355x import synth  # noqa: E402


def load_ref(exp):
    spec = importlib.util.spec_from_file_location(f"ref_{exp}", os.path.join(REF, exp, "run_sr.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_lr(ref, truth, psf, shifts, f, seed):
    clean = np.stack([ref.forward_model(truth, psf, s, f) for s in shifts])
    return synth.sensor_frames(clean, seed=seed)


def ibp_trace(ref, lr, shifts, psf, init, f, its, step=0.5):
    out = {}
    for n in its:
        hr, errs = quiet(ref.ibp, list(lr), shifts, psf, init.copy(), factor=f, n_iter=n, step=step)
        out[n] = hr
    return out, np.asarray(errs)


def frame_goldens(mono, psf_g, psf_m):
    """delta = 0 cases LARGE enough for the one-launch frame kernel (k_ibp_ztile: H, W >= 128; tiles of 52 x 244 valid pixels, so
    144 x 280 gives 3 x 2 tiles, all ragged), run through the reference's ibp for the full 80 iterations
    (mono_cal_target/run_sr.py:190-209, shifts :59-66).  (a) seeded synthetic frames, N = 5 nominal, Gaussian PSF and the
    reference's measured (non-separable) PSF; (b) a 72 x 140 crop of the committed mono_cal_target frames."""
    f, h, w = 2, 72, 140
    nom5 = synth.NOMINAL_5
    truth = synth.truth_image(h * f, w * f, seed=synth.SEED_TRUTH + 7)
    lr = make_lr(mono, truth, psf_g, nom5, f, synth.SEED_NOISE + 7)
    d = dict(truth=truth.astype(np.float32), psf_g=psf_g, psf_m=psf_m, shifts5=np.array(nom5), lr5=lr.astype(np.uint8))
    saa = mono.shift_and_add(list(lr), nom5, factor=f, order=3)
    d["saa5"] = saa
    tr, errs = ibp_trace(mono, lr, nom5, psf_g, saa, f, (1, 10, 80))
    d["ibp5_1"], d["ibp5_10"], d["ibp5_80"] = tr[1].astype(np.float32), tr[10].astype(np.float32), tr[80]
    d["ibp5_errors"] = errs
    hr, errs = quiet(mono.ibp, list(lr), nom5, psf_m, saa.copy(), factor=f, n_iter=80, step=0.5)
    d["ibp5m_80"], d["ibp5m_errors"] = hr, np.asarray(errs)
    sess = glob.glob(os.path.join(REF, "mono_cal_target", "data", "*"))[0]
    frames, shifts = quiet(mono.load_session, sess)
    y0, x0 = 640, 900
    crop = [fr[y0:y0 + h, x0:x0 + w].copy() for fr in frames]
    saa = mono.shift_and_add(crop, shifts, factor=f, order=3)
    hr, errs = quiet(mono.ibp, crop, shifts, psf_g, saa.copy(), factor=f, n_iter=80, step=0.5)
    d["real_lr"], d["real_shifts"], d["real_saa"] = np.stack(crop).astype(np.uint8), np.array(shifts), saa
    d["real_ibp80"], d["real_errors"] = hr, np.asarray(errs)
    np.savez_compressed(os.path.join(OUT, "frame_zero.npz"), **d)


def window_goldens(mono, rgb, psf_g, psf_m):
    """Round 4: reference-generated goldens LARGE enough for the three window kernels (the round-3 review: they were held to the
    oracle only).  Four files, each <= ~1 MB (inputs uint8, big outputs float32, every hr_init the float32-ROUNDED shift_and_add so
    that the stored init is exactly what the reference iterated from):
      win_btile.npz  a 64 x 96 red-LR crop of the committed rgb_cal_target frames through the reference's load_combo (rep means:
                     non-integer frames; measured shifts; rgb_cal_target/run_sr.py:78-113) and its ibp for its 50 iterations with
                     the DEFAULT Gaussian PSF (:59, :204-223) and with the measured, non-separable PSF (--psf measured, :128-166)
      win_dtile.npz  72 x 80 LR at x4, all 16 phases (288 x 320 HR = 2 x 2 windows): ibp at 1 / 80 iterations + the 80-entry trace
                     (mono_cal_target/run_sr.py:190-209); win_dtile_float.npz: the same with half-integer frames (mean of two sensor
                     draws), 80 iterations
      win_atile.npz  40 x 50 LR at x4, all 16 phases (160 x 200 HR): 1 / 10 / 80 iterations"""
    f32 = np.float32
    # ---- (a) rgb_cal_target crop ----
    combo = glob.glob(os.path.join(REF, "rgb_cal_target", "data", "*"))[0]
    frames, shifts = quiet(rgb.load_combo, combo)
    h, w, y0, x0 = 64, 96, 352, 470
    raw = []
    for idx in range(4):
        reps = sorted(glob.glob(os.path.join(combo, f"corner{idx}_rep*.png")))
        raw.append(np.stack([np.array(Image.open(r))[2 * y0:2 * (y0 + h), 2 * x0:2 * (x0 + w)] for r in reps]))
    crop = [fr[y0:y0 + h, x0:x0 + w].copy() for fr in frames]
    saa = rgb.shift_and_add(crop, shifts, factor=2, order=3)
    init = saa.astype(f32).astype(np.float64)
    d = dict(raw=np.stack(raw), shifts=np.array(shifts), psf_g=psf_g, psf_m=psf_m, saa=saa.astype(f32), crop_yx=np.array([y0, x0]))
    for tag, psf in (("g", psf_g), ("m", psf_m)):
        hr, errs = quiet(rgb.ibp, crop, shifts, psf, init.copy(), factor=2, n_iter=50, step=0.5)
        d[f"ibp50_{tag}"], d[f"errors_{tag}"] = hr.astype(f32), np.asarray(errs)
    hr, _ = quiet(rgb.ibp, crop, shifts, psf_g, init.copy(), factor=2, n_iter=1, step=0.5)
    d["ibp1_g"] = hr.astype(f32)
    np.savez_compressed(os.path.join(OUT, "win_btile.npz"), **d)
    # ---- (b), (c): x4 phase grids ----
    ph = synth.phase_shifts(4)
    for name, (h, w), seed in (("win_dtile", (72, 80), 31), ("win_atile", (40, 50), 33)):
        f = 4
        truth = synth.truth_image(h * f, w * f, seed=synth.SEED_TRUTH + seed)
        lr = make_lr(mono, truth, psf_g, ph, f, synth.SEED_NOISE + seed)
        saa = mono.shift_and_add(list(lr), ph, factor=f, order=3)
        init = saa.astype(f32).astype(np.float64)
        tr, errs = ibp_trace(mono, lr, ph, psf_g, init, f, (1, 10, 80))
        # (truth is synth.truth_image(h * f, w * f, seed=truth_seed): numpy only, regenerated by the tests)
        d = dict(truth_seed=np.array(synth.SEED_TRUTH + seed), psf_g=psf_g, shifts16=np.array(ph), lr16=lr.astype(np.uint8),
                 saa16=saa.astype(f32), ibp16_1=tr[1].astype(f32), ibp16_80=tr[80].astype(f32), ibp16_errors=errs)
        if name == "win_atile":
            d["ibp16_10"] = tr[10].astype(f32)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        if name == "win_dtile":  # half-integer frames: the float form of the mosaic (k_ibp_dtile<false, .>)
            lr_b = make_lr(mono, truth, psf_g, ph, f, synth.SEED_NOISE + seed + 1)
            lrh = 0.5 * (lr + lr_b)
            saa = mono.shift_and_add(list(lrh), ph, factor=f, order=3)
            init = saa.astype(f32).astype(np.float64)
            hr, errs = quiet(mono.ibp, list(lrh), ph, psf_g, init.copy(), factor=f, n_iter=80, step=0.5)
            np.savez_compressed(os.path.join(OUT, "win_dtile_float.npz"), truth_seed=d["truth_seed"], psf_g=psf_g, shifts16=np.array(ph),
                                lr16_a=lr.astype(np.uint8), lr16_b=lr_b.astype(np.uint8), saa16=saa.astype(f32), ibp16_80=hr.astype(f32),
                                ibp16_errors=np.asarray(errs))


def write_manifest(meta):
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as fp:
        json.dump({"generated_by": "tools/make_golden.py", "reference": "benedikthoward/ENPH459-Super-Resolution",
                   "functions": "mono_cal_target/run_sr.py:157-209 (+rgb_cal_target loaders) imported by path",
                   "versions": meta,
                   "files": sorted(os.path.basename(p) for p in glob.glob(os.path.join(OUT, "*.npz")))}, fp, indent=1)
    for p in sorted(glob.glob(os.path.join(OUT, "*.npz"))):
        print(f"{os.path.basename(p):24s} {os.path.getsize(p) / 1024:8.1f} KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    mono = load_ref("mono_cal_target")
    rgb = load_ref("rgb_cal_target")
    psf_g = quiet(mono.make_gaussian_psf)
    psf_m = quiet(mono.load_measured_psf, os.path.join(REF, "calibration_beam_shift", "data"))
    meta = {"scipy": __import__("scipy").__version__, "numpy": np.__version__}
    if "--only-windows" in sys.argv:  # round 4: the window kernels' files; everything else stays byte for byte
        window_goldens(mono, rgb, psf_g, psf_m)
        write_manifest(meta)
        return
    frame_goldens(mono, psf_g, psf_m)
    window_goldens(mono, rgb, psf_g, psf_m)
    if "--only-frames" in sys.argv:  # the other files are unchanged since round 1: leave them byte for byte
        write_manifest(meta)
        return

    # ---------------- C1: f=2, N=4, 32x32 LR -> 64x64 -------------------------------
    f, h, w = 2, 32, 32
    truth = synth.truth_image(h * f, w * f)
    nom, meas = synth.NOMINAL_4, synth.MEASURED_4
    lr_nom = make_lr(mono, truth, psf_g, nom, f, synth.SEED_NOISE)
    lr_meas = make_lr(mono, truth, psf_m, meas, f, synth.SEED_NOISE + 1)
    # rep-averaged (non-integer) frames, like rgb_cal_target/run_sr.py:107-108
    reps = np.stack([make_lr(mono, truth, psf_m, meas, f, 1000 + r) for r in range(5)])
    lr_avg = reps.mean(axis=0)
    d = dict(truth=truth, psf_g=psf_g, psf_m=psf_m, shifts_nom=np.array(nom), shifts_meas=np.array(meas),
             lr_nom=lr_nom.astype(np.uint8), lr_meas=lr_meas.astype(np.uint8), lr_reps=reps.astype(np.uint8))
    d["blur_g"] = mono.blur(truth, psf_g)
    d["blur_m"] = mono.blur(truth, psf_m)
    d["shift_frac_arg"] = np.array([0.9445, -0.8677])
    d["shift_frac"] = mono.ndi_shift(truth, (0.9445, -0.8677), order=3, mode="nearest")
    d["shift_int_arg"] = np.array([1.0, -1.0])
    d["shift_int"] = mono.ndi_shift(truth, (1.0, -1.0), order=3, mode="nearest")
    d["zoom2"] = mono.ndi_zoom(lr_nom[0], 2, order=3)
    d["native2"] = mono.ndi_zoom(np.mean(list(lr_nom), axis=0), 2, order=3)
    d["fwd_nom"] = np.stack([mono.forward_model(truth, psf_g, s, f) for s in nom])
    d["fwd_meas"] = np.stack([mono.forward_model(truth, psf_m, s, f) for s in meas])
    err = lr_meas - d["fwd_meas"]
    d["bp_meas"] = np.stack([mono.back_project(e, psf_m, s, f, truth.shape) for e, s in zip(err, meas)])
    err = lr_nom - d["fwd_nom"]
    d["bp_nom"] = np.stack([mono.back_project(e, psf_g, s, f, truth.shape) for e, s in zip(err, nom)])
    d["saa_nom"] = mono.shift_and_add(list(lr_nom), nom, factor=f, order=3)
    d["saa_meas"] = mono.shift_and_add(list(lr_avg), meas, factor=f, order=3)
    tr, errs = ibp_trace(mono, lr_nom, nom, psf_g, d["saa_nom"], f, (1, 2, 10, 80))
    for n, hr in tr.items():
        d[f"ibp_nom_{n}"] = hr
    d["ibp_nom_errors"] = errs
    tr, errs = ibp_trace(mono, lr_avg, meas, psf_m, d["saa_meas"], f, (1, 2, 10, 50))
    for n, hr in tr.items():
        d[f"ibp_meas_{n}"] = hr
    d["ibp_meas_errors"] = errs
    np.savez_compressed(os.path.join(OUT, "synth_c1.npz"), **d)

    # ---------------- C2 small: f=4, 24x24 LR -> 96x96 ------------------------------
    f, h, w = 4, 24, 24
    truth = synth.truth_image(h * f, w * f, seed=synth.SEED_TRUTH + 1)
    ph = synth.phase_shifts(4)
    nom4 = synth.NOMINAL_4
    lr16 = make_lr(mono, truth, psf_g, ph, f, synth.SEED_NOISE + 2)
    lr4 = make_lr(mono, truth, psf_m, nom4, f, synth.SEED_NOISE + 3)
    d = dict(truth=truth, psf_g=psf_g, psf_m=psf_m, shifts16=np.array(ph), shifts4=np.array(nom4),
             lr16=lr16.astype(np.uint8), lr4=lr4.astype(np.uint8))
    d["zoom4"] = mono.ndi_zoom(lr16[3], 4, order=3)
    d["fwd16"] = np.stack([mono.forward_model(truth, psf_g, s, f) for s in ph])
    d["saa16"] = mono.shift_and_add(list(lr16), ph, factor=f, order=3)
    d["saa4"] = mono.shift_and_add(list(lr4), nom4, factor=f, order=3)
    tr, errs = ibp_trace(mono, lr16, ph, psf_g, d["saa16"], f, (1, 10, 80))
    for n, hr in tr.items():
        d[f"ibp16_{n}"] = hr
    d["ibp16_errors"] = errs
    tr, errs = ibp_trace(mono, lr4, nom4, psf_m, d["saa4"], f, (1, 10, 80))
    for n, hr in tr.items():
        d[f"ibp4_{n}"] = hr
    d["ibp4_errors"] = errs
    np.savez_compressed(os.path.join(OUT, "synth_c2_small.npz"), **d)

    # ---------------- C2 full patch: f=4, N=16, 64x64 LR -> 256x256 ------------------
    f, h, w = 4, 64, 64
    truth = synth.truth_image(h * f, w * f, seed=synth.SEED_TRUTH + 2)
    lr16 = make_lr(mono, truth, psf_g, ph, f, synth.SEED_NOISE + 4)
    saa = mono.shift_and_add(list(lr16), ph, factor=f, order=3)
    hr80, errs = quiet(mono.ibp, list(lr16), ph, psf_g, saa.copy(), factor=f, n_iter=80, step=0.5)
    np.savez_compressed(os.path.join(OUT, "synth_c2_full.npz"), truth=truth.astype(np.float32), psf_g=psf_g,
                        shifts16=np.array(ph), lr16=lr16.astype(np.uint8), saa16=saa, ibp16_80=hr80,
                        ibp16_errors=np.asarray(errs))

    # ---------------- ragged shapes: hr_init not f * lr shape -------------------------
    f = 2
    truth = synth.truth_image(65, 67, seed=synth.SEED_TRUTH + 3)
    sh = [(0.31, -0.2), (-0.5, 0.5), (0.0, 0.75)]
    sim = np.stack([mono.forward_model(truth, psf_m, s, f) for s in sh])  # [3, 33, 34]
    lr = synth.sensor_frames(sim, seed=5)[:, :32, :33]  # lr smaller than sim -> cropping path :199-201
    hr10, errs = quiet(mono.ibp, list(lr), sh, psf_m, truth * 0.9, factor=f, n_iter=10, step=0.5)
    bp = mono.back_project(lr[0] - sim[0, :32, :33], psf_m, sh[0], f, truth.shape)  # pad path :172-175
    # a 5x3 kernel and a big shift to pin the generic paths
    k53 = psf_m[1:6, 2:5] / psf_m[1:6, 2:5].sum()
    np.savez_compressed(os.path.join(OUT, "ragged.npz"), truth=truth, psf_m=psf_m, shifts=np.array(sh),
                        lr=lr.astype(np.uint8), fwd=sim, bp0=bp, hr_init=truth * 0.9, ibp_10=hr10,
                        ibp_errors=np.asarray(errs), k53=k53, blur53=mono.blur(truth, k53),
                        shift_big_arg=np.array([13.7, -20.25]),
                        shift_big=mono.ndi_shift(truth, (13.7, -20.25), order=3, mode="nearest"),
                        zoom3=mono.ndi_zoom(truth[:21, :17], 3, order=3))

    # ---------------- crops of the reference's committed real inputs ------------------
    d = {}
    sess = glob.glob(os.path.join(REF, "mono_cal_target", "data", "*"))[0]
    frames, shifts = quiet(mono.load_session, sess)
    Hh, Ww = frames[0].shape
    for name, (y0, x0) in {"tl": (0, 0), "br": (Hh - 48, Ww - 48), "mid": (700, 1000)}.items():
        crop = [fr[y0:y0 + 48, x0:x0 + 48].copy() for fr in frames]
        saa = mono.shift_and_add(crop, shifts, factor=2, order=3)
        hr, errs = quiet(mono.ibp, crop, shifts, psf_g, saa.copy(), factor=2, n_iter=10, step=0.5)
        d[f"mono_{name}_lr"] = np.stack(crop).astype(np.uint8)
        d[f"mono_{name}_native"] = mono.ndi_zoom(np.mean(crop, axis=0), 2, order=3)
        d[f"mono_{name}_saa"] = saa
        d[f"mono_{name}_ibp10"] = hr
        d[f"mono_{name}_errors"] = np.asarray(errs)
    d["mono_shifts"] = np.array(shifts)
    combo = glob.glob(os.path.join(REF, "rgb_cal_target", "data", "*"))[0]
    raw = {}
    for idx in range(4):
        reps = sorted(glob.glob(os.path.join(combo, f"corner{idx}_rep*.png")))
        raw[idx] = np.stack([np.array(Image.open(r)) for r in reps])  # uint8 [R, 1536, 2048]
    frames, shifts = quiet(rgb.load_combo, combo)
    for name, (y0, x0) in {"tr": (0, 1024 - 48), "mid": (300, 500)}.items():
        crop = [fr[y0:y0 + 48, x0:x0 + 48].copy() for fr in frames]
        saa = rgb.shift_and_add(crop, shifts, factor=2, order=3)
        hr, errs = quiet(rgb.ibp, crop, shifts, psf_m, saa.copy(), factor=2, n_iter=10, step=0.5)
        # raw Bayer crops (uint8) of every rep so the build can redo extract_red + rep mean
        d[f"rgb_{name}_raw"] = np.stack([raw[i][:, 2 * y0:2 * y0 + 96, 2 * x0:2 * x0 + 96] for i in range(4)])
        d[f"rgb_{name}_lr"] = np.stack(crop)
        d[f"rgb_{name}_native"] = rgb.ndi_zoom(np.mean(crop, axis=0), 2, order=3)
        d[f"rgb_{name}_saa"] = saa
        d[f"rgb_{name}_ibp10"] = hr
        d[f"rgb_{name}_errors"] = np.asarray(errs)
    d["rgb_shifts"] = np.array(shifts)
    d["psf_m"] = psf_m
    d["psf_g"] = psf_g
    np.savez_compressed(os.path.join(OUT, "real_crops.npz"), **d)

    # ---------------- measured PSF: windows around the pinhole peaks + the reference's kernel -----------
    psf_dir = os.path.join(REF, "calibration_beam_shift", "data")
    wins = []
    for sweep in sorted(os.listdir(psf_dir)):
        path = os.path.join(psf_dir, sweep, "pos4_(0,0).png")
        if os.path.isdir(os.path.join(psf_dir, sweep)) and os.path.exists(path):
            img = np.array(Image.open(path))
            pr, pc = np.unravel_index(img.argmax(), img.shape)
            assert 20 <= pr < img.shape[0] - 20 and 20 <= pc < img.shape[1] - 20
            wins.append(img[pr - 20:pr + 21, pc - 20:pc + 21].copy())  # 41x41, same arg-max as the full frame
    np.savez_compressed(os.path.join(OUT, "pinholes.npz"), windows=np.stack(wins), psf_m=psf_m)

    write_manifest(meta)


if __name__ == "__main__":
    main()
