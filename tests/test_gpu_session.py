"""Session driver on the GPU: a session directory is rebuilt from the golden crops of the reference's committed
inputs (PNG files in the reference's own naming), run through sr_mi355x.session.process_session and the PNGs
it writes are compared with what the reference's functions gave on the same frames (tests/golden/real_crops.npz):
uint8 outputs within 1 LSB and >= 99.9 % identical (truncating quantiser), done.flag skip honoured."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from PIL import Image  # noqa: E402

import sr_mi355x as S  # noqa: E402
from sr_mi355x import session  # noqa: E402


def u8(x):
    return np.clip(x, 0, 255).astype(np.uint8)


def same_u8(path, ref_float):
    got = np.array(Image.open(path)).astype(np.int16)
    ref = u8(ref_float).astype(np.int16)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1 and (got == ref).mean() >= 0.999


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_mono_cal_session(tmp_path, g_real, prec):
    S.set_precision(prec)
    try:
        sess = tmp_path / "data" / "cal_target_crop"
        sess.mkdir(parents=True)
        for (fname, _), frame in zip(session.IMAGE_SHIFTS, g_real["mono_mid_lr"]):
            Image.fromarray(frame).save(sess / fname)
        out = tmp_path / "results"
        written = session.process_session(str(sess), g_real["psf_g"], str(out), n_iter=10, verbose=False)
        assert len(written) == 1
        d = written[0]
        same_u8(os.path.join(d, "native_2x.png"), g_real["mono_mid_native"])
        same_u8(os.path.join(d, "SAA.png"), g_real["mono_mid_saa"])
        same_u8(os.path.join(d, "SAA_IBP.png"), g_real["mono_mid_ibp10"])
        same_u8(os.path.join(d, "LR_mean.png"), g_real["mono_mid_lr"].astype(np.float64).mean(axis=0))
        mse = json.load(open(os.path.join(d, "convergence.json")))["ibp_mse"]
        np.testing.assert_allclose(mse, g_real["mono_mid_errors"], rtol=1e-4)
        assert os.path.exists(os.path.join(d, "done.flag"))
        assert session.process_session(str(sess), g_real["psf_g"], str(out), n_iter=10, verbose=False) == []  # skipped
    finally:
        S.set_precision("f32")


def test_rgb_cal_combo(tmp_path, g_real):
    combo = tmp_path / "data" / "combo_crop"
    combo.mkdir(parents=True)
    raw = g_real["rgb_mid_raw"]  # uint8 [4 corners, 5 reps, 96, 96] Bayer crops
    for c in range(4):
        for r in range(raw.shape[1]):
            Image.fromarray(raw[c, r]).save(combo / f"corner{c}_rep{r:02d}.png")
    shifts = g_real["rgb_shifts"]
    meta = {"expected_shifts": {lab: {"dy_px": 2.0 * float(s[0]), "dx_px": 2.0 * float(s[1])}
                                for lab, s in zip(session.CORNER_ORDER, shifts)}}
    (combo / "metadata.json").write_text(json.dumps(meta))
    out = tmp_path / "results"
    written = session.process_session(str(combo), g_real["psf_m"], str(out), kind="rgb_cal_target", n_iter=10,
                                      verbose=False)
    d = written[0]
    same_u8(os.path.join(d, "native_2x.png"), g_real["rgb_mid_native"])
    same_u8(os.path.join(d, "SAA.png"), g_real["rgb_mid_saa"])
    same_u8(os.path.join(d, "SAA_IBP.png"), g_real["rgb_mid_ibp10"])
    same_u8(os.path.join(d, "LR_red_mean.png"), g_real["rgb_mid_lr"].mean(axis=0))
    sj = json.load(open(os.path.join(d, "shifts.json")))
    assert sj["corner_labels"] == session.CORNER_ORDER
    np.testing.assert_allclose(np.array(sj["shifts_lr_yx"]), shifts, rtol=0, atol=1e-15)


def test_barcode_reps(tmp_path, g_real):
    """corner{c}_rep{rr}.png layout, one reconstruction per rep into rep{idx}/ (mono_barcodes/run_sr.py:301-351)."""
    sess = tmp_path / "data" / "barcodes"
    sess.mkdir(parents=True)
    frames = g_real["mono_tl_lr"][1:]  # 4 frames standing in for the 4 corners
    for rep in range(2):
        for c in range(4):
            Image.fromarray(np.roll(frames[c], rep, axis=1)).save(sess / f"corner{c}_rep{rep:02d}.png")
    out = tmp_path / "results"
    written = session.process_session(str(sess), g_real["psf_g"], str(out), kind="mono_barcodes", n_iter=3, verbose=False)
    assert [os.path.basename(w) for w in written] == ["rep0", "rep1"]
    from oracle import sr_oracle as O
    for rep, d in enumerate(written):
        lr = np.stack([np.roll(frames[c], rep, axis=1) for c in range(4)]).astype(np.float64)
        saa = O.shift_and_add(list(lr), session.CORNER_SHIFTS, 2)
        hr, _ = O.ibp(list(lr), session.CORNER_SHIFTS, g_real["psf_g"], saa, 2, 3, 0.5)
        same_u8(os.path.join(d, "SAA.png"), saa)
        same_u8(os.path.join(d, "SAA_IBP.png"), hr)
