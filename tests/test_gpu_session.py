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


def _png_bytes(d):
    return {n: open(os.path.join(d, n), "rb").read() for n in ("native_2x.png", "SAA.png", "SAA_IBP.png", "LR_mean.png")}


def test_barcode_reps_batched_equals_per_rep(tmp_path, g_real):
    """The reps of a barcode session in ONE B = reps library call (reconstruct_batch) give, file for file and bit for bit, what one
    call per rep gives (mono_barcodes/run_sr.py:301-351 processes them one after the other); a rep already done is skipped and
    the rest still batched; the prefetching multi-session loop writes the same files."""
    frames = g_real["mono_tl_lr"][1:]
    for sname in ("bc_a", "bc_b"):
        sess = tmp_path / "data" / sname
        sess.mkdir(parents=True)
        for rep in range(3):
            for c in range(4):
                Image.fromarray(np.roll(frames[c], rep + (5 if sname == "bc_b" else 0), axis=1)).save(sess / f"corner{c}_rep{rep:02d}.png")
    sa = str(tmp_path / "data" / "bc_a")
    one = session.process_session(sa, g_real["psf_g"], str(tmp_path / "per_rep"), kind="mono_barcodes", n_iter=6, verbose=False, batch_reps=False)
    bat = session.process_session(sa, g_real["psf_g"], str(tmp_path / "batched"), kind="mono_barcodes", n_iter=6, verbose=False)
    assert [os.path.basename(w) for w in bat] == ["rep0", "rep1", "rep2"] and len(one) == 3
    for a, b in zip(one, bat):
        assert _png_bytes(a) == _png_bytes(b)
        assert json.load(open(os.path.join(a, "convergence.json"))) == json.load(open(os.path.join(b, "convergence.json")))
    # rep1 already done elsewhere: skipped, rep0 and rep2 still go through one call
    part = tmp_path / "partial" / "bc_a" / "rep1"
    part.mkdir(parents=True)
    (part / "done.flag").write_text("")
    got = session.process_session(sa, g_real["psf_g"], str(tmp_path / "partial"), kind="mono_barcodes", n_iter=6, verbose=False)
    assert [os.path.basename(w) for w in got] == ["rep0", "rep2"]
    assert _png_bytes(got[1]) == _png_bytes(bat[2])
    # the multi-session loop (host decode of the next session overlapped with the device work of the current one), and its
    # two-rank sharding: rank 0 of 2 owns session 0, rank 1 session 1
    sessions = session.discover_sessions(str(tmp_path / "data"), "mono_barcodes")
    assert [os.path.basename(s) for s in sessions] == ["bc_a", "bc_b"]
    all_w = session.process_sessions(sessions, g_real["psf_g"], str(tmp_path / "loop"), "mono_barcodes", n_iter=6, verbose=False)
    assert len(all_w) == 6 and _png_bytes(all_w[2]) == _png_bytes(bat[2])
    r1 = session.process_sessions(sessions, g_real["psf_g"], str(tmp_path / "rank1"), "mono_barcodes", n_iter=6, verbose=False, rank=1, world=2)
    assert len(r1) == 3 and all("bc_b" in w for w in r1) and _png_bytes(r1[0]) == _png_bytes(all_w[3])


def test_run_sr_two_ranks_on_one_gpu(tmp_path, g_real):
    """`python -m sr_mi355x.run_sr` under torch.distributed.run with two ranks (both on cuda:0 here; one GPU each on a node):
    session i goes to rank i mod 2, no collective on the data path, the files equal a single-process run's."""
    import subprocess
    import sys
    frames = g_real["mono_tl_lr"][1:]
    for k, sname in enumerate(("s0", "s1", "s2")):
        sess = tmp_path / "data" / sname
        sess.mkdir(parents=True)
        for rep in range(2):
            for c in range(4):
                Image.fromarray(np.roll(frames[c], rep + 3 * k, axis=0)).save(sess / f"corner{c}_rep{rep:02d}.png")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRX_ONE_GPU="1", PYTHONPATH=os.path.join(root, "enph459-super-resolution_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    args = ["-m", "sr_mi355x.run_sr", "--kind", "mono_barcodes", "--data-dir", str(tmp_path / "data")]
    one = subprocess.run([sys.executable] + args + ["--output-dir", str(tmp_path / "one")], capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541"] + args + ["--output-dir", str(tmp_path / "two")], capture_output=True, text=True,
                         timeout=600, env=env)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    assert "on 2 GPUs" in two.stdout and "[rank 1: 1/1] s1" in two.stdout and "[rank 0: 2/2] s2" in two.stdout
    for sname in ("s0", "s1", "s2"):
        for rep in ("rep0", "rep1"):
            assert _png_bytes(str(tmp_path / "one" / sname / rep)) == _png_bytes(str(tmp_path / "two" / sname / rep))


def test_run_sr_row_bands_two_ranks_on_one_gpu(tmp_path, g_real):
    """`run_sr --row-bands` under torch.distributed.run: both ranks work on the SAME mono_cal_target session, each iterating its row
    band of the one image (sr_mi355x/rowband.py), rank 0 writes.  Against the single-process run of the same command line."""
    import subprocess
    import sys
    sess = tmp_path / "data" / "cal_target_tall"
    sess.mkdir(parents=True)
    tall = np.concatenate([g_real["mono_tl_lr"], g_real["mono_mid_lr"], g_real["mono_br_lr"]], axis=1)  # 5 frames of 144 x 48
    for (fname, _), frame in zip(session.IMAGE_SHIFTS, tall):
        Image.fromarray(frame).save(sess / fname)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRX_ONE_GPU="1", PYTHONPATH=os.path.join(root, "enph459-super-resolution_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    args = ["-m", "sr_mi355x.run_sr", "--kind", "mono_cal_target", "--data-dir", str(tmp_path / "data")]
    one = subprocess.run([sys.executable] + args + ["--output-dir", str(tmp_path / "one")], capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547"] + args + ["--row-bands", "--output-dir", str(tmp_path / "two")], capture_output=True,
                         text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    d1, d2 = tmp_path / "one" / "cal_target_tall", tmp_path / "two" / "cal_target_tall"
    for name in ("native_2x.png", "SAA.png", "LR_mean.png"):
        assert np.array_equal(np.array(Image.open(d1 / name)), np.array(Image.open(d2 / name)))
    a, b = np.array(Image.open(d1 / "SAA_IBP.png")).astype(np.int16), np.array(Image.open(d2 / "SAA_IBP.png")).astype(np.int16)
    assert a.shape == (288, 96) and np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.999
    m1, m2 = (json.load(open(d / "convergence.json"))["ibp_mse"] for d in (d1, d2))
    np.testing.assert_allclose(m2, m1, rtol=2e-5)
    assert os.path.exists(d2 / "done.flag")


def test_metrics_json_from_device_tensors_equals_the_file_based_report(tmp_path):
    """`run_sr --metrics`: metrics.json is written from the device tensors the PNGs were quantised from (session.write_metrics_device:
    every pass over a frame / ROI in libsrx) and equals the notebook summary computed on the host from the PNGs themselves
    (session.write_metrics, analysis.ipynb cells 3-10), on a full-size synthetic mono_cal_target session (1536 x 2048 -> 3072 x 4096)."""
    from sr_mi355x import metrics as M
    S.set_precision("f32")
    h, w = 1536, 2048
    yy, xx = np.mgrid[0:h, 0:w]
    (r0, r1), (c0, c1) = M.ROI2_LR
    d = (xx - 0.5 * (c0 + c1)) * np.cos(0.35) + (yy - 0.5 * (r0 + r1)) * np.sin(0.35)
    img = 40.0 + 170.0 / (1.0 + np.exp(-(np.abs(d) - 12.0) / 0.4))      # a dark diagonal line through ROI 2 (sharp: MTF50 inside the sampled band)
    c = M.ROI1_COL_LR
    img[:, c - 20:c + 20] = (128.0 + 90.0 * np.sign(np.sin(yy / 2.3)))[:, c - 20:c + 20]  # horizontal bars through ROI 1
    rng = np.random.default_rng(12)
    sess = tmp_path / "data" / "cal_target_full"
    sess.mkdir(parents=True)
    for k, (fname, _) in enumerate(session.IMAGE_SHIFTS):
        Image.fromarray(np.clip(np.roll(img, (k % 2, k // 2), axis=(0, 1)) + rng.normal(0, 1.0, img.shape), 0, 255).astype(np.uint8)).save(sess / fname)
    got = {}
    written = session.process_session(str(sess), S.make_gaussian_psf(), str(tmp_path / "results"), n_iter=5, verbose=False,
                                      on_images=lambda out_dir, images: got.update(session.write_metrics_device(out_dir, images)))
    dev = json.load(open(os.path.join(written[0], "metrics.json")))
    assert dev == json.loads(json.dumps(got))
    host = session.write_metrics(written[0])  # from the PNGs, host numpy
    for title in ("native_2x", "SAA", "SAA_IBP"):
        for k, v in host[title].items():
            same_nan = np.isnan(v) and np.isnan(dev[title][k])  # (an MTF that never falls through 0.1 in the sampled band: nan on both sides)
            assert same_nan or abs(v - dev[title][k]) <= 1e-6 * max(1.0, abs(v)), (title, k, v, dev[title][k])
        assert np.isfinite(host[title]["mtf50"]) and host[title]["mean_contrast"] > 0.1
    q = {n: np.array(Image.open(os.path.join(written[0], n + ".png"))) for n in ("native_2x", "SAA", "SAA_IBP")}
    for n in ("native_2x", "SAA"):
        assert abs(dev["psnr_db"][f"SAA_IBP_vs_{n}"]["affine_fit"] - M.psnr_affine(q[n], q["SAA_IBP"])) < 1e-6
        assert abs(dev["psnr_db"][f"SAA_IBP_vs_{n}"]["plain"] - M.psnr(q[n], q["SAA_IBP"])) < 1e-9
