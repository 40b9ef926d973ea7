"""Host-side pieces of the session driver (no GPU): measured-PSF estimation against the kernel the
reference's load_measured_psf (mono_cal_target/run_sr.py:114-152) produced from the same pinhole frames, the
shift tables and the discovery rules."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden

session = pytest.importorskip("sr_mi355x.session")


def test_measured_psf_matches_reference():
    g = load_golden("pinholes.npz")
    k = session.psf_from_pinhole_images(list(g["windows"]))
    assert k.shape == (7, 7)
    assert np.abs(k - g["psf_m"]).max() < 1e-15
    assert abs(k.sum() - 1.0) < 1e-15 and (k >= 0).all()
    assert not np.allclose(k, k[::-1, ::-1])  # asymmetric: the flipped kernel of back_project is not a no-op


def test_psf_skips_peaks_near_the_edge():
    g = load_golden("pinholes.npz")
    wins = list(g["windows"])
    bad = np.zeros((41, 41))
    bad[2, 20] = 255.0  # peak closer than 9 px to the border
    k = session.psf_from_pinhole_images(wins + [bad])
    assert np.abs(k - g["psf_m"]).max() < 1e-15
    with pytest.raises(FileNotFoundError):
        session.psf_from_pinhole_images([bad])


def test_shift_tables_match_reference_constants():
    assert [s for _, s in session.IMAGE_SHIFTS] == [(0.0, 0.0), (0.5, -0.5), (0.5, 0.5), (-0.5, -0.5), (-0.5, 0.5)]
    assert session.CORNER_SHIFTS == [(0.5, -0.5), (0.5, 0.5), (-0.5, -0.5), (-0.5, 0.5)]
    assert session.IBP_ITERATIONS == {"mono_cal_target": 80, "rgb_cal_target": 50, "mono_barcodes": 80,
                                      "rgb_barcodes": 80}


def test_discovery(tmp_path):
    (tmp_path / "a").mkdir()
    (tmp_path / "a" / "center.png").write_bytes(b"")
    (tmp_path / "b").mkdir()
    (tmp_path / "b" / "corner0_rep00.png").write_bytes(b"")
    (tmp_path / "b" / "metadata.json").write_text(json.dumps({}))
    (tmp_path / "c.txt").write_text("x")
    assert [os.path.basename(p) for p in session.discover_sessions(str(tmp_path), "mono_cal_target")] == ["a"]
    assert [os.path.basename(p) for p in session.discover_sessions(str(tmp_path), "rgb_cal_target")] == ["b"]
    assert [os.path.basename(p) for p in session.discover_sessions(str(tmp_path), "mono_barcodes")] == ["b"]
    assert session.detect_kind(str(tmp_path / "a")) == "mono_cal_target"
    assert session.detect_kind(str(tmp_path / "b")) is None


def test_oracle_interleave4_is_a_pixel_shuffle():
    """The numpy restatement of the vendor interleave: interior = depth-to-space of the four frames."""
    from oracle import sr_oracle as O
    rng = np.random.default_rng(2)
    fr = rng.integers(0, 256, (4, 12, 10), dtype=np.uint8)
    out = O.interleave4(fr)
    assert out.shape == (24, 20) and out.dtype == np.uint8
    assert np.array_equal(out[0::2, 0::2], fr[0])                    # (tx, ty) = (0, 0)
    assert np.array_equal(out[3::2, 0::2], fr[1][1:])                # (0, +1): rows 2i+1
    assert np.array_equal(out[3::2, 1:-1:2], fr[2][1:, 1:])          # (-1, +1): rows 2i+1, columns 2j-1
    assert np.array_equal(out[0::2, 1:-1:2], fr[3][:, 1:])           # (-1, 0)


def test_png_writer_round_trips(tmp_path):
    """session.write_png_u8 (Up filter + one zlib pass, instead of PIL's per-row filter search): the file is a valid 8-bit greyscale
    PNG whose pixels read back exactly -- smooth, noisy and constant images, odd sizes, one row / one column."""
    from PIL import Image
    from sr_mi355x import session
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:97, 0:131]
    cases = [((np.sin(yy / 9.0) * np.cos(xx / 7.0) * 120 + 128).astype(np.uint8)), rng.integers(0, 256, (64, 33), dtype=np.uint8),
             np.full((5, 7), 255, np.uint8), np.zeros((1, 40), np.uint8), rng.integers(0, 256, (40, 1), dtype=np.uint8)]
    for i, a in enumerate(cases):
        path = str(tmp_path / f"t{i}.png")
        session.write_png_u8(path, a)
        im = Image.open(path)
        assert im.mode == "L" and im.size == (a.shape[1], a.shape[0])
        assert np.array_equal(np.array(im), a)
    rgb = rng.integers(0, 256, (6, 5, 3), dtype=np.uint8)  # not a greyscale plane: PIL writes it
    session.write_png_u8(str(tmp_path / "rgb.png"), rgb)
    assert np.array_equal(np.array(Image.open(str(tmp_path / "rgb.png"))), rgb)
