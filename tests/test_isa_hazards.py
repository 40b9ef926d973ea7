"""Static lint of the shipped machine code for the two gfx950 hazards round 3 found by determinism tests:
a wide buffer store with a register soffset whose data registers the next vector instruction overwrites, and a scalar write of M0
directly in front of an LDS add-TID access.  No GPU needed: the code object inside libsrx.so is disassembled with llvm-objdump,
and a probe built from the product's own helpers (tests/isa/probe_hazards.hip) shows the lint turning red on the two broken forms."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

from sr_mi355x import _lib  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(os.path.join(isa_lint.LLVM_BIN, "llvm-objdump"))),
                                reason="ROCm toolchain not installed")


def _probe_listing(tmp_path, tag, defines):
    out = str(tmp_path / f"probe_{tag}.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-slp-vectorize", "-w",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "enph459-super-resolution_amd", "csrc"),
                           "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "tests", "isa", "probe_hazards.hip")] + defines)
    return open(out).read()


def test_lint_recognises_the_sequences():
    bad = """
k:
\tbuffer_store_dwordx4 v[26:29], v68, s[56:59], s3 offen offset:16
\tv_mov_b64 v[28:29], v[2:3]
\ts_mov_b32 m0, s8
\tds_write_addtid_b32 v5 offset:264
"""
    assert [f[0] for f in isa_lint.lint(bad)] == ["H1", "H2"]
    good = """
k:
\tbuffer_store_dwordx4 v[26:29], v68, s[56:59], 0 offen offset:16
\tv_mov_b64 v[28:29], v[2:3]
\tbuffer_store_dwordx4 v[26:29], v68, s[56:59], s3 offen
\tv_mov_b64 v[30:31], v[2:3]
\tbuffer_store_dwordx4 v[26:29], v68, s[56:59], s3 offen
\ts_nop 0
\tv_mov_b64 v[28:29], v[2:3]
\ts_mov_b32 m0, s8
\ts_nop 0
\tds_write_addtid_b32 v5 offset:264
"""
    assert isa_lint.lint(good) == []


def test_probe_is_green_as_shipped_and_red_when_broken(tmp_path):
    ok = _probe_listing(tmp_path, "ok", [])
    assert "ds_write_addtid_b32" in ok and "buffer_store_dwordx4" in ok
    assert isa_lint.lint(ok) == []
    bad = isa_lint.lint(_probe_listing(tmp_path, "bad", ['-DSRX_M0_NOP=""', "-DSRX_PROBE_SOFFSET_STORE"]))
    rules = {(r, fn) for r, fn, _, _ in bad}
    assert ("H2", "k_probe_transpose") in rules and ("H1", "k_probe_store") in rules
    assert any(r == "H2" and "k_ibp_patch" in fn for r, fn in rules)  # the product kernel itself turns red without the wait state


def test_shipped_library_has_neither_hazard():
    _lib.build()
    text = isa_lint.disassemble_library(_lib.SO_PATH)
    ins = isa_lint.instructions(text)
    assert sum(1 for _, _, mn, _ in ins if mn.startswith("ds_write_addtid_b32")) > 1000   # the transposes are in there ...
    assert sum(1 for _, _, mn, _ in ins if mn.startswith("buffer_store_dwordx4")) > 100    # ... and so are the 16-byte stores
    found = isa_lint.lint(text)
    assert found == [], "\n".join(f"{r} {fn}:{ln}: {msg}" for r, fn, ln, msg in found[:20])
