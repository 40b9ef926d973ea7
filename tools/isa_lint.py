#!/usr/bin/env python3
"""Static lint of gfx950 machine code for the two hazards round 3 found on MI355X that neither the assembler nor the compiler's
hazard recogniser covers (reproducers in tools/microbench/):

  H1  a 96- or 128-bit buffer store whose scalar offset operand is a register (not the literal 0 / `off`), directly followed by a
      vector instruction that writes one of its data registers: GCNHazardRecognizer::createsVALUHazard exempts exactly this addressing
      form, and with many waves storing at once gfx950 does send the overwritten values (tools/microbench/store_data_war.hip).
  H2  a scalar write of M0 directly followed by an LDS add-TID access (ds_write_addtid_b32 / ds_read_addtid_b32): the ISA asks for one
      wait state; inside an asm block nobody inserts it (srx_patch.hpp: SRX_M0_NOP).

Works on `hipcc -S --cuda-device-only` listings and on `llvm-objdump -d` output of the code object inside libsrx.so.

    python tools/isa_lint.py                     # lint enph459-super-resolution_amd/sr_mi355x/libsrx.so
    python tools/isa_lint.py file.s [file2.s]    # lint listings
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = os.environ.get("SRX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

_REG = re.compile(r"^(v|a)(?:(\d+)|\[(\d+):(\d+)\])$")
_WIDE_STORE = re.compile(r"^buffer_store_(dwordx3|dwordx4|b96|b128)\b")
_ADDTID = re.compile(r"^ds_(write|read|store|load)_addtid_b32\b")


def instructions(text):
    """[(function, line number, mnemonic, [operands])] in program order; labels, directives and comments dropped."""
    out, fn = [], "?"
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split("//")[0].split(";")[0].strip()
        if not line or line.startswith("."):
            continue
        m = re.match(r"^(?:[0-9a-fA-F]+\s+)?<?([A-Za-z_$][\w$.]*)>?:$", line)
        if m:  # `name:` (-S) or `0000000000001000 <name>:` (objdump)
            if not re.match(r"^(\.?L|BB)\w*$", m.group(1)):
                fn = m.group(1)
            continue
        if line.endswith(":"):
            continue
        parts = line.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        out.append((fn, ln, parts[0], ops))
    return out


def _regs(op):
    """the set of ('v' | 'a', index) an operand token names"""
    m = _REG.match(op.split()[0]) if op else None
    if not m:
        return set()
    if m.group(2) is not None:
        return {(m.group(1), int(m.group(2)))}
    return {(m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1)}


def _vector_dests(mn, ops):
    """registers a vector instruction writes (VALU: the first operand; v_swap: both; loads into VGPRs count as writers too)"""
    if mn.startswith("v_swap"):
        return _regs(ops[0]) | _regs(ops[1])
    if mn.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    if mn.startswith("v_"):
        return _regs(ops[0]) if ops else set()
    if re.match(r"^(buffer|global|flat|scratch)_load|^ds_(read|load|bpermute|permute|swizzle)|^ds_\w+_rtn", mn):
        return _regs(ops[0]) if ops else set()
    return set()


def _soffset_is_register(ops):
    """buffer_store vdata, vaddr, srsrc, soffset [modifiers]: soffset other than the literal 0 / `off` / `null`"""
    if len(ops) < 4:
        return False
    tok = ops[3].split()[0] if ops[3] else ""
    if tok in ("0", "off", "null", ""):
        return False
    return not re.match(r"^(0x0+|0)$", tok)


def _writes_m0(mn, ops):
    return bool(ops) and ops[0].split()[0] == "m0" and mn.startswith("s_") and not mn.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_nop"))


def lint(text):
    """-> list of (rule, function, line number, message)"""
    ins = instructions(text)
    found = []
    for i in range(len(ins) - 1):
        fn, ln, mn, ops = ins[i]
        fn2, _, mn2, ops2 = ins[i + 1]
        if fn2 != fn:
            continue
        if _WIDE_STORE.match(mn) and _soffset_is_register(ops):
            hit = _regs(ops[0]) & _vector_dests(mn2, ops2)
            if hit:
                found.append(("H1", fn, ln, f"{mn} {', '.join(ops)}  ->  {mn2} {', '.join(ops2)} overwrites {sorted(hit)}"))
        if _writes_m0(mn, ops) and _ADDTID.match(mn2):
            found.append(("H2", fn, ln, f"{mn} {', '.join(ops)}  ->  {mn2} with no wait state"))
    return found


def disassemble_library(so_path):
    """device code of a hipcc-built shared library as text (copies it to a scratch directory: llvm-objdump --offloading writes the
    extracted bundles next to its input)"""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "lib.so")
        shutil.copy(so_path, tmp)
        subprocess.check_call([objdump, "--offloading", tmp], stdout=subprocess.DEVNULL, cwd=td)
        cos = [os.path.join(td, f) for f in os.listdir(td) if "amdgcn" in f]
        if not cos:
            raise RuntimeError(f"no amdgcn code object inside {so_path}")
        return "".join(subprocess.check_output([objdump, "-d", "--no-show-raw-insn", co], text=True) for co in cos)


def main(argv):
    if argv:
        texts = [(p, open(p).read()) for p in argv]
    else:
        so = os.path.join(ROOT, "enph459-super-resolution_amd", "sr_mi355x", "libsrx.so")
        texts = [(so, disassemble_library(so))]
    bad = 0
    for name, text in texts:
        ins, found = instructions(text), lint(text)
        wide = sum(1 for _, _, mn, ops in ins if _WIDE_STORE.match(mn))
        wide_s = sum(1 for _, _, mn, ops in ins if _WIDE_STORE.match(mn) and _soffset_is_register(ops))
        addtid = sum(1 for _, _, mn, _ in ins if _ADDTID.match(mn))
        print(f"{name}: {len(ins)} instructions, {wide} wide buffer stores ({wide_s} with a register soffset), {addtid} add-TID LDS accesses, "
              f"{len(found)} findings")
        for rule, fn, ln, msg in found:
            print(f"  {rule} {fn}:{ln}: {msg}")
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
