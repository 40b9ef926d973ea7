"""The drop-in boundary is a C ABI: a plain C++ program (HIP runtime + include/srx.h, no Python, no torch) links libsrx.so,
runs shift_and_add + IBP in both precisions on its own stream and checks the HR image and the MSE trace against the reference's
outputs for the same inputs (tests/c_abi/host_example.cpp, tests/golden/c_abi_c1.bin)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_links_and_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.join(ROOT, "enph459-super-resolution_amd", "sr_mi355x")
    assert os.path.exists(os.path.join(libdir, "libsrx.so")), "build libsrx.so first (python -c 'import __graft_entry__ as g; g.build()')"
    exe = str(tmp_path / "host_example")
    subprocess.check_call([hipcc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "host_example.cpp"),
                           "-L", libdir, "-l:libsrx.so", f"-Wl,-rpath,{libdir}", "-o", exe])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "c_abi_c1.bin")], capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI host example OK" in out.stdout and "path=mosaic" in out.stdout
