import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "enph459-super-resolution_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def g_c1():
    return load_golden("synth_c1.npz")


@pytest.fixture(scope="session")
def g_c2s():
    return load_golden("synth_c2_small.npz")


@pytest.fixture(scope="session")
def g_c2f():
    return load_golden("synth_c2_full.npz")


@pytest.fixture(scope="session")
def g_rag():
    return load_golden("ragged.npz")


@pytest.fixture(scope="session")
def g_real():
    return load_golden("real_crops.npz")


@pytest.fixture(scope="session")
def g_frame():
    return load_golden("frame_zero.npz")
