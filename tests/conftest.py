import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """The built .so is kept out of git; build it in-tree when a fresh checkout has none (hipcc cross-compiles without a GPU)."""
    from flowdec_amd import build as _build
    if not os.path.exists(_build.LIB):
        _build.build(verbose=False)
    yield


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, b):
    """||a-b|| / ||b|| (Frobenius)."""
    a = np.asarray(a); b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


@pytest.fixture(scope="session")
def golden():
    return load_golden
