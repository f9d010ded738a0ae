"""Shared test plumbing: marker registration, import paths, golden-fixture loader."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "uav-autonomous-control_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")

# A test run on a fresh checkout builds the HIP library itself (the product only does so when asked: uav_ac/_native.py)
os.environ.setdefault("UAVAC_AUTOBUILD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def col_err(a, b):
    """SURVEY.md 8(c) metric: per column max|a-b| / max(1, max|b|); returns the worst column."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    a2 = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a.reshape(-1, 1)
    b2 = b.reshape(-1, b.shape[-1]) if b.ndim > 1 else b.reshape(-1, 1)
    scale = np.maximum(1.0, np.max(np.abs(b2), axis=0))
    return float(np.max(np.max(np.abs(a2 - b2), axis=0) / scale))


@pytest.fixture(scope="session")
def golden():
    return load_golden
