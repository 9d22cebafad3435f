import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_loader():
    return golden


# ---- parity measures shared by the GPU tests (north star: "within 1e-4 relative") ----
def rel(a, b):
    """normwise: max |delta| / max |ref| over the array"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def rel_large(a, b, frac=0.05):
    """worst element-wise relative error over the entries with |ref| >= frac * max |ref| of the array"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = np.abs(b) >= frac * np.abs(b).max()
    return (np.abs(a - b)[m] / np.abs(b)[m]).max() if m.any() else 0.0


def rel_rows(a, b, frac=0.05, floor=1e-25):
    """worst ELEMENT-WISE relative error over the entries that carry their ROW: |ref| >= frac * max |ref| of the row
    (a component's row of F / S, a session's row of stat0 / stat1, one i-vector).  A 1-D array is one row per entry
    (frac has no effect: every non-negligible entry is judged).  Rows whose largest entry is below `floor` times the
    largest entry of the array are skipped: float32 products of posteriors that small are denormal."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.ndim == 1:
        a, b = a[:, None], b[:, None]
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    rmax = np.abs(b).max(axis=1, keepdims=True)
    m = (np.abs(b) >= frac * rmax) & (rmax > floor * max(np.abs(b).max(), 1e-300)) & (np.abs(b) > 0)
    return (np.abs(a - b)[m] / np.abs(b)[m]).max() if m.any() else 0.0
