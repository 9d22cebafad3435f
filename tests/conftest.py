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


def _rows(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.ndim == 1:                 # a vector (N, w, one session) is ONE row
        a, b = a[None, :], b[None, :]
    return a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)


ROW_FLOOR = 1e-3     # rows (components, sessions) a thousand times below the largest row are not judged


def rel_rows(a, b, frac=0.05, floor=ROW_FLOOR):
    """worst ELEMENT-WISE relative error over the entries that carry their ROW: |ref| >= frac * max |ref| of the row
    (a component's row of F / S, a session's row of stat0 / stat1, one i-vector; a 1-D array is one row).  Rows whose
    largest entry is below `floor` times the largest entry of the array are skipped: a component a thousand times
    lighter than the heaviest one (a fraction of a frame in these test sizes) gets its whole mass from the far tail of
    a few posteriors, whose float32 logits differ in their leading bits from those of the frame's dominant components
    (no common rounding to cancel), and nothing downstream -- the M-step floors its variance, the TV weighs it by N_c
    -- reads it to four digits."""
    a, b = _rows(a, b)
    rmax = np.abs(b).max(axis=1, keepdims=True)
    m = (np.abs(b) >= frac * rmax) & (rmax > floor * max(np.abs(b).max(), 1e-300)) & (np.abs(b) > 0)
    return (np.abs(a - b)[m] / np.abs(b)[m]).max() if m.any() else 0.0


def rel_rownorm(a, b, floor=ROW_FLOOR):
    """worst ROW-normwise error: max over rows of (max |delta| of the row / max |ref| of the row) -- every row is judged
    on its own scale, a row 100 times below the largest one cannot hide."""
    a, b = _rows(a, b)
    rmax = np.abs(b).max(axis=1)
    keep = rmax > floor * max(np.abs(b).max(), 1e-300)
    return (np.abs(a - b).max(axis=1)[keep] / rmax[keep]).max() if keep.any() else 0.0


# First-order statistics are sums of terms of both signs: F[c, d] = sum_t gamma[t, c] x[t, d].  An error of relative size
# eps in the posteriors (float32 logits: eps ~ 1e-5 at 60 columns, the reference's float64 logits: 1e-16) moves F[c, d] by up
# to eps * sum_t gamma |x|, which an entry at 5 % of its row's maximum sees 20 times enlarged.  F rows therefore carry three
# measures: row-normwise at 1e-4, element-wise against the sum of magnitudes at 1e-4 (where the test has the posteriors),
# and the plain element-wise ratio on the entries >= 5 % of their row's maximum at CANCEL_TOL.
CANCEL_TOL = 1e-3


def check_first_order(F, rF, tol=1e-4, what=""):
    assert rel_rownorm(F, rF) < tol, (what, "row-normwise", rel_rownorm(F, rF))
    assert rel_rows(F, rF) < CANCEL_TOL, (what, "element-wise on entries >= 5 % of their row", rel_rows(F, rF))
