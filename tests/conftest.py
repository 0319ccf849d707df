import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "f2cnn_golden.npz"))


@pytest.fixture(scope="session")
def golden_eval():
    """G5: the tensor the reference hands to model.predict for one 1 s file (tests/golden/make_golden.py eval)"""
    return np.load(os.path.join(ROOT, "tests", "golden", "f2cnn_golden_eval.npz"))


def chan_relerr(a, b):
    """Parity norm of SURVEY section 8d: per-channel max|a-b| / max|b| (rows = channels)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    a = a.reshape(-1, a.shape[-1])
    b = b.reshape(-1, b.shape[-1])
    den = np.abs(b).max(axis=1)
    den[den == 0] = 1.0
    return (np.abs(a - b).max(axis=1) / den).max() if a.size else 0.0
