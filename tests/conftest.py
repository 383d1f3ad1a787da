import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    """norm-wise relative error |a-b|_2 / |b|_2 (compare norm-wise, not element-wise)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt((b ** 2).sum())
    num = np.sqrt(((a - b) ** 2).sum())
    return num / den if den > 0 else num


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return _load
