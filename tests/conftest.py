"""pytest configuration: markers + import paths (repo root, tests/, the package dir)."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "wavecap-sdr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


def peak_rel_err(a, b):
    """max|a-b| / max|b| -- the 'relative' error used for every float parity check
    (north_star tolerance 1e-5).  Peak-relative, so zero crossings do not blow it up."""
    import numpy as np

    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    den = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / (den if den > 0 else 1.0)
