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


MARGINS: dict[str, list[float]] = {}     # test id -> [worst peak-relative, worst per-row-relative, comparisons]


def peak_rel_err(a, b):
    """max|a-b| / max|b| -- the 'relative' error used for every float parity check
    (north_star tolerance 1e-5).  Peak-relative, so zero crossings do not blow it up.
    Every call is recorded (worst value per test, plus the worst PER-ROW relative error for 2-D arrays: each row --
    hop, frame, channel -- against its own peak, which catches a small-magnitude row that is wrong in relative terms);
    the session ends with the table of achieved margins (also written to gpurun_out/parity_margins.json)."""
    import numpy as np

    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    den = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b))) / (den if den > 0 else 1.0)
    row = err
    if a.ndim >= 2 and a.shape[-1] > 1:
        a2, b2 = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
        rd = np.max(np.abs(b2), axis=1)
        ok = rd > 0
        if ok.any():
            row = float(np.max(np.max(np.abs(a2 - b2), axis=1)[ok] / rd[ok]))
    tid = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    m = MARGINS.setdefault(tid, [0.0, 0.0, 0])
    m[0], m[1], m[2] = max(m[0], err), max(m[1], row), m[2] + 1
    return err


def pytest_terminal_summary(terminalreporter):
    if not MARGINS:
        return
    import json

    tr = terminalreporter
    tr.section("achieved float parity margins (tolerance 1e-5 unless the test says otherwise)")
    tr.write_line(f"{'test':88s} {'peak-rel':>10s} {'row-rel':>10s} {'n':>4s}")
    for tid in sorted(MARGINS):
        e, r, n = MARGINS[tid]
        tr.write_line(f"{tid[-88:]:88s} {e:10.2e} {r:10.2e} {n:4d}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        try:
            json.dump(MARGINS, open(os.path.join(out, "parity_margins.json"), "w"), indent=1)
        except OSError:
            pass
