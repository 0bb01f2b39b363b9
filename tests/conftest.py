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


OTHER: dict[str, list[float]] = {}       # "test id [label]" -> [worst value, tolerance, comparisons]


def record_margin(err: float, tol: float, label: str = "") -> float:
    """Record an achieved margin against a tolerance that is NOT the 1e-5 peak-relative bar (metrics in dB, the
    statistic's float32 block sums, the noise-reduction row edges ...): printed with the parity table and written to
    gpurun_out/parity_margins.json under "other"."""
    tid = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + (f" [{label}]" if label else "")
    m = OTHER.setdefault(tid, [0.0, float(tol), 0])
    if float(err) / max(float(tol), 1e-300) >= m[0] / max(m[1], 1e-300):      # keep the comparison closest to its tolerance
        m[0], m[1] = float(err), float(tol)
    m[2] += 1
    return float(err)


def db_close(got, want, tol: float = 2e-4, label: str = "dB") -> bool:
    """|got - want| <= tol for metrics in dB (rssi / signal power / snr: 2e-4 dB absolute = 4.6e-5 relative in power --
    the float32 power sum is taken in another order than numpy's pairwise mean); the achieved margin is recorded."""
    import numpy as np

    err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))))
    record_margin(err, tol, label)
    return err <= tol


def pytest_terminal_summary(terminalreporter):
    if OTHER:
        tr = terminalreporter
        tr.section("achieved margins against other stated tolerances")
        tr.write_line(f"{'test [what]':88s} {'worst':>10s} {'tolerance':>10s} {'n':>4s}")
        for tid in sorted(OTHER):
            e, t, n = OTHER[tid]
            tr.write_line(f"{tid[-88:]:88s} {e:10.2e} {t:10.2e} {int(n):4d}")
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
            json.dump(dict(MARGINS, other=OTHER), open(os.path.join(out, "parity_margins.json"), "w"), indent=1)
        except OSError:
            pass
