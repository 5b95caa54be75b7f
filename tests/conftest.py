import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Writes what every row-wise comparison of the session measured (tests.helpers.PARITY_LOG) to
    gpurun_out/parity_report.json: per test and comparison, the tolerance asked for, the worst error in units of the
    row scale and the worst pure relative error."""
    try:
        from tests import helpers
    except Exception:
        return
    if not helpers.PARITY_LOG:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    gpu = any("gpu" in item.keywords for item in session.items)
    worst = {}
    for r in helpers.PARITY_LOG:
        k = (r["test"], r["what"])
        w = worst.setdefault(k, dict(r))
        w["worst_over_row_scale"] = max(w["worst_over_row_scale"], r["worst_over_row_scale"])
        w["worst_pure_relative"] = max(w["worst_pure_relative"], r["worst_pure_relative"])
    rows = sorted(worst.values(), key=lambda r: -r["worst_over_row_scale"])
    doc = {"note": "one row per (test, comparison) made through tests.helpers.assert_rows_close; worst_over_row_scale is "
                   "|a-b| / max(|b|, largest |b| of the row) (what the assertion bounds by rtol), worst_pure_relative is "
                   "|a-b| / |b| over entries >= 1e-3 of their row's largest", "comparisons": rows}
    with open(os.path.join(out, "parity_report_gpu.json" if gpu else "parity_report_cpu.json"), "w") as f:
        json.dump(doc, f, indent=1)
