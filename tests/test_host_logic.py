"""Host-side logic that needs no GPU: workload shapes, sequential metric sums, the CPU-baseline helpers, and the rule
that nothing shipped imports the oracle on its own."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from diffgraph_amd import data as dgdata
from diffgraph_amd.evaluate import _seq_sum

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workload_shapes_follow_the_published_counts():
    wl = dgdata.rec_workload("ijcai_readme")           # README.md: 17 435 x 35 920, 799 368 links
    links = [len(wl.target[0])] + [len(b[0]) for b in wl.behaviours]
    assert (wl.n_user, wl.n_item, sum(links)) == (17435, 35920, 799368)
    for (u, i) in [wl.target] + wl.behaviours:         # distinct pairs, inside the shape
        key = u.astype(np.int64) * wl.n_item + i
        assert len(np.unique(key)) == len(key) and u.max() < wl.n_user and i.max() < wl.n_item
    tiny = dgdata.rec_workload("tiny")
    assert (tiny.n_user, tiny.n_item) == (40, 60)
    s, d = dgdata.bipartite_edges(*tiny.target, tiny.n_user)
    assert len(s) == 2 * len(tiny.target[0]) and set(zip(s.tolist(), d.tolist())) == set(zip(d.tolist(), s.tolist()))


def test_sequential_sum_is_the_python_loop():
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 1023, 31882):
        v = rng.random(n) * rng.choice([1e-6, 1.0, 1e6], n)
        want = 0
        for x in v.tolist():
            want += x
        assert _seq_sum(v) == want


def test_cpu_baseline_helpers_run_and_agree_on_the_tiny_workload():
    sys.path.insert(0, ROOT)
    import bench
    wl = dgdata.rec_workload("tiny")
    res = bench.cpu_baseline(wl, budget_s=0.5)
    assert res["kind"] == "port" and res["value"] > 0 and res["value"] == max(res["index_add_value"], res["csr_value"])
    assert set(res["spmm_variants"]) == {"coo_sparse_mm", "csr_sparse_mm", "index_add"}
    assert all(isinstance(v, float) and v > 0 for v in res["spmm_variants"].values())


def test_nothing_shipped_imports_the_oracle_by_itself():
    """The package, bench.py and its legs may use oracle/ only inside their CPU-baseline code: importing them must
    not pull it in."""
    code = ("import sys; sys.path.insert(0, %r); import diffgraph_amd, diffgraph_amd.train, diffgraph_amd.dist, bench, bench_next; "
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; assert not bad, bad" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]


def test_f1_scores_equal_sklearn():
    """train_nc.f1_scores (confusion matrix) against sklearn.metrics.f1_score, which the reference calls
    (DiffGraph_NC/Utils/Utils.py:121-122): classes that never occur, classes only predicted, perfect and empty cases."""
    from sklearn.metrics import f1_score
    from diffgraph_amd.train_nc import f1_scores
    rng = np.random.default_rng(4)
    cases = [(rng.integers(0, 4, 300), rng.integers(0, 4, 300)), (rng.integers(0, 2, 50), rng.integers(0, 4, 50)),
             (rng.integers(1, 3, 40), rng.integers(0, 2, 40)), (np.arange(4), np.arange(4)), (np.zeros(7, int), np.ones(7, int))]
    for y, p in cases:
        macro, micro = f1_scores(torch.from_numpy(y), torch.from_numpy(p), 4)
        assert macro == pytest.approx(f1_score(y, p, average="macro"), abs=1e-12)
        assert micro == pytest.approx(f1_score(y, p, average="micro"), abs=1e-12)
