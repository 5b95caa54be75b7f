"""bench.py's contract on a tiny workload: one JSON line with the agreed keys, the roofline and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_training_step_line_on_the_tiny_workload():
    r = _run("--workload", "tiny", "--steps", "3", "--warmup", "2", "--other-workloads", "tiny", "--one-gpu-forward", "tiny")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 2 and r["unit"] == "edges/s" and r["value"] > 0
    assert r["vs_baseline"] is None and r["dtype"] == "f32" and "workload" in r["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "frac_compulsory", "bytes_compulsory",
              "per_layer"):
        assert k in r["roofline"], k
    modes = {(p["mode"], p["layer"]) for p in r["roofline"]["per_layer"]}
    assert {("inference", 1), ("inference", 2), ("training step (forward)", 1), ("training step (forward)", 2)} <= modes
    for p in r["roofline"]["per_layer"]:      # what a launch has to move never exceeds the algorithmic model of a full layer
        assert 0 < p["bytes_compulsory"] <= p["bytes_algorithmic"] and p["ms_per_launch"] > 0
    assert r["roofline"]["bound"] == "hbm" and abs(r["roofline"]["frac"] - r["roofline"]["achieved"] / r["roofline"]["peak"]) < 1e-12
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in r["cpu_baseline"], k
    assert r["cpu_baseline"]["kind"] == "port" and r["config"]["loss"] == r["config"]["loss"]   # a number, not NaN
    # a like-for-like CPU figure for `value` (the whole step), the other workloads' layer fractions, and the one-GPU
    # partner of the multi-GPU experiment -- so that two lines of different --gpus can be compared without a third run
    assert r["cpu_baseline"]["step_value"] > 0 and "step_sample" in r["cpu_baseline"]
    ow = r["roofline"]["other_workloads"]
    assert 0 < ow["tiny"]["frac"] and {p["layer"] for p in ow["tiny"]["per_layer"]} == {1, 2}
    assert r["config"]["tiny_1gpu_edges_per_s"] > 0 and "value` is the tiny workload" in r["config"]["step"]


def test_eager_step_gives_the_same_kind_of_line():
    r = _run("--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--other-workloads", "",
             "--one-gpu-forward", "")
    assert r["config"]["launch"] == "eager" and "cpu_baseline" not in r and r["value"] > 0


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself and relays rank 0's
    line.  (Both ranks share this box's one GPU and exchange over gloo: a rehearsal of the control flow, not a
    measurement.)"""
    r = _run("--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", env={"DG_DIST_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["value"] > 0
    assert r["config"]["speedup_over_1gpu"] > 0 and "x2" in r["config"]["parallelism"]
    assert r["config"]["same_workload_1gpu_edges_per_s"] > 0 and "one-GPU partner" in r["config"]["value_is"]


@pytest.mark.parametrize("flags, launch", [((), "hip-graph replay"), (("--no-graph",), "eager")])
def test_nc_training_step_leg(flags, launch):
    """`--workload nc_train_dblp`: one DiffGraph_NC training step on the committed DBLP data, replayed as a HIP graph
    by default; the same kind of line, finite losses."""
    r = _run("--workload", "nc_train_dblp", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", *flags)
    assert r["config"]["launch"] == launch and r["unit"] == "edges/s" and r["value"] > 0 and r["n_gpus"] == 1
    assert all(x == x and abs(x) < 1e3 for x in r["config"]["loss"])
    # the as-coded DBLP graphs (3 x the 11 113-edge co-author graph) are sparse: no bitmap graph, nothing for the matrix cores
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and "hbm_secondary" in rf and rf["bitmap_graphs"] in (0, 2)
