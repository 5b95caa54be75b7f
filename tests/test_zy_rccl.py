"""The RCCL calls of the row-sharded propagation, executed: `dist.all_gather_start`'s asynchronous
all_gather_into_tensor (alone and inside a whole sharded forward pass) and the C entry point dg_allgather_rows, each on
a ONE-rank communicator (a one-GPU box cannot
hold two ranks: RCCL refuses two ranks on one device).  The multi-rank control flow is covered over gloo
(tests/test_dist_cpu.py) and with emulated ranks (tests/test_gpu_parity.py).  Runs in a process of its own and late in
the collection order; a box on which no communicator can be made skips."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_call_sites_on_a_single_rank_communicator():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "rccl_single_rank.py")
    try:
        res = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL did not come up within 150 s on this box")
    if res.returncode == 3:
        pytest.skip("no RCCL communicator on this box: " + res.stdout.strip().splitlines()[-1])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    for what in ("torch.distributed", "sharded forward with", "sharded forward recorded as a graph", "dg_allgather_rows"):
        assert "RCCL-OK " + what in res.stdout, res.stdout
