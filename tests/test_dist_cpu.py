"""The row-sharded propagation (diffgraph_amd.dist) with world_size 2 over gloo on CPU: partition,
global degree norms, the one exchange per layer and the fixed summation order, against the
single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffgraph_amd import data as dgdata
from diffgraph_amd.dist import RowBlockPropagation, ShardedRecPropagation, shard_bipartite, user_ranges
from oracle import ref_cpu as O
from tests.cpu_backend import CpuBackend
from tests.helpers import assert_rows_close, edges_t


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _workload():
    n_user, n_item = 301, 23
    inter = [dgdata.synth_interactions(n_user, n_item, m, seed) for m, seed in ((900, 1), (2500, 2), (400, 3))]
    ue = torch.from_numpy(dgdata.xavier_uniform(n_user, 64, seed=5))
    ie = torch.from_numpy(dgdata.xavier_uniform(n_item, 64, seed=6))
    return n_user, n_item, inter, ue, ie


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    n_user, n_item, inter, ue, ie = _workload()
    prop = ShardedRecPropagation(inter, n_user, n_item, rank, world, CpuBackend())
    e0 = torch.cat([ue[prop.u0:prop.u1], ie])
    ui, het = prop.forward(e0)
    q.put((rank, prop.u0, prop.u1, ui.numpy(), het.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_users_and_keeps_global_degrees():
    assert user_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    n_user, n_item, inter, *_ = _workload()
    u, i = inter[1]
    total = 0
    for (u0, u1) in user_ranges(n_user, 4):
        lg = shard_bipartite(u, i, n_user, n_item, u0, u1)
        total += len(lg.src)
        np.testing.assert_array_equal(lg.degrees[: u1 - u0], np.bincount(u, minlength=n_user)[u0:u1])
        np.testing.assert_array_equal(lg.degrees[u1 - u0:], np.bincount(i, minlength=n_item))
        assert lg.src.max() < (u1 - u0) + n_item and lg.n_edges_global == 2 * len(u)
    assert total == 2 * len(u)


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_forward_matches_oracle(world):
    n_user, n_item, inter, ue, ie = _workload()
    tu, ti, hu, hi = O.rec_forward(edges_t(*inter[0], n_user), [edges_t(*b, n_user) for b in inter[1:]], ue, ie, 2)
    if world == 1:
        prop = ShardedRecPropagation(inter, n_user, n_item, 0, 1, CpuBackend())
        ui, het = prop.forward(torch.cat([ue, ie]))
        results = [(0, 0, n_user, ui.numpy(), het.numpy())]
    else:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        results = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    ui_users = np.concatenate([r[3][: r[2] - r[1]] for r in results])
    het_users = np.concatenate([r[4][: r[2] - r[1]] for r in results])
    assert_rows_close(ui_users, tu, what="target users")
    assert_rows_close(het_users, hu, what="heter users")
    for r in results:  # item rows: replicated, identical on every rank
        assert_rows_close(r[3][r[2] - r[1]:], ti, what="target items")
        assert_rows_close(r[4][r[2] - r[1]:], hi, what="heter items")
        np.testing.assert_array_equal(r[3][r[2] - r[1]:], results[0][3][results[0][2] - results[0][1]:])


# ---------------------------------------------------------------- general row-block partition (all-gather per layer)
def _rb_workload():
    n_user, n_item = 157, 140      # neither side small: the case the item-partial-sum scheme does not cover
    inter = [dgdata.synth_interactions(n_user, n_item, m, seed) for m, seed in ((700, 1), (1900, 2), (300, 3))]
    ue = torch.from_numpy(dgdata.xavier_uniform(n_user, 64, seed=5))
    ie = torch.from_numpy(dgdata.xavier_uniform(n_item, 64, seed=6))
    return n_user, n_item, inter, ue, ie


def _rb_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    n_user, n_item, inter, ue, ie = _rb_workload()
    graphs = [dgdata.bipartite_edges(u, i, n_user) for (u, i) in inter]
    prop = RowBlockPropagation(graphs, n_user + n_item, rank, world, CpuBackend())
    ui, het = prop.forward(torch.cat([ue, ie])[prop.r0:prop.r1])
    q.put((rank, prop.r0, prop.r1, ui.numpy(), het.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_block_forward_matches_oracle(world):
    n_user, n_item, inter, ue, ie = _rb_workload()
    tu, ti, hu, hi = O.rec_forward(edges_t(*inter[0], n_user), [edges_t(*b, n_user) for b in inter[1:]], ue, ie, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rb_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [min(n_user + n_item, k * -(-(n_user + n_item) // world)) for k in range(world)]
    assert results[-1][2] == n_user + n_item
    ui = np.concatenate([r[3] for r in results])
    het = np.concatenate([r[4] for r in results])
    assert_rows_close(ui, torch.cat([tu, ti]), what="row-block target")
    assert_rows_close(het, torch.cat([hu, hi]), what="row-block heter")


# ---------------------------------------------------------------- the bench leg for --gpus N, end to end over gloo
def _bench_worker(rank, world, port, q):
    from types import SimpleNamespace
    from diffgraph_amd import dist as dgdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    calls = []

    class Recording(CpuBackend):     # the order of the launches and exchanges of one forward pass, as this rank issues them
        def prepare_layer(self, tasks):
            kind = "partial" if tasks[0]["flags"] & dgdist.DG_ROWS_PARTIAL_ONLY else "full" if tasks[0]["flags"] & dgdist.DG_ROWS_FULL_ONLY else "layer"
            fn = super().prepare_layer(tasks)
            return lambda: (calls.append(kind), fn())

        def prepare_finalize(self, tasks):
            fn = super().prepare_finalize(tasks)
            return lambda: (calls.append("finalize"), fn())
    start, wait = dgdist.all_gather_start, dgdist._Pending.wait
    dgdist.all_gather_start = lambda *a, **k: (calls.append("start"), start(*a, **k))[1]
    dgdist._Pending.wait = lambda self: (calls.append("wait"), wait(self))[1]
    args = SimpleNamespace(workload="tiny", steps=3, warmup=1)
    res = dgdist.bench_sharded(args, rank, world, "cpu", backend=Recording())
    q.put((rank, res, calls[:10]))


def test_bench_sharded_runs_end_to_end_over_gloo():
    """bench.py's multi-rank leg (diffgraph_amd.dist.bench_sharded) with two gloo ranks on host tensors: one JSON
    record from rank 0 with the whole-job rate and the one-rank comparison, and the split-row order of a forward
    pass -- item partial sums, exchange started, user rows beside it, next layer's partial sums ahead of this
    layer's wait + finalize."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict((r[0], r[1:]) for r in (q.get(timeout=180) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res, calls = results[0]
    assert results[1][0] is None                      # only rank 0 reports
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["value"] > 0 and res["unit"] == "edges/s"
    assert res["config"]["speedup_over_1gpu"] > 0 and res["config"]["parallelism"] == "user-row shards x2"
    assert res["roofline"]["ms_per_launch"] > 0
    # forward pass of 2 layers: A1 start (B1 | A2 start) wait F1 B2 wait F2
    assert calls == ["partial", "start", "full", "partial", "start", "wait", "finalize", "full", "wait", "finalize"]
    assert results[1][1] == calls
