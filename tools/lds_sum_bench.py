"""The last layer's user rows on the Industry shape (rank 0 of a W-rank job): one walk over all graphs that writes the
two layer sums (DG_ACC_SHARED) against a layer sum per graph followed by dg_sum_tables.  HIP events, median of 30.
Usage: python tools/lds_sum_bench.py [W ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgraph_amd import data as dgdata, ops  # noqa: E402
from diffgraph_amd._lib import DG_ACC_REBUILD, DG_ACC_SHARED, DG_IN_PRESCALED, DG_ROWS_FULL_ONLY  # noqa: E402
from diffgraph_amd.dist import HipBackend, shard_bipartite, user_ranges  # noqa: E402

dev = torch.device("cuda:0")
wl = dgdata.rec_workload("industry")
inter = [wl.target] + wl.behaviours


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


for w in [int(a) for a in sys.argv[1:]] or [1, 8]:
    u0, u1 = user_ranges(wl.n_user, w)[0]
    n_loc, n = u1 - u0, u1 - u0 + wl.n_item
    be = HipBackend(dev)
    graphs = [be.make_graph(shard_bipartite(u, i, wl.n_user, wl.n_item, u0, u1), n) for (u, i) in inter]
    G = len(graphs)
    e0 = torch.randn(n, 64, device=dev)
    x = [torch.randn(n, 64, device=dev) for _ in range(G)]
    part = [torch.empty(wl.n_item, 64, device=dev) for _ in range(G)]
    sums = [torch.empty(n, 64, device=dev) for _ in range(G)]
    ui, het, het2 = (torch.empty(n, 64, device=dev) for _ in range(3))
    flags = DG_IN_PRESCALED | DG_ACC_REBUILD | DG_ROWS_FULL_ONLY

    def task(g, out, fl):
        return ops.LayerTask(graph=graphs[g], x=x[g], e_out=None, acc_in=e0, acc_out=out, flags=fl, slope=0.5,
                             partial_out=part[g], partial_from=n_loc)
    fused = ops.prepare_gcn_layer_batched([task(g, ui if g == 0 else het, flags | DG_ACC_SHARED) for g in range(G)])
    per_graph = ops.prepare_gcn_layer_batched([task(g, sums[g], flags) for g in range(G)])
    add = ops.prepare_sum_tables(sums[1:], het2)

    def separate():
        per_graph()
        add()
    t_f, t_p, t_s = timed(fused), timed(per_graph), timed(separate)
    torch.cuda.synchronize()
    same = torch.equal(ui[:n_loc], sums[0][:n_loc]) and torch.equal(het[:n_loc], het2[:n_loc])
    nbytes = n_loc * (256 + G * (64 + 256) + 512)
    print(f"W={w}: {n_loc} user rows x {G} graphs: one walk {t_f:.1f} us ({nbytes / t_f / 1e6:.2f} TB/s of what it must move)   "
          f"per graph {t_p:.1f} us + sum = {t_s:.1f} us   same bits: {same}", flush=True)
