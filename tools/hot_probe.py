"""Bound for LDS reuse on a Rec workload: times the fused layer launch (all graphs) with the edges an LDS path would
serve taken out of the gather kernel -- user rows' edges to the H hottest items, and the rows of items with degree
>= THR -- to see what is left for the L2 gathers and the streamed epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diffgraph_amd as dg
from diffgraph_amd import data as dgdata, ops
from diffgraph_amd._lib import DG_IN_PRESCALED, DG_OUT_PRESCALE, DG_ACC_REBUILD
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "tmall"
wl = dgdata.rec_workload(name)
n, nu = wl.n, wl.n_user


def timeit(fn, reps=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


e0 = torch.randn(n, 64, device=dev)


def variant(label, hot_items, item_thr):
    graphs, kept = [], 0
    for (u, i) in [wl.target] + wl.behaviours:
        u = u.astype(np.int32); it = (i + nu).astype(np.int32)
        deg_item = np.bincount(i, minlength=wl.n_item)
        order = np.argsort(-deg_item, kind="stable")
        hot = np.zeros(wl.n_item, bool)
        hot[order[:hot_items]] = hot_items > 0
        keep_user_dir = ~hot[i]                      # user rows gather items: hot items come from LDS
        keep_item_dir = deg_item[i] < item_thr       # item rows gather users: long rows go to the window kernel
        src = np.concatenate([it[keep_user_dir], u[keep_item_dir]])
        dst = np.concatenate([u[keep_user_dir], it[keep_item_dir]])
        kept += src.size
        graphs.append(dg.CSRGraph(src, dst, n))
    prev = [torch.randn(n, 64, device=dev) for _ in graphs]
    sums = [torch.empty(n, 64, device=dev) for _ in graphs]
    outs = [torch.empty(n, 64, device=dev) for _ in graphs]
    l1 = [ops.LayerTask(g, e0, o, None, None, DG_OUT_PRESCALE, 0.5) for g, o in zip(graphs, outs)]
    l2i = [ops.LayerTask(g, p, None, e0, s, DG_IN_PRESCALED | DG_ACC_REBUILD, 0.5) for g, p, s in zip(graphs, prev, sums)]
    l2t = [ops.LayerTask(g, p, o, e0, s, DG_IN_PRESCALED | DG_ACC_REBUILD, 0.5) for g, p, s, o in zip(graphs, prev, sums, outs)]
    f1, f2, f3 = (ops.prepare_gcn_layer_batched(t) for t in (l1, l2i, l2t))
    print(f"{label:46s} edges {kept:8d}  layer1 {timeit(f1):6.1f}  layer2(inference) {timeit(f2):6.1f}  layer2(train) {timeit(f3):6.1f} us",
          flush=True)


INF = 1 << 30
variant("all edges", 0, INF)
if len(sys.argv) > 2 and sys.argv[2] == "quick":
    variant("no edges", wl.n_item, 0)
    sys.exit(0)
variant("user side: top 255 items out", 255, INF)
variant("user side: top 383 items out", 383, INF)
variant("user side: top 639 items out", 639, INF)
variant("item side: rows >= 375 out", 0, 375)
variant("item side: rows >= 128 out", 0, 128)
variant("both: top 383 / rows >= 375", 383, 375)
variant("both: top 639 / rows >= 128", 639, 128)
variant("both: top 639 / rows >= 64", 639, 64)
variant("no edges", wl.n_item, 0)
