"""A/B of the propagation over a graph set (one row walk over all graphs) against the task-per-graph path, same
process, same tables: forward layers with and without the activations a backward pass needs, and the backward steps.
Usage: python tools/set_bench.py [workload] [group_max,seg_len ...]      (HIP events, median of 50 after 10 warm-up)"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffgraph_amd as dg  # noqa: E402
from diffgraph_amd import data as dgdata  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "tmall"
knobs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(0, 0)]
wl = dgdata.rec_workload(name)
graphs = [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item) for (u, i) in [wl.target] + wl.behaviours]
handler = SimpleNamespace(userNum=wl.n_user, itemNum=wl.n_item, target_adj=graphs[0], behavior_mats=graphs[1:])
model = dg.rec.HGDM(handler, device=dev).eval()
nnz = sum(g.num_edges() for g in graphs)
algo = sum(4 * g.num_edges() + 1032 * wl.n for g in graphs)
ev = []


def hook(phase, layer, tasks=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev.append((phase, layer, e))


def timed(fn, reps=50, warm=10):
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


FLUSH = None
if os.environ.get("DG_BENCH_FLUSH_MB"):      # a table of that size rewritten before every pass: the passes start with cold caches
    FLUSH = torch.empty(int(os.environ["DG_BENCH_FLUSH_MB"]) << 18, dtype=torch.float32, device=dev)


def layers(keep):
    """median us of layer 0 / layer 1 launches (all kernels of the layer) and of the whole forward"""
    global ev
    model.layer_hook = None

    def run():
        if FLUSH is not None:
            FLUSH.add_(1.0)
        return model._propagate(model.embedding_table.detach(), keep=keep)
    whole = timed(run)
    ev = []
    model.layer_hook = hook
    for _ in range(40):
        run()
    torch.cuda.synchronize()
    model.layer_hook = None
    t = {0: [], 1: []}
    for k in range(0, len(ev), 2):
        t[ev[k][1]].append(ev[k][2].elapsed_time(ev[k + 1][2]) * 1e3)
    return float(np.median(t[0])), float(np.median(t[1])), whole


def backward_time():
    e0 = model.embedding_table
    d_ui, d_het = torch.randn_like(e0), torch.randn_like(e0)
    model.defer_embedding_grad = True      # (as the training step runs it: no final sum of the gradient parts)

    def run():
        ui, het = model.propagate(e0)
        torch.autograd.backward([ui, het], [d_ui, d_het])
    both = timed(run)
    model.defer_embedding_grad = False
    return both


with torch.no_grad():
    for fuse, kn in [(False, (0, 0))] + [(True, k) for k in knobs]:
        model.fuse_graphs = fuse
        if fuse:
            os.environ["DG_SET_SPLIT"] = str(kn[2]) if len(kn) > 2 else "0"      # third knob: sides dealt to separate XCDs
            model._gset, model._gset_graphs = dg.GraphSet(graphs, group_max=kn[0], seg_len=kn[1]), list(graphs)
            what = f"graph set group_max={kn[0]} seg_len={kn[1]} {model._gset.counts()}"
        else:
            what = "task per graph"
        for keep in (False, True):
            l0, l1, whole = layers(keep)
            print(f"{name} {what}: keep={keep}: layer1 {l0:.1f} us  layer2 {l1:.1f} us  forward {whole:.1f} us  "
                  f"frac {algo / ((l0 + l1) / 2 * 1e-6) / 8e12:.3f}", flush=True)
for fuse, split in ((False, 0), (True, 0), (True, 1)):
    model.fuse_graphs = fuse
    if fuse:
        os.environ["DG_SET_SPLIT"] = str(split)
        model._gset = None
    print(f"{name} {'graph set' if fuse else 'task per graph'} split={split}: forward(keep) + backward {backward_time():.1f} us", flush=True)
