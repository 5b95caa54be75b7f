"""Where the Tmall layer's time goes: each graph alone, and its user-row / item-row halves alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diffgraph_amd as dg
from diffgraph_amd import data as dgdata, ops
from diffgraph_amd._lib import DG_IN_PRESCALED, DG_OUT_PRESCALE
dev = torch.device("cuda", 0)
wl = dgdata.rec_workload("tmall")
n = wl.n
def timeit(tasks_fn, reps=40):
    for _ in range(5): tasks_fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): tasks_fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
x = torch.randn(n, 64, device=dev)
names = ["buy"] + wl.behaviour_names
allg = []
for nm, (u, i) in zip(names, [wl.target] + wl.behaviours):
    u = u.astype(np.int32); it = (i + wl.n_user).astype(np.int32)
    variants = {"both": (np.concatenate([u, it]), np.concatenate([it, u])), "user rows": (it, u), "item rows": (u, it)}
    for vn, (src, dst) in variants.items():
        g = dg.CSRGraph(src, dst, n)
        e_out, acc_in, acc_out = torch.empty_like(x), torch.randn_like(x), torch.empty_like(x)
        t = ops.LayerTask(g, x, e_out, acc_in, acc_out, DG_IN_PRESCALED | DG_OUT_PRESCALE, 0.5)
        us = timeit(lambda: ops.gcn_layer_batched([t]))
        s = g.schedule()
        print(f"{nm:5s} {vn:10s} edges {src.size:8d}  {us:7.1f} us   hub/wave/group/seg {s['n_hub']},{s['n_wave']},{s['n_group']},{s['n_seg']}", flush=True)
        if vn == "both": allg.append(t)
print("all four graphs, one launch:", timeit(lambda: ops.gcn_layer_batched(allg)))
g0 = dg.CSRGraph(np.zeros(0, np.int32), np.zeros(0, np.int32), n)
acc_in = torch.randn_like(x)
t0 = ops.LayerTask(g0, x, torch.empty_like(x), acc_in, torch.empty_like(x), DG_IN_PRESCALED | DG_OUT_PRESCALE, 0.5)
print("no edges (epilogue only), one graph:", timeit(lambda: ops.gcn_layer_batched([t0])))

from diffgraph_amd.ops import BwdTask
bt = BwdTask(g0, None, x, acc_in, torch.empty_like(x), 0, 0.5)
print("rowwise (bwd epilogue only), one graph:", timeit(lambda: ops.gcn_layer_bwd_batched([bt])))
print("rowwise, four:", timeit(lambda: ops.gcn_layer_bwd_batched([bt, bt, bt, bt])))
print("no edges, four graphs:", timeit(lambda: ops.gcn_layer_batched([t0, t0, t0, t0])))
y = torch.empty_like(x)
print("torch copy 16MB:", timeit(lambda: y.copy_(x)))
print("sum_tables 3->1:", timeit(lambda: ops.sum_tables([x, acc_in, y])))
