"""pv item-row direction of Tmall: cost of the hub / wave / group row classes separately."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diffgraph_amd as dg
from diffgraph_amd import data as dgdata, ops
from diffgraph_amd._lib import DG_IN_PRESCALED, DG_OUT_PRESCALE
dev = torch.device("cuda", 0)
wl = dgdata.rec_workload("tmall")
n = wl.n
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
x = torch.randn(n, 64, device=dev)
acc_in = torch.randn_like(x)
u, i = wl.behaviours[0]
u = u.astype(np.int32); it = (i + wl.n_user).astype(np.int32)
deg_item = np.bincount(it, minlength=n)
deg_user = np.bincount(u, minlength=n)
def run(label, src, dst):
    g = dg.CSRGraph(src, dst, n)
    t = ops.LayerTask(g, x, torch.empty_like(x), acc_in, torch.empty_like(x), DG_IN_PRESCALED | DG_OUT_PRESCALE, 0.5)
    s = g.schedule()
    print(f"{label:34s} edges {src.size:8d} {timeit(lambda: ops.gcn_layer_batched([t])):7.1f} us  hub/wave/group/seg {s['n_hub']},{s['n_wave']},{s['n_group']},{s['n_seg']}", flush=True)
d = deg_item[it]
run("item rows: all", u, it)
run("item rows: hub rows only (>256)", u[d > 256], it[d > 256])
run("item rows: wave rows only (49..256)", u[(d > 48) & (d <= 256)], it[(d > 48) & (d <= 256)])
run("item rows: group rows only (<=48)", u[d <= 48], it[d <= 48])
run("item rows: deg<=13", u[d <= 13], it[d <= 13])
run("item rows: 14..48", u[(d > 13) & (d <= 48)], it[(d > 13) & (d <= 48)])
du = deg_user[u]
run("user rows: all", it, u)
run("no edges", u[:0], it[:0])
