"""Does the order of the graphs in the fused layer launch matter?  Times the Tmall layer with the tasks in several orders."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diffgraph_amd as dg
from diffgraph_amd import data as dgdata, ops
from diffgraph_amd._lib import DG_IN_PRESCALED, DG_OUT_PRESCALE, DG_ACC_REBUILD
dev = torch.device("cuda", 0)
wl = dgdata.rec_workload(sys.argv[1] if len(sys.argv) > 1 else "tmall")
n = wl.n
def timeit(fn, reps=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
e0 = torch.randn(n, 64, device=dev)
graphs = [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item) for (u, i) in [wl.target] + wl.behaviours]
names = ["buy"] + wl.behaviour_names
prev = [torch.randn(n, 64, device=dev) for _ in graphs]
sums = [torch.empty(n, 64, device=dev) for _ in graphs]
outs = [torch.empty(n, 64, device=dev) for _ in graphs]
for order in [(0, 1, 2, 3), (1, 0, 3, 2), (2, 3, 0, 1), (3, 2, 0, 1), (1, 3, 0, 2)]:
    l1 = [ops.LayerTask(graphs[g], e0, outs[g], None, None, DG_OUT_PRESCALE, 0.5) for g in order]
    l2 = [ops.LayerTask(graphs[g], prev[g], outs[g], e0, sums[g], DG_IN_PRESCALED | DG_ACC_REBUILD, 0.5) for g in order]
    f1, f2 = ops.prepare_gcn_layer_batched(l1), ops.prepare_gcn_layer_batched(l2)
    print([names[g] for g in order], f"layer1 {timeit(f1):6.1f}  layer2(train) {timeit(f2):6.1f} us", flush=True)
# each graph as a launch of its own, back to back
fs = [ops.prepare_gcn_layer_batched([ops.LayerTask(graphs[g], prev[g], outs[g], e0, sums[g], DG_IN_PRESCALED | DG_ACC_REBUILD, 0.5)]) for g in range(4)]
print("four launches, layer2(train):", timeit(lambda: [f() for f in fs]))
