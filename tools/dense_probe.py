"""Times one fused NC layer over the DBLP graphs (APA sparse, APCPA / APTPA dense) and its dense part alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diffgraph_amd as dg
from diffgraph_amd import ops
from diffgraph_amd._lib import DG_IN_PRESCALED, DG_OUT_PRESCALE, DG_OUT_NORM, DG_ACC_REBUILD
from tests.test_oracle_golden import _dblp_graphs
dev = torch.device("cuda", 0)
NA, apa, apcpa, aptpa, features = _dblp_graphs()
graphs = [dg.CSRGraph.from_scipy(m) for m in (apa, apcpa, aptpa)]
print([g.schedule()["dense_splits"] for g in graphs])
def timeit(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
x = torch.randn(NA, 64, device=dev) * 0.1
prev = [torch.randn(NA, 64, device=dev) for _ in graphs]
outs = [torch.empty(NA, 64, device=dev) for _ in graphs]
sums = [torch.empty(NA, 64, device=dev) for _ in graphs]
for label, sel in (("all three", [0, 1, 2]), ("dense two", [1, 2]), ("APCPA", [1]), ("APTPA", [2]), ("APA", [0])):
    l1 = ops.prepare_gcn_layer_batched([ops.LayerTask(graphs[g], x, outs[g], None, None, DG_OUT_NORM | DG_OUT_PRESCALE, 0.5) for g in sel])
    l2 = ops.prepare_gcn_layer_batched([ops.LayerTask(graphs[g], prev[g], outs[g], x, sums[g], DG_IN_PRESCALED | DG_OUT_NORM | DG_OUT_PRESCALE | DG_ACC_REBUILD, 0.5) for g in sel])
    print(f"{label:10s} layer1 {timeit(l1):6.1f} us   layer2 {timeit(l2):6.1f} us", flush=True)
