"""Times the fused propagation (2 layers, all graphs) of a Rec workload for several schedule knobs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import diffgraph_amd as dg
from diffgraph_amd import data as dgdata
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "tmall"
knobs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(0, 0)]
wl = dgdata.rec_workload(name)
for gm, sl in knobs:
    graphs = [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item, group_max=gm, seg_len=sl) for (u, i) in [wl.target] + wl.behaviours]
    sched = [g.schedule() for g in graphs]
    handler = SimpleNamespace(userNum=wl.n_user, itemNum=wl.n_item, target_adj=graphs[0], behavior_mats=graphs[1:])
    model = dg.rec.HGDM(handler, device=dev).eval()
    ev = []
    def hook(phase, layer, tasks=None):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((phase, layer, e))
    with torch.no_grad():
        for _ in range(5):
            model.propagate(model.embedding_table)
        model.layer_hook = hook
        for _ in range(30):
            model.propagate(model.embedding_table)
        torch.cuda.synchronize()
    t = {0: [], 1: []}
    for k in range(0, len(ev), 2):
        t[ev[k][1]].append(ev[k][2].elapsed_time(ev[k + 1][2]) * 1e3)
    print(f"group_max={gm} seg_len={sl}: layer0 {np.median(t[0]):.1f} us  layer1 {np.median(t[1]):.1f} us   "
          f"hub/wave/group/seg per graph: {[(s['n_hub'], s['n_wave'], s['n_group'], s['n_seg']) for s in sched]}", flush=True)
