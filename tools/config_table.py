"""One fused propagation layer (all graphs of the configuration in one launch) for every configuration of the
benchmark: HIP events round each layer launch, 20 warm-up + 100 timed forward passes, median and p10 / p90 per layer,
directed edges/s and algorithmic GB/s (4 nnz + 1032 N bytes per graph-layer) against the 8 TB/s HBM peak.
Usage: python tools/config_table.py out.json [config ...]     (retail_rocket tmall ijcai ijcai_readme dblp industry)"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffgraph_amd as dg  # noqa: E402
from diffgraph_amd import data as dgdata  # noqa: E402

DEV = torch.device("cuda", 0)
WARM, TIMED = 20, 100


def rec_model(name):
    wl = dgdata.rec_workload(name)
    graphs = [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item) for (u, i) in [wl.target] + wl.behaviours]
    handler = SimpleNamespace(userNum=wl.n_user, itemNum=wl.n_item, target_adj=graphs[0], behavior_mats=graphs[1:])
    model = dg.rec.HGDM(handler, device=DEV).eval()
    return model, graphs, wl.n, lambda: model.propagate(model.embedding_table), wl.note


def dblp_model():
    import scipy.sparse as sp
    from diffgraph_amd import metapath as mp
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dblp_graph.npz"))
    NA, NP, NC, NT = (int(v) for v in g["counts"])
    mk = lambda e, shape: sp.coo_matrix((np.ones(len(e), np.float32), (e[:, 0], e[:, 1])), shape=shape).tocsr()
    AP, PC, PT = mk(g["ap"], (NA, NP)), mk(g["pc"], (NP, NC)), mk(g["pt"], (NP, NT))
    D = mp.DevicePattern.from_scipy
    pats = [mp.compose([D(AP), D(AP.T)]), mp.compose([D(AP), D(PC), D(PC.T), D(AP.T)]), mp.compose([D(AP), D(PT), D(PT.T), D(AP.T)])]
    adjs = [dg.CSRGraph.from_scipy(p.to_scipy()) for p in pats]
    torch.manual_seed(5)
    model = dg.nc.HGDM(334).eval()
    embed = torch.randn(NA, 64, device=DEV) * 0.1
    return model, adjs, NA, lambda: model.propagate(adjs, embed), "DBLP author graphs APA / APCPA / APTPA composed on the device"


def measure(name):
    model, graphs, n, run, note = dblp_model() if name == "dblp" else rec_model(name)
    events = []

    def hook(phase, layer, tasks=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        events.append((layer, e))
    with torch.no_grad():
        for _ in range(WARM):
            run()
        torch.cuda.synchronize()
        model.layer_hook = hook
        for _ in range(TIMED):
            run()
        torch.cuda.synchronize()
        model.layer_hook = None
    per_layer = {}
    for k in range(0, len(events), 2):
        per_layer.setdefault(events[k][0], []).append(events[k][1].elapsed_time(events[k + 1][1]) * 1e3)
    nnz = [g.num_edges() for g in graphs]
    layer_bytes = sum(4 * e + 1032 * n for e in nnz)
    rows = []
    for layer in sorted(per_layer):
        t = np.array(per_layer[layer])
        med = float(np.median(t))
        rows.append({"layer": layer, "median_us": med, "p10_us": float(np.percentile(t, 10)), "p90_us": float(np.percentile(t, 90)),
                     "edges_per_s": sum(nnz) / (med * 1e-6), "algorithmic_GBps": layer_bytes / (med * 1e-6) / 1e9,
                     "frac_of_8TBps": layer_bytes / (med * 1e-6) / 1e9 / 8000.0})
    return {"config": name, "note": note, "nodes": n, "graphs": len(graphs), "directed_edges_per_graph": nnz,
            "algorithmic_bytes_per_layer": layer_bytes, "layers": rows}


def main():
    out = sys.argv[1]
    names = sys.argv[2:] or ["retail_rocket", "tmall", "tmall_skewed", "ijcai", "ijcai_readme", "dblp", "industry"]
    res = []
    for nm in names:
        r = measure(nm)
        res.append(r)
        for row in r["layers"]:
            print(f"{nm:14s} layer {row['layer']}: {row['median_us']:8.1f} us (p10 {row['p10_us']:.1f}, p90 {row['p90_us']:.1f})  "
                  f"{row['edges_per_s'] / 1e9:6.1f} G edges/s  {row['algorithmic_GBps']:7.0f} GB/s  frac {row['frac_of_8TBps']:.3f}", flush=True)
    with open(out, "w") as f:
        json.dump({"protocol": f"HIP events round each fused layer launch, {WARM} warm-up + {TIMED} timed forward passes", "configs": res}, f, indent=1)


if __name__ == "__main__":
    main()
