"""The two single-step denoiser kernels of the recommendation model's training step -- the fused q_sample + denoiser +
loss launch and its backward (with the weight gradients' partial sums) -- alone, on tables of the Tmall shape: recorded
ten at a time as a HIP graph, replays timed by HIP events (median of 30).   Usage: python tools/loss_bench.py [workload]"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffgraph_amd as dg  # noqa: E402
from diffgraph_amd import data as dgdata, ops  # noqa: E402

dev = torch.device("cuda", 0)
wl = dgdata.rec_workload(sys.argv[1] if len(sys.argv) > 1 else "tmall")
graphs = [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item) for (u, i) in [wl.target]]
handler = SimpleNamespace(userNum=wl.n_user, itemNum=wl.n_item, target_adj=graphs[0], behavior_mats=[])
model = dg.rec.HGDM(handler, device=dev).eval()
gd, nu, n = model.diffusion_process, wl.n_user, wl.n
spans = [ops.LossSpan(dn.params(bool(dn.norm)), b, m, 0.0) for dn, b, m in
         ((model.usr_denoiser, 0, nu), (model.item_denoiser, nu, wl.n_item))]
params = [sp.dn for sp in spans]
ops.refresh_time_bias(params, gd.steps)
ops.wgrad_operand_tables(params, gd.steps)
sa, sb = gd._dev("sqrt_alphas_cumprod", dev), gd._dev("sqrt_one_minus_alphas_cumprod", dev)
lw = gd._dev("loss_weight", dev, torch.float64)
ui, het, noise, d_out, d_add = (torch.randn(n, 64, device=dev) for _ in range(5))
ts = torch.randint(0, gd.steps, (n,), device=dev)
g_rl = torch.rand(n, dtype=torch.float64, device=dev)


def timed(fn, reps=30, per_graph=10):
    """us per call: `per_graph` calls recorded as one HIP graph (no host time between the launches), replays timed by events"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(per_graph):
            fn()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / per_graph)
    return float(np.median(out))


fwd = timed(lambda: ops.diffusion_loss_fwd_spans(spans, ui, het, noise, ts, sa, sb, lw, validate=False))
bwd = timed(lambda: ops.diffusion_loss_bwd_spans(spans, ui, het, noise, ts, sa, sb, lw, g_rl, d_out, d_target_add=d_add,
                                                 fused_wgrad=True))
print(f"{wl.name if hasattr(wl, 'name') else ''} n={n}: loss forward {fwd:.1f} us   loss backward {bwd:.1f} us", flush=True)
