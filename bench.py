#!/usr/bin/env python3
"""bench.py -- DiffGraph hot path on MI355X: SpMM edges/s + achieved HBM GB/s per propagation layer.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tmall|ijcai|retail_rocket|industry]
    python bench.py --workload predict_ijcai|nc_dblp                                   (the other benchmark configurations)
    python bench.py --workload eval_tmall|eval_ijcai|negsample_tmall|metapath_dblp     (neighbouring stages)

One JSON line on stdout (rank 0).  A "step" is one pass of the hot path over one synthetic
batch: HGDM.cal_loss -- (1+B)*L fused propagation launches, q_sample + denoiser + SNR loss for
users and items, BPR glue.  At N=1 the workload is the Tmall-shaped graph (the configuration the
headline metric is quoted on); with N>1 ranks it is the Industry-shaped graph row-sharded over the ranks.
Inputs are resident in HBM before the timed region.  `value` = edges gathered by all SpMM
launches of the K steps / wall time (max over ranks).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def source_digest() -> str:
    """Digest of the kernel sources: a traffic file counts only for the kernels it was measured on (the loaded library
    carries the same digest and is refused when it differs: diffgraph_amd._lib.load)."""
    from diffgraph_amd import build
    return build.source_digest()


def measured_traffic(workload: str):
    """(bytes, source) -- HBM-side bytes per layer launch from the newest profiles/rNN_traffic.json, which
    tools/collect_profiles.sh writes from rocprofv3 --pmc passes of this same command (counters cannot be read from
    inside the run being timed).  A file measured on other kernel sources than the ones in the tree is refused:
    (None, why)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    if not files:
        return None, "no profiles/rNN_traffic.json"
    name = os.path.relpath(files[-1], ROOT)
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
        if doc.get("source_digest") != source_digest():
            return None, f"{name} was measured on other kernel sources (digest {doc.get('source_digest')}): rerun tools/collect_profiles.sh"
        return float(doc[workload]["traffic_bytes_per_layer_launch"]), f"{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, bytes per layer launch (FETCH_SIZE doubled on gfx950)"
    except (OSError, KeyError, ValueError) as exc:
        return None, f"{name}: {exc!r}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-workloads", default="ijcai", help="Rec workloads whose layer timings ride along in roofline.other_workloads "
                    "(comma list; '' for none)")
    ap.add_argument("--one-gpu-forward", default="industry", help="workload of the multi-GPU experiment (forward propagation) timed on "
                    "this one GPU as config.<name>_1gpu_edges_per_s ('' for none)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-side-overlap", action="store_true",
                    help="keep the step's short side launches on the main stream (what running them beside the propagation buys)")
    return ap.parse_args()


class LayerTimer:
    """HIP events around every propagation-layer launch, on the stream the kernels run on, keyed by (layer, which
    tables the launch reads and writes)."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def __call__(self, phase, layer, tasks=None):
        if not self.enabled:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if phase == "begin":
            self.records.append([layer, ev, None, layer_bytes(tasks)])
        else:
            self.records[-1][2] = ev

    def by_key(self):
        """{(layer, writes): {"ms": mean, "n": launches, "bytes": (algorithmic, compulsory)}}"""
        out = {}
        for layer, a, b, by in self.records:
            key = (layer, by[2])
            d = out.setdefault(key, {"ms": [], "bytes": by[:2]})
            d["ms"].append(a.elapsed_time(b))
        return {k: {"ms": float(np.mean(v["ms"])), "n": len(v["ms"]), "bytes": v["bytes"]} for k, v in out.items()}


def layer_bytes(tasks):
    """(algorithmic, compulsory, label) bytes of one fused layer launch over `tasks` (diffgraph_amd.ops.LayerTask).
    Algorithmic: 4 nnz + 1032 N per graph -- indices, one read of each source row, write E, read-modify-write the
    running sum -- whatever the launch's flags.  Compulsory: what THIS launch has to move given its flags: indices
    (+ per-edge norms while the input is not scaled yet), the 64-B row records (+ their norms), 12 B of per-row
    scalars, every DISTINCT table it reads once (source table, running sum or e0, the row's own input under
    DG_ACC_REBUILD) and every table it writes."""
    from diffgraph_amd._lib import DG_IN_PRESCALED
    algo = comp = 0
    reads, writes, label = {}, {}, set()
    for k, t in enumerate(tasks):
        n, nnz = t.graph.num_nodes(), t.graph.num_edges()
        algo += 4 * nnz + 1032 * n
        pre = bool(t.flags & DG_IN_PRESCALED)
        comp += (4 if pre else 8) * nnz
        # row records: one per graph and row -- or, in a launch over a graph set, one per row for all its graphs
        if not t.shared_rows or k == 0:
            comp += (64 if pre else 128) * n
        comp += (8 if t.shared_rows else 12) * n
        reads[t.x.data_ptr()] = 256 * n
        if t.acc_out is not None:
            reads[t.acc_in.data_ptr()] = 256 * n
            writes[t.acc_out.data_ptr()] = 256 * n       # (the behaviour graphs of a set share ONE sum table)
            label.add("sum")
        if t.e_out is not None:
            writes[t.e_out.data_ptr()] = 256 * n
            label.add("E")
    return algo, comp + sum(reads.values()) + sum(writes.values()), "+".join(sorted(label))


def time_layers(model, timer, step, iters: int, graphs):
    """Kernel-level timing for the roofline, in the process that timed the steps: HIP events around each fused layer
    launch of (a) the training step, launched eagerly (events cannot be read back from inside a replayed graph; the
    kernels and their arguments are the same), and (b) the inference-mode propagation (HGDM.predict's), back to back.
    A timed step runs its short side launches (random draws, time-bias tables, buffer clearing) BESIDE the first layer
    launch; here they stay in line on the one stream, so that an interval holds the layer kernels alone.  Returns one
    row per (mode, layer)."""
    timer.records = []
    timer.enabled = True
    overlap, model.side_overlap = model.side_overlap, False
    for _ in range(iters):
        torch.cuda._sleep(10_000_000)   # the GPU spins ~5 ms while the host queues the step: launches then run back to back,
        step()                          # and an interval holds kernel time, not the host's launch latency
    torch.cuda.synchronize()
    model.side_overlap = overlap
    train_layers = timer.by_key()
    timer.records = []
    with torch.no_grad():
        for _ in range(iters):
            model.propagate(model.embedding_table)
    torch.cuda.synchronize()
    timer.enabled = False
    infer_layers = timer.by_key()
    timer.records = []
    nnz = [g.num_edges() for g in graphs]

    def rows(d, mode):
        out = []
        for (layer, writes), v in sorted(d.items()):
            algo, comp = v["bytes"]
            sec = v["ms"] * 1e-3
            out.append({"mode": mode, "layer": layer + 1, "writes": writes, "launches": v["n"], "ms_per_launch": v["ms"],
                        "bytes_algorithmic": algo, "bytes_compulsory": comp, "frac": algo / sec / 1e9 / HBM_PEAK_GBS,
                        "frac_compulsory": comp / sec / 1e9 / HBM_PEAK_GBS, "spmm_edges_per_s": sum(nnz) / sec})
        return out
    return rows(train_layers, "training step (forward)") + rows(infer_layers, "inference")


def rec_step(workload: str, device):
    """(wl, graphs, model, timer, step): the model of a Rec workload with a LayerTimer hooked in and `step`, one
    training step as Coach.trainEpoch runs it (DiffGraph-Rec/main.py:114-128: cal_loss -- timesteps and noise drawn
    inside training_losses2, Model.py:352-353 --, backward, Adam) over a batch of uniform random triples, returning
    float64[4] = loss, bprLoss, regLoss, diff_loss."""
    import diffgraph_amd as dg
    from diffgraph_amd.train import training_step
    wl, graphs, model = build_rec(workload, device)
    gen = torch.Generator().manual_seed(1)
    B = model.args.batch
    # (uniform random triples; `--workload epoch_tmall` runs whole epochs over the real, popularity-skewed pairs)
    ancs = torch.randint(0, wl.n_user, (B,), generator=gen).to(device)
    poss = torch.randint(0, wl.n_item, (B,), generator=gen).to(device)
    negs = torch.randint(0, wl.n_item, (B,), generator=gen).to(device)
    timer = LayerTimer()
    model.layer_hook = timer
    opt = dg.optim.Adam(model, lr=1e-3, weight_decay=0)  # DiffGraph-Rec/main.py:117-119
    seed_vec = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=device)
    return wl, graphs, model, timer, lambda: training_step(model, opt, ancs, poss, negs, seed_vec)


def other_workload(workload: str, device, iters: int = 10):
    """The per-layer roofline rows of another Rec workload, measured the same way as the headline's (eager training
    steps behind a device-side spin, then inference-mode propagation): {"frac": mean over the forward launches of the
    training step, "per_layer": [...]}."""
    wl, graphs, model, timer, step = rec_step(workload, device)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rows = time_layers(model, timer, step, iters, graphs)
    fwd = [r for r in rows if r["mode"].startswith("training")]
    ms = float(np.mean([r["ms_per_launch"] for r in fwd]))
    return {"workload": f"{workload}: {wl.note}", "bytes_per_launch": fwd[0]["bytes_algorithmic"], "ms_per_launch": ms,
            "frac": fwd[0]["bytes_algorithmic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_inference": float(np.mean([r["frac"] for r in rows if r["mode"] == "inference"])), "per_layer": rows}


def cpu_step_baseline(wl, model, budget_s: float = 20.0):
    """The oracle's whole training step minus Adam -- rec_cal_loss + backward under torch autograd, the work `value`
    divides by -- on the host cores: directed edges of the step's 2 L propagation passes per second."""
    from oracle import ref_cpu as O
    from diffgraph_amd import data as dgdata
    T = torch.from_numpy
    mk = lambda ui: tuple(T(a.astype(np.int64)) for a in dgdata.bipartite_edges(ui[0], ui[1], wl.n_user))
    target, behs = mk(wl.target), [mk(b) for b in wl.behaviours]
    gen = torch.Generator().manual_seed(1)
    B = model.args.batch
    ancs, poss, negs = (torch.randint(0, k, (B,), generator=gen) for k in (wl.n_user, wl.n_item, wl.n_item))
    ts_u, ts_i = (torch.randint(0, model.args.steps, (k,), generator=gen) for k in (wl.n_user, wl.n_item))
    noise_u, noise_i = torch.randn(wl.n_user, 64, generator=gen), torch.randn(wl.n_item, 64, generator=gen)

    def weights(dn):
        g = lambda p: p.detach().cpu().clone().requires_grad_(True)
        return O.DenoiseWeights(g(dn.emb_layer.weight), g(dn.emb_layer.bias), g(dn.in_layers[0].weight), g(dn.in_layers[0].bias),
                                g(dn.out_layers[0].weight), g(dn.out_layers[0].bias))
    a = model.args
    tab = O.DiffusionTables(a.noise_scale, a.noise_min, a.noise_max, a.steps)
    edges_per_step = 2 * model.n_layers * sum(2 * len(ui[0]) for ui in [wl.target] + wl.behaviours)
    t0, n = time.perf_counter(), 0
    while True:
        ue = T(dgdata.xavier_uniform(wl.n_user, 64, seed=0)).requires_grad_(True)
        ie = T(dgdata.xavier_uniform(wl.n_item, 64, seed=1)).requires_grad_(True)
        loss = O.rec_cal_loss(tab, weights(model.usr_denoiser), weights(model.item_denoiser), target, behs, ue, ie, ancs, poss, negs,
                              ts_u, noise_u, ts_i, noise_i, float(a.batch), float(a.reg))[0]
        loss.backward()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"step_value": edges_per_step * n / el, "step_sample": f"{n} x (rec_cal_loss + backward) of the {wl.name} workload under torch "
            f"autograd, no optimizer step, {el:.1f} s"}


def build_rec(workload: str, device):
    import diffgraph_amd as dg
    from diffgraph_amd import data as dgdata
    wl = dgdata.rec_workload(workload)
    graphs = [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item) for (u, i) in [wl.target] + wl.behaviours]
    handler = SimpleNamespace(userNum=wl.n_user, itemNum=wl.n_item, target_adj=graphs[0], behavior_mats=graphs[1:])
    torch.manual_seed(3)
    model = dg.rec.HGDM(handler, device=device).eval()
    model.set_embeddings(torch.from_numpy(dgdata.xavier_uniform(wl.n_user, 64, seed=0)).to(device),
                         torch.from_numpy(dgdata.xavier_uniform(wl.n_item, 64, seed=1)).to(device))
    return wl, graphs, model


def cpu_baseline(wl, budget_s=20.0, passes_per_call=2):
    """The oracle's torch-CPU restatement of the same propagation (index_add_ form, what the
    reference computes through DGL) timed on the host cores: whole forward passes until the
    budget is used."""
    from oracle import ref_cpu as O
    from diffgraph_amd import data as dgdata
    T = torch.from_numpy
    ue = T(dgdata.xavier_uniform(wl.n_user, 64, seed=0))
    ie = T(dgdata.xavier_uniform(wl.n_item, 64, seed=1))
    mk = lambda ui: tuple(T(a.astype(np.int64)) for a in dgdata.bipartite_edges(ui[0], ui[1], wl.n_user))
    target, behs = mk(wl.target), [mk(b) for b in wl.behaviours]
    edges_per_pass = 2 * sum(2 * len(ui[0]) for ui in [wl.target] + wl.behaviours)
    if passes_per_call > 1:
        O.rec_forward(target, behs, ue, ie, 2)  # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        O.rec_forward(target, behs, ue, ie, 2)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s / 2 or n >= 500:
            break
    res = {"value": edges_per_pass * n / el, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n} full propagation passes ({2 * (1 + len(wl.behaviours))} SpMM each, index_add_ form) of the {wl.name} workload, "
                     f"torch {torch.__version__} CPU, {el:.1f} s"}
    res["spmm_variants"] = cpu_spmm_variants(target, wl.n, torch.cat([ue, ie]))
    # the same propagation over prebuilt CSR A_hat matrices (torch's fastest CPU form on some hosts): the better of
    # the two is the baseline
    try:
        import torch.nn.functional as F
        mats = [O.normalised_adj_csr(s_, d_, wl.n) for (s_, d_) in [target] + behs]
        e0 = torch.cat([ue, ie])

        def forward_csr():
            sums = []
            for a in mats:
                e, acc = e0, e0
                for _ in range(2):
                    e = F.leaky_relu(a @ e, 0.5)
                    acc = acc + F.normalize(e, p=2, dim=1)
                sums.append(acc)
            return sums[0], sum(sums[1:])
        forward_csr()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s / 2 and k < 2000:
            forward_csr()
            k += 1
        el2 = time.perf_counter() - t0
        res["index_add_value"], res["csr_value"] = res["value"], edges_per_pass * k / el2
        if res["csr_value"] > res["value"]:
            res["value"] = res["csr_value"]
            res["sample"] = (f"{k} full propagation passes ({2 * (1 + len(wl.behaviours))} SpMM each) of the {wl.name} workload over prebuilt "
                             f"fp32 CSR A_hat matrices (torch.sparse CSR @ dense; faster here than the index_add_ form, "
                             f"{res['index_add_value']:.3g} edges/s), torch {torch.__version__} CPU, {el2:.1f} s")
    except Exception as exc:   # torch builds without CSR kernels keep the index_add_ figure
        res["csr_value"] = repr(exc)[:200]
    return res


def cpu_spmm_variants(edges, n, x, budget_s=2.0):
    """The bare SpMM A_hat . X of the target graph on the host cores in the three forms it can take in torch: the
    fp32 COO A_hat the reference's DataHandler builds (DiffGraph-Rec/DataHandler.py:103-118), the same as CSR, and
    the index_add_ form of the oracle.  Directed edges per second each (or the error text where torch refuses)."""
    from oracle import ref_cpu as O
    src, dst = edges
    out = {}
    try:
        csr = O.normalised_adj_csr(src, dst, n)
        coo = csr.to_sparse_coo().coalesce()
        no = torch.pow(O.degrees(src, dst, n)[0].float().clamp(min=1), -0.5).view(-1, 1)
        forms = {"coo_sparse_mm": lambda: torch.sparse.mm(coo, x), "csr_sparse_mm": lambda: csr @ x,
                 "index_add": lambda: torch.zeros_like(x).index_add_(0, dst, (x * no)[src])}
    except Exception as exc:        # a torch build without sparse CPU kernels: say so instead of failing the bench
        return {"error": repr(exc)[:200]}
    for name, fn in forms.items():
        try:
            fn()
            t0, k = time.perf_counter(), 0
            while time.perf_counter() - t0 < budget_s and k < 50:
                fn()
                k += 1
            out[name] = len(src) * k / (time.perf_counter() - t0)
        except Exception as exc:
            out[name] = repr(exc)[:200]
    return out


def launch_ranks(n: int) -> int:
    """Runs this command line as N ranks, one per GPU, under torch.distributed.run on this node; relays rank 0's JSON
    line (the ranks' other output goes to stderr) and returns non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    records = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in lines:
        if ln not in records:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: torch.distributed.run exited with code {proc.returncode}", file=sys.stderr)
        return proc.returncode
    if len(records) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(records)}", file=sys.stderr)
        return 1
    print(records[0])
    return 0


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks (nothing has touched the GPU yet in this process)
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # (> 1 rank per GPU only happens in single-GPU rehearsals over gloo)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or args.workload == "industry":
        if world > 1:
            import torch.distributed as dist
            backend = os.environ.get("DG_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
        from diffgraph_amd import dist as dgdist
        res = dgdist.bench_sharded(args, rank, world, device)
        if res is not None:
            if world == 1:
                res["roofline"]["traffic"], res["roofline"]["traffic_source"] = measured_traffic(args.workload)
            if world == 1 and not args.no_cpu_baseline:
                from diffgraph_amd import data as dgdata
                res["cpu_baseline"] = cpu_baseline(dgdata.rec_workload(args.workload), passes_per_call=1)
            print(json.dumps(res))
        return

    workload = args.workload or "tmall"
    if workload.startswith(("eval_", "negsample_", "metapath", "predict_", "nc_dblp", "nc_train", "epoch_")):
        # the stages either side of the propagation path (evaluation, sampling, graph construction)
        import bench_next
        if workload.startswith("eval_"):
            res = bench_next.bench_eval(workload[5:], args.steps, args.warmup, not args.no_cpu_baseline)
        elif workload.startswith("epoch_"):
            res = bench_next.bench_epoch(workload[6:], args.steps, args.warmup, not args.no_cpu_baseline)
        elif workload.startswith("predict_"):
            res = bench_next.bench_predict(workload[8:], args.steps, args.warmup, not args.no_cpu_baseline)
        elif workload.startswith("nc_train"):
            res = bench_next.bench_nc_train(args.steps, args.warmup, not args.no_cpu_baseline, not args.no_graph)
        elif workload.startswith("nc_dblp"):
            res = bench_next.bench_nc_dblp(args.steps, args.warmup, not args.no_cpu_baseline)
        elif workload.startswith("negsample_"):
            res = bench_next.bench_negsample(workload[10:], args.steps, args.warmup, not args.no_cpu_baseline)
        else:
            res = bench_next.bench_metapath(args.steps, args.warmup, not args.no_cpu_baseline)
        print(json.dumps(res))
        return
    wl, graphs, model, timer, step = rec_step(workload, device)
    B = model.args.batch
    if args.no_side_overlap:
        model.side_overlap = False

    if args.no_graph:
        run = step
        for _ in range(args.warmup):
            out = run()
    else:
        # The step is ~150 short launches; replaying it as one HIP graph removes the host launch cost.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step()
        run = graph.replay
        for _ in range(args.warmup):
            run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    per_layer = time_layers(model, timer, step, max(args.steps // 2, 10), graphs)
    L = model.n_layers
    nnz = [g.num_edges() for g in graphs]
    edges_per_step = 2 * L * sum(nnz)  # L forward + L backward (transposed) propagation launches per graph
    fwd = [r for r in per_layer if r["mode"].startswith("training")]
    layer_bytes_algo = fwd[0]["bytes_algorithmic"]
    layer_ms = float(np.mean([r["ms_per_launch"] for r in fwd]))          # the forward layer launches of the timed step
    comp_mean = float(np.mean([r["bytes_compulsory"] for r in fwd]))
    achieved = layer_bytes_algo / (layer_ms * 1e-3) / 1e9
    traffic, traffic_source = measured_traffic(workload)
    res = {
        "metric": "SpMM edges/sec + achieved HBM GB/s per propagation layer",
        "value": edges_per_step * args.steps / elapsed,
        "unit": "edges/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{workload}: {wl.note}", "n_user": wl.n_user, "n_item": wl.n_item,
                   "graphs": len(graphs), "directed_edges_per_graph": nnz, "layers": L,
                   "step": f"`value` is the {workload} workload on one GPU: one training step = HGDM.cal_loss (propagation, "
                           "q_sample + denoiser + SNR loss, BPR) + backward + Adam; eval-mode dropout (the reference's mode from "
                           "its second epoch on); the edges of the step's 2 L propagation passes per graph divided by the WHOLE "
                           "step's time.  Runs with --gpus N > 1 time another experiment, the Industry forward propagation "
                           "sharded over N ranks: compare those with `industry_1gpu_edges_per_s` here (the same forward on one "
                           "GPU), not with `value`",
                   "spmm_only_edges_per_s": sum(nnz) / (layer_ms * 1e-3),
                   "batch": B, "loss": float(out[0].detach()), "launch": "eager" if args.no_graph else "hip-graph replay"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "set_layer_kernel + set_hub_kernel (one propagation layer as ONE row walk over all graphs; mean "
                               "over the forward layer launches of the training step, timed with nothing running beside them -- "
                               "in the replayed step the first one shares the GPU with the step's short side launches, see the "
                               "step timeline under profiles/)",
                     "bytes_per_launch": layer_bytes_algo, "ms_per_launch": layer_ms,
                     "bytes_compulsory": comp_mean, "frac_compulsory": comp_mean / (layer_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "per_layer": per_layer},
    }
    others = {}
    for name in [w for w in args.other_workloads.split(",") if w]:
        # the other configuration the propagation target is quoted on, same method, a few seconds
        others[name] = other_workload(name, device)
    if args.one_gpu_forward:
        # the experiment the multi-GPU runs time (Industry forward propagation), here on this one GPU, so that a line of
        # `--gpus N` has its one-GPU partner in the default run
        from diffgraph_amd import dist as dgdist
        name = args.one_gpu_forward
        one = dgdist.bench_sharded(SimpleNamespace(workload=name, steps=10, warmup=3), 0, 1, device)
        res["config"][f"{name}_1gpu_edges_per_s"] = one["value"]
        res["config"][f"{name}_1gpu_ms_per_forward"] = one["ms_per_step"]
        others[f"{name} (forward propagation, the --gpus N experiment on one GPU)"] = {
            k: one["roofline"][k] for k in ("frac", "frac_compulsory", "bytes_per_launch", "ms_per_launch")}
    if others:
        res["roofline"]["other_workloads"] = others
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(wl)
        res["cpu_baseline"].update(cpu_step_baseline(wl, model))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
