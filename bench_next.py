"""bench.py legs for the stages either side of the propagation path: all-rank evaluation, negative sampling,
meta-path composition.  Each returns the JSON-able result dict of one workload (synthetic data of the
published shapes, inputs resident in HBM before the timed region, HIP events on the launch stream)."""
from __future__ import annotations

import os
import time

import numpy as np
import scipy.sparse as sp
import torch

from diffgraph_amd import data as dgdata
from diffgraph_amd import evaluate as ev
from diffgraph_amd import metapath as mp
from diffgraph_amd.sampling import PairwiseTrnData

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X fp32 MFMA / vector peak
HBM_PEAK_GBS = 8000.0
BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 matrix peak (no sparsity)


def _timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, a.elapsed_time(b) / steps


def _coo(u, i, nu, ni):
    return sp.coo_matrix((np.ones(len(u), np.float32), (u, i)), shape=(nu, ni))


def eval_roofline(flop: float, dev_ms: float) -> dict:
    """The scoring kernel against the matrix pipes: `flop` = the scores' fp32 work (2 x users x items x 64).  The default
    form (k <= 20) executes six bf16 products per fp32 product on v_mfma_f32_32x32x16_bf16 (dense bf16 peak); with
    DG_EVAL_FP32=1 the exact-fp32 instruction executes `flop` itself.  Either way most of the kernel's time is the top-k
    selection on the vector unit, which does not overlap the matrix instructions on this chip."""
    fp32_rate = flop / (dev_ms * 1e-3) / 1e12
    if os.environ.get("DG_EVAL_FP32") == "1":
        return {"bound": "mfma", "achieved": fp32_rate, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fp32_rate / FP32_MATRIX_PEAK_TFLOPS, "traffic": None,
                "kernel": "score_topk_kernel<., false> + merge_topk_kernel (v_mfma_f32_32x32x2_f32)", "flop_per_launch": flop,
                "ms_per_launch": dev_ms}
    return {"bound": "mfma", "achieved": 6.0 * fp32_rate, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": 6.0 * fp32_rate / BF16_DENSE_PEAK_TFLOPS, "traffic": None,
            "kernel": "score_topk_kernel<., true> + merge_topk_kernel (v_mfma_f32_32x32x16_bf16 over fp32 values cut into three "
                      "bf16 numbers, six products per score term)",
            "flop_per_launch": 6.0 * flop, "ms_per_launch": dev_ms,
            "fp32_equivalent": {"achieved": fp32_rate, "fp32_instruction_peak": FP32_MATRIX_PEAK_TFLOPS,
                                "frac_of_fp32_instruction_peak": fp32_rate / FP32_MATRIX_PEAK_TFLOPS, "flop_per_launch": flop}}


def bench_eval(workload: str, steps: int, warmup: int, cpu_baseline: bool):
    """Top-20 of every user against every item with the target behaviour's training pairs masked
    (Coach.testEpoch, DiffGraph-Rec/main.py:153-160)."""
    wl = dgdata.rec_workload(workload)
    nu, ni, k = wl.n_user, wl.n_item, 20
    usr = (torch.from_numpy(dgdata.xavier_uniform(nu, 64, seed=0)) * 3)
    itm = (torch.from_numpy(dgdata.xavier_uniform(ni, 64, seed=1)) * 3)
    trn_u, trn_i = wl.target
    rng = np.random.default_rng(11)
    tst_u, tst_i = np.arange(nu, dtype=np.int32), rng.integers(0, ni, nu).astype(np.int32)
    data = ev.AllRankTestData(_coo(tst_u, tst_i, nu, ni), _coo(trn_u, trn_i, nu, ni))
    ud, idv = usr.cuda(), itm.cuda()
    wall_ms, dev_ms = _timed(lambda: ev.all_rank_topk(ud, idv, data.test_users_dev, data, k), steps, warmup)
    flop = 2.0 * nu * ni * 64
    res = {"metric": "all-rank evaluation: user-item scores/sec through mask + top-20", "value": nu * float(ni) / (wall_ms * 1e-3),
           "unit": "scores/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": wall_ms, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32" if os.environ.get("DG_EVAL_FP32") == "1" else "f32 (as 3 x bf16)",
           "data": "synthetic",
           "config": {"workload": f"eval_{workload}: every user of the {workload}-shaped graph against every item, "
                                  f"target-behaviour training pairs masked, top-{k}", "n_user": nu, "n_item": ni, "topk": k},
           "roofline": eval_roofline(flop, dev_ms)}
    if cpu_baseline:
        from oracle import ref_cpu as O      # the CPU baseline leg: the only use of the oracle here
        n_s = 2048
        users = torch.from_numpy(data.test_users[:n_s]).long()
        mask = np.zeros((n_s, ni), np.float32)
        for r, u in enumerate(data.test_users[:n_s]):
            mask[r, trn_i[trn_u == u]] = 1.0
        m = torch.from_numpy(mask)
        O.all_rank_topk(usr, itm, users, m, k)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0:
            O.all_rank_topk(usr, itm, users, m, k)
            n += 1
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": n * n_s * float(ni) / el, "unit": "scores/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{n} batches of {n_s} users x {ni} items: torch.mm + dense mask + topk (mask rows prebuilt, "
                                         f"outside the timed loop), {el:.1f} s"}
    return res


def bench_negsample(workload: str, steps: int, warmup: int, cpu_baseline: bool):
    """One negative per training pair of the target behaviour (PairwiseTrnData.negSampling, DataHandler.py:282-289)."""
    wl = dgdata.rec_workload(workload)
    u, i = wl.target
    ds = PairwiseTrnData(_coo(u, i, wl.n_user, wl.n_item), seed=1)
    wall_ms, dev_ms = _timed(ds.negSampling, steps, warmup)
    n = len(ds)
    deg = np.bincount(u, minlength=wl.n_user)
    probes = float(np.sum(np.ceil(np.log2(np.maximum(deg[u], 1) + 1))))
    alg_bytes = 16.0 * n + 4.0 * probes          # pair's user id, two row pointers, the negative; one id per search step
    res = {"metric": "negative sampling: training pairs/sec", "value": n / (wall_ms * 1e-3), "unit": "pairs/s", "n_gpus": 1,
           "steps": steps, "warmup": warmup, "ms_per_step": wall_ms, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": f"negsample_{workload}: one negative per pair of the {workload}-shaped target behaviour",
                      "pairs": n, "n_item": wl.n_item},
           "roofline": {"bound": "hbm", "achieved": alg_bytes / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "neg_sampling_kernel",
                        "bytes_per_launch": alg_bytes, "ms_per_launch": dev_ms,
                        "note": "a launch-latency-sized kernel: 1e5-1e6 pairs of a few dependent 4-byte probes each"}}
    if cpu_baseline:
        from oracle import ref_cpu as O      # the CPU baseline leg: the only use of the oracle here
        n_s = min(n, 100000)
        t0 = time.perf_counter()
        O.neg_sampling(u[:n_s], i[:n_s], wl.n_item, np.random.default_rng(0))
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": n_s / el, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"the first {n_s} pairs, Python loop with set look-ups as the reference, {el:.1f} s"}
    return res


def bench_metapath(steps: int, warmup: int, cpu_baseline: bool):
    """Author x author meta-path graphs on a DBLP-shaped heterogeneous graph (DiffGraph_NC/DataHandler.py:147-165):
    APA, APCPA, APTPA as chains of boolean products (the reference's DBLP edge lists do not travel; the graph is
    synthetic with DBLP's node and edge counts)."""
    NA, NP, NC, NT = 4057, 14328, 20, 7723
    rng = np.random.default_rng(0)
    def rel(n_src, n_dst, n_edges, zipf):
        src = rng.integers(0, n_src, n_edges)
        w = 1.0 / np.arange(1, n_dst + 1) ** zipf
        dst = rng.choice(n_dst, n_edges, p=w / w.sum())
        m = sp.coo_matrix((np.ones(n_edges, np.float32), (src, dst)), shape=(n_src, n_dst)).tocsr()
        m.data[:] = 1.0
        return m
    AP = rel(NP, NA, 19645, 0.5).T.tocsr()
    PC, PT = rel(NP, NC, 14328, 0.3), rel(NP, NT, 85810, 0.9)
    D = mp.DevicePattern.from_scipy
    chains = [[D(AP), D(AP.T.tocsr())], [D(AP), D(PC), D(PC.T.tocsr()), D(AP.T.tocsr())], [D(AP), D(PT), D(PT.T.tocsr()), D(AP.T.tocsr())]]
    out = []
    def run():
        out.clear()
        for c in chains:
            out.append(mp.compose(c))
    wall_ms, dev_ms = _timed(run, steps, warmup)
    nnz_out = sum(o.nnz for o in out)
    words = (NA + 31) // 32
    alg_bytes = sum(4.0 * (c[0].nnz + c[-1].nnz) for c in chains) + 3 * NA * words * 8.0 + 4.0 * nnz_out
    res = {"metric": "meta-path composition: output entries/sec", "value": nnz_out / (wall_ms * 1e-3), "unit": "entries/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": wall_ms, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "metapath_dblp: APA, APCPA, APTPA of a DBLP-shaped graph", "nodes": [NA, NP, NC, NT],
                      "output_entries": [o.nnz for o in out]},
           "roofline": {"bound": "hbm", "achieved": alg_bytes / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "spgemm_bitmap_kernel + spgemm_expand_kernel (7 products)", "bytes_per_launch": alg_bytes,
                        "ms_per_launch": dev_ms}}
    if cpu_baseline:
        from oracle import ref_cpu as O      # the CPU baseline leg: the only use of the oracle here
        binz = lambda m: (m != 0).astype(np.float32)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0:
            apc, apt = binz(AP @ PC), binz(AP @ PT)
            _ = binz(AP @ AP.T), binz(apc @ apc.T), binz(apt @ apt.T)
            n += 1
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": n * nnz_out / el, "unit": "entries/s", "cores": 1, "kind": "port",
                               "sample": f"{n} x the three compositions with scipy sparse products, {el:.1f} s"}
    return res


def predict_roofline(flop: float, loop_ms: float) -> dict:
    """The reverse loop against the matrix pipes.  `flop` = the loop's fp32 work.  On the exact-fp32 instruction that
    is what the pipes execute (peak 157.3 TFLOP/s).  The default form executes SIX bf16 products per fp32 product (peak
    2 500 TFLOP/s dense bf16): `achieved` / `frac` count what the pipes execute, `fp32_equivalent` says what that is
    worth as fp32 work -- more than the fp32 instruction's own peak allows."""
    fp32_rate = flop / (loop_ms * 1e-3) / 1e12
    if os.environ.get("DG_PSAMPLE_FP32") == "1":
        return {"bound": "mfma", "achieved": fp32_rate, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fp32_rate / FP32_MATRIX_PEAK_TFLOPS, "traffic": None,
                "kernel": "denoise_kernel<MODE_PSAMPLE> (both denoisers, one launch; v_mfma_f32_32x32x2_f32)",
                "flop_per_launch": flop, "ms_per_launch": loop_ms}
    return {"bound": "mfma", "achieved": 6.0 * fp32_rate, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": 6.0 * fp32_rate / BF16_DENSE_PEAK_TFLOPS, "traffic": None,
            "kernel": "psample_b3_kernel (both denoisers, one launch; v_mfma_f32_32x32x16_bf16 over fp32 values cut into three "
                      "bf16 numbers, six products per contraction)",
            "flop_per_launch": 6.0 * flop, "ms_per_launch": loop_ms,
            "fp32_equivalent": {"achieved": fp32_rate, "fp32_instruction_peak": FP32_MATRIX_PEAK_TFLOPS,
                                "frac_of_fp32_instruction_peak": fp32_rate / FP32_MATRIX_PEAK_TFLOPS, "flop_per_launch": flop}}


def bench_predict(workload: str, steps: int, warmup: int, cpu_baseline: bool):
    """HGDM.predict (DiffGraph-Rec/Model.py:112-118): forward propagation + the 200-step reverse diffusion loop
    for every user and item row; the roofline entry is the reverse loop's: 200 x 17 408 FLOP per row as fp32 work,
    carried out as six bf16 matrix products per contraction (every fp32 value cut into three bf16 numbers) unless
    DG_PSAMPLE_FP32=1 keeps it on the exact-fp32 matrix instruction."""
    from types import SimpleNamespace
    from diffgraph_amd import CSRGraph, rec
    dev = torch.device("cuda", torch.cuda.current_device())
    wl = dgdata.rec_workload(workload)
    graphs = [CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item) for (u, i) in [wl.target] + wl.behaviours]
    handler = SimpleNamespace(userNum=wl.n_user, itemNum=wl.n_item, target_adj=graphs[0], behavior_mats=graphs[1:])
    torch.manual_seed(3)
    model = rec.HGDM(handler, device=dev).eval()
    model.set_embeddings(torch.from_numpy(dgdata.xavier_uniform(wl.n_user, 64, seed=0)).to(dev),
                         torch.from_numpy(dgdata.xavier_uniform(wl.n_item, 64, seed=1)).to(dev))
    with torch.no_grad():
        wall_ms, _ = _timed(model.predict, steps, warmup)
        ui, het = model.propagate(model.embedding_table)
        dp = model.diffusion_process
        pairs = [(model.usr_denoiser, het[: wl.n_user]), (model.item_denoiser, het[wl.n_user:])]
        _, loop_ms = _timed(lambda: dp.p_sample_many(pairs, model.args.sampling_steps), steps, warmup)
    n = wl.n
    flop = float(n) * dp.steps * 17408.0
    nnz = [g.num_edges() for g in graphs]
    res = {"metric": "HGDM.predict: embedding rows/sec through propagation + 200-step reverse diffusion", "value": n / (wall_ms * 1e-3),
           "unit": "rows/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": wall_ms, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32" if os.environ.get("DG_PSAMPLE_FP32") == "1" else "f32 (as 3 x bf16)",
           "data": "synthetic",
           "config": {"workload": f"predict_{workload}: {wl.note}", "n_user": wl.n_user, "n_item": wl.n_item,
                      "directed_edges_per_graph": nnz, "layers": model.n_layers, "diffusion_steps": dp.steps,
                      "edges_per_s": 2.0 * sum(nnz) / (wall_ms * 1e-3)},
           "roofline": predict_roofline(flop, loop_ms)}
    if cpu_baseline:
        from oracle import ref_cpu as O      # the CPU baseline leg: the only use of the oracle here
        n_s = 512
        tab = O.DiffusionTables(noise_scale=1e-4, noise_min=1e-4, noise_max=1e-3, steps=dp.steps)
        dn = model.usr_denoiser
        w = O.DenoiseWeights(*[t.detach().cpu() for t in (dn.emb_layer.weight, dn.emb_layer.bias, dn.in_layers[0].weight,
                                                          dn.in_layers[0].bias, dn.out_layers[0].weight, dn.out_layers[0].bias)])
        x = het[:n_s].cpu()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0:
            O.p_sample(tab, w, x, 0)
            k += 1
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": k * n_s / el, "unit": "rows/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{k} x the 200-step reverse loop over {n_s} rows (reverse loop only, torch CPU ops as the "
                                         f"reference's p_sample), {el:.1f} s"}
    return res


def _nc_dblp_setup():
    """DBLP from the committed edge lists: the three author graphs composed on the device, row-normalised author
    features (DiffGraph_NC/DataHandler.py:147-174, 254-259), a random-init model."""
    import os
    from diffgraph_amd import nc
    from diffgraph_amd.graph import CSRGraph
    dev = torch.device("cuda", torch.cuda.current_device())
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "dblp_graph.npz"))
    NA, NP, NC, NT = (int(v) for v in g["counts"])
    mk = lambda e, shape: sp.coo_matrix((np.ones(len(e), np.float32), (e[:, 0], e[:, 1])), shape=shape).tocsr()
    AP, PC, PT = mk(g["ap"], (NA, NP)), mk(g["pc"], (NP, NC)), mk(g["pt"], (NP, NT))
    D = mp.DevicePattern.from_scipy
    pats = [mp.compose([D(AP), D(AP.T)]), mp.compose([D(AP), D(PC), D(PC.T), D(AP.T)]), mp.compose([D(AP), D(PT), D(PT.T), D(AP.T)])]
    mats = [p.to_scipy() for p in pats]
    adjs = [CSRGraph.from_scipy(m) for m in mats]
    feats = sp.csr_matrix((g["feat_data"], g["feat_indices"], g["feat_indptr"]), shape=tuple(g["feat_shape"]))
    rowsum = np.array(feats.sum(1)).flatten()
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    features = torch.from_numpy(np.asarray(sp.diags(r_inv).dot(feats).todense(), dtype=np.float32))
    torch.manual_seed(5)
    model = nc.HGDM(features.shape[1]).eval()
    return dev, NA, mats, adjs, features, model, torch.from_numpy(np.asarray(g["labels"])).long()


def bench_nc_train(steps: int, warmup: int, cpu_baseline: bool, use_graph: bool = True):
    """One training step of DiffGraph_NC on DBLP (DiffGraph_NC/main.py:372-386): cal_loss over a batch of 8 labelled
    authors (3 x 3 propagation layers, q_sample + denoiser + SNR loss, classifier, NLL), backward, Adam."""
    dev, NA, mats, adjs, features, model, labels = _nc_dblp_setup()
    model.train()      # the reference never leaves training mode (no .eval() in DiffGraph_NC/main.py): dropout is on
    fd = features.to(dev)
    onehot = torch.zeros(NA, 4, device=dev)
    onehot[torch.arange(NA, device=dev), labels.to(dev)] = 1.0
    gen = torch.Generator().manual_seed(1)
    ancs = torch.sort(torch.randperm(NA, generator=gen)[: model.args.batch])[0].to(dev)
    # one launch for the ten parameter tensors, Adam's step count on the device: the whole step can be recorded
    from diffgraph_amd.optim import SmallAdam
    opt = SmallAdam(model.parameters(), lr=3e-3, weight_decay=0)

    def step():
        nll, diff = model.cal_loss(ancs, onehot, adjs, fd)
        opt.zero_grad(set_to_none=True)
        (nll + diff).backward()
        opt.step()
        return nll, diff
    if use_graph:
        # the ~60 launches of a step recorded once and replayed: the timesteps and the noise come from torch's
        # generator, whose offset a replay advances, so every replay trains on fresh draws like an eager step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            losses = step()
        wall_ms, dev_ms = _timed(graph.replay, steps, warmup)
        graph.replay()
        nll, diff = losses
    else:
        wall_ms, dev_ms = _timed(step, steps, warmup)
        nll, diff = step()
    with torch.no_grad():
        embed = model.transform_layer(fd)
        _, prop_ms = _timed(lambda: model.propagate(adjs, embed), max(steps, 10), 3)
    nnz = [m.nnz for m in mats]
    L = model.args.gcn_layer
    layer_bytes = sum(4 * e + 1032 * NA for e in nnz)
    res = {"metric": "DiffGraph_NC training step: SpMM edges/sec through cal_loss + backward + Adam", "value": 2 * L * sum(nnz) / (wall_ms * 1e-3),
           "unit": "edges/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": wall_ms, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "DBLP edge lists and author features (tests/golden/dblp_graph.npz), random-init weights",
           "config": {"workload": "nc_train_dblp: APA / APCPA / APTPA author graphs, batch of 8 authors", "authors": NA,
                      "directed_edges_per_graph": nnz, "layers": L, "device_ms_per_step": dev_ms,
                      "loss": [float(nll.detach()), float(diff.detach())],
                      "launch": "hip-graph replay" if use_graph else "eager"},
           "roofline": dense_roofline(nnz, NA, L, prop_ms, layer_bytes, "dense_rows_kernel (+ dense_split / dense_reduce / hub_reduce "
                                      "and the co-author graph's gather workgroups), forward propagation: 3 layers x 3 graphs")}
    if cpu_baseline:
        from oracle import ref_cpu as O      # the CPU baseline leg: the only use of the oracle here
        T = lambda m: tuple(torch.from_numpy(a.astype(np.int64)) for a in (m.tocoo().row, m.tocoo().col))
        adj_t = [T(m) for m in mats]
        w_t = model.transform_layer.weight.detach().cpu().clone().requires_grad_(True)
        b_t = model.transform_layer.bias.detach().cpu().clone().requires_grad_(True)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0 and k < 10:
            outs = O.nc_forward(adj_t, features, w_t, b_t, L)
            sum(o.sum() for o in outs).backward()
            k += 1
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": k * 2 * L * sum(nnz) / el, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{k} x forward + autograd backward of the propagation (projection + {L} layers x 3 graphs, "
                                         f"index_add_ form), {el:.1f} s"}
    return res




def dense_roofline(nnz, n_nodes: int, n_layers: int, prop_ms: float, layer_bytes: int, what: str) -> dict:
    """Roofline of a propagation whose large graphs run on the bitmap path (dense_rows_kernel: A[n x n] bitmap times
    the table as three bf16 planes on v_mfma_f32_32x32x16_bf16): matrix-core bound.  FLOP per bitmap graph and layer
    = 2 n^2 64 x 3 planes; `achieved` divides by the whole layer launch (split + contraction + reduce kernels, and
    the gather-path graphs riding along), so it is a lower bound for the contraction kernel itself -- its own time is
    in the rocprofv3 kernel statistics under profiles/.  The HBM figures stay as secondary fields."""
    dense = [e for e in nnz if e >= 0.10 * n_nodes * n_nodes]           # DG_DENSE_MIN_DENSITY
    flop = len(dense) * 2.0 * n_nodes * n_nodes * 64 * 3
    achieved = flop / (prop_ms / n_layers * 1e-3) / 1e12
    hbm = n_layers * layer_bytes / (prop_ms * 1e-3) / 1e9
    return {"bound": "mfma", "achieved": achieved, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / BF16_DENSE_PEAK_TFLOPS, "traffic": None, "kernel": what,
            "flop_per_launch": flop, "ms_per_launch": prop_ms / n_layers, "bitmap_graphs": len(dense),
            "hbm_secondary": {"achieved_GBps": hbm, "frac_of_8TBps": hbm / HBM_PEAK_GBS, "bytes_per_launch": layer_bytes,
                              "note": "the 1 MB tables sit in L2; the algorithmic-bytes model says little for a dense graph"}}



def bench_nc_dblp(steps: int, warmup: int, cpu_baseline: bool):
    """DiffGraph_NC on DBLP (4 057 authors; meta-path graphs APA / APCPA / APTPA composed on the device from the
    committed edge lists): HGDM.get_allembeds = input projection, 3 x 3 propagation layers, reverse loop, classifier."""
    dev, NA, mats, adjs, features, model, _ = _nc_dblp_setup()
    fd = features.to(dev)
    with torch.no_grad():
        wall_ms, _ = _timed(lambda: model.get_allembeds(adjs, fd), steps, warmup)
        embed = model.transform_layer(fd)
        _, prop_ms = _timed(lambda: model.propagate(adjs, embed), steps, warmup)
    nnz = [m.nnz for m in mats]
    L = model.args.gcn_layer
    layer_bytes = sum(4 * e + 1032 * NA for e in nnz)
    res = {"metric": "DiffGraph_NC get_allembeds: SpMM edges/sec through projection + propagation + reverse diffusion + classifier",
           "value": L * sum(nnz) / (wall_ms * 1e-3), "unit": "edges/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": wall_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "DBLP edge lists and author features (tests/golden/dblp_graph.npz), random-init weights",
           "config": {"workload": "nc_dblp: APA / APCPA / APTPA author graphs", "authors": NA, "directed_edges_per_graph": nnz,
                      "layers": L, "propagation_ms": prop_ms},
           "roofline": dense_roofline(nnz, NA, L, prop_ms, layer_bytes, "dense_rows_kernel (+ dense_split / dense_reduce / hub_reduce "
                                      "and the co-author graph's gather workgroups), 3 layers x 3 graphs")}
    if cpu_baseline:
        from oracle import ref_cpu as O      # the CPU baseline leg: the only use of the oracle here
        T = lambda m: tuple(torch.from_numpy(a.astype(np.int64)) for a in (m.tocoo().row, m.tocoo().col))
        adj_t = [T(m) for m in mats]
        w_t, b_t = model.transform_layer.weight.detach().cpu(), model.transform_layer.bias.detach().cpu()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0 and k < 20:
            O.nc_forward(adj_t, features, w_t, b_t, L)
            k += 1
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": k * L * sum(nnz) / el, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{k} x the forward (projection + {L} layers x 3 graphs, index_add_ form), {el:.1f} s"}
    return res


def bench_epoch(workload: str, steps: int, warmup: int, cpu_baseline: bool):
    """One step = one EPOCH of Coach (DiffGraph-Rec/main.py:78-94): trainEpoch -- negative sampling, every shuffled
    batch of the target behaviour through cal_loss + backward + Adam -- then testEpoch (predict with the 200-step
    reverse loop, all-rank top-20, Recall / NDCG), as the reference tests after every epoch."""
    import diffgraph_amd as dg
    from types import SimpleNamespace
    from diffgraph_amd.train import Coach
    wl = dgdata.rec_workload(workload)
    nu, ni = wl.n_user, wl.n_item
    graphs = [dg.CSRGraph.from_bipartite(u, i, nu, ni) for (u, i) in [wl.target] + wl.behaviours]
    rng = np.random.default_rng(11)
    tst_u, tst_i = np.arange(nu, dtype=np.int32), rng.integers(0, ni, nu).astype(np.int32)
    trn = _coo(*wl.target, nu, ni)
    handler = SimpleNamespace(userNum=nu, itemNum=ni, target_adj=graphs[0], behavior_mats=graphs[1:],
                              trn_data=PairwiseTrnData(trn, seed=1), tst_data=ev.AllRankTestData(_coo(tst_u, tst_i, nu, ni), trn))
    torch.manual_seed(3)
    coach = Coach(handler, seed=5)
    coach.prepareModel()
    model, B = coach.model, coach.args.batch
    times = {"train": 0.0, "test": 0.0}
    last = {}

    def epoch(timed: bool):
        t0 = time.perf_counter()
        last["train"] = coach.trainEpoch()            # ends with the epoch's one read-back of the loss sums
        t1 = time.perf_counter()
        last["test"] = coach.testEpoch()              # ends with the metrics on the host
        t2 = time.perf_counter()
        if timed:
            times["train"] += t1 - t0
            times["test"] += t2 - t1
    for _ in range(max(warmup, 1)):                   # the first epoch records the step's HIP graph
        epoch(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        epoch(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # the propagation layer's launch time for the roofline: HIP events round each fused layer launch, back to back
    pairs = []

    def hook(phase, layer, tasks=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        pairs.append(e)
    with torch.no_grad():
        for _ in range(5):
            model.propagate(model.embedding_table)
        torch.cuda.synchronize()
        model.layer_hook = hook
        for _ in range(20):
            model.propagate(model.embedding_table)
        torch.cuda.synchronize()
        model.layer_hook = None
    layer_ms = float(np.mean([pairs[k].elapsed_time(pairs[k + 1]) for k in range(0, len(pairs), 2)]))
    L, n = model.n_layers, wl.n
    nnz = [g.num_edges() for g in graphs]
    n_pairs = len(handler.trn_data)
    n_batches = -(-n_pairs // B)
    edges_per_epoch = (2 * L * n_batches + L) * sum(nnz)      # forward + backward per batch, one forward in predict
    layer_bytes = sum(4 * e + 1032 * n for e in nnz)
    achieved = layer_bytes / (layer_ms * 1e-3) / 1e9
    res = {"metric": "SpMM edges/sec + achieved HBM GB/s per propagation layer", "value": edges_per_epoch * steps / elapsed,
           "unit": "edges/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"epoch_{workload}: {wl.note}; one step = one full epoch (trainEpoch + testEpoch)",
                      "n_user": nu, "n_item": ni, "graphs": len(graphs), "directed_edges_per_graph": nnz, "layers": L,
                      "train_pairs": n_pairs, "batch": B, "batches_per_epoch": n_batches,
                      "train_ms_per_epoch": times["train"] / steps * 1e3, "test_ms_per_epoch": times["test"] / steps * 1e3,
                      "last_epoch": {**last["train"], **last["test"]},
                      "launch": "hip-graph replay (one recording per batch size: full and last)"},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "kernel": "gcn_layer_kernel + hub_reduce_kernel (one fused propagation layer, all graphs)",
                        "bytes_per_launch": layer_bytes, "ms_per_launch": layer_ms}}
    if cpu_baseline:
        import bench      # (bench.cpu_baseline is the one place that times the oracle's propagation: index_add_ and CSR forms)
        res["cpu_baseline"] = bench.cpu_baseline(wl)
        res["cpu_baseline"]["sample"] += (f"; forward propagation only -- an epoch of the reference adds the backward pass, "
                                          f"{n_batches} times, and its Python-side sampling and evaluation")
    return res
