"""bench.py legs for the stages either side of the propagation path: all-rank evaluation, negative sampling,
meta-path composition.  Each returns the JSON-able result dict of one workload (synthetic data of the
published shapes, inputs resident in HBM before the timed region, HIP events on the launch stream)."""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp
import torch

from . import data as dgdata
from . import evaluate as ev
from . import metapath as mp
from .sampling import PairwiseTrnData

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X fp32 MFMA / vector peak
HBM_PEAK_GBS = 8000.0


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


def bench_eval(workload: str, steps: int, warmup: int, cpu_baseline: bool):
    """Top-20 of every user against every item with the target behaviour's training pairs masked
    (Coach.testEpoch, DiffGraph-Rec/main.py:153-160)."""
    from oracle import ref_cpu as O  # checker / CPU baseline only
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
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"eval_{workload}: every user of the {workload}-shaped graph against every item, "
                                  f"target-behaviour training pairs masked, top-{k}", "n_user": nu, "n_item": ni, "topk": k},
           "roofline": {"bound": "mfma", "achieved": flop / (dev_ms * 1e-3) / 1e12, "peak": FP32_MATRIX_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flop / (dev_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, "traffic": None,
                        "kernel": "score_topk_kernel + merge_topk_kernel", "flop_per_launch": flop, "ms_per_launch": dev_ms}}
    if cpu_baseline:
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
    from oracle import ref_cpu as O
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
