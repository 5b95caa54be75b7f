"""Row-sharded propagation over the GPUs of one node (one process per GPU, torch.distributed).

The reference is single-device; this is the multi-GPU design for Industry-shaped graphs
(README.md:73: 1M users x 361 items): the USER rows of the bipartite adjacency are
partitioned into contiguous blocks, one per rank; the small ITEM block is replicated.

Per propagation layer and rank:
  1. one fused launch over all graphs: every local user row is finished locally (its
     neighbours are items, and the item table is replicated); every item row yields the
     partial neighbour sum over this rank's users only;
  2. ONE exchange: all-gather of the item partial sums  [graphs, n_item, 64]  (361 x 64
     floats per graph -- 92 KB -- instead of an all-gather of the 256 MB user table);
  3. a small launch adds the partial sums in rank order (fixed order: the result does not
     depend on timing) and applies the layer epilogue to the item rows, identically on
     every rank.
User embeddings never leave their rank.  The compute calls go through a `backend` object so
that the orchestration (partition, global degrees, exchange, summation order) is testable on
CPU with torch.distributed's gloo backend; the product backend is HipBackend (HIP kernels).
"""
from __future__ import annotations

import json
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import DG_ACC_REBUILD, DG_D, DG_IN_PRESCALED, DG_OUT_PRESCALE


def user_ranges(n_user: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal user blocks (user degrees are near-uniform in the Industry shape)."""
    base, rem = divmod(n_user, world)
    out, u = [], 0
    for r in range(world):
        m = base + (1 if r < rem else 0)
        out.append((u, u + m))
        u += m
    return out


@dataclass
class LocalGraph:
    """One behaviour graph restricted to a rank: local users [0,n_loc) + all items [n_loc, n_loc+n_item)."""
    src: np.ndarray
    dst: np.ndarray
    degrees: np.ndarray     # GLOBAL degrees of the local nodes (symmetric graph: in == out)
    n_edges_global: int     # directed edges of the whole graph
    in_degrees: Optional[np.ndarray] = None   # if set: in-degrees, with -1 for rows this rank does not own


def shard_bipartite(u: np.ndarray, i: np.ndarray, n_user: int, n_item: int, u0: int, u1: int) -> LocalGraph:
    deg_u = np.bincount(u, minlength=n_user).astype(np.int32)
    deg_i = np.bincount(i, minlength=n_item).astype(np.int32)
    keep = (u >= u0) & (u < u1)
    lu = (u[keep] - u0).astype(np.int32)
    li = (i[keep] + (u1 - u0)).astype(np.int32)
    src = np.concatenate([lu, li])
    dst = np.concatenate([li, lu])
    return LocalGraph(src, dst, np.concatenate([deg_u[u0:u1], deg_i]), 2 * len(u))


class HipBackend:
    """The product backend: CSRGraph handles and fused HIP launches."""

    def __init__(self, device):
        self.device = device

    def make_graph(self, lg: LocalGraph, n_nodes: int):
        from .graph import CSRGraph
        return CSRGraph(lg.src, lg.dst, n_nodes, out_degrees=lg.degrees,
                        in_degrees=lg.degrees if lg.in_degrees is None else lg.in_degrees)

    def empty(self, rows: int) -> torch.Tensor:
        return torch.empty((rows, DG_D), dtype=torch.float32, device=self.device)

    def layer(self, tasks) -> None:
        from . import ops
        ops.gcn_layer_batched([ops.LayerTask(**t) for t in tasks])

    def finalize(self, tasks) -> None:
        from . import ops
        ops.gcn_partial_finalize_batched([ops.FinalizeTask(**t) for t in tasks])

    def sum_tables(self, tables):
        from . import ops
        return ops.sum_tables(tables)


def all_gather_stack(t: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """[...] on every rank -> [world, ...] in rank order (one RCCL all-gather on GPU tensors)."""
    if world == 1:
        return t.unsqueeze(0)
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # single-GPU rehearsal of the multi-process path: gloo moves host memory
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return torch.stack(parts, dim=0).to(t.device)
    if t.is_cuda:
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)   # no re-packing
        return out
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.stack(parts, dim=0)


class RowBlockPropagation:
    """The general partition: contiguous, equal blocks of the rows of A_hat (and of every table) per rank,
    ONE all-gather of each graph's activation table per layer.  Works for any symmetric graph set; it is
    what a Tmall-shaped graph would use (neither side is small).  A rank holds the edges that end in its
    block, full-height tables (only its rows of the outputs are valid) and the whole graph's degrees.
    Same layer semantics as HGDM.forward (DiffGraph-Rec/Model.py:53-92)."""

    def __init__(self, graphs: Sequence[Tuple[np.ndarray, np.ndarray]], n: int, rank: int, world: int, backend,
                 n_layers: int = 2, slope: float = 0.5, group=None):
        self.rank, self.world, self.backend, self.group = rank, world, backend, group
        self.n, self.n_layers, self.slope = n, n_layers, slope
        self.blk = -(-n // world)
        self.r0 = min(n, rank * self.blk)
        self.r1 = min(n, self.r0 + self.blk)
        self.n_pad = self.blk * world          # table height: every rank's slice has blk rows
        self.graphs, self.edges_global = [], []
        self.hook = None
        for (src, dst) in graphs:
            src, dst = np.asarray(src, dtype=np.int32), np.asarray(dst, dtype=np.int32)
            deg_out = np.bincount(src, minlength=n).astype(np.int32)
            deg_in = np.bincount(dst, minlength=n).astype(np.int32)
            keep = (dst >= self.r0) & (dst < self.r1)
            owned = np.zeros(n, dtype=bool)
            owned[self.r0:self.r1] = True
            lg = LocalGraph(src[keep], dst[keep], deg_out, len(src), np.where(owned, deg_in, -1).astype(np.int32))
            self.graphs.append(backend.make_graph(lg, n))
            self.edges_global.append(len(src))

    def _gather_table(self, table: torch.Tensor) -> torch.Tensor:
        """table [n_pad, 64] with this rank's block valid -> the whole table."""
        mine = table[self.rank * self.blk:(self.rank + 1) * self.blk]
        return all_gather_stack(mine, self.world, self.group).reshape(self.n_pad, DG_D)

    def forward(self, e0_local: torch.Tensor):
        """e0_local: rows [r0, r1) of the layer-0 table.  Returns this rank's rows of
        (layer sum of graph 0, sum over the other graphs of their layer sums)."""
        be, G, L = self.backend, len(self.graphs), self.n_layers
        stage = be.empty(self.n_pad)
        stage[self.r0:self.r1] = e0_local
        e0 = self._gather_table(stage)
        sums = [be.empty(self.n_pad) for _ in range(G)]
        prev: List[Optional[torch.Tensor]] = [None] * G
        for l in range(L):
            last = l == L - 1
            flags = (DG_IN_PRESCALED if l > 0 else 0) | (0 if last else DG_OUT_PRESCALE)
            skip_sum = l == 0 and L >= 2
            if l == 1:
                flags |= DG_ACC_REBUILD
            outs, tasks = [], []
            v = lambda t: None if t is None else t[: self.n]   # the kernels take tables of exactly n rows
            for g in range(G):
                e_out = None if last else be.empty(self.n_pad)
                outs.append(e_out)
                acc_in = None if skip_sum else (e0 if l <= 1 else sums[g])
                tasks.append(dict(graph=self.graphs[g], x=v(e0 if l == 0 else prev[g]), e_out=v(e_out), acc_in=v(acc_in),
                                  acc_out=None if skip_sum else v(sums[g]), flags=flags, slope=self.slope))
            if self.hook:
                self.hook("begin", l)
            be.layer(tasks)
            prev = [None if o is None else self._gather_table(o) for o in outs]
            if self.hook:
                self.hook("end", l)
        het = be.sum_tables([t[: self.n] for t in sums[1:]])
        return sums[0][self.r0:self.r1], het[self.r0:self.r1]


class ShardedRecPropagation:
    """The propagation of HGDM.forward (DiffGraph-Rec/Model.py:53-92) for one rank's user block."""

    def __init__(self, interactions: Sequence[Tuple[np.ndarray, np.ndarray]], n_user: int, n_item: int, rank: int,
                 world: int, backend, n_layers: int = 2, slope: float = 0.5, group=None):
        self.rank, self.world, self.backend, self.group = rank, world, backend, group
        self.n_layers, self.slope = n_layers, slope
        self.u0, self.u1 = user_ranges(n_user, world)[rank]
        self.n_loc, self.n_item = self.u1 - self.u0, n_item
        self.n = self.n_loc + n_item
        self.local = [shard_bipartite(u, i, n_user, n_item, self.u0, self.u1) for (u, i) in interactions]
        self.graphs = [backend.make_graph(lg, self.n) for lg in self.local]
        self.edges_global = [lg.n_edges_global for lg in self.local]
        self.hook = None

    def _exchange(self, partials: torch.Tensor) -> torch.Tensor:
        """[G, n_item, 64] on every rank -> [world, G, n_item, 64] (partial sums in rank order)."""
        return all_gather_stack(partials, self.world, self.group)

    def forward(self, e0: torch.Tensor):
        """e0: [n_loc + n_item, 64] = this rank's user rows, then the (replicated) item rows.
        Returns (ui, het) in the same layout; item rows are identical on every rank."""
        be, G, L = self.backend, len(self.graphs), self.n_layers
        sums = [be.empty(self.n) for _ in range(G)]
        prev: List[Optional[torch.Tensor]] = [None] * G
        for l in range(L):
            last = l == L - 1
            flags = (DG_IN_PRESCALED if l > 0 else 0) | (0 if last else DG_OUT_PRESCALE)
            skip_sum = l == 0 and L >= 2          # the second layer rebuilds e0 + normalize(E_1) itself
            if l == 1:
                flags |= DG_ACC_REBUILD
            partials = torch.empty((G, self.n_item, DG_D), dtype=torch.float32, device=e0.device)
            outs, tasks = [], []
            for g in range(G):
                e_out = None if last else be.empty(self.n)
                outs.append(e_out)
                acc_in = None if skip_sum else (e0 if l <= 1 else sums[g])
                tasks.append(dict(graph=self.graphs[g], x=e0 if l == 0 else prev[g], e_out=e_out,
                                  acc_in=acc_in, acc_out=None if skip_sum else sums[g], flags=flags, slope=self.slope,
                                  partial_out=partials[g], partial_from=self.n_loc))
            if self.hook:
                self.hook("begin", l)
            be.layer(tasks)
            gathered = self._exchange(partials)
            be.finalize([dict(graph=self.graphs[g], partials=gathered[:, g], row_begin=self.n_loc, e_out=outs[g],
                              acc_in=tasks[g]["acc_in"], acc_out=tasks[g]["acc_out"], flags=flags, slope=self.slope,
                              x=tasks[g]["x"]) for g in range(G)])
            if self.hook:
                self.hook("end", l)
            prev = outs
        return sums[0], be.sum_tables(sums[1:])


# --------------------------------------------------------------------------------------
# bench.py leg for --gpus N > 1 (and for --workload industry at N = 1)
# --------------------------------------------------------------------------------------
def bench_sharded(args, rank: int, world: int, device) -> Optional[dict]:
    from . import data as dgdata
    import torch.distributed as dist
    workload = args.workload or "industry"
    wl = dgdata.rec_workload(workload)
    inter = [wl.target] + wl.behaviours
    ie = torch.from_numpy(dgdata.xavier_uniform(wl.n_item, 64, seed=1))
    # item side small enough to replicate (Industry): user shards + item partial sums; otherwise row blocks
    row_blocks = wl.n_item > 4096

    def make(rk, wd):
        if row_blocks:
            edges = [dgdata.bipartite_edges(u, i, wl.n_user) for (u, i) in inter]
            p = RowBlockPropagation(edges, wl.n, rk, wd, HipBackend(device))
            full = torch.cat([torch.from_numpy(dgdata.xavier_uniform(wl.n_user, 64, seed=0)), ie])
            return p, full[p.r0:p.r1].to(device)
        p = ShardedRecPropagation(inter, wl.n_user, wl.n_item, rk, wd, HipBackend(device))
        ue = torch.from_numpy(dgdata.xavier_uniform(wl.n_user, 64, seed=0)[p.u0:p.u1])
        return p, torch.cat([ue, ie]).to(device)
    prop, e0 = make(rank, world)

    events = []

    def hook(phase, layer):
        if hook.enabled:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            events.append(ev)
    hook.enabled = False
    prop.hook = hook

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        prop.forward(e0)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = prop.forward(e0)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    hook.enabled = True
    for _ in range(max(10, args.steps // 2)):
        prop.forward(e0)
    torch.cuda.synchronize()
    hook.enabled = False
    layer_ms = float(np.mean([events[k].elapsed_time(events[k + 1]) for k in range(0, len(events), 2)]))

    L = prop.n_layers
    nnz = prop.edges_global
    n_total = wl.n
    layer_bytes_global = sum(4 * e + 1032 * n_total for e in nnz)
    per_rank_bytes = layer_bytes_global / world
    achieved = per_rank_bytes / (layer_ms * 1e-3) / 1e9
    value = L * sum(nnz) * args.steps / elapsed

    # The same workload on ONE GPU, measured by rank 0 in this very run (the other ranks wait), so that the
    # scaling of this line can be read without a second run.
    single = None
    if world > 1 and rank == 0:
        del prop, out
        p1, e1 = make(0, 1)
        for _ in range(max(2, args.warmup // 2)):
            p1.forward(e1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            p1.forward(e1)
        torch.cuda.synchronize()
        single = L * sum(nnz) * args.steps / (time.perf_counter() - t1)

    res = None
    if rank == 0:
        res = {
            "metric": "SpMM edges/sec + achieved HBM GB/s per propagation layer",
            "value": value, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}: {wl.note}", "n_user": wl.n_user, "n_item": wl.n_item,
                       "graphs": len(nnz), "directed_edges_per_graph": nnz, "layers": L,
                       "step": ("HGDM.forward propagation, equal row blocks per rank, one all-gather of each graph's "
                                "activation table per layer") if row_blocks else
                               ("HGDM.forward propagation, user rows sharded over the ranks, one all-gather of the "
                                "item partial sums per layer"),
                       "parallelism": f"row blocks x{world}" if row_blocks else f"user-row shards x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "kernel": "one sharded propagation layer on rank 0 (propagation kernels + "
                         "exchange + partial_finalize_kernel)",
                         "bytes_per_launch": per_rank_bytes, "ms_per_launch": layer_ms},
        }
        if single is not None:
            res["config"]["same_workload_1gpu_edges_per_s"] = single
            res["config"]["speedup_over_1gpu"] = value / single
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res
