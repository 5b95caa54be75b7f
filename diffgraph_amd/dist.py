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

import os
import sys
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import (DG_ACC_REBUILD, DG_ACC_SHARED, DG_D, DG_IN_PRESCALED, DG_OUT_PRESCALE, DG_ROWS_FULL_ONLY,
                   DG_ROWS_PARTIAL_ONLY)


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
    sums_in_layer = True      # layer() takes DG_ACC_SHARED

    def __init__(self, device):
        self.device = device

    def can_split_rows(self, graphs, split: int) -> bool:
        """Whether layer() accepts DG_ROWS_PARTIAL_ONLY / DG_ROWS_FULL_ONLY for these graphs with the rows from
        `split` on as the partial-sum rows: every schedule must hold exactly that side in LDS."""
        return all(g.schedule()["lds_split"] == split for g in graphs)

    def make_graph(self, lg: LocalGraph, n_nodes: int):
        from .graph import CSRGraph
        return CSRGraph(lg.src, lg.dst, n_nodes, out_degrees=lg.degrees,
                        in_degrees=lg.degrees if lg.in_degrees is None else lg.in_degrees)

    def empty(self, rows: int) -> torch.Tensor:
        return torch.empty((rows, DG_D), dtype=torch.float32, device=self.device)

    def layer(self, tasks) -> None:
        from . import ops
        ops.gcn_layer_batched([ops.LayerTask(**t) for t in tasks])

    # prepared launches: argument tables built once, for passes repeated over the same buffers
    def prepare_layer(self, tasks):
        from . import ops
        return ops.prepare_gcn_layer_batched([ops.LayerTask(**t) for t in tasks])

    def prepare_finalize(self, tasks):
        from . import ops
        return ops.prepare_gcn_partial_finalize_batched([ops.FinalizeTask(**t) for t in tasks])

    def prepare_sum(self, tables, out):
        from . import ops
        return ops.prepare_sum_tables(tables, out)

    def finalize(self, tasks) -> None:
        from . import ops
        ops.gcn_partial_finalize_batched([ops.FinalizeTask(**t) for t in tasks])

    def sum_tables(self, tables):
        from . import ops
        return ops.sum_tables(tables)


class _Pending:
    """An exchange under way: `wait()` makes the current stream wait for it and returns the gathered tensor."""

    def __init__(self, out: torch.Tensor, work=None):
        self.out, self.work = out, work

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out


def _gather_async(t: torch.Tensor, world: int, group=None, out: Optional[torch.Tensor] = None) -> _Pending:
    """The collective itself: one all_gather_into_tensor, enqueued only (`async_op`)."""
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    flat = out.view((world * t.shape[0],) + tuple(t.shape[1:]))    # the concatenated form: every backend takes it
    return _Pending(out, dist.all_gather_into_tensor(flat, t.contiguous(), group=group, async_op=True))


def all_gather_start(t: torch.Tensor, world: int, group=None, out: Optional[torch.Tensor] = None,
                     always_collective: bool = False) -> _Pending:
    """As all_gather_stack (into `out` [world, ...] if given), but on GPU tensors over RCCL the collective is only
    enqueued (on the communicator's own stream, after everything already enqueued on the current stream): kernels
    launched before `wait()` run beside it."""
    import torch.distributed as dist
    # RCCL on GPU tensors, gloo on host tensors (always_collective: a one-rank job too goes through the collective)
    if (world > 1 or always_collective) and (t.is_cuda != (dist.get_backend(group) == "gloo")):
        return _gather_async(t, world, group, out)
    res = all_gather_stack(t, world, group)
    if out is None:
        return _Pending(res)
    out.copy_(res)
    return _Pending(out)


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
                 world: int, backend, n_layers: int = 2, slope: float = 0.5, group=None, reuse_buffers: bool = False):
        self.rank, self.world, self.backend, self.group = rank, world, backend, group
        self.n_layers, self.slope = n_layers, slope
        self.u0, self.u1 = user_ranges(n_user, world)[rank]
        self.n_loc, self.n_item = self.u1 - self.u0, n_item
        self.n = self.n_loc + n_item
        self.local = [shard_bipartite(u, i, n_user, n_item, self.u0, self.u1) for (u, i) in interactions]
        self.graphs = [backend.make_graph(lg, self.n) for lg in self.local]
        self.edges_global = [lg.n_edges_global for lg in self.local]
        self.hook = None
        # two launches per layer (item partial sums, then the user rows beside the exchange) where the kernels allow
        self.split_rows = backend.can_split_rows(self.graphs, self.n_loc)
        # ... and then the last layer's user rows of all graphs in one walk that writes `ui` and `het` directly
        self.sum_in_layer = n_layers == 2 and len(self.graphs) >= 2 and getattr(backend, "sums_in_layer", False)
        self.reuse_buffers = reuse_buffers
        self.always_collective = False    # True: a one-rank job issues the (trivial) collective as well
        self._programs = {}

    def _exchange(self, partials: torch.Tensor, out: torch.Tensor) -> _Pending:
        """[G, n_item, 64] on every rank -> out [world, G, n_item, 64] (partial sums in rank order), valid after
        `.wait()`."""
        return all_gather_start(partials, self.world, self.group, out, self.always_collective)

    def _program(self, e0: torch.Tensor):
        """Allocates the buffers of a forward pass over `e0` and prepares its launches.
        Returns (steps, (ui, het)): calling the steps in order fills the two output tables."""
        be, G, L = self.backend, len(self.graphs), self.n_layers
        sum_in_layer = self.split_rows and self.sum_in_layer
        sums = [be.empty(self.n) for _ in range(G)]
        prev: List[Optional[torch.Tensor]] = [None] * G
        steps, tail = [], []       # tail: the previous layer's wait + finalize, still to be placed
        for l in range(L):
            last = l == L - 1
            flags = (DG_IN_PRESCALED if l > 0 else 0) | (0 if last else DG_OUT_PRESCALE)
            skip_sum = l == 0 and L >= 2          # the second layer rebuilds e0 + normalize(E_1) itself
            if l == 1:
                flags |= DG_ACC_REBUILD
            partials = torch.empty((G, self.n_item, DG_D), dtype=torch.float32, device=e0.device)
            gathered = torch.empty((self.world,) + tuple(partials.shape), dtype=torch.float32, device=e0.device)
            outs, tasks = [], []
            for g in range(G):
                e_out = None if last else be.empty(self.n)
                outs.append(e_out)
                acc_in = None if skip_sum else (e0 if l <= 1 else sums[g])
                tasks.append(dict(graph=self.graphs[g], x=e0 if l == 0 else prev[g], e_out=e_out,
                                  acc_in=acc_in, acc_out=None if skip_sum else sums[g], flags=flags, slope=self.slope,
                                  partial_out=partials[g], partial_from=self.n_loc))
            pending: List[Optional[_Pending]] = [None]

            def start(p=partials, o=gathered, slot=pending):
                slot[0] = self._exchange(p, o)

            def wait(slot=pending):
                slot[0].wait()
                slot[0] = None
            fin = be.prepare_finalize(
                [dict(graph=self.graphs[g], partials=gathered[:, g], row_begin=self.n_loc, e_out=outs[g],
                      acc_in=tasks[g]["acc_in"], acc_out=tasks[g]["acc_out"], flags=flags, slope=self.slope,
                      x=tasks[g]["x"]) for g in range(G)])
            if self.split_rows:
                # Two launches: the item partial sums, then the user rows.  The first needs only the user rows of the
                # previous layer, so it goes ahead of that layer's wait + finalize: an exchange runs beside the user
                # rows of its own layer and the partial sums of the next.
                steps += [be.prepare_layer([dict(t, flags=t["flags"] | DG_ROWS_PARTIAL_ONLY) for t in tasks]), start]
                steps += tail
                if last and sum_in_layer:
                    # the user rows of `het` come straight out of the layer: no layer sum per behaviour graph is
                    # written for them (the item rows' sums, 361 rows per graph on the Industry shape, are added below)
                    het = sums[1] if G == 2 else be.empty(self.n)
                    steps.append(be.prepare_layer([dict(t, flags=t["flags"] | DG_ROWS_FULL_ONLY | DG_ACC_SHARED,
                                                        acc_out=sums[0] if g == 0 else het) for g, t in enumerate(tasks)]))
                else:
                    steps.append(be.prepare_layer([dict(t, flags=t["flags"] | DG_ROWS_FULL_ONLY) for t in tasks]))
            else:
                steps += tail + [be.prepare_layer(tasks), start]
            tail = [wait, fin]
            prev = outs
        steps += tail
        steps.insert(0, lambda: self.hook and self.hook("begin"))     # (timing hook round the layers, not the final sum)
        steps.append(lambda: self.hook and self.hook("end"))
        if G > 2 and sum_in_layer:
            steps.append(be.prepare_sum([t[self.n_loc:] for t in sums[1:]], het[self.n_loc:]))   # the item rows
        elif G > 2:
            het = be.empty(self.n)
            steps.append(be.prepare_sum(sums[1:], het))
        else:
            het = sums[1] if G == 2 else None
        return steps, (sums[0], het)

    def forward(self, e0: torch.Tensor):
        """e0: [n_loc + n_item, 64] = this rank's user rows, then the (replicated) item rows.
        Returns (ui, het) in the same layout; item rows are identical on every rank.  With `reuse_buffers` the
        buffers and prepared launches of the first call over a table are kept: the returned tables are then
        overwritten by the next call over the same table."""
        key = (e0.data_ptr(), tuple(e0.shape), str(e0.device))
        prog = self._programs.get(key) if self.reuse_buffers else None
        if prog is None:
            prog = self._program(e0)
            if self.reuse_buffers:
                self._programs = {key: prog + (e0,)}    # one table at a time; e0 is kept so its address stays taken
        for step in prog[0]:
            step()
        return prog[1]


    def capture(self, e0: torch.Tensor):
        """Records one forward pass over `e0` -- its prepared launches and its exchanges -- as a HIP graph (needs
        `reuse_buffers`: the recording holds the buffers' addresses).  Returns (graph, (ui, het)): `graph.replay()`
        refills the two tables.  Every rank must record, and later replay, the same number of times: the exchanges
        are part of the recording.  (RCCL's collectives can be recorded once the communicator has run one eagerly,
        which the two passes before the recording see to; `thread_local`: the communicator's watchdog thread polls
        its events while this thread records.)"""
        if not self.reuse_buffers:
            raise ValueError("capture() records addresses: construct with reuse_buffers=True")
        for _ in range(2):
            self.forward(e0)
        torch.cuda.synchronize(e0.device)
        hook, self.hook = self.hook, None
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.forward(e0)
        finally:
            self.hook = hook
        return graph, out


# --------------------------------------------------------------------------------------
# bench.py leg for --gpus N > 1 (and for --workload industry at N = 1)
# --------------------------------------------------------------------------------------
class _HostEvent:
    """Stand-in for a HIP event when the leg runs on host tensors (the gloo tests): wall-clock stamps."""

    def __init__(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other: "_HostEvent") -> float:
        return (other.t - self.t) * 1e3


def bench_sharded(args, rank: int, world: int, device, backend=None) -> Optional[dict]:
    """The sharded-propagation leg of bench.py.  `backend` defaults to the HIP kernels on `device`; the tests pass
    a host backend and a cpu device to run the same control flow over gloo."""
    from . import data as dgdata
    import torch.distributed as dist
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    backend = backend or HipBackend(device)
    workload = args.workload or "industry"
    wl = dgdata.rec_workload(workload)
    inter = [wl.target] + wl.behaviours
    ie = torch.from_numpy(dgdata.xavier_uniform(wl.n_item, 64, seed=1))
    # item side small enough to replicate (Industry): user shards + item partial sums; otherwise row blocks
    row_blocks = wl.n_item > 4096

    def make(rk, wd):
        if row_blocks:
            edges = [dgdata.bipartite_edges(u, i, wl.n_user) for (u, i) in inter]
            p = RowBlockPropagation(edges, wl.n, rk, wd, backend)
            full = torch.cat([torch.from_numpy(dgdata.xavier_uniform(wl.n_user, 64, seed=0)), ie])
            return p, full[p.r0:p.r1].to(device)
        p = ShardedRecPropagation(inter, wl.n_user, wl.n_item, rk, wd, backend, reuse_buffers=True)
        ue = torch.from_numpy(dgdata.xavier_uniform(wl.n_user, 64, seed=0)[p.u0:p.u1])
        return p, torch.cat([ue, ie]).to(device)
    prop, e0 = make(rank, world)

    events = []

    def hook(phase, layer=None):
        if hook.enabled:
            if on_gpu:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream())
            else:
                ev = _HostEvent()
            events.append(ev)
    hook.enabled = False
    prop.hook = hook

    def device_sync():
        if on_gpu:
            torch.cuda.synchronize()

    def sync():
        device_sync()
        if world > 1:
            dist.barrier()
            device_sync()

    def recorded(p, table):
        """The forward pass as a replayable HIP graph where that is on offer (GPU, user-row shards, not switched off
        with DG_SHARD_GRAPH=0), else the eager call; a recording that fails falls back to the eager call."""
        # (over gloo -- the one-GPU rehearsal of several ranks -- the exchange goes through host memory: nothing to record)
        rccl = p.world == 1 or dist.get_backend(p.group) == "nccl"
        if on_gpu and rccl and hasattr(p, "capture") and os.environ.get("DG_SHARD_GRAPH", "1") != "0":
            try:
                graph, res = p.capture(table)
                return (lambda: (graph.replay(), res)[1]), True
            except Exception as exc:       # noqa: BLE001 -- whatever the runtime or the communicator objects to
                print(f"[bench_sharded] forward pass not recorded as a graph ({exc!r}); running it eagerly", file=sys.stderr)
                device_sync()
        return (lambda: p.forward(table)), False
    run, as_graph = recorded(prop, e0)

    for _ in range(args.warmup):
        run()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    hook.enabled = True
    for _ in range(max(10, args.steps // 2)):
        prop.forward(e0)
    device_sync()
    hook.enabled = False
    L = prop.n_layers
    # row blocks: one event pair per layer; user-row shards: one pair round the layers of a forward pass (they interleave)
    layer_ms = float(np.mean([events[k].elapsed_time(events[k + 1]) for k in range(0, len(events), 2)]))
    if not row_blocks:
        layer_ms /= L
    nnz = prop.edges_global
    n_total = wl.n
    layer_bytes_global = sum(4 * e + 1032 * n_total for e in nnz)
    per_rank_bytes = layer_bytes_global / world
    achieved = per_rank_bytes / (layer_ms * 1e-3) / 1e9
    value = L * sum(nnz) * args.steps / elapsed
    # What one layer of this forward pass cannot avoid moving, per graph (mean over the layers, the rank's share):
    # user-row shards -- a 64-B record per user row, every user row of the layer's input read once by the item side
    # (256 B) plus a byte per link, the activation written (all layers but the last), from layer 2 on the running sum
    # rebuilt from e0 and the stored activation and written (3 x 256 B); row blocks -- the general kernel's streams.
    if row_blocks:
        per_layer = [sum(4 * e + n_total * (128 + 256 + (256 if l < L - 1 else 0) + (0 if l == 0 and L >= 2 else 768))
                         for e in nnz) for l in range(L)]
    else:
        per_layer = [sum(wl.n_user * (64 + 256 + (256 if l < L - 1 else 0) + (0 if l == 0 and L >= 2 else 768)) + e // 2
                         for e in nnz) for l in range(L)]
    bytes_compulsory = float(np.mean(per_layer)) / world

    # The same workload on ONE GPU, measured by rank 0 in this very run (the other ranks wait), so that the
    # scaling of this line can be read without a second run.
    single = None
    if world > 1 and rank == 0:
        del prop, out, run
        p1, e1 = make(0, 1)
        run1, _ = recorded(p1, e1)
        for _ in range(max(2, args.warmup // 2)):
            run1()
        device_sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            run1()
        device_sync()
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
                       "value_is": f"the {workload} forward propagation over {world} rank(s): its one-GPU partner is "
                                   "`same_workload_1gpu_edges_per_s` of this line (measured by rank 0 in this run) or "
                                   "`industry_1gpu_edges_per_s` of the default one-GPU run -- NOT the `value` of that run, "
                                   "which times a Tmall training step",
                       "parallelism": f"row blocks x{world}" if row_blocks else f"user-row shards x{world}",
                       "replayed_as_hip_graph": as_graph},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "kernel": "one sharded propagation layer on rank 0 (propagation kernels + "
                         "exchange + partial_finalize_kernel; user-row shards: a forward pass / its layers)",
                         "bytes_per_launch": per_rank_bytes, "ms_per_launch": layer_ms,
                         "bytes_compulsory": bytes_compulsory,
                         "frac_compulsory": bytes_compulsory / (layer_ms * 1e-3) / 1e9 / 8000.0},
        }
        if single is not None:
            res["config"]["same_workload_1gpu_edges_per_s"] = single
            res["config"]["speedup_over_1gpu"] = value / single
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res
