"""Functional wrappers: torch tensors in, HIP kernels through the C ABI, torch tensors out.

torch is plumbing here (device memory, streams); every op below is one call into
libdiffgraph_hip.so.  All wrappers validate shapes/dtypes/devices on the host before a
kernel is launched.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import DG_D, DiffGraphHipError
from .graph import CSRGraph, GraphSet


def _table(x: torch.Tensor, name: str, rows: Optional[int] = None) -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise DiffGraphHipError(f"{name}: expected a tensor on the GPU")
    if x.dtype != torch.float32:
        raise DiffGraphHipError(f"{name}: expected float32, got {x.dtype}")
    if x.dim() != 2 or x.shape[1] != DG_D:
        raise DiffGraphHipError(f"{name}: expected shape [rows,{DG_D}], got {tuple(x.shape)}")
    if rows is not None and x.shape[0] != rows:
        raise DiffGraphHipError(f"{name}: expected {rows} rows, got {x.shape[0]}")
    if not x.is_contiguous():
        raise DiffGraphHipError(f"{name}: tensor must be contiguous")
    return x


def _vec(x: torch.Tensor, name: str, dtype, count: Optional[int] = None) -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != dtype or x.dim() != 1 or not x.is_contiguous():
        raise DiffGraphHipError(f"{name}: expected a contiguous 1-D {dtype} tensor on the GPU")
    if count is not None and x.numel() != count:
        raise DiffGraphHipError(f"{name}: expected {count} elements, got {x.numel()}")
    return x


def _ptr(x: Optional[torch.Tensor]) -> Optional[int]:
    return None if x is None else x.data_ptr()


# --------------------------------------------------------------------------------------
# propagation
# --------------------------------------------------------------------------------------
def spmm_norm_act(graph: CSRGraph, x: torch.Tensor, slope: float = 0.5, transpose: bool = False) -> torch.Tensor:
    """y = LeakyReLU_slope(dinv_dst * sum_{j->i} dinv_src[j] x[j])  (DGLLayer.forward, Model.py:142-174)."""
    _table(x, "x", graph.num_nodes())
    y = torch.empty_like(x)
    rc = _lib.load().dg_spmm_norm_act_fwd(graph.handle, x.data_ptr(), y.data_ptr(), DG_D, float(slope),
                                          int(transpose), _lib.stream_ptr(x.device))
    _lib.check(rc, "dg_spmm_norm_act_fwd")
    return y


@dataclass
class LayerTask:
    graph: CSRGraph
    x: torch.Tensor
    e_out: Optional[torch.Tensor] = None
    acc_in: Optional[torch.Tensor] = None
    acc_out: Optional[torch.Tensor] = None
    flags: int = 0
    slope: float = 0.5
    partial_out: Optional[torch.Tensor] = None   # sharded runs: raw neighbour sums of rows >= partial_from
    partial_from: int = 0
    row_norm_out: Optional[torch.Tensor] = None  # [n] ||y|| before normalisation (for a normalised-feed backward)
    shared_rows: bool = False   # descriptive only: the task is one graph's share of a launch over a GraphSet


def _row_vector(v: torch.Tensor, name: str, n: int) -> None:
    if v.dtype != torch.float32 or not v.is_cuda or not v.is_contiguous() or v.numel() != n:
        raise DiffGraphHipError(f"{name}: expected a contiguous float32 GPU vector of {n} elements")


def gcn_layer_batched(tasks: Sequence[LayerTask]) -> None:
    """One fused propagation layer (SpMM + LeakyReLU + normalise + layer sum) over several graphs."""
    if tasks:
        prepare_gcn_layer_batched(tasks)()


def prepare_gcn_layer_batched(tasks: Sequence[LayerTask]):
    """Checks the tasks and builds the call's argument table once; returns a function that launches the layer on the
    current stream.  For loops that run the same launch over the same buffers many times."""
    n_tasks = len(tasks)
    if n_tasks > _lib.DG_MAX_TASKS:
        raise DiffGraphHipError(f"at most {_lib.DG_MAX_TASKS} graphs per batched layer launch")
    arr = (_lib.LayerTask * n_tasks)()
    dev = tasks[0].x.device
    for k, t in enumerate(tasks):
        n = t.graph.num_nodes()
        _table(t.x, "x", n)
        for nm in ("e_out", "acc_in", "acc_out"):
            v = getattr(t, nm)
            if v is not None:
                _table(v, nm, t.graph.num_rows())
        if t.partial_out is not None:
            _table(t.partial_out, "partial_out", n - int(t.partial_from))
        if t.row_norm_out is not None:
            _row_vector(t.row_norm_out, "row_norm_out", n)
        arr[k] = _lib.LayerTask(t.graph.handle, t.x.data_ptr(), _ptr(t.e_out), _ptr(t.acc_in), _ptr(t.acc_out),
                                int(t.flags), float(t.slope), _ptr(t.partial_out), int(t.partial_from),
                                _ptr(t.row_norm_out))
    fn, keep = _lib.load().dg_gcn_layer_fwd, list(tasks)   # the table holds raw pointers: keep the tensors alive

    def launch(_keep=keep):
        _lib.check(fn(arr, n_tasks, DG_D, _lib.stream_ptr(dev)), "dg_gcn_layer_fwd")
    return launch


@dataclass
class FinalizeTask:
    graph: CSRGraph
    partials: torch.Tensor          # [n_parts, n_rows, 64]; the parts may be strided (a slice of a gathered buffer)
    row_begin: int
    e_out: Optional[torch.Tensor]
    acc_in: Optional[torch.Tensor]
    acc_out: Optional[torch.Tensor]
    flags: int = 0
    slope: float = 0.5
    x: Optional[torch.Tensor] = None  # the layer's gather source (read under DG_ACC_REBUILD)


def gcn_partial_finalize_batched(tasks: Sequence[FinalizeTask]) -> None:
    """Second half of a row-sharded layer for all its graphs in one launch (see dg_gcn_partial_finalize)."""
    if tasks:
        prepare_gcn_partial_finalize_batched(tasks)()


def prepare_gcn_partial_finalize_batched(tasks: Sequence[FinalizeTask]):
    """As prepare_gcn_layer_batched, for the second half of a row-sharded layer."""
    n_tasks = len(tasks)
    if n_tasks > _lib.DG_MAX_TASKS:
        raise DiffGraphHipError(f"at most {_lib.DG_MAX_TASKS} graphs per batched launch")
    arr = (_lib.FinalizeTask * n_tasks)()
    for k, t in enumerate(tasks):
        p = t.partials
        if p.dim() != 3 or p.shape[2] != DG_D or p.dtype != torch.float32 or not p.is_cuda \
                or p.stride(2) != 1 or p.stride(1) != DG_D or p.stride(0) % DG_D != 0:
            raise DiffGraphHipError("partials: expected a float32 GPU tensor [n_parts, n_rows, 64] with contiguous rows")
        n = t.graph.num_nodes()
        for nm, v in (("e_out", t.e_out), ("acc_in", t.acc_in), ("acc_out", t.acc_out), ("x", t.x)):
            if v is not None:
                _table(v, nm, n)
        arr[k] = _lib.FinalizeTask(t.graph.handle, p.data_ptr(), p.shape[0], p.stride(0) // DG_D, int(t.row_begin),
                                   p.shape[1], _ptr(t.x), _ptr(t.e_out), _ptr(t.acc_in), _ptr(t.acc_out), int(t.flags),
                                   float(t.slope))
    fn, keep, dev = _lib.load().dg_gcn_partial_finalize_batched, list(tasks), tasks[0].partials.device

    def launch(_keep=keep):
        _lib.check(fn(arr, n_tasks, DG_D, _lib.stream_ptr(dev)), "dg_gcn_partial_finalize_batched")
    return launch


def gcn_partial_finalize(graph: CSRGraph, partials: torch.Tensor, row_begin: int, e_out, acc_in, acc_out, flags: int,
                         slope: float, x: Optional[torch.Tensor] = None) -> None:
    """One graph's share of gcn_partial_finalize_batched."""
    gcn_partial_finalize_batched([FinalizeTask(graph, partials, row_begin, e_out, acc_in, acc_out, flags, slope, x)])


@dataclass
class BwdTask:
    graph: CSRGraph
    w_in: Optional[torch.Tensor]      # scaled dZ of the layer above, or None for the last layer
    e_saved: Optional[torch.Tensor]   # this layer's forward activation, or None for layer 0
    d_sum: Optional[torch.Tensor]     # gradient wrt the graph's layer sum
    w_out: torch.Tensor
    flags: int = 0
    slope: float = 0.5
    row_norm: Optional[torch.Tensor] = None   # [n] saved ||y||, read under DG_BWD_NORM_FEED


def gcn_layer_bwd_batched(tasks: Sequence[BwdTask]) -> None:
    """Backward of one fused propagation layer over several graphs (see dg_gcn_layer_bwd)."""
    if not tasks:
        return
    n_tasks = len(tasks)
    if n_tasks > _lib.DG_MAX_TASKS:
        raise DiffGraphHipError(f"at most {_lib.DG_MAX_TASKS} graphs per batched layer launch")
    arr = (_lib.BwdTask * n_tasks)()
    for k, t in enumerate(tasks):
        n = t.graph.num_nodes()
        _table(t.w_out, "w_out", n)
        for nm in ("w_in", "e_saved", "d_sum"):
            v = getattr(t, nm)
            if v is not None:
                _table(v, nm, n)
        if t.row_norm is not None:
            _row_vector(t.row_norm, "row_norm", n)
        arr[k] = _lib.BwdTask(t.graph.handle, _ptr(t.w_in), _ptr(t.e_saved), _ptr(t.d_sum), t.w_out.data_ptr(),
                              int(t.flags), float(t.slope), _ptr(t.row_norm))
    rc = _lib.load().dg_gcn_layer_bwd(arr, n_tasks, DG_D, _lib.stream_ptr(tasks[0].w_out.device))
    _lib.check(rc, "dg_gcn_layer_bwd")


def sum_tables(tables: Sequence[torch.Tensor]) -> torch.Tensor:
    """((t0 + t1) + t2) + ... like Python's sum(list) (Model.py:87)."""
    if len(tables) == 1:
        return tables[0]
    out = torch.empty_like(tables[0])
    prepare_sum_tables(tables, out)()
    return out


def prepare_sum_tables(tables: Sequence[torch.Tensor], out: torch.Tensor):
    """Returns a function that writes ((t0 + t1) + ...) into `out` on the current stream."""
    for t in list(tables) + [out]:
        _table(t, "table", tables[0].shape[0])
    ptrs = (C.c_void_p * len(tables))(*[t.data_ptr() for t in tables])
    fn, keep, n_in = _lib.load().dg_sum_tables, list(tables) + [out], len(tables)

    def launch(_keep=keep):
        _lib.check(fn(ptrs, n_in, out.data_ptr(), out.numel(), _lib.stream_ptr(out.device)), "dg_sum_tables")
    return launch


def _stacked(x: torch.Tensor, name: str, gset: GraphSet) -> torch.Tensor:
    want = (gset.num_graphs(), gset.num_nodes(), DG_D)
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32 or tuple(x.shape) != want \
            or not x.is_contiguous():
        raise DiffGraphHipError(f"{name}: expected a contiguous float32 GPU tensor of shape {want}")
    return x


def set_layer_first(gset: GraphSet, x: torch.Tensor, e_out: torch.Tensor, slope: float = 0.5) -> None:
    """First layer over all graphs of the set from the one table x [n,64]: e_out[g] = the activation of graph g,
    times the row's source norm (what the next layer gathers)."""
    _table(x, "x", gset.num_nodes())
    _stacked(e_out, "e_out", gset)
    rc = _lib.load().dg_set_layer_first(gset.handle, x.data_ptr(), e_out.data_ptr(), float(slope), DG_D, _lib.stream_ptr(x.device))
    _lib.check(rc, "dg_set_layer_first")


def set_layer_last(gset: GraphSet, x: torch.Tensor, e0: torch.Tensor, sum_first: torch.Tensor, sum_rest: torch.Tensor,
                   e_out: Optional[torch.Tensor] = None, slope: float = 0.5) -> None:
    """Second (last) layer over the set: gathers graph g's rows from x[g] (set_layer_first's output), writes the
    first graph's layer sum e0 + normalize(E_1) + normalize(E_2) and the sum of the other graphs' layer sums, and --
    for a backward pass -- the activations E_2 into e_out[g]."""
    n = gset.num_nodes()
    _stacked(x, "x", gset)
    _table(e0, "e0", n)
    _table(sum_first, "sum_first", n)
    _table(sum_rest, "sum_rest", n)
    if e_out is not None:
        _stacked(e_out, "e_out", gset)
    rc = _lib.load().dg_set_layer_last(gset.handle, x.data_ptr(), e0.data_ptr(), sum_first.data_ptr(), sum_rest.data_ptr(),
                                       _ptr(e_out), float(slope), DG_D, _lib.stream_ptr(x.device))
    _lib.check(rc, "dg_set_layer_last")


def set_layer_bwd_top(gset: GraphSet, e_saved: torch.Tensor, d_first: torch.Tensor, d_rest: torch.Tensor,
                      w_out: torch.Tensor, slope: float = 0.5, row_flags: Optional[torch.Tensor] = None) -> None:
    """Backward through the last layer's LeakyReLU / normalize / layer sum for every graph of the set (no gather).
    row_flags (int32 [nodes]): 0 where the rows of d_first and d_rest are zero -- those rows of w_out become zeros unread."""
    n = gset.num_nodes()
    if row_flags is not None:
        _vec(row_flags, "row_flags", torch.int32, n)
    _stacked(e_saved, "e_saved", gset)
    _stacked(w_out, "w_out", gset)
    _table(d_first, "d_first", n)
    _table(d_rest, "d_rest", n)
    rc = _lib.load().dg_set_layer_bwd_top(gset.handle, e_saved.data_ptr(), d_first.data_ptr(), d_rest.data_ptr(),
                                          w_out.data_ptr(), _ptr(row_flags), float(slope), DG_D, _lib.stream_ptr(w_out.device))
    _lib.check(rc, "dg_set_layer_bwd_top")


def set_layer_bwd_mid(gset: GraphSet, w_in: torch.Tensor, e_saved: torch.Tensor, d_first: torch.Tensor,
                      d_rest: torch.Tensor, w_out: torch.Tensor, slope: float = 0.5) -> None:
    """Backward through the first layer for every graph of the set: the transposed propagation of w_in (the top
    step's output) fused with the LeakyReLU / normalize backward at e_saved = set_layer_first's output."""
    n = gset.num_nodes()
    for nm, v in (("w_in", w_in), ("e_saved", e_saved), ("w_out", w_out)):
        _stacked(v, nm, gset)
    _table(d_first, "d_first", n)
    _table(d_rest, "d_rest", n)
    rc = _lib.load().dg_set_layer_bwd_mid(gset.handle, w_in.data_ptr(), e_saved.data_ptr(), d_first.data_ptr(),
                                          d_rest.data_ptr(), w_out.data_ptr(), float(slope), DG_D,
                                          _lib.stream_ptr(w_out.device))
    _lib.check(rc, "dg_set_layer_bwd_mid")


def set_layer_bwd_last(gset: GraphSet, w_in: torch.Tensor, g_out: torch.Tensor) -> None:
    """The last backward step over the set: what reaches the layer-0 table through all graphs' propagation, as one table
    (the step is linear: no activation or normalisation between the graphs' terms)."""
    _stacked(w_in, "w_in", gset)
    _table(g_out, "g_out", gset.num_nodes())
    rc = _lib.load().dg_set_layer_bwd_last(gset.handle, w_in.data_ptr(), g_out.data_ptr(), DG_D, _lib.stream_ptr(g_out.device))
    _lib.check(rc, "dg_set_layer_bwd_last")


# --------------------------------------------------------------------------------------
# diffusion
# --------------------------------------------------------------------------------------
def set_layer_bwd_last_adam(gset: GraphSet, w_in: torch.Tensor, d_first: torch.Tensor, d_rest: torch.Tensor, param: torch.Tensor,
                            exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, state: torch.Tensor, lr: float, beta1: float,
                            beta2: float, eps: float, weight_decay: float) -> None:
    """set_layer_bwd_last with the table's Adam step in its epilogue: the gradient d_first + d_rest (once per behaviour
    graph) + (the gathered sum) updates param / exp_avg / exp_avg_sq in place, at step count state[0] + 1, and is never
    written; the launch moves the step count state[0] on itself."""
    n = gset.num_nodes()
    _stacked(w_in, "w_in", gset)
    for nm, t in (("d_first", d_first), ("d_rest", d_rest), ("param", param), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _table(t, nm, n)
    _vec(state, "Adam state", torch.float32, 4)
    rc = _lib.load().dg_set_layer_bwd_last_adam(gset.handle, w_in.data_ptr(), d_first.data_ptr(), d_rest.data_ptr(), param.data_ptr(),
                                                exp_avg.data_ptr(), exp_avg_sq.data_ptr(), state.data_ptr(), float(lr), float(beta1),
                                                float(beta2), float(eps), float(weight_decay), DG_D, _lib.stream_ptr(param.device))
    _lib.check(rc, "dg_set_layer_bwd_last_adam")


class DenoiserParams:
    """Device pointers + host frequencies of one Denoise module, and its per-timestep bias table."""

    def __init__(self, w_emb, b_emb, w_in, b_in, w_out, b_out, norm: bool = True):
        if tuple(w_emb.shape) != (8, 8) or tuple(w_in.shape) != (DG_D, DG_D + 8) or tuple(w_out.shape) != (DG_D, DG_D):
            raise DiffGraphHipError("denoiser weights must be emb 8x8, in 64x72, out 64x64 (dims='[64]', d_emb_size=8)")
        self.tensors = [t.detach().contiguous() for t in (w_emb, b_emb, w_in, b_in, w_out, b_out)]
        for t in self.tensors:
            if not t.is_cuda or t.dtype != torch.float32:
                raise DiffGraphHipError("denoiser weights must be float32 GPU tensors")
        half = 4
        # Model.py:215 -- evaluated with the same torch CPU ops as the reference
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
        self._freqs = (C.c_float * 4)(*[float(f) for f in freqs])
        self.c = _lib.Denoiser(*[t.data_ptr() for t in self.tensors], self._freqs, int(bool(norm)))
        self.device = self.tensors[0].device
        self._tb = None
        self._tb_steps = 0
        self.wg_tab = None      # [n_steps, 16] operand table of the fused weight gradients (wgrad_operand_tables)

    def time_bias(self, n_steps: int, refresh: bool = False) -> torch.Tensor:
        if self._tb is None or self._tb_steps != n_steps or refresh:
            tb = torch.empty((n_steps, DG_D), dtype=torch.float32, device=self.device)
            rc = _lib.load().dg_time_bias_table(C.byref(self.c), n_steps, tb.data_ptr(), _lib.stream_ptr(self.device))
            _lib.check(rc, "dg_time_bias_table")
            self._tb, self._tb_steps = tb, n_steps
        return self._tb


def refresh_time_bias(params: Sequence["DenoiserParams"], n_steps: int) -> None:
    """DenoiserParams.time_bias(n_steps, refresh=True) for several denoisers (at most four) in one launch."""
    dev = params[0].device
    tabs = [torch.empty((n_steps, DG_D), dtype=torch.float32, device=dev) for _ in params]
    dns = (C.POINTER(_lib.Denoiser) * len(params))(*[C.pointer(p.c) for p in params])
    ptrs = (C.c_void_p * len(params))(*[t.data_ptr() for t in tabs])
    rc = _lib.load().dg_time_bias_tables(dns, ptrs, len(params), n_steps, _lib.stream_ptr(dev))
    _lib.check(rc, "dg_time_bias_tables")
    for p, t in zip(params, tabs):
        p._tb, p._tb_steps = t, n_steps


def _check_t(t: torch.Tensor, n: int, n_steps: int, validate: bool) -> torch.Tensor:
    _vec(t, "timesteps", torch.int64, n)
    if validate and n > 0:
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= n_steps:
            raise DiffGraphHipError(f"timesteps must lie in [0,{n_steps}); got [{lo},{hi}]")
    return t


def q_sample(x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor, sqrt_ac: torch.Tensor, sqrt_1mac: torch.Tensor,
             validate: bool = True) -> torch.Tensor:
    _table(x0, "x_start")
    _table(noise, "noise", x0.shape[0])
    n_steps = sqrt_ac.numel()
    _vec(sqrt_ac, "sqrt_alphas_cumprod", torch.float32)
    _vec(sqrt_1mac, "sqrt_one_minus_alphas_cumprod", torch.float32, n_steps)
    _check_t(t, x0.shape[0], n_steps, validate)
    out = torch.empty_like(x0)
    rc = _lib.load().dg_q_sample(x0.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(),
                                 n_steps, x0.shape[0], DG_D, out.data_ptr(), _lib.stream_ptr(x0.device))
    _lib.check(rc, "dg_q_sample")
    return out


def denoise_fwd(dn: DenoiserParams, x: torch.Tensor, t: torch.Tensor, n_steps: int, validate: bool = True, drop_p: float = 0.0,
                keep_mask: Optional[torch.Tensor] = None, drop_seed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Denoise.forward; with drop_p > 0 the train-mode form: dropout over keep_mask (uint8 / bool [n, 64]) or over a mask
    drawn from drop_seed (int64[1] on the GPU)."""
    _table(x, "x")
    _check_t(t, x.shape[0], n_steps, validate)
    tb = dn.time_bias(n_steps)
    out = torch.empty_like(x)
    if drop_p > 0:
        if keep_mask is not None:
            if not keep_mask.is_cuda or keep_mask.dtype not in (torch.uint8, torch.bool) or tuple(keep_mask.shape) != tuple(x.shape):
                raise DiffGraphHipError("keep_mask: expected a uint8 / bool GPU tensor of x's shape")
            keep_mask = keep_mask.contiguous().view(torch.uint8)
        elif drop_seed is not None:
            _vec(drop_seed, "drop_seed", torch.int64, 1)
        rc = _lib.load().dg_denoise_fwd_train(C.byref(dn.c), tb.data_ptr(), n_steps, x.data_ptr(), t.data_ptr(), x.shape[0], DG_D,
                                              _ptr(keep_mask), _ptr(drop_seed), float(drop_p), out.data_ptr(),
                                              _lib.stream_ptr(x.device))
        _lib.check(rc, "dg_denoise_fwd_train")
        return out
    rc = _lib.load().dg_denoise_fwd(C.byref(dn.c), tb.data_ptr(), n_steps, x.data_ptr(), t.data_ptr(), x.shape[0], DG_D,
                                    out.data_ptr(), _lib.stream_ptr(x.device))
    _lib.check(rc, "dg_denoise_fwd")
    return out


def p_sample_loop(dn: DenoiserParams, x_T: torch.Tensor, coef1: torch.Tensor, coef2: torch.Tensor) -> torch.Tensor:
    _table(x_T, "x_T")
    n_steps = coef1.numel()
    _vec(coef1, "posterior_mean_coef1", torch.float32)
    _vec(coef2, "posterior_mean_coef2", torch.float32, n_steps)
    tb = dn.time_bias(n_steps)
    out = torch.empty_like(x_T)
    rc = _lib.load().dg_p_sample_loop(C.byref(dn.c), tb.data_ptr(), coef1.data_ptr(), coef2.data_ptr(), n_steps,
                                      x_T.data_ptr(), x_T.shape[0], DG_D, out.data_ptr(), _lib.stream_ptr(x_T.device))
    _lib.check(rc, "dg_p_sample_loop")
    return out


def p_sample_loop_batched(pairs, coef1: torch.Tensor, coef2: torch.Tensor) -> List[torch.Tensor]:
    """pairs: [(DenoiserParams, x_T), ...] -> [x_0, ...]; all reverse loops in one launch."""
    n_steps = coef1.numel()
    _vec(coef1, "posterior_mean_coef1", torch.float32)
    _vec(coef2, "posterior_mean_coef2", torch.float32, n_steps)
    arr = (_lib.PSampleTask * len(pairs))()
    outs, keep = [], []
    for k, (dn, x_T) in enumerate(pairs):
        _table(x_T, "x_T")
        tb = dn.time_bias(n_steps)
        out = torch.empty_like(x_T)
        outs.append(out)
        keep.append(tb)
        arr[k] = _lib.PSampleTask(C.pointer(dn.c), tb.data_ptr(), x_T.data_ptr(), out.data_ptr(), x_T.shape[0])
    rc = _lib.load().dg_p_sample_loop_batched(arr, len(pairs), coef1.data_ptr(), coef2.data_ptr(), n_steps, DG_D,
                                              _lib.stream_ptr(coef1.device))
    _lib.check(rc, "dg_p_sample_loop_batched")
    return outs


def diffusion_loss_fwd(dn: DenoiserParams, target: torch.Tensor, x_start: torch.Tensor, noise: Optional[torch.Tensor],
                       ts: torch.Tensor, sqrt_ac: Optional[torch.Tensor], sqrt_1mac: Optional[torch.Tensor],
                       loss_w: torch.Tensor, want_x_t: bool = False, validate: bool = True):
    """Returns (row_loss fp64 [n], out [n,64], x_t or None)."""
    n = x_start.shape[0]
    _table(x_start, "x_start")
    _table(target, "targetEmbeds", n)
    n_steps = loss_w.numel()
    _vec(loss_w, "loss weights", torch.float64)
    _check_t(ts, n, n_steps, validate)
    if sqrt_ac is not None:
        _table(noise, "noise", n)
        _vec(sqrt_ac, "sqrt_alphas_cumprod", torch.float32, n_steps)
        _vec(sqrt_1mac, "sqrt_one_minus_alphas_cumprod", torch.float32, n_steps)
    tb = dn.time_bias(n_steps)
    out = torch.empty_like(x_start)
    row_loss = torch.empty(n, dtype=torch.float64, device=x_start.device)
    x_t = torch.empty_like(x_start) if (want_x_t and sqrt_ac is not None) else None
    rc = _lib.load().dg_diffusion_loss_fwd(C.byref(dn.c), tb.data_ptr(), n_steps, target.data_ptr(), x_start.data_ptr(),
                                           _ptr(noise) if sqrt_ac is not None else None, ts.data_ptr(), _ptr(sqrt_ac),
                                           _ptr(sqrt_1mac), loss_w.data_ptr(), n, DG_D, out.data_ptr(),
                                           row_loss.data_ptr(), _ptr(x_t), _lib.stream_ptr(x_start.device))
    _lib.check(rc, "dg_diffusion_loss_fwd")
    return row_loss, out, x_t


def diffusion_loss_bwd(dn: DenoiserParams, target, x_start, noise, ts, sqrt_ac, sqrt_1mac, loss_w, g_row_loss, g_out):
    """Backward of diffusion_loss_fwd.  Returns (d_x_start, d_target, dout_tot, hid, dpre, xn)."""
    n = x_start.shape[0]
    n_steps = loss_w.numel()
    _table(x_start, "x_start")
    _table(target, "targetEmbeds", n)
    _vec(g_row_loss, "grad of row_loss", torch.float64, n)
    if g_out is not None:
        _table(g_out, "grad of out", n)
    tb = dn.time_bias(n_steps)
    outs = [torch.empty_like(x_start) for _ in range(6)]
    rc = _lib.load().dg_diffusion_loss_bwd(C.byref(dn.c), tb.data_ptr(), n_steps, target.data_ptr(), x_start.data_ptr(),
                                           _ptr(noise) if sqrt_ac is not None else None, ts.data_ptr(), _ptr(sqrt_ac),
                                           _ptr(sqrt_1mac), loss_w.data_ptr(), g_row_loss.data_ptr(), _ptr(g_out), n, DG_D,
                                           *[o.data_ptr() for o in outs], _lib.stream_ptr(x_start.device))
    _lib.check(rc, "dg_diffusion_loss_bwd")
    return tuple(outs)


@dataclass
class LossSpan:
    """Rows [begin, begin + n) of the shared tables go through denoiser `dn` (diffusion_loss_*_spans).
    drop_p > 0: nn.Dropout(drop_p) in training mode between F.normalize and the first Linear (Model.py:221-224),
    with the span's rows of the `keep_mask` table given to the call, or a mask drawn in the kernel from its `drop_seed`."""
    dn: DenoiserParams
    begin: int
    n: int
    drop_p: float = 0.0
    # only these rows of the span take part (int32 row numbers inside the span, ascending, on the device) ...
    rows: Optional[torch.Tensor] = None
    n_rows: Optional[torch.Tensor] = None     # ... this many of them (int32[1] on the device); see active_rows()


def new_draw_state(device, seed: Optional[int] = None) -> torch.Tensor:
    """int64[4] on `device`: {seed (default: torch's initial seed), draws made so far, dropout key of the last draw, 0}."""
    seed = torch.initial_seed() if seed is None else int(seed)
    return torch.tensor([seed & (2 ** 63 - 1), 0, 0, 0], dtype=torch.int64, device=device)


def diffusion_draw(state: torch.Tensor, n: int, n_steps: int, ts: bool = True, noise: bool = True, out_ts=None, out_noise=None):
    """(ts int64[n], noise fp32[n, 64]) of one training_losses2 call from `state` (new_draw_state), which moves on by
    one draw; state[2:3] afterwards keys the dropout masks of the same call.  out_ts / out_noise: buffers to fill."""
    _vec(state, "draw state", torch.int64, 4)
    dev = state.device
    t = (torch.empty(n, dtype=torch.int64, device=dev) if out_ts is None else out_ts) if ts else None
    z = (torch.empty((n, DG_D), dtype=torch.float32, device=dev) if out_noise is None else out_noise) if noise else None
    if t is not None:
        _vec(t, "ts", torch.int64, n)
    if z is not None:
        _table(z, "noise", n)
    rc = _lib.load().dg_diffusion_draw(state.data_ptr(), n, int(n_steps), DG_D, None if t is None else t.data_ptr(),
                                       None if z is None else z.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_diffusion_draw")
    return t, z


def dropout_keep_mask(drop_seed: torch.Tensor, task: int, drop_p: float, n: int) -> torch.Tensor:
    """The keep mask (uint8 [n, 64]) the loss kernels draw for span number `task` from `drop_seed` (int64[1] on the GPU)."""
    _vec(drop_seed, "drop_seed", torch.int64, 1)
    mask = torch.empty((n, DG_D), dtype=torch.uint8, device=drop_seed.device)
    rc = _lib.load().dg_dropout_keep_mask(drop_seed.data_ptr(), int(task), float(drop_p), n, DG_D, mask.data_ptr(),
                                          _lib.stream_ptr(drop_seed.device))
    _lib.check(rc, "dg_dropout_keep_mask")
    return mask


def _check_dropout(spans: Sequence[LossSpan], n: int, keep_mask, drop_seed):
    if not any(sp.drop_p for sp in spans):
        return
    if keep_mask is not None:
        if not keep_mask.is_cuda or keep_mask.dtype not in (torch.uint8, torch.bool) or tuple(keep_mask.shape) != (n, DG_D) \
                or not keep_mask.is_contiguous():
            raise DiffGraphHipError(f"keep_mask: expected a contiguous uint8/bool GPU tensor [{n},{DG_D}]")
    elif drop_seed is None:
        raise DiffGraphHipError("dropout needs a keep_mask or a drop_seed")
    else:
        _vec(drop_seed, "drop_seed", torch.int64, 1)


def _span_tasks(spans: Sequence[LossSpan], n_steps: int, tables: dict, keep_mask=None, drop_seed=None):
    arr = (_lib.LossTask * len(spans))()
    for k, sp in enumerate(spans):
        t = _lib.LossTask()
        t.dn = C.pointer(sp.dn.c)
        t.time_bias = sp.dn.time_bias(n_steps).data_ptr()
        t.n = sp.n
        t.drop_p = float(sp.drop_p)
        if sp.drop_p:
            t.keep_mask = None if keep_mask is None else keep_mask.data_ptr() + sp.begin * DG_D
            t.drop_seed = None if drop_seed is None else drop_seed.data_ptr()
        for name, (tensor, row_bytes) in tables.items():
            setattr(t, name, None if tensor is None else tensor.data_ptr() + sp.begin * row_bytes)
        if sp.rows is not None:
            if sp.rows.dtype != torch.int32 or not sp.rows.is_cuda or not sp.rows.is_contiguous() or sp.n_rows is None \
                    or sp.n_rows.dtype != torch.int32 or sp.n_rows.numel() != 1 or not 0 < sp.rows.numel():
                raise DiffGraphHipError("a span's row list: contiguous int32 GPU vector with its int32[1] count")
            t.rows, t.n_rows, t.rows_cap = sp.rows.data_ptr(), sp.n_rows.data_ptr(), min(sp.rows.numel(), sp.n)
        arr[k] = t
    return arr


def active_rows_buffers(n_batch: int, n: int, device):
    """(flags scratch int32[n], rows_u int32[B], rows_i int32[2B], counts int32[2]) for active_rows(out=...)."""
    mk = lambda m: torch.empty(m, dtype=torch.int32, device=device)
    return mk(n), mk(n_batch), mk(2 * n_batch), mk(2)


def clear_table(t: torch.Tensor) -> None:
    """Zeroes a table with the runtime's memset on the current stream (a memset node in a recorded step)."""
    rc = _lib.load().dg_clear(t.data_ptr(), t.numel() * t.element_size(), _lib.stream_ptr(t.device))
    _lib.check(rc, "dg_clear")


def active_rows(ancs, poss, negs, n_user: int, n_item: int, out=None):
    """The table rows a batch of HGDM.cal_loss names (Model.py:96-104): (rows_u int32[B] = the distinct ancs ascending,
    rows_i int32[2B] = the distinct poss and negs ascending, counts int32[2]), all on the device -- for LossSpan.rows:
    no other row of training_losses2's tables is read by the loss, and no other row gets a gradient from it.
    out: active_rows_buffers(...) made beforehand (e.g. on another stream than the one this is launched on)."""
    B = ancs.numel()
    for v, nm in ((ancs, "ancs"), (poss, "poss"), (negs, "negs")):
        _index_vec(v, nm, B)
    dev = ancs.device
    flags, rows_u, rows_i, counts = active_rows_buffers(B, n_user + n_item, dev) if out is None else out
    rc = _lib.load().dg_active_rows(ancs.data_ptr(), poss.data_ptr(), negs.data_ptr(), B, int(n_user), int(n_item), flags.data_ptr(),
                                    rows_u.data_ptr(), rows_i.data_ptr(), counts.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_active_rows")
    return rows_u, rows_i, counts


def diffusion_loss_fwd_spans(spans: Sequence[LossSpan], target, x_start, noise, ts, sqrt_ac, sqrt_1mac, loss_w,
                             validate: bool = True, keep_mask=None, drop_seed=None):
    """diffusion_loss_fwd over one set of [N,64] tables whose row spans use different denoisers (the user rows
    and the item rows of the recommendation model), in ONE launch.  Returns (row_loss fp64 [N], out [N,64])."""
    n = x_start.shape[0]
    _table(x_start, "x_start")
    _table(target, "targetEmbeds", n)
    n_steps = loss_w.numel()
    _vec(loss_w, "loss weights", torch.float64)
    _check_t(ts, n, n_steps, validate)
    if sqrt_ac is not None:
        _table(noise, "noise", n)
        _vec(sqrt_ac, "sqrt_alphas_cumprod", torch.float32, n_steps)
        _vec(sqrt_1mac, "sqrt_one_minus_alphas_cumprod", torch.float32, n_steps)
    if sum(sp.n for sp in spans) != n or any(sp.begin < 0 or sp.begin + sp.n > n for sp in spans):
        raise DiffGraphHipError("row spans must tile the tables")
    _check_dropout(spans, n, keep_mask, drop_seed)
    out = torch.empty_like(x_start)
    row_loss = torch.empty(n, dtype=torch.float64, device=x_start.device)
    rb = DG_D * 4
    arr = _span_tasks(spans, n_steps, {"target": (target, rb), "x_start": (x_start, rb),
                                       "noise": (noise if sqrt_ac is not None else None, rb), "ts": (ts, 8), "out": (out, rb),
                                       "row_loss": (row_loss, 8)}, keep_mask, drop_seed)
    rc = _lib.load().dg_diffusion_loss_fwd_batched(arr, len(spans), n_steps, _ptr(sqrt_ac), _ptr(sqrt_1mac), loss_w.data_ptr(),
                                                   DG_D, _lib.stream_ptr(x_start.device))
    _lib.check(rc, "dg_diffusion_loss_fwd_batched")
    return row_loss, out


WG_ROWS_PER_PARTIAL = 128    # table rows behind one partial result of the fused weight gradients (a workgroup's rows)


def wgrad_operand_tables(params: Sequence["DenoiserParams"], n_steps: int) -> None:
    """The per-timestep operand table [n_steps, 16] = emb_layer(te_t) | te_t of up to two denoisers, in one launch;
    kept on the DenoiserParams (wg_tab) for diffusion_loss_bwd_spans(fused_wgrad=True).  Depends on emb_layer only."""
    dev = params[0].device
    tabs = [torch.empty((n_steps, 16), dtype=torch.float32, device=dev) for _ in params]
    dns = (C.POINTER(_lib.Denoiser) * len(params))(*[C.pointer(p.c) for p in params])
    ptrs = (C.c_void_p * len(params))(*[t.data_ptr() for t in tabs])
    rc = _lib.load().dg_denoiser_wgrad_tab(dns, ptrs, len(params), n_steps, _lib.stream_ptr(dev))
    _lib.check(rc, "dg_denoiser_wgrad_tab")
    for p, t in zip(params, tabs):
        p.wg_tab = t


def diffusion_loss_bwd_spans(spans: Sequence[LossSpan], target, x_start, noise, ts, sqrt_ac, sqrt_1mac, loss_w, g_row_loss,
                             g_out, d_target_add=None, keep_mask=None, drop_seed=None, fused_wgrad: bool = False,
                             d_x_start_out=None, d_target_out=None):
    """Backward of diffusion_loss_fwd_spans in ONE launch.  d_target_add ([N,64] or None) is added into d_target.
    Spans with a row list write only those rows of d_x_start / d_target: d_x_start_out / d_target_out are the (zeroed)
    tables to write into -- d_target_out may be d_target_add itself -- else zeroed tables are made here.
    Returns (d_x_start, d_target, dout_tot, hid, dpre, xn), all [N,64] -- or, with fused_wgrad=True (every span's
    denoiser then has its wg_tab, see wgrad_operand_tables), (d_x_start, d_target, partials): the kernel forms the
    weight gradients' partial sums itself, partials[k] = [ceil(span k rows / 128), partial floats] for
    denoiser_wgrad_reduce, and the four operand tables never go through memory."""
    n = x_start.shape[0]
    n_steps = loss_w.numel()
    _vec(g_row_loss, "grad of row_loss", torch.float64, n)
    for nm, t in (("grad of out", g_out), ("d_target_add", d_target_add)):
        if t is not None:
            _table(t, nm, n)
    names = ("d_x_start", "d_target") if fused_wgrad else ("d_x_start", "d_target", "dout_tot", "hid", "dpre", "xn")
    listed = any(sp.rows is not None for sp in spans)
    if listed and not fused_wgrad:
        raise DiffGraphHipError("row lists go with fused_wgrad (the operand tables of the other form cover every row)")
    outs = [torch.empty_like(x_start) for _ in names]
    if listed:
        for k, given in enumerate((d_x_start_out, d_target_out)):
            if given is not None:
                _table(given, names[k] + " table", n)
            outs[k] = torch.zeros_like(x_start) if given is None else given
    rb = DG_D * 4
    tables = {"target": (target, rb), "x_start": (x_start, rb), "noise": (noise if sqrt_ac is not None else None, rb),
              "ts": (ts, 8), "g_row_loss": (g_row_loss, 8), "g_out": (g_out, rb), "d_target_add": (d_target_add, rb)}
    tables.update({nm: (o, rb) for nm, o in zip(names, outs)})
    _check_dropout(spans, n, keep_mask, drop_seed)
    arr = _span_tasks(spans, n_steps, tables, keep_mask, drop_seed)
    partials = []
    if fused_wgrad:
        pf = int(_lib.load().dg_wgrad_partial_floats())
        for k, sp in enumerate(spans):
            tab = getattr(sp.dn, "wg_tab", None)
            if tab is None or tuple(tab.shape) != (n_steps, 16):
                raise DiffGraphHipError("fused_wgrad: the span's denoiser has no operand table (wgrad_operand_tables)")
            rows_here = sp.n if sp.rows is None else min(sp.rows.numel(), sp.n)
            part = torch.empty((-(-rows_here // WG_ROWS_PER_PARTIAL), pf), dtype=torch.float32, device=x_start.device)
            arr[k].wg_tab, arr[k].wg_partials = tab.data_ptr(), part.data_ptr()
            partials.append(part)
    rc = _lib.load().dg_diffusion_loss_bwd_batched(arr, len(spans), n_steps, _ptr(sqrt_ac), _ptr(sqrt_1mac), loss_w.data_ptr(),
                                                   DG_D, _lib.stream_ptr(x_start.device))
    _lib.check(rc, "dg_diffusion_loss_bwd_batched")
    if fused_wgrad:
        return outs[0], outs[1], partials
    return tuple(outs)


def denoiser_wgrad_reduce(jobs, launch: bool = True):
    """The denoisers' weight gradients from the partial sums of diffusion_loss_bwd_spans(fused_wgrad=True) -- jobs:
    (dn, partials) for up to two denoisers, one set of launches.  Returns one 6-tuple per job (d_w_emb [8,8], d_b_emb
    [8], d_w_in [64,72], d_b_in [64], d_w_out [64,64], d_b_out [64]); with launch=False (results, go): the tensors
    are allocated and go() launches the kernels that fill them, on the stream current at that time."""
    lib = _lib.load()
    arr = (_lib.WgradReduceTask * len(jobs))()
    results, keep = [], []
    for k, (dn, part) in enumerate(jobs):
        dev = part.device
        flat = torch.empty(512 + 64 + 8 + DG_D * 72 + DG_D + DG_D * DG_D + DG_D, dtype=torch.float32, device=dev)
        parts, o = [], 512                                   # the first 512 floats: scratch of the launches
        for shape in ((8, 8), (8,), (DG_D, 72), (DG_D,), (DG_D, DG_D), (DG_D,)):
            m = math.prod(shape)
            parts.append(flat[o: o + m].view(shape))
            o += m
        arr[k] = _lib.WgradReduceTask(C.pointer(dn.c), part.data_ptr(), part.shape[0], flat.data_ptr(), *[p.data_ptr() for p in parts])
        keep.append((flat, part))
        results.append(tuple(parts))

    def go(_keep=keep):
        rc = lib.dg_denoiser_wgrad_reduce(arr, len(jobs), _lib.stream_ptr(jobs[0][1].device))
        _lib.check(rc, "dg_denoiser_wgrad_reduce")

    if not launch:
        return results, go
    go()
    return results


def denoiser_wgrad(dn: DenoiserParams, n_steps: int, ts, dout, hid, dpre, xn):
    """Weight gradients of the denoiser from the tables of diffusion_loss_bwd.
    Returns (d_w_emb [8,8], d_b_emb [8], d_w_in [64,72], d_b_in [64], d_w_out [64,64], d_b_out [64])."""
    return denoiser_wgrad_many([(dn, ts, dout, hid, dpre, xn)], n_steps)[0]


def denoiser_wgrad_many(jobs, n_steps: int, launch: bool = True):
    """denoiser_wgrad for up to two denoisers -- jobs: (dn, ts, dout, hid, dpre, xn) each -- in one set of launches.
    Returns one 6-tuple of gradients per job; with launch=False (results, go): the gradient tensors are allocated
    and go() launches the kernels that fill them, on the stream current at that time."""
    lib = _lib.load()
    ws_bytes = int(lib.dg_denoiser_wgrad_workspace(n_steps))
    arr = (_lib.WgradTask * len(jobs))()
    results, keep = [], []
    for k, (dn, ts, dout, hid, dpre, xn) in enumerate(jobs):
        n = dout.shape[0]
        for nm, t in (("dout", dout), ("hid", hid), ("dpre", dpre), ("xn", xn)):
            _table(t, nm, n)
        _vec(ts, "timesteps", torch.int64, n)
        dev = dout.device
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
        flat = torch.empty(64 + 8 + DG_D * 72 + DG_D + DG_D * DG_D + DG_D, dtype=torch.float32, device=dev)
        parts, o = [], 0
        for shape in ((8, 8), (8,), (DG_D, 72), (DG_D,), (DG_D, DG_D), (DG_D,)):
            m = math.prod(shape)
            parts.append(flat[o: o + m].view(shape))
            o += m
        arr[k] = _lib.WgradTask(C.pointer(dn.c), ts.data_ptr(), dout.data_ptr(), hid.data_ptr(), dpre.data_ptr(), xn.data_ptr(),
                                n, ws.data_ptr(), ws_bytes, *[p.data_ptr() for p in parts])
        keep.append(ws)
        results.append(tuple(parts))
    def go(_keep=(keep, jobs)):
        rc = lib.dg_denoiser_wgrad_batched(arr, len(jobs), n_steps, DG_D, _lib.stream_ptr(jobs[0][2].device))
        _lib.check(rc, "dg_denoiser_wgrad_batched")

    if not launch:
        return results, go
    go()
    return results


# ------------------------------------------------------------------------------ loss head (a10)
def _index_vec(v: torch.Tensor, name: str, n: int) -> None:
    if v.dtype != torch.int64 or not v.is_cuda or not v.is_contiguous() or v.numel() != n:
        raise DiffGraphHipError(f"{name}: expected a contiguous int64 GPU vector of {n} elements")


def rec_head_fwd(ui, out_u, out_i, row_loss_u, row_loss_i, ancs, poss, negs, n_user: int, batch_div: float, reg: float,
                 reduce_later: bool = False):
    """Loss head of HGDM.cal_loss (Model.py:101-111).  Returns (results fp64[4] = loss, bpr, reg, diff; entries_f).
    out_u / out_i and row_loss_u / row_loss_i may be the two row spans of one table.  reduce_later=True returns
    (results, entries_f, go): the last launch -- the sums over the batch entries, which nothing of the backward pass
    reads -- is left to go(), on the stream current at that time; results are valid after it."""
    n_item = ui.shape[0] - n_user
    B = ancs.numel()
    _table(ui, "ui")
    _table(out_u, "out_u", n_user)
    _table(out_i, "out_i", n_item)
    _vec(row_loss_u, "row_loss_u", torch.float64, n_user)
    _vec(row_loss_i, "row_loss_i", torch.float64, n_item)
    for v, nm in ((ancs, "ancs"), (poss, "poss"), (negs, "negs")):
        _index_vec(v, nm, B)
    dev = ui.device
    ent_d = torch.empty(2 * B + 4, dtype=torch.float64, device=dev)   # per-entry diffusion losses, then the results
    ent_f = torch.empty(8 * B, dtype=torch.float32, device=dev)
    results = ent_d[2 * B:]
    lib = _lib.load()
    rc = lib.dg_rec_head_fwd(ui.data_ptr(), out_u.data_ptr(), out_i.data_ptr(), row_loss_u.data_ptr(),
                             row_loss_i.data_ptr(), ancs.data_ptr(), poss.data_ptr(), negs.data_ptr(), B, n_user,
                             n_item, float(batch_div), float(reg), DG_D, ent_f.data_ptr(), ent_d.data_ptr(),
                             None if reduce_later else results.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_rec_head_fwd")
    if reduce_later:
        def go(_keep=(ent_f, ent_d)):
            rc = lib.dg_rec_head_reduce(ent_f.data_ptr(), ent_d.data_ptr(), B, float(batch_div), float(reg), results.data_ptr(),
                                        _lib.stream_ptr(dev))
            _lib.check(rc, "dg_rec_head_reduce")
        return results, ent_f, go
    return results, ent_f


def rec_head_bwd_buffer(n: int, device, clear: bool = True) -> torch.Tensor:
    """The buffer rec_head_bwd writes its five outputs and the per-row tags into (every part 16-byte aligned);
    all but the last 24 n bytes (entry slots) have to be zero when it is handed over -- clear=False leaves that to
    the caller (clear_rec_head_bwd_buffer, e.g. on another stream ahead of time)."""
    buf = torch.empty(2 * n * DG_D * 4 + 40 * n, dtype=torch.uint8, device=device)
    if clear:
        clear_rec_head_bwd_buffer(buf, n)
    return buf


def clear_rec_head_bwd_buffer(buf: torch.Tensor, n: int) -> None:
    rc = _lib.load().dg_clear(buf.data_ptr(), 2 * n * DG_D * 4 + 16 * n, _lib.stream_ptr(buf.device))
    _lib.check(rc, "dg_clear")


def rec_head_bwd(ui, out_u, out_i, ancs, poss, negs, n_user: int, batch_div: float, reg: float, ent_f, g, full: bool = False,
                 buffer: Optional[torch.Tensor] = None):
    """Backward of rec_head_fwd; g: fp64[4] upstream gradients.  Returns dense (d_ui, d_out_u, d_out_i,
    d_row_loss_u, d_row_loss_i), zero outside the batch rows; with full=True (d_ui, d_out [N,64], d_row_loss [N])
    -- the user and item parts as one table each.  buffer: a cleared rec_head_bwd_buffer(n) to use."""
    n, n_item = ui.shape[0], ui.shape[0] - n_user
    B = ancs.numel()
    _vec(g, "upstream gradients", torch.float64, 4)
    dev = ui.device
    tb = n * DG_D * 4
    zero = rec_head_bwd_buffer(n, dev) if buffer is None else buffer
    if zero.numel() != 2 * tb + 40 * n or zero.dtype != torch.uint8 or zero.device != dev:
        raise DiffGraphHipError("rec_head_bwd: buffer is not a rec_head_bwd_buffer of this table size")
    d_ui = zero[:tb].view(torch.float32).view(n, DG_D)
    d_out = zero[tb: 2 * tb].view(torch.float32).view(n, DG_D)
    d_rl = zero[2 * tb: 2 * tb + 8 * n].view(torch.float64)
    tags = zero[2 * tb + 8 * n:].view(torch.int32)   # 2 n zeroed words, then 6 n slot words
    d_out_u, d_out_i = d_out[:n_user], d_out[n_user:]
    d_rl_u, d_rl_i = d_rl[:n_user], d_rl[n_user:]
    scratch = torch.empty(3 * B * (DG_D + 1) + B + 4, dtype=torch.float32, device=dev)
    rc = _lib.load().dg_rec_head_bwd(ui.data_ptr(), out_u.data_ptr(), out_i.data_ptr(), ancs.data_ptr(), poss.data_ptr(),
                                     negs.data_ptr(), B, n_user, n_item, float(batch_div), float(reg), DG_D, ent_f.data_ptr(),
                                     g.data_ptr(), scratch.data_ptr(), tags.data_ptr(), d_ui.data_ptr(), d_out_u.data_ptr(),
                                     d_out_i.data_ptr(), d_rl_u.data_ptr(), d_rl_i.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_rec_head_bwd")
    if full:
        return d_ui, d_out, d_rl
    return d_ui, d_out_u, d_out_i, d_rl_u, d_rl_i


# ------------------------------------------------------------------------------ NC: the two dense layers
def _mat(x: torch.Tensor, name: str, rows: Optional[int] = None, cols: Optional[int] = None) -> torch.Tensor:
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or not x.is_contiguous() \
            or (rows is not None and x.shape[0] != rows) or (cols is not None and x.shape[1] != cols):
        raise DiffGraphHipError(f"{name}: expected a contiguous float32 GPU matrix [{rows or 'n'},{cols or 'k'}], got "
                                f"{tuple(x.shape)} {x.dtype} on {x.device}")
    return x


def linear_fwd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """x [n, k] . w [m, k]^T + b [m]: nn.Linear's forward (DiffGraph_NC/Model.py:47, transform_layer)."""
    _mat(x, "x")
    n, k = x.shape
    _mat(w, "weight", None, k)
    m = w.shape[0]
    if b is not None:
        _vec(b, "bias", torch.float32, m)
    y = torch.empty((n, m), dtype=torch.float32, device=x.device)
    rc = _lib.load().dg_linear_fwd(x.data_ptr(), w.data_ptr(), _ptr(b), n, k, m, y.data_ptr(), _lib.stream_ptr(x.device))
    _lib.check(rc, "dg_linear_fwd")
    return y


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor):
    """(d weight [m, k], d bias [m]) of y = x . w^T + b from dy [n, m] and x [n, k]; fixed summation order."""
    _mat(dy, "dy")
    n, m = dy.shape
    _mat(x, "x", n)
    k = x.shape[1]
    lib = _lib.load()
    ws_bytes = int(lib.dg_linear_wgrad_workspace(k, m))
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    d_w = torch.empty((m, k), dtype=torch.float32, device=x.device)
    d_b = torch.empty(m, dtype=torch.float32, device=x.device)
    rc = lib.dg_linear_wgrad(dy.data_ptr(), x.data_ptr(), n, k, m, ws.data_ptr(), ws_bytes, d_w.data_ptr(), d_b.data_ptr(),
                             _lib.stream_ptr(x.device))
    _lib.check(rc, "dg_linear_wgrad")
    return d_w, d_b


def nc_logits(e: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """self.dense(all_embeddings) (DiffGraph_NC/Model.py:84,107): e [n, 64] . w [n_cls, 64]^T + b."""
    _table(e, "embeddings")
    _mat(w, "dense weight", None, DG_D)
    n_cls = w.shape[0]
    _vec(b, "dense bias", torch.float32, n_cls)
    out = torch.empty((e.shape[0], n_cls), dtype=torch.float32, device=e.device)
    rc = _lib.load().dg_nc_logits(e.data_ptr(), w.data_ptr(), b.data_ptr(), e.shape[0], DG_D, n_cls, out.data_ptr(),
                                  _lib.stream_ptr(e.device))
    _lib.check(rc, "dg_nc_logits")
    return out


def nc_nll_fwd(logits: torch.Tensor, ancs: torch.Tensor, labels: torch.Tensor):
    """F.nll_loss(F.log_softmax(logits, 1)[ancs], labels) (Model.py:84-89).  Returns (loss fp32[1], entries)."""
    _mat(logits, "logits")
    n, n_cls = logits.shape
    B = ancs.numel()
    _index_vec(ancs, "ancs", B)
    _index_vec(labels, "labels", B)
    entries = torch.empty((B, n_cls), dtype=torch.float32, device=logits.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    rc = _lib.load().dg_nc_nll_fwd(logits.data_ptr(), ancs.data_ptr(), labels.data_ptr(), B, n, n_cls, entries.data_ptr(),
                                   loss.data_ptr(), _lib.stream_ptr(logits.device))
    _lib.check(rc, "dg_nc_nll_fwd")
    return loss, entries


def nc_nll_bwd(e: torch.Tensor, w: torch.Tensor, ancs, labels, entries, g: torch.Tensor):
    """Backward of nc_logits + nc_nll_fwd wrt (e, w, b); g: fp32[1] upstream gradient of the loss."""
    _table(e, "embeddings")
    n_cls = w.shape[0]
    _vec(g, "upstream gradient", torch.float32, 1)
    d_e = torch.zeros_like(e)
    d_w = torch.empty_like(w)
    d_b = torch.empty(n_cls, dtype=torch.float32, device=e.device)
    rc = _lib.load().dg_nc_nll_bwd(e.data_ptr(), w.data_ptr(), ancs.data_ptr(), labels.data_ptr(), entries.data_ptr(),
                                   g.data_ptr(), ancs.numel(), e.shape[0], DG_D, n_cls, d_e.data_ptr(), d_w.data_ptr(),
                                   d_b.data_ptr(), _lib.stream_ptr(e.device))
    _lib.check(rc, "dg_nc_nll_bwd")
    return d_e, d_w, d_b


def nc_head_fwd(target: torch.Tensor, diff: torch.Tensor, row_loss: torch.Tensor, w: torch.Tensor, b: torch.Tensor,
                ancs: torch.Tensor, label: torch.Tensor):
    """The loss head of DiffGraph_NC's cal_loss over the batch rows (Model.py:79-89): nll_loss(log_softmax(dense(target +
    diff))[ancs], labels) and mean(row_loss[ancs]).  `label`: the one-hot table [n, n_cls] (float32; argmax inside) or
    the batch's labels (int64 [B]).  Returns (nll fp32[1], diff_loss fp64[1], entries, labels int32[B])."""
    _table(target, "target embeddings")
    n = target.shape[0]
    _table(diff, "denoised embeddings", n)
    _vec(row_loss, "row losses", torch.float64, n)
    _mat(w, "dense weight", None, DG_D)
    n_cls = w.shape[0]
    _vec(b, "dense bias", torch.float32, n_cls)
    B = ancs.numel()
    _index_vec(ancs, "ancs", B)
    onehot = labels = None
    if label.dtype == torch.int64:
        _index_vec(label, "labels", B)
        labels = label
    else:
        _mat(label, "one-hot labels", n, n_cls)
        onehot = label
    dev = target.device
    entries = torch.empty((B, n_cls), dtype=torch.float32, device=dev)
    label_out = torch.empty(B, dtype=torch.int32, device=dev)
    nll = torch.empty(1, dtype=torch.float32, device=dev)
    diff_loss = torch.empty(1, dtype=torch.float64, device=dev)
    rc = _lib.load().dg_nc_head_fwd(target.data_ptr(), diff.data_ptr(), row_loss.data_ptr(), w.data_ptr(), b.data_ptr(),
                                    ancs.data_ptr(), _ptr(onehot), _ptr(labels), B, n, DG_D, n_cls, entries.data_ptr(),
                                    label_out.data_ptr(), nll.data_ptr(), diff_loss.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_nc_head_fwd")
    return nll, diff_loss, entries, label_out


def nc_head_bwd(target, diff, w, ancs, label_out, entries, g_nll: torch.Tensor, g_diff: torch.Tensor):
    """Backward of nc_head_fwd.  Returns (d_e -- the gradient of target and of diff alike --, d_row_loss fp64[n], d_w, d_b)."""
    n, n_cls, dev = target.shape[0], w.shape[0], target.device
    _vec(g_nll, "upstream gradient of nll", torch.float32, 1)
    _vec(g_diff, "upstream gradient of diff_loss", torch.float64, 1)
    # one cleared buffer for both tables that arrive zeroed: [n, 64] fp32, then [n] fp64
    buf = torch.zeros(n * DG_D + 2 * n, dtype=torch.float32, device=dev)
    d_e = buf[: n * DG_D].view(n, DG_D)
    d_row_loss = buf[n * DG_D:].view(torch.float64)
    d_w = torch.empty_like(w)
    d_b = torch.empty(n_cls, dtype=torch.float32, device=dev)
    rc = _lib.load().dg_nc_head_bwd(target.data_ptr(), diff.data_ptr(), w.data_ptr(), ancs.data_ptr(), label_out.data_ptr(),
                                    entries.data_ptr(), g_nll.data_ptr(), g_diff.data_ptr(), ancs.numel(), n, DG_D, n_cls,
                                    d_e.data_ptr(), d_row_loss.data_ptr(), d_w.data_ptr(), d_b.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_nc_head_bwd")
    return d_e, d_row_loss, d_w, d_b
