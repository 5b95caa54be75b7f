"""HGDM for recommendation with the reference's interface (DiffGraph-Rec/Model.py:16-118).

forward() -> (target_user, target_item, heter_user, heter_item)
cal_loss(ancs, poss, negs) -> (loss, bprLoss, regLoss, diff_loss)
predict() -> (usrEmbeds, itmEmbeds)

`data_handler` needs userNum, itemNum, target_adj (CSRGraph) and behavior_mats (sequence or
dict of CSRGraph), as the reference's DataHandler provides with DGL graphs.

The propagation runs as one fused launch per layer over all (1 + B) graphs: layer l of graph
g reads the previous layer's activation already scaled by the source-node norm, writes the
next one, and adds the L2-normalised activation into that graph's running sum.
"""
from __future__ import annotations

import os
from contextlib import contextmanager
from dataclasses import replace
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import ops
from .graph import CSRGraph, GraphSet
from ._lib import (DG_ACC_REBUILD, DG_ACC_SHARED, DG_BWD_SAVED_PRESCALED, DG_D, DG_IN_PRESCALED, DG_OUT_PRESCALE,
                   DG_ROWS_FULL_ONLY, DG_ROWS_PARTIAL_ONLY)
from .diffusion import Denoise, GaussianDiffusion, dropout_args, dropout_rate
from .layers import DGLLayer

DEFAULT_ARGS = SimpleNamespace(latdim=64, gcn_layer=2, dims="[64]", d_emb_size=8, norm=True, steps=200,
                               noise_scale=1e-4, noise_min=1e-4, noise_max=1e-3, sampling_steps=0, batch=2048,
                               reg=3e-2, lr=1e-3, tstBat=1024, topk=20, epoch=100, patience=20)  # DiffGraph-Rec/params.py:5-39


def pairPredict(ancEmbeds, posEmbeds, negEmbeds):  # DiffGraph-Rec/Utils/Utils.py:4-8
    return (ancEmbeds * posEmbeds).sum(-1) - (ancEmbeds * negEmbeds).sum(-1)


APPLIED = "applied"     # HGDM.embedding_grad_parts after a backward pass that stepped the table itself


def _same_objects(a, b) -> bool:
    return a is not None and len(a) == len(b) and all(x is y for x, y in zip(a, b))


class _SavedStack(list):
    """Per-graph [E_1, E_2] views of a forward pass over a graph set, with the set and the stacked tensors."""
    gset = None
    stacked = None


class _PropagateFn(torch.autograd.Function):
    """HGDM.propagate as one differentiable op.  The forward keeps, per graph, every layer's
    activation (scaled by the source norm, as the next layer consumed it; the last layer's
    unscaled); the backward walks the layers down with the transposed propagation kernel."""

    @staticmethod
    def forward(ctx, e0, model):
        e0 = e0.contiguous()
        ui, het, saved = model._propagate(e0, keep=True)
        ctx.model, ctx.saved = model, saved
        # (a forward over the graph set keeps the graphs' activations stacked: the backward then walks the set too)
        ctx.gset, ctx.stacked = getattr(saved, "gset", None), getattr(saved, "stacked", None)
        return ui, het

    @staticmethod
    def backward(ctx, d_ui, d_het):
        model, saved = ctx.model, ctx.saved
        # (cal_loss_vector, whose loss reaches these layer sums at the batch's rows only, has said which on this very
        # node: elsewhere the incoming gradients, and with them dZ_2, are zero)
        flags = getattr(ctx, "row_flags", None)
        graphs = [model.target_adj] + model.behavior_mats
        slope, L = float(model.act.negative_slope), model.n_layers
        ref = saved[0][0]
        d_ui = torch.zeros_like(ref) if d_ui is None else d_ui.contiguous()
        d_het = torch.zeros_like(ref) if d_het is None else d_het.contiguous()
        d_sum = [d_ui] + [d_het] * (len(graphs) - 1)
        w = [None] * len(graphs)
        stack = None
        if ctx.gset is not None:
            # both steps over the set: d_ui / d_het are read once per row, not once per graph.  Each step writes its
            # result over the activations it has just consumed (a row is read, then written, by the same 16 lanes): the
            # step's footprint is 128 MB smaller on the Tmall shape, which leaves more of it in the memory-side cache
            if ctx.stacked is None:
                raise RuntimeError("HGDM.propagate: its saved activations were consumed by an earlier backward pass "
                                   "(the graph-set path overwrites them; run the forward again)")
            e1, e2 = ctx.stacked
            ctx.stacked = None
            ops.set_layer_bwd_top(ctx.gset, e2, d_ui, d_het, e2, slope, row_flags=flags)
            ops.set_layer_bwd_mid(ctx.gset, e2, e1, d_ui, d_het, e1, slope)
            stack = e1
            w = [stack[g] for g in range(len(graphs))]
        for l in range(L if ctx.gset is None else 0, 0, -1):
            tasks = []
            # the tables of the step before the last lie one after the other: the merged last step reads them as one
            stack = torch.empty((len(graphs),) + tuple(ref.shape), dtype=ref.dtype, device=ref.device) if l == 1 else None
            for g, graph in enumerate(graphs):
                out = stack[g] if l == 1 else torch.empty_like(ref)
                tasks.append(ops.BwdTask(graph, w[g], saved[g][l - 1], d_sum[g], out,
                                         DG_BWD_SAVED_PRESCALED if l < L else 0, slope))
                w[g] = out
            ops.gcn_layer_bwd_batched(tasks)
        merged = None if ctx.gset is not None else model.merged_last_step()
        step_now = getattr(model, "table_step_in_backward", None)
        if ctx.gset is not None and step_now is not None:
            # ... and the optimizer's step of the table inside that launch (a caller that runs backward() and step()
            # back to back -- diffgraph_amd.train.training_step -- asks for it): the table's gradient is never written
            step_now.step_table_in(lambda *adam: ops.set_layer_bwd_last_adam(ctx.gset, stack, d_ui, d_het, *adam))
            model.embedding_grad_parts = APPLIED
            return None, None
        if ctx.gset is not None:
            # the last step over the set as well: the lists BWD_MID has just walked, one output table
            g0 = torch.empty_like(ref)
            ops.set_layer_bwd_last(ctx.gset, stack, g0)
            w = [g0]
        elif merged is not None:
            # G0 = sum_g A_g^T dZ_g as ONE table from one launch (no per-graph tables to write and add up later)
            g0 = torch.empty_like(ref)
            ops.gcn_layer_batched([ops.LayerTask(merged, stack.view(-1, ref.shape[1]), g0, None, None, 0, 1.0)])
            w = [g0]
        else:
            tasks = []
            for g, graph in enumerate(graphs):
                out = torch.empty_like(ref)
                tasks.append(ops.BwdTask(graph, w[g], None, None, out, 0, slope))
                w[g] = out
            ops.gcn_layer_bwd_batched(tasks)
        # every graph's layer sum starts from e0 (Model.py:57,74): d e0 = sum_g (d_sum_g + G0_g)
        if getattr(model, "defer_embedding_grad", False):
            model.embedding_grad_parts = d_sum + w   # diffgraph_amd.optim.Adam adds them up inside its step
            return None, None
        return ops.sum_tables(d_sum + w), None


class _RecHeadFn(torch.autograd.Function):
    """The loss head of cal_loss (Model.py:101-111) as one differentiable op over the whole tables: residual
    add, the three batch gathers, pairPredict, BPR, the regulariser and the batch means of the two diffusion
    losses.  Returns fp64[4] = (loss, bprLoss, regLoss, diff_loss)."""

    @staticmethod
    def forward(ctx, ui, out_u, out_i, row_loss_u, row_loss_i, ancs, poss, negs, n_user, batch_div, reg):
        ui, out_u, out_i = ui.contiguous(), out_u.contiguous(), out_i.contiguous()
        results, ent_f = ops.rec_head_fwd(ui, out_u, out_i, row_loss_u.contiguous(), row_loss_i.contiguous(), ancs, poss,
                                          negs, n_user, batch_div, reg)
        ctx.save_for_backward(ui, out_u, out_i, ancs, poss, negs, ent_f)
        ctx.consts = (n_user, batch_div, reg)
        return results

    @staticmethod
    def backward(ctx, g):
        ui, out_u, out_i, ancs, poss, negs, ent_f = ctx.saved_tensors
        n_user, batch_div, reg = ctx.consts
        d_ui, d_out_u, d_out_i, d_rl_u, d_rl_i = ops.rec_head_bwd(ui, out_u, out_i, ancs, poss, negs, n_user, batch_div, reg,
                                                                  ent_f, g.contiguous())
        return d_ui, d_out_u, d_out_i, d_rl_u, d_rl_i, None, None, None, None, None, None


class _RecLossFn(torch.autograd.Function):
    """Everything of cal_loss after the propagation (Model.py:96-111) as ONE differentiable op over the two
    [n_user + n_item, 64] tables: both training_losses2 calls (one launch, user rows through usr_denoiser,
    item rows through item_denoiser) and the loss head.  The backward runs the head, one launch for both
    diffusion losses -- the head's gradient of the target embeddings is added inside it -- and the weight
    gradients of the two denoisers.  Returns fp64[4] = (loss, bprLoss, regLoss, diff_loss)."""

    @staticmethod
    def forward(ctx, ui, het, noise, ts, ancs, poss, negs, model, validate, drop, *weights):
        gd, args, nu = model.diffusion_process, model.args, model.n_user
        dev = ui.device
        ui, het = ui.contiguous(), het.contiguous()
        drop_p, keep_mask, drop_seed = drop
        spans = []
        # only the rows the batch names go through the denoisers (cal_loss_vector listed them beside the propagation):
        # diff_loss is read at [ancs] / [poss] and the outputs at [ancs], [poss], [negs] alone (Model.py:96-104)
        act, model._active = model._active, None
        for k, (dn, begin, n) in enumerate(((model.usr_denoiser, 0, nu), (model.item_denoiser, nu, model.n_item))):
            rows = {} if act is None else dict(rows=act[1 + k], n_rows=act[3][k: k + 1])
            spans.append(ops.LossSpan(dn.params(bool(dn.norm)), begin, n, drop_p, **rows))   # (time-bias tables: cal_loss_vector)
        sa = gd._dev("sqrt_alphas_cumprod", dev) if gd.noise_scale != 0 else None
        sb = gd._dev("sqrt_one_minus_alphas_cumprod", dev) if gd.noise_scale != 0 else None
        lw = gd._dev("loss_weight", dev, torch.float64)
        row_loss, out = ops.diffusion_loss_fwd_spans(spans, ui, het, noise, ts, sa, sb, lw, validate=validate,
                                                     keep_mask=keep_mask, drop_seed=drop_seed)
        # (a caller that joins the side stream before it reads the four values -- training_step -- gets them from there)
        if model.side_join_deferred:
            results, ent_f, go = ops.rec_head_fwd(ui, out[:nu], out[nu:], row_loss[:nu], row_loss[nu:], ancs, poss, negs, nu,
                                                  float(args.batch), float(args.reg), reduce_later=True)
            model._late_side.append((model.fork_point(), go))
        else:
            results, ent_f = ops.rec_head_fwd(ui, out[:nu], out[nu:], row_loss[:nu], row_loss[nu:], ancs, poss, negs, nu,
                                              float(args.batch), float(args.reg))
        ctx.save_for_backward(ui, het, noise if noise is not None else het, ts, out, ancs, poss, negs, ent_f)
        ctx.aux = (model, spans, sa, sb, lw, keep_mask, drop_seed, act)
        return results

    @staticmethod
    def backward(ctx, g):
        ui, het, noise, ts, out, ancs, poss, negs, ent_f = ctx.saved_tensors
        model, spans, sa, sb, lw, keep_mask, drop_seed, act = ctx.aux
        gd, args, nu = model.diffusion_process, model.args, model.n_user
        head_buf, model._head_buf = model._head_buf, None     # cleared beside the forward propagation (cal_loss_vector)
        dhet_buf, model._dhet_buf = model._dhet_buf, None     # (so was this one: d het of the rows outside the batch is zero)
        d_ui_head, d_out, d_rl = ops.rec_head_bwd(ui, out[:nu], out[nu:], ancs, poss, negs, nu, float(args.batch),
                                                  float(args.reg), ent_f, g.contiguous(), full=True, buffer=head_buf)
        # one launch for both diffusion losses; it also sums the weight gradients' operands over its 128-row strips
        listed = {} if act is None else dict(d_x_start_out=dhet_buf, d_target_out=d_ui_head)   # (d ui written over the head's)
        d_het, d_ui, partials = ops.diffusion_loss_bwd_spans(spans, ui, het, noise, ts, sa, sb, lw, d_rl, d_out,
                                                             d_target_add=d_ui_head, keep_mask=keep_mask,
                                                             drop_seed=drop_seed, fused_wgrad=True, **listed)
        jobs = [(sp.dn, part) for sp, part in zip(spans, partials)]
        if model.side_join_deferred:
            # a caller that promises run_late_side_work() before it touches the gradients (training_step): the strips'
            # sums are added up after the propagation's backward pass is launched and beside it, from here on
            res, go = ops.denoiser_wgrad_reduce(jobs, launch=False)
            model._late_side.append((model.fork_point(), go))
        else:
            res = ops.denoiser_wgrad_reduce(jobs)               # both denoisers, one set of launches
        grads = [g for r in res for g in r]
        return (d_ui, d_het, None, None, None, None, None, None, None, None) + tuple(grads)


class HGDM(nn.Module):
    def __init__(self, data_handler, args=None, device="cuda"):
        super().__init__()
        self.args = args = args or DEFAULT_ARGS
        if args.latdim < 1:
            raise ValueError(f"latdim must be positive (got {args.latdim})")
        # latdim > 64: the table has its true width and the model runs through diffgraph_amd.wide (column blocks of 64
        # through the propagation kernel, what couples them in torch)
        self.wide = args.latdim > DG_D
        self.n_user, self.n_item = data_handler.userNum, data_handler.itemNum
        bm = data_handler.behavior_mats
        self.behavior_mats = [bm[k] for k in sorted(bm)] if isinstance(bm, dict) else list(bm)
        self.target_adj = data_handler.target_adj
        self.n_hid, self.n_layers = args.latdim, args.gcn_layer
        # one [n_user+n_item, 64] table; user_emb / item_emb are views, so the concatenation
        # at Model.py:55-56 costs nothing.  With latdim < 64 the columns from latdim on are zero and stay zero: every
        # kernel on the path sums over a row's columns or treats them alike, the zero columns of the denoisers' outputs
        # and of the noise (cal_loss_vector) keep their gradient at exactly zero, and Adam leaves a zero with a zero
        # gradient where it is.
        table = torch.zeros(self.n_user + self.n_item, max(DG_D, self.n_hid), device=device)
        nn.init.xavier_uniform_(table[: self.n_user, : self.n_hid])
        nn.init.xavier_uniform_(table[self.n_user:, : self.n_hid])
        self.embedding_table = nn.Parameter(table)
        self.act = nn.LeakyReLU(0.5, inplace=True)
        self.layers = nn.ModuleList(
            [DGLLayer(self.n_hid, self.n_hid, weight=False, bias=False, activation=self.act) for _ in range(self.n_layers)])
        self.diffusion_process = GaussianDiffusion(args.noise_scale, args.noise_min, args.noise_max, args.steps)
        out_dims = eval(args.dims) + [args.latdim]
        in_dims = out_dims[::-1]
        self.usr_denoiser = Denoise(in_dims, out_dims, args.d_emb_size, norm=args.norm).to(device)
        self.item_denoiser = Denoise(in_dims, out_dims, args.d_emb_size, norm=args.norm).to(device)
        # {seed, draws made, dropout key, 0} of the timesteps / noise / dropout masks drawn on the device; part of the
        # state_dict, so that a resumed run goes on with the draws that follow instead of repeating the first ones
        self.register_buffer("draw_state", ops.new_draw_state(device), persistent=True)
        self._side: Optional[torch.cuda.Stream] = None
        self._side_pending: list = []
        self._late_side: list = []
        self._head_buf: Optional[torch.Tensor] = None
        self._dhet_buf: Optional[torch.Tensor] = None
        self._active = None
        self.loss_on_batch_rows = os.environ.get("DG_LOSS_ALL_ROWS") != "1"   # False: both denoisers over every row, as the reference runs them
        self.side_join_deferred = False
        self.side_overlap = True       # False: side_work() stays on the current stream (one kernel at a time, e.g. to time them)
        self.merge_last_step = True    # False: the backward's last propagation step runs graph by graph (see merged_last_step)
        self._merged, self._merged_key = None, None
        self.fuse_graphs = True        # False: one layer task per graph instead of the row walk over the graph set (graph_set)
        self._gset, self._gset_graphs = None, None

    def merged_last_step(self) -> Optional[CSRGraph]:
        """The last step of the backward pass is linear: d e0 gets sum_g A_g^T dZ_g, with no activation or
        normalisation between the graphs' terms.  Stacking the graphs' dZ tables and putting all their edges into one
        operator [n, G n] -- entry (r, g n + c) = the norm of r in graph g, for every edge of graph g; dZ_g carries its
        own side's norm already -- makes that one launch with one output table instead of G tables the optimizer adds
        up.  Built at first use; None for graphs the gather kernel does not run (one side held in LDS, bitmap graphs,
        directed graphs) and when switched off."""
        if not self.merge_last_step:
            return None
        graphs = [self.target_adj] + self.behavior_mats
        if self._merged is None or not _same_objects(self._merged_key, graphs):      # (graphs swapped since: build it anew)
            n = self.n_user + self.n_item
            self._merged, self._merged_key = False, list(graphs)   # the graphs themselves: an id() can be handed out again
            if all(g.is_symmetric and g.schedule()["lds_split"] == 0 and g.schedule()["dense_splits"] == 0 for g in graphs) \
                    and len(graphs) * n * DG_D * 4 < 2 ** 32:
                rows, cols, weights = [], [], []
                for k, g in enumerate(graphs):
                    rowptr, colidx = g.csr()
                    r = np.repeat(np.arange(n, dtype=np.int32), np.diff(rowptr))
                    rows.append(r)
                    cols.append(colidx + np.int32(k * n))
                    weights.append(g.norms()[0][r])
                self._merged = CSRGraph.weighted(np.concatenate(rows), np.concatenate(cols), np.concatenate(weights), n,
                                                 len(graphs) * n, device=self.embedding_table.device)
        return self._merged or None

    # -- work beside the main stream --------------------------------------------------------------------
    # The propagation kernels wait on memory most of the time; short launches that do not depend on them (the
    # step's random draws and time-bias tables in the forward pass, the denoisers' weight gradients and their Adam
    # update in the backward pass) run on a second stream meanwhile.  Recorded into a HIP graph the fork and the
    # join become branches of the graph.
    def _side_stream(self) -> torch.cuda.Stream:
        dev = self.embedding_table.device
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(dev)
        return self._side

    def fork_point(self) -> torch.cuda.Event:
        """Marks the current position of the current stream: side_work(after=...) starts behind it."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.embedding_table.device))
        return ev

    @contextmanager
    def side_work(self, keep=None, after: Optional[torch.cuda.Event] = None):
        """Runs the block on the side stream, after everything launched so far on the current one (or up to the
        fork_point `after`).  The current stream does not wait for it until join_side_work(); `keep` holds what the
        block reads alive until then, so that its memory is not handed out again meanwhile."""
        if not self.side_overlap:      # everything in launch order on the current stream
            yield
            return
        dev = self.embedding_table.device
        main, side = torch.cuda.current_stream(dev), self._side_stream()
        if after is None:
            side.wait_stream(main)
        else:
            side.wait_event(after)
        self._side_pending.append(keep)
        with torch.cuda.stream(side):
            yield

    def run_late_side_work(self, then=None) -> None:
        """Launches what a backward pass under side_join_deferred left for later (and `then`, e.g. the update of the
        parameters whose gradients it forms) on the side stream, and joins."""
        late, self._late_side = self._late_side, []
        if late:
            # one fork for all of them, behind the last one's fork point (every dependency between the two streams
            # costs the main chain a few microseconds)
            with self.side_work(after=late[-1][0]):
                for _, go in late:
                    go()
                if then is not None:
                    then()
        elif then is not None:
            then()
        self.join_side_work()

    def join_side_work(self) -> None:
        if self._side_pending:
            torch.cuda.current_stream(self.embedding_table.device).wait_stream(self._side_stream())
            self._side_pending = []

    @property
    def embedding_dict(self):
        """The reference's two tables, [n_user, latdim] and [n_item, latdim], as views of the one table."""
        L = self.n_hid
        return {"user_emb": self.embedding_table[: self.n_user, :L], "item_emb": self.embedding_table[self.n_user:, :L]}

    def set_embeddings(self, user_emb: torch.Tensor, item_emb: torch.Tensor) -> None:
        """Takes the reference's [rows, latdim] tables (or 64-wide ones whose columns from latdim on are zero)."""
        L = self.n_hid
        with torch.no_grad():
            self.embedding_table.zero_()
            self.embedding_table[: self.n_user, :L].copy_(user_emb[:, :L])
            self.embedding_table[self.n_user:, :L].copy_(item_emb[:, :L])

    def propagate(self, e0: torch.Tensor):
        """Returns (ui_embeddings [N,64], all_heter_embeddings [N,64])  (Model.py:61-87)."""
        if self.wide:
            from . import wide
            return wide.propagate([self.target_adj] + self.behavior_mats, e0, self.n_layers, float(self.act.negative_slope))
        if torch.is_grad_enabled() and e0.requires_grad:
            return _PropagateFn.apply(e0, self)
        ui, het, _ = self._propagate(e0.detach().contiguous(), keep=False)
        return ui, het

    def graph_set(self) -> Optional[GraphSet]:
        """The target graph and the behaviour graphs behind one row schedule (GraphSet), when the propagation can run
        over it: two layers, 2..4 symmetric graphs on the plain gather path.  Built at first use; None otherwise and
        when switched off (fuse_graphs = False: one task per graph, as for every other shape)."""
        if not self.fuse_graphs or self.n_layers != 2:
            return None
        graphs = [self.target_adj] + self.behavior_mats
        if self._gset is None or not _same_objects(self._gset_graphs, graphs):      # (graphs swapped since: build it anew)
            n = self.n_user + self.n_item
            self._gset, self._gset_graphs = False, list(graphs)
            if 2 <= len(graphs) <= 4 and len(graphs) * n * DG_D * 4 < 2 ** 32 and all(
                    g.num_nodes() == n and g.is_symmetric and g.schedule()["lds_split"] == 0
                    and g.schedule()["dense_splits"] == 0 for g in graphs):
                self._gset = GraphSet(graphs, device=self.embedding_table.device)
        return self._gset or None

    def _propagate_set(self, gset: GraphSet, e0: torch.Tensor, keep: bool):
        """Both layers over the graph set: two launches (+ one each for the rows summed in pieces)."""
        graphs = [self.target_adj] + self.behavior_mats
        slope = float(self.act.negative_slope)
        G, n = len(graphs), e0.shape[0]
        hook = getattr(self, "layer_hook", None)
        e1 = torch.empty((G, n, DG_D), dtype=e0.dtype, device=e0.device)
        e2 = torch.empty_like(e1) if keep else None
        ui, het = torch.empty_like(e0), torch.empty_like(e0)
        if hook is not None:    # what the launch does, per graph, in the terms of the task-per-graph path
            hook("begin", 0, [ops.LayerTask(g, e0, e1[k], None, None, DG_OUT_PRESCALE, slope, shared_rows=True)
                              for k, g in enumerate(graphs)])
        ops.set_layer_first(gset, e0, e1, slope)
        if hook is not None:
            hook("end", 0, None)
            hook("begin", 1, [ops.LayerTask(g, e1[k], None if e2 is None else e2[k], e0, ui if k == 0 else het,
                                            DG_IN_PRESCALED | DG_ACC_REBUILD, slope, shared_rows=True)
                              for k, g in enumerate(graphs)])
        ops.set_layer_last(gset, e1, e0, ui, het, e2, slope)
        if hook is not None:
            hook("end", 1, None)
        saved = _SavedStack([[e1[k], e2[k]] for k in range(G)] if keep else [])
        saved.gset, saved.stacked = gset, (e1, e2)
        return ui, het, saved

    def _propagate(self, e0: torch.Tensor, keep: bool):
        gset = self.graph_set()
        if gset is not None:
            return self._propagate_set(gset, e0, keep)
        graphs = [self.target_adj] + self.behavior_mats
        slope = float(self.act.negative_slope)
        L = self.n_layers
        sums = [torch.empty_like(e0) for _ in graphs]
        prev: List[Optional[torch.Tensor]] = [None] * len(graphs)
        saved = [[] for _ in graphs]
        for l in range(L):
            last = l == L - 1
            tasks = []
            for g, graph in enumerate(graphs):
                e_out = None if (last and not keep) else torch.empty_like(e0)
                flags = (DG_IN_PRESCALED if l > 0 else 0) | (0 if last else DG_OUT_PRESCALE)
                # with L >= 2 the first layer writes no running sum; the second rebuilds e0 + normalize(E_1)
                # from e0 and the stored first-layer activation (see DG_ACC_REBUILD)
                if l == 0 and L >= 2:
                    acc_in = acc_out = None
                elif l == 1:
                    acc_in, acc_out, flags = e0, sums[g], flags | DG_ACC_REBUILD
                else:
                    acc_in, acc_out = (e0 if l == 0 else sums[g]), sums[g]
                tasks.append(ops.LayerTask(graph, e0 if l == 0 else prev[g], e_out, acc_in, acc_out, flags, slope))
                prev[g] = e_out
                saved[g].append(e_out)
            hook = getattr(self, "layer_hook", None)
            if hook is not None:
                hook("begin", l, tasks)
            split = self._lds_split(graphs) if (last and l == 1 and not keep and len(graphs) >= 2) else 0
            if split:
                # Graphs with one small side held in LDS (the Industry shape): the small side's rows per graph, then
                # the large side's rows of ALL graphs in one walk that writes `ui` and `het` directly -- e0 read once
                # per row, no layer sum per behaviour graph for them (DG_ACC_SHARED)
                het = sums[1] if len(graphs) == 2 else torch.empty_like(e0)
                ops.gcn_layer_batched([replace(t, flags=t.flags | DG_ROWS_PARTIAL_ONLY) for t in tasks])
                ops.gcn_layer_batched([replace(t, flags=t.flags | DG_ROWS_FULL_ONLY | DG_ACC_SHARED,
                                               acc_out=sums[0] if g == 0 else het) for g, t in enumerate(tasks)])
                if len(graphs) > 2:
                    ops.prepare_sum_tables([t[split:] for t in sums[1:]], het[split:])()
            else:
                ops.gcn_layer_batched(tasks)
            if hook is not None:
                hook("end", l, tasks)
            if split:
                return sums[0], het, saved
        return sums[0], ops.sum_tables(sums[1:]), saved

    def _lds_split(self, graphs) -> int:
        """The first row of the side every graph holds in LDS, when all of them do and agree on it (else 0)."""
        kept = getattr(self, "_lds_split_of", None)
        if kept is None or not _same_objects(kept[0], graphs):
            splits = {int(g.schedule()["lds_split"]) for g in graphs}
            kept = self._lds_split_of = (list(graphs), splits.pop() if len(splits) == 1 and 0 not in splits else 0)
        return kept[1]

    def forward(self):
        """(target_user, target_item, heter_user, heter_item), each [rows, latdim] (Model.py:53-92)."""
        ui, het = self.propagate(self.embedding_table)
        nu, L = self.n_user, self.n_hid
        if L < DG_D:
            return ui[:nu, :L], ui[nu:, :L], het[:nu, :L], het[nu:, :L]
        return ui[:nu], ui[nu:], het[:nu], het[nu:]

    def cal_loss(self, ancs, poss, negs, ts_u=None, noise_u=None, ts_i=None, noise_i=None, keep_u=None, keep_i=None):  # Model.py:94-111
        res = self.cal_loss_vector(ancs, poss, negs, ts_u, noise_u, ts_i, noise_i, keep_u, keep_i)
        return res[0], res[1], res[2], res[3]

    def cal_loss_vector(self, ancs, poss, negs, ts_u=None, noise_u=None, ts_i=None, noise_i=None, keep_u=None,
                        keep_i=None, validate: Optional[bool] = None) -> torch.Tensor:
        """cal_loss as one float64[4] tensor (loss, bprLoss, regLoss, diff_loss).  `res.backward(e)` with
        e = [1, 0, 0, 0] is `loss.backward()` without the kernels autograd spends on slicing the scalar out.
        In training mode the denoisers' dropout (Model.py:223-224) runs inside the loss kernels, over the
        elements keep_u [n_user, 64] / keep_i [n_item, 64] keep (bool or uint8) or over a fresh draw."""
        dp = self.diffusion_process
        dev, n = self.embedding_table.device, self.n_user + self.n_item
        denoisers = [self.usr_denoiser, self.item_denoiser]
        if self.wide:
            from . import wide
            return wide.cal_loss(self, ancs, poss, negs, ts_u, noise_u, ts_i, noise_i, keep_u, keep_i)
        if not all(dn.fused for dn in denoisers):
            return self._cal_loss_layers(ancs, poss, negs, ts_u, noise_u, ts_i, noise_i, keep_u, keep_i)
        if validate is None:   # caller-supplied timesteps are range-checked (one host sync) unless the caller vouches for them
            validate = ts_u is not None or ts_i is not None
        keep = None
        if keep_u is not None or keep_i is not None:
            keep = torch.cat([keep_u.to(torch.uint8), keep_i.to(torch.uint8)]).contiguous()
        # what the call has to draw itself: timesteps and noise of both training_losses2 calls at once
        # (Model.py:352-353) and the key of the dropout masks
        need_ts = ts_u is None and ts_i is None
        need_noise = dp.noise_scale != 0 and noise_u is None and noise_i is None
        need_key = dropout_rate(denoisers) > 0 and keep is None
        ts = torch.empty(n, dtype=torch.int64, device=dev) if need_ts else torch.cat([ts_u, ts_i]).long().contiguous()
        noise = None
        if dp.noise_scale != 0:
            noise = torch.empty((n, DG_D), device=dev) if need_noise else torch.cat([noise_u, noise_i]).contiguous()
        self._head_buf = ops.rec_head_bwd_buffer(n, dev, clear=False) if torch.is_grad_enabled() else None
        ancs, poss, negs = (v.long().contiguous() for v in (ancs, poss, negs))
        listed = self.loss_on_batch_rows and 0 < ancs.numel() <= 8192 and ancs.numel() == poss.numel() == negs.numel()
        act = ops.active_rows_buffers(ancs.numel(), n, dev) if listed else None
        self._dhet_buf = torch.empty((n, DG_D), device=dev) if listed and torch.is_grad_enabled() else None
        start = self.fork_point()
        ui, het = self.propagate(self.embedding_table)   # [users | items] tables: target-graph sums, behaviour sums
        # beside the propagation (launched after it, started before it): the draws and both denoisers' time-bias tables
        with self.side_work(after=start):
            if need_ts or need_noise or need_key:
                ops.diffusion_draw(self.draw_state, n, dp.steps, need_ts, need_noise, ts if need_ts else None,
                                   noise if need_noise else None)
            params = [dn.params(bool(dn.norm)) for dn in denoisers]
            ops.refresh_time_bias(params, dp.steps)
            if act is not None:
                ops.active_rows(ancs, poss, negs, self.n_user, self.n_item, out=act)   # the rows the batch names
            if torch.is_grad_enabled():
                ops.wgrad_operand_tables(params, dp.steps)     # the backward pass's time-embedding operands
                ops.clear_rec_head_bwd_buffer(self._head_buf, n)   # ... and the loss head's cleared gradient tables
                if self._dhet_buf is not None:
                    ops.clear_table(self._dhet_buf)
        self.join_side_work()
        self._active = act
        if act is not None and ui.grad_fn is not None:
            # ui and het have no reader but the loss below, which touches them at the batch's rows only: the backward
            # pass of THIS propagation may take every other row's incoming gradient as zero
            ui.grad_fn.row_flags = act[0]
        drop = dropout_args(denoisers, keep, dev, self.draw_state[2:3])
        weights = []
        for dn in (self.usr_denoiser, self.item_denoiser):
            weights += [dn.emb_layer.weight, dn.emb_layer.bias, dn.in_layers[0].weight, dn.in_layers[0].bias,
                        dn.out_layers[0].weight, dn.out_layers[0].bias]
        return _RecLossFn.apply(ui, het, noise, ts, ancs, poss, negs, self, validate, drop, *weights)

    def _cal_loss_layers(self, ancs, poss, negs, ts_u, noise_u, ts_i, noise_i, keep_u, keep_i) -> torch.Tensor:
        """cal_loss (Model.py:94-111) for denoisers outside the fused kernels' shape (another --dims list or
        --d_emb_size): the propagation and the loss head as always, the two training_losses2 calls layer by layer."""
        dp, nu = self.diffusion_process, self.n_user
        ui, het = self.propagate(self.embedding_table)
        rl_u, out_u = dp.training_losses_rows(self.usr_denoiser, ui[:nu], het[:nu], ts_u, noise_u, keep_u)
        rl_i, out_i = dp.training_losses_rows(self.item_denoiser, ui[nu:], het[nu:], ts_i, noise_i, keep_i)
        ancs, poss, negs = (v.long().contiguous() for v in (ancs, poss, negs))
        return _RecHeadFn.apply(ui, out_u, out_i, rl_u, rl_i, ancs, poss, negs, nu, float(self.args.batch), float(self.args.reg))

    def predict(self):  # Model.py:112-118
        ui, het = self.propagate(self.embedding_table)
        nu, L = self.n_user, self.n_hid
        dp = self.diffusion_process
        denoised_u, denoised_i = dp.p_sample_many([(self.usr_denoiser, het[:nu]), (self.item_denoiser, het[nu:])],
                                                  self.args.sampling_steps)
        pu, pi = ui[:nu] + denoised_u, ui[nu:] + denoised_i
        return (pu, pi) if L >= DG_D else (pu[:, :L], pi[:, :L])
