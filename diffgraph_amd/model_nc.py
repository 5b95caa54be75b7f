"""HGDM for node classification with the reference's interface (DiffGraph_NC/Model.py:18-108).

forward(he_adjs, feature_list, is_training=True) -> (source1, source2, target)
cal_loss(ancs, label, he_adjs, initial_feature) -> (nll_loss, diff_loss)
get_allembeds(he_adjs, initial_feature) -> (all_embeddings, scores)

he_adjs is a list of three CSRGraph (author-author meta-path graphs).  Unlike the Rec
model, the NORMALISED activation of a layer feeds the next one (Model.py:53,60,67).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from ._lib import (DG_ACC_REBUILD, DG_BWD_NORM_FEED, DG_BWD_SAVED_PRESCALED, DG_D, DG_IN_PLANES, DG_IN_PRESCALED, DG_OUT_NORM,
                   DG_OUT_PLANES, DG_OUT_PRESCALE)
from .diffusion import Denoise, GaussianDiffusion
from .layers import UUGCNLayer

DEFAULT_ARGS = SimpleNamespace(latdim=64, gcn_layer=3, uugcn_layer=3, dims="[64]", d_emb_size=8, norm=True, steps=200,
                               noise_scale=1e-5, noise_min=1e-4, noise_max=1e-3, sampling_steps=0,
                               batch=8, lr=3e-3, epoch=10)  # DiffGraph_NC/params.py:5-68


class _LinearFn(torch.autograd.Function):
    """nn.Linear on the HIP path: y = x . w^T + b and the gradients autograd forms for it."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = x.contiguous(), w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return ops.linear_fwd(x, w, None if b is None else b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        d_w = d_b = d_x = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            d_w, d_b = ops.linear_wgrad(dy, x)
        if ctx.needs_input_grad[0]:
            d_x = ops.linear_fwd(dy, w.t().contiguous(), None)        # dy [n, m] . w [m, k]
        return d_x, d_w, (d_b if ctx.has_bias else None)


class _NLLHeadFn(torch.autograd.Function):
    """F.nll_loss(F.log_softmax(dense(e), dim=1)[ancs], labels) (Model.py:84-89) as one differentiable op."""

    @staticmethod
    def forward(ctx, e, w, b, ancs, labels):
        e, w, b = e.contiguous(), w.contiguous(), b.contiguous()
        logits = ops.nc_logits(e, w, b)
        loss, entries = ops.nc_nll_fwd(logits, ancs, labels)
        ctx.save_for_backward(e, w, ancs, labels, entries)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        e, w, ancs, labels, entries = ctx.saved_tensors
        d_e, d_w, d_b = ops.nc_nll_bwd(e, w, ancs, labels, entries, g.reshape(1).float().contiguous())
        return d_e, d_w, d_b, None, None


class _LossHeadFn(torch.autograd.Function):
    """Everything of cal_loss after training_losses2's rows (Model.py:80-89) as one differentiable op over the batch
    rows: all_embeddings = target + diffused, dense, log_softmax, nll_loss against argmax(label[ancs]), and the batch
    mean of the per-row diffusion losses.  Returns (nll fp32 scalar, diff_loss fp64 scalar)."""

    @staticmethod
    def forward(ctx, target, diffused, row_loss, w, b, ancs, label):
        target, diffused, row_loss = target.contiguous(), diffused.contiguous(), row_loss.contiguous()
        w, b = w.contiguous(), b.contiguous()
        nll, diff_loss, entries, labels = ops.nc_head_fwd(target, diffused, row_loss, w, b, ancs, label)
        ctx.save_for_backward(target, diffused, w, ancs, labels, entries)
        return nll[0], diff_loss[0]

    @staticmethod
    def backward(ctx, g_nll, g_diff):
        target, diffused, w, ancs, labels, entries = ctx.saved_tensors
        g_nll = torch.zeros(1, dtype=torch.float32, device=w.device) if g_nll is None else g_nll.reshape(1).float().contiguous()
        g_diff = torch.zeros(1, dtype=torch.float64, device=w.device) if g_diff is None else g_diff.reshape(1).double().contiguous()
        d_e, d_row_loss, d_w, d_b = ops.nc_head_bwd(target, diffused, w, ancs, labels, entries, g_nll, g_diff)
        return d_e, d_e, d_row_loss, d_w, d_b, None, None


def _plane_flags(graphs):
    """(out, in) hints for layer tasks whose output the same graph gathers next: bitmap graphs keep that table as
    bf16 planes, and the producing launch can leave them ready.  A graph handle holds ONE set of planes, so the hints
    are only given when every channel has a handle of its own."""
    return (DG_OUT_PLANES, DG_IN_PLANES) if len({id(g) for g in graphs}) == len(graphs) else (0, 0)


def _launch_distinct(tasks, launch) -> None:
    """Graphs that share a handle cannot share a launch (one hub slab per handle)."""
    seen, batch = set(), []
    for t in tasks:
        if id(t.graph) in seen:
            launch(batch)
            seen, batch = set(), []
        seen.add(id(t.graph))
        batch.append(t)
    launch(batch)


class _PropagateFn(torch.autograd.Function):
    """HGDM.propagate as one differentiable op of the layer-0 table (the output of
    transform_layer).  The forward keeps each layer's normalised activation and its length
    before normalisation; the backward walks the layers down with the transposed kernel:
        G_L = dS,   dZ_l = LeakyReLU'(y_l) * normalize_backward(G_l; y_l),   G_{l-1} = dS + A^T dZ_l."""

    @staticmethod
    def forward(ctx, embed, model, he_adjs):
        embed = embed.contiguous()
        sums, saved, norms = model._propagate(he_adjs, embed, keep=True)
        ctx.model, ctx.he_adjs, ctx.saved, ctx.norms = model, list(he_adjs), saved, norms
        return tuple(sums)

    @staticmethod
    def backward(ctx, *d_sums):
        model, graphs, saved, norms = ctx.model, ctx.he_adjs, ctx.saved, ctx.norms
        slope, Ls = float(model.act.negative_slope), model.layer_counts(len(graphs))
        ref = saved[0][0]
        d_sum = [torch.zeros_like(ref) if d is None else d.contiguous() for d in d_sums]
        w = [None] * len(graphs)
        planes_out, planes_in = _plane_flags(graphs)
        for l in range(max(Ls), 0, -1):
            tasks = []
            for g, graph in enumerate(graphs):
                L = Ls[g]
                if l > L:
                    continue
                out = torch.empty_like(ref)
                flags = DG_BWD_NORM_FEED | (DG_BWD_SAVED_PRESCALED if l < L else 0) | planes_out | (planes_in if l < L else 0)
                tasks.append(ops.BwdTask(graph, w[g], saved[g][l - 1], d_sum[g], out, flags, slope, norms[g][l - 1]))
                w[g] = out
            _launch_distinct(tasks, ops.gcn_layer_bwd_batched)
        tasks = []
        for g, graph in enumerate(graphs):
            out = torch.empty_like(ref)
            tasks.append(ops.BwdTask(graph, w[g], None, None, out, planes_in, slope))
            w[g] = out
        _launch_distinct(tasks, ops.gcn_layer_bwd_batched)
        # every channel's layer sum starts from the same table (Model.py:48-50)
        return ops.sum_tables(d_sum + w), None, None


class HGDM(nn.Module):
    def __init__(self, f_dim, args=None, device="cuda"):
        super().__init__()
        self.args = args = args or DEFAULT_ARGS
        if args.latdim < 1:
            raise ValueError(f"latdim must be positive (got {args.latdim})")
        # latdim > 64: through diffgraph_amd.wide (column blocks of 64 through the propagation kernel, the rest in torch)
        self.wide = args.latdim > DG_D
        # latdim < 64: the tables stay 64 wide with zeros from column latdim on (transform_layer's output and the
        # classification weights are padded, under autograd: the zero columns neither change a sum nor get a gradient)
        self.n_hid = args.latdim
        out_dims = eval(args.dims) + [args.latdim]
        in_dims = out_dims[::-1]
        self.user_denoise_model = Denoise(in_dims, out_dims, args.d_emb_size, norm=args.norm).to(device)
        # (a denoiser outside the fused kernels' shape -- another --dims list or --d_emb_size -- runs layer by layer:
        # GaussianDiffusion.training_losses_rows and p_sample take either kind)
        self.diffusion_model = GaussianDiffusion(args.noise_scale, args.noise_min, args.noise_max, args.steps)
        self.act = nn.LeakyReLU(0.5, inplace=True)
        mk = lambda: UUGCNLayer(args.latdim, args.latdim, weight=False, bias=False, activation=self.act)
        self.helayers1 = nn.ModuleList([mk() for _ in range(args.gcn_layer)])
        self.helayers2 = nn.ModuleList([mk() for _ in range(args.gcn_layer)])
        self.main_layers = nn.ModuleList([mk() for _ in range(args.uugcn_layer)])
        self.transform_layer = nn.Linear(f_dim, args.latdim, bias=True).to(device)
        nn.init.xavier_normal_(self.transform_layer.weight, gain=1.414)
        self.dense = nn.Linear(args.latdim, 4).to(device)

    def layer_counts(self, n_graphs: int) -> List[int]:
        """Layers per channel: he_adjs[0] goes through main_layers (--uugcn_layer), the others through helayers1 / 2
        (--gcn_layer) -- Model.py:31-36,52-71."""
        return [int(self.args.uugcn_layer)] + [int(self.args.gcn_layer)] * (n_graphs - 1)

    def propagate(self, he_adjs, embed: torch.Tensor):
        """(sum over layers for he_adjs[0], [1], [2]) -- Model.py:52-71."""
        if torch.is_grad_enabled() and embed.requires_grad:
            return list(_PropagateFn.apply(embed, self, he_adjs))
        return self._propagate(he_adjs, embed.detach().contiguous(), keep=False)[0]

    def _propagate(self, he_adjs, embed: torch.Tensor, keep: bool):
        slope = float(self.act.negative_slope)
        Ls = self.layer_counts(len(he_adjs))
        sums = [torch.empty_like(embed) for _ in he_adjs]
        prev: List[Optional[torch.Tensor]] = [None] * len(he_adjs)
        saved = [[] for _ in he_adjs]
        norms = [[] for _ in he_adjs]
        planes_out, planes_in = _plane_flags(he_adjs)
        for l in range(max(Ls)):
            tasks = []
            for g, graph in enumerate(he_adjs):
                L = Ls[g]
                if l >= L:       # (this channel has fewer layers: Model.py:31-36)
                    continue
                last = l == L - 1
                e_out = None if (last and not keep) else torch.empty_like(embed)
                flags = (DG_IN_PRESCALED | planes_in if l > 0 else 0) | DG_OUT_NORM | (0 if last else DG_OUT_PRESCALE | planes_out)
                if l == 0 and L >= 2:      # no running sum from the first layer; the second rebuilds it
                    acc_in = acc_out = None
                elif l == 1:
                    acc_in, acc_out, flags = embed, sums[g], flags | DG_ACC_REBUILD
                else:
                    acc_in, acc_out = (embed if l == 0 else sums[g]), sums[g]
                rn = torch.empty(embed.shape[0], dtype=torch.float32, device=embed.device) if keep else None
                tasks.append(ops.LayerTask(graph, embed if l == 0 else prev[g], e_out, acc_in, acc_out, flags, slope,
                                           row_norm_out=rn))
                prev[g] = e_out
                saved[g].append(e_out)
                norms[g].append(rn)
            hook = getattr(self, "layer_hook", None)     # timing hook, as in the Rec model
            if hook is not None:
                hook("begin", l, tasks)
            _launch_distinct(tasks, ops.gcn_layer_batched)
            if hook is not None:
                hook("end", l, tasks)
        return sums, saved, norms

    def _forward_wide(self, he_adjs, feature_list):
        """forward() on the kernels' 64-wide rows (zeros from column latdim on)."""
        embed = _LinearFn.apply(feature_list, self.transform_layer.weight, self.transform_layer.bias)
        if self.n_hid != DG_D:
            embed = torch.nn.functional.pad(embed, (0, DG_D - self.n_hid))
        target, s1, s2 = self.propagate(he_adjs, embed)
        return s1, s2, target

    def forward(self, he_adjs, feature_list, is_training=True):
        if self.wide:
            from . import wide
            embed = _LinearFn.apply(feature_list, self.transform_layer.weight, self.transform_layer.bias)
            target, s1, s2 = wide.nc_propagate(he_adjs, embed, self.layer_counts(len(he_adjs)), float(self.act.negative_slope))
            return s1, s2, target
        s1, s2, target = self._forward_wide(he_adjs, feature_list)
        if self.n_hid != DG_D:
            L = self.n_hid
            return s1[:, :L], s2[:, :L], target[:, :L]
        return s1, s2, target

    def cal_loss(self, ancs, label, he_adjs, initial_feature, ts=None, noise=None, keep_mask=None,
                 validate=None, drop_seed=None):  # Model.py:75-89
        """`ts` / `noise` / `keep_mask`: the draws of training_losses2 handed in instead of drawn (`drop_seed`, int64[1]
        on the device: the key the kernels draw the dropout mask from when no mask is given); `validate=False` skips
        the range check of caller-supplied timesteps (a host synchronisation a recorded step cannot have)."""
        if self.wide:
            from . import wide
            return wide.nc_cal_loss(self, ancs, label, he_adjs, initial_feature, ts, noise, keep_mask)
        s1, s2, target_embedding = self._forward_wide(he_adjs, initial_feature)
        source_embeddings = s1 + s2
        # training_losses2 before its [batch] gather: the head below gathers the batch rows of everything at once
        row_loss, diff_embeddings = self.diffusion_model.training_losses_rows(
            self.user_denoise_model, target_embedding, source_embeddings, ts, noise, keep_mask, validate, drop_seed)
        ancs = ancs.long().contiguous()
        if not (label.dtype == torch.float32 and label.dim() == 2 and label.is_contiguous()):
            label = torch.argmax(label[ancs], dim=-1)        # (any other label table: the batch's labels by torch)
        # all_embeddings = target + diffused, log_softmax(dense(.))[ancs], nll_loss and diff_loss[ancs].mean() in one op:
        # only the batch rows' scores are ever used
        return _LossHeadFn.apply(target_embedding, diff_embeddings, row_loss, self._dense_weight(), self.dense.bias, ancs, label)

    def _dense_weight(self) -> torch.Tensor:
        """dense.weight [classes, latdim] as the kernels take it: [classes, 64], zeros behind the columns."""
        w = self.dense.weight
        return w if self.n_hid == DG_D else torch.nn.functional.pad(w, (0, DG_D - self.n_hid))

    def get_embeds(self, ancs, label, he_adjs, initial_feature):  # Model.py:90-99
        all_embeddings, _ = self.get_allembeds(he_adjs, initial_feature)
        return all_embeddings[ancs]

    def get_allembeds(self, he_adjs, initial_feature):  # Model.py:100-108
        if self.wide:
            s1, s2, target_embedding = self.forward(he_adjs, initial_feature)
            all_embeddings = target_embedding + self.diffusion_model.p_sample(self.user_denoise_model, s1 + s2, self.args.sampling_steps)
            return all_embeddings, self.dense(all_embeddings)
        s1, s2, target_embedding = self._forward_wide(he_adjs, initial_feature)
        source_embeddings = s1 + s2
        diff_embeddings = self.diffusion_model.p_sample(self.user_denoise_model, source_embeddings, self.args.sampling_steps)
        all_embeddings = target_embedding + diff_embeddings
        scores = ops.nc_logits(all_embeddings.contiguous(), self._dense_weight().detach().contiguous(),
                               self.dense.bias.detach().contiguous())
        return (all_embeddings if self.n_hid == DG_D else all_embeddings[:, : self.n_hid]), scores
