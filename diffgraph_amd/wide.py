"""The recommendation model at a latent width above the kernels' 64 columns (`--latdim` > 64, DiffGraph-Rec/params.py:14).

The propagation is column-separable up to the row normalisation: y = LeakyReLU(A_hat x) holds column block by column
block, so every 64-column block of a table goes through the same HIP propagation kernel (dg_spmm_norm_act_fwd, and its
transposed form in the backward pass), and what couples the blocks -- F.normalize over the whole row, the layer sums,
the loss head's dot products, the evaluation scores -- is formed by torch device ops under autograd, written as the
reference writes them (Model.py:53-111).  The denoisers run layer by layer in the HIP dense-layer kernels
(diffusion.Denoise.forward_layers).  Correct for every width; tuned for none: the fused kernels are built for 64.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn.functional as F

from . import ops
from ._lib import DG_D


class _SpmmFn(torch.autograd.Function):
    """y = LeakyReLU_slope(A_hat x) on one [n, 64] column block (DGLLayer.forward, Model.py:142-174), differentiable:
    d x = A_hat^T (LeakyReLU'(y) * d y), through the same kernel on the transposed graph with slope 1."""

    @staticmethod
    def forward(ctx, x, graph, slope):
        y = ops.spmm_norm_act(graph, x.contiguous(), slope)
        ctx.save_for_backward(y)
        ctx.graph, ctx.slope = graph, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dz = torch.where(y > 0, dy, dy * ctx.slope).contiguous()        # LeakyReLU passes `slope` at 0 and below
        return ops.spmm_norm_act(ctx.graph, dz, 1.0, transpose=True), None, None


def propagate(graphs: Sequence, e0: torch.Tensor, n_layers: int, slope: float):
    """HGDM.forward's two tables (Model.py:53-92) for e0 [n, latdim], latdim any: (layer sum over graphs[0], sum over
    the other graphs of their layer sums), each [n, latdim]."""
    n, L = e0.shape
    B = -(-L // DG_D)
    x0 = e0 if L == B * DG_D else F.pad(e0, (0, B * DG_D - L))          # zero columns stay zero through every layer
    sums: List[torch.Tensor] = []
    for g in graphs:
        acc = x0
        blocks = [x0[:, DG_D * b: DG_D * (b + 1)] for b in range(B)]
        for _ in range(n_layers):
            blocks = [_SpmmFn.apply(xb, g, slope) for xb in blocks]
            acc = acc + F.normalize(torch.cat(blocks, dim=1), p=2, dim=1)          # Model.py:67-70
        sums.append(acc)
    het = sums[1] if len(sums) > 1 else torch.zeros_like(x0)
    for s in sums[2:]:
        het = het + s                                                               # sum(list), left to right (Model.py:87)
    return sums[0][:, :L], het[:, :L]


def cal_loss(model, ancs, poss, negs, ts_u=None, noise_u=None, ts_i=None, noise_i=None, keep_u=None, keep_i=None) -> torch.Tensor:
    """HGDM.cal_loss (Model.py:94-111) as float64[4] = (loss, bprLoss, regLoss, diff_loss)."""
    dp, nu, args = model.diffusion_process, model.n_user, model.args
    ui, het = propagate([model.target_adj] + model.behavior_mats, model.embedding_table, model.n_layers,
                        float(model.act.negative_slope))
    rl_u, out_u = dp.training_losses_rows(model.usr_denoiser, ui[:nu], het[:nu], ts_u, noise_u, keep_u)
    rl_i, out_i = dp.training_losses_rows(model.item_denoiser, ui[nu:], het[nu:], ts_i, noise_i, keep_i)
    ancs, poss, negs = ancs.long(), poss.long(), negs.long()
    diff = rl_u[ancs].mean() + rl_i[poss].mean()
    usr, itm = ui[:nu] + out_u, ui[nu:] + out_i
    a, p, n = usr[ancs], itm[poss], itm[negs]
    score = (a * p).sum(-1) - (a * n).sum(-1)                                       # pairPredict (Utils/Utils.py:4-8)
    bpr = -score.sigmoid().log().sum() / args.batch
    reg = ((torch.norm(a) ** 2 + torch.norm(p) ** 2 + torch.norm(n) ** 2) * args.reg) / args.batch
    return torch.stack([bpr + reg + diff, bpr.double(), reg.double(), diff])


def all_rank_topk(usr: torch.Tensor, itm: torch.Tensor, users: torch.Tensor, trn_rowptr: torch.Tensor, trn_col: torch.Tensor,
                  topk: int, batch: int = 1024):
    """t.topk(t.mm(usr[users], itm^T) * (1 - trnMask) - trnMask * 1e8, topk) (main.py:157-158) in batches of users,
    the mask rows scattered from the training CSR.  Returns (values [len(users), topk], item ids int32)."""
    n_item = itm.shape[0]
    vals, idx = [], []
    deg = (trn_rowptr[1:] - trn_rowptr[:-1]).long()
    for b in range(0, users.numel(), batch):
        u = users[b: b + batch]
        mask = torch.zeros((u.numel(), n_item), dtype=usr.dtype, device=usr.device)
        cnt = deg[u]
        rows = torch.repeat_interleave(torch.arange(u.numel(), device=usr.device), cnt)
        if rows.numel():
            first = trn_rowptr[u].long()
            offs = torch.arange(rows.numel(), device=usr.device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
            mask[rows, trn_col[torch.repeat_interleave(first, cnt) + offs].long()] = 1.0
        preds = torch.mm(usr[u], itm.t()) * (1 - mask) - mask * 1e8
        v, i = torch.topk(preds, min(topk, n_item))
        vals.append(v)
        idx.append(i.to(torch.int32))
    return torch.cat(vals), torch.cat(idx)


# ------------------------------------------------------------------------------------------------ DiffGraph_NC
def nc_propagate(he_adjs: Sequence, embed: torch.Tensor, layer_counts: Sequence[int], slope: float) -> List[torch.Tensor]:
    """The three layer sums of DiffGraph_NC's HGDM.forward (Model.py:52-71) for embed [n, latdim], latdim any: the
    NORMALISED activation feeds the next layer; channel c runs layer_counts[c] layers."""
    n, L = embed.shape
    B = -(-L // DG_D)
    x0 = embed if L == B * DG_D else F.pad(embed, (0, B * DG_D - L))
    sums = []
    for g, count in zip(he_adjs, layer_counts):
        acc = cur = x0
        for _ in range(count):
            blocks = [_SpmmFn.apply(cur[:, DG_D * b: DG_D * (b + 1)], g, slope) for b in range(B)]
            cur = F.normalize(torch.cat(blocks, dim=1), p=2, dim=1)
            acc = acc + cur
        sums.append(acc[:, :L])
    return sums


def nc_cal_loss(model, ancs, label, he_adjs, initial_feature, ts=None, noise=None, keep_mask=None):
    """DiffGraph_NC's HGDM.cal_loss (Model.py:75-89): (nll_loss, diff_loss)."""
    s1, s2, target = model.forward(he_adjs, initial_feature)
    ancs = ancs.long()
    row_loss, diffused = model.diffusion_model.training_losses_rows(model.user_denoise_model, target, s1 + s2, ts, noise, keep_mask)
    scores = F.log_softmax(model.dense(target + diffused), dim=1)
    return F.nll_loss(scores[ancs], torch.argmax(label[ancs], dim=-1)), row_loss[ancs].mean()
