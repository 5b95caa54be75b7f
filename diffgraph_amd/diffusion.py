"""GaussianDiffusion and Denoise with the reference's method signatures
(DiffGraph-Rec/Model.py:176-399; DiffGraph_NC/Model.py:217-415 is the same code minus
importance sampling), driving the HIP kernels.

Differences a caller can see:
  * training_losses2 / q_sample / p_sample accept optional `ts` / `noise` tensors so runs
    are reproducible; left as None they are drawn with torch like the reference does.
  * The noise-schedule tables are built in fp64 on the host with the reference's torch ops,
    then cast to fp32 device tables once (the reference indexes the fp64 table and casts
    the gathered values, which gives the same fp32 numbers).
  * Two ways through a Denoise.  The reference's default shape -- dims='[64]', d_emb_size=8: one 72->64 and one
    64->64 layer -- runs in the fused kernels (csrc/diffusion.hip: denoiser, loss and reverse loop as single
    launches).  Any other `--dims` list or `--d_emb_size` (params.py:32-33) runs layer by layer: every nn.Linear
    through the HIP dense-layer kernels (dg_linear_fwd / dg_linear_wgrad, csrc/linear.hip), the elementwise steps
    between them (normalise, dropout, tanh, the loss) as torch device ops under autograd.  Correct for every
    configuration the reference accepts at latdim <= 64; only the default shape is tuned.
  * `--latdim` above 64: see diffgraph_amd/wide.py (column blocks through the propagation kernel, the rest in torch).
  * `--latdim` below 64 (params.py:14): the embedding tables stay 64 floats wide with zeros from column latdim on
    (the propagation, the loss head and the evaluation kernels are built for that width, and zero columns change
    none of their sums); a denoiser of in_dims[0] = latdim takes such rows, works on their first latdim columns layer
    by layer and returns rows of the width it was given.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import DG_D, DiffGraphHipError


class _LinearFn(torch.autograd.Function):
    """nn.Linear (y = x . w^T + b) through the HIP dense-layer kernels, forward and backward."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = x.contiguous(), w.contiguous()
        ctx.save_for_backward(x, w)
        return ops.linear_fwd(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        d_x = ops.linear_fwd(dy, w.t().contiguous(), None) if ctx.needs_input_grad[0] else None
        d_w, d_b = ops.linear_wgrad(dy, x)
        return d_x, d_w, d_b


class Denoise(nn.Module):
    """Denoise(in_dims, out_dims, emb_size, norm, dropout) -- Model.py:176-234.  `fused`: the layer shapes the fused
    kernels are built for (in_dims = out_dims = [64, 64], emb_size 8); other shapes run layer by layer."""

    def __init__(self, in_dims, out_dims, emb_size, norm=False, dropout=0.5):
        super().__init__()
        in_dims, out_dims = list(in_dims), list(out_dims)
        if len(in_dims) < 2 or len(out_dims) < 2 or in_dims[0] < 1 or out_dims[-1] != in_dims[0] or emb_size < 1:
            raise NotImplementedError(f"a denoiser takes and returns rows of one width (latdim), got in_dims {in_dims}, "
                                      f"out_dims {out_dims}")
        self.in_dims, self.out_dims = in_dims, out_dims
        self.width = in_dims[0]             # latdim: the columns of a row that carry values
        self.time_emb_dim = emb_size
        self.norm = norm
        self.fused = in_dims == [DG_D, DG_D] and out_dims == [DG_D, DG_D] and emb_size == 8
        self.emb_layer = nn.Linear(emb_size, emb_size)
        in_temp = [in_dims[0] + emb_size] + in_dims[1:]                                       # Model.py:186-191
        self.in_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(in_temp[:-1], in_temp[1:])])
        self.out_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(out_dims[:-1], out_dims[1:])])
        self.drop = nn.Dropout(dropout)
        self.init_weights()
        self._params = {}

    def init_weights(self):  # Model.py:196-212
        for layer in list(self.in_layers) + list(self.out_layers) + [self.emb_layer]:
            size = layer.weight.size()
            std = np.sqrt(2.0 / (size[0] + size[1]))
            layer.weight.data.normal_(0.0, std)
            layer.bias.data.normal_(0.0, 0.001)

    def params(self, norm: bool) -> ops.DenoiserParams:
        """Kernel-side view of the weights (pointers stay valid across in-place optimiser steps)."""
        if not self.fused:
            raise DiffGraphHipError("Denoise.params(): the fused kernels take the default layer shapes only")
        ws = (self.emb_layer.weight, self.emb_layer.bias, self.in_layers[0].weight, self.in_layers[0].bias,
              self.out_layers[0].weight, self.out_layers[0].bias)
        key = (bool(norm),) + tuple(w.data_ptr() for w in ws)
        p = self._params.get(key)
        if p is None:
            self._params.clear()
            p = ops.DenoiserParams(*[w.data for w in ws], norm=norm)
            self._params[key] = p
        return p

    def forward(self, x, timesteps, mess_dropout=True, keep_mask=None):
        """keep_mask (uint8 / bool [n, 64]): the dropout mask to apply in train() mode instead of a fresh draw."""
        if not self.fused:
            return self.forward_layers(x, timesteps, mess_dropout, keep_mask)
        n_steps = int(getattr(self, "n_steps", 0)) or int(timesteps.max().item()) + 1
        p = self.params(bool(self.norm))
        p.time_bias(n_steps, refresh=True)
        if mess_dropout and self.training and self.drop.p > 0:
            # dropout sits between F.normalize and the first Linear (Model.py:221-224): inside the kernel, as in the
            # loss kernels -- over the caller's mask or over one drawn from a fresh device word
            seed = None if keep_mask is not None else torch.randint(0, 2 ** 62, (1,), device=x.device, dtype=torch.int64)
            return ops.denoise_fwd(p, x.contiguous(), timesteps, n_steps, drop_p=float(self.drop.p), keep_mask=keep_mask,
                                   drop_seed=seed)
        return ops.denoise_fwd(p, x.contiguous(), timesteps, n_steps)

    def forward_layers(self, x, timesteps, mess_dropout=True, keep_mask=None):
        """Model.py:214-234 layer by layer, differentiable: the Linear layers in the HIP dense-layer kernels.  Rows wider
        than latdim (the 64-wide tables of a narrower model) are cut to their first latdim columns and the result is
        padded back with zeros."""
        wide = x.shape[1]
        if wide != self.width:
            x = x[:, : self.width]
            if keep_mask is not None:
                keep_mask = keep_mask[:, : self.width]
        E = self.time_emb_dim
        half = E // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=x.device) / half)
        temp = timesteps[:, None].float() * freqs[None]
        time_emb = torch.cat([torch.cos(temp), torch.sin(temp)], dim=-1)
        if E % 2:
            time_emb = torch.cat([time_emb, torch.zeros_like(time_emb[:, :1])], dim=-1)
        emb = _LinearFn.apply(time_emb, self.emb_layer.weight, self.emb_layer.bias)
        if self.norm:
            x = torch.nn.functional.normalize(x)
        if mess_dropout and self.training and self.drop.p > 0:
            if keep_mask is not None:
                x = x * (keep_mask.to(x.dtype) * (1.0 / (1.0 - float(self.drop.p))))
            else:
                x = self.drop(x)
        h = torch.cat([x, emb], dim=-1)
        for layer in self.in_layers:
            h = torch.tanh(_LinearFn.apply(h, layer.weight, layer.bias))
        for i, layer in enumerate(self.out_layers):
            h = _LinearFn.apply(h, layer.weight, layer.bias)
            if i != len(self.out_layers) - 1:
                h = torch.tanh(h)
        return h if wide == self.width else torch.nn.functional.pad(h, (0, wide - self.width))


def dropout_rate(models) -> float:
    """The dropout probability of a loss call through `models` (Denoise modules): p while they are in training mode
    (Model.py:223-224), else 0."""
    ps = {float(m.drop.p) if m.training else 0.0 for m in models}
    if len(ps) != 1:
        raise DiffGraphHipError("the denoisers of one loss call must share their training mode and dropout rate")
    return max(ps.pop(), 0.0)


def dropout_args(models, keep_mask, device, drop_seed=None):
    """(drop_p, keep_mask, drop_seed) for a loss call through `models`.  Without a caller-supplied mask the kernels
    draw one from the key `drop_seed` (int64[1] on the device; default: a fresh one from torch's generator -- a device
    word, so a captured step draws a new mask per replay)."""
    p = dropout_rate(models)
    if p <= 0.0:
        return 0.0, None, None
    if keep_mask is not None:
        return p, keep_mask.contiguous().view(torch.uint8) if keep_mask.dtype == torch.bool else keep_mask.contiguous(), None
    if drop_seed is None:
        drop_seed = torch.randint(0, 2 ** 62, (1,), device=device, dtype=torch.int64)
    return p, None, drop_seed


class _DiffusionLossFn(torch.autograd.Function):
    """training_losses2 as one differentiable op: HIP kernels for the per-row work in both
    directions, the weight gradients included (ops.denoiser_wgrad over tables the backward
    kernel writes)."""

    @staticmethod
    def forward(ctx, target, x_start, noise, ts, w_emb, b_emb, w_in, b_in, w_out, b_out, gd, model, validate, drop):
        dev = x_start.device
        p = model.params(bool(model.norm))
        p.time_bias(gd.steps, refresh=True)
        sa = gd._dev("sqrt_alphas_cumprod", dev) if gd.noise_scale != 0 else None
        sb = gd._dev("sqrt_one_minus_alphas_cumprod", dev) if gd.noise_scale != 0 else None
        lw = gd._dev("loss_weight", dev, torch.float64)
        target, x_start = target.contiguous(), x_start.contiguous()
        drop_p, keep_mask, drop_seed = drop
        spans = [ops.LossSpan(p, 0, x_start.shape[0], drop_p)]
        row_loss, out = ops.diffusion_loss_fwd_spans(spans, target, x_start, noise, ts, sa, sb, lw, validate=validate,
                                                     keep_mask=keep_mask, drop_seed=drop_seed)
        ctx.save_for_backward(target, x_start, noise if noise is not None else x_start, ts)
        ctx.aux = (gd, spans, sa, sb, lw, keep_mask, drop_seed)
        return row_loss, out

    @staticmethod
    def backward(ctx, g_row, g_out):
        target, x_start, noise, ts = ctx.saved_tensors
        gd, spans, sa, sb, lw, keep_mask, drop_seed = ctx.aux
        dev = x_start.device
        n = x_start.shape[0]
        if g_row is None:
            g_row = torch.zeros(n, dtype=torch.float64, device=dev)
        g_out = None if g_out is None else g_out.contiguous()
        d_x, d_tgt, dout, hid, dpre, xn = ops.diffusion_loss_bwd_spans(spans, target, x_start, noise, ts, sa, sb, lw,
                                                                       g_row.contiguous(), g_out, keep_mask=keep_mask,
                                                                       drop_seed=drop_seed)
        d_w_emb, d_b_emb, d_w_in, d_b_in, d_w_out, d_b_out = ops.denoiser_wgrad(spans[0].dn, gd.steps, ts, dout, hid, dpre, xn)
        return d_tgt, d_x, None, None, d_w_emb, d_b_emb, d_w_in, d_b_in, d_w_out, d_b_out, None, None, None, None


class GaussianDiffusion(nn.Module):
    """GaussianDiffusion(noise_scale, noise_min, noise_max, steps) -- Model.py:236-399."""

    def __init__(self, noise_scale, noise_min, noise_max, steps, beta_fixed=True):
        super().__init__()
        self.noise_scale, self.noise_min, self.noise_max, self.steps = noise_scale, noise_min, noise_max, steps
        self._dev_tables = {}
        if noise_scale != 0:
            self.betas = torch.tensor(self.get_betas(), dtype=torch.float64)
            if beta_fixed:
                self.betas[0] = 0.0001
            self.calculate_for_diffusion()

    def get_betas(self):  # Model.py:257-266
        start = self.noise_scale * self.noise_min
        end = self.noise_scale * self.noise_max
        variance = np.linspace(start, end, self.steps, dtype=np.float64)
        alpha_bar = 1 - variance
        betas = [1 - alpha_bar[0]]
        for i in range(1, self.steps):
            betas.append(min(1 - alpha_bar[i] / alpha_bar[i - 1], 0.999))
        return np.array(betas)

    def calculate_for_diffusion(self):  # Model.py:268-285
        alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(alphas, dim=0)
        self.alphas_cumprod_prev = torch.cat([torch.tensor([1.0], dtype=torch.float64), self.alphas_cumprod[:-1]])
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        self.posterior_variance = self.betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = torch.log(
            torch.cat([self.posterior_variance[1].unsqueeze(0), self.posterior_variance[1:]]))
        self.posterior_mean_coef1 = self.betas * torch.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * torch.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        # SNR(t-1) - SNR(t), 1.0 at t == 0 (Model.py:362-363), fp64
        snr = self.alphas_cumprod / (1 - self.alphas_cumprod)
        w = torch.roll(snr, 1) - snr
        w[0] = 1.0
        self.loss_weight = w

    # -- device copies ----------------------------------------------------------------
    def _dev(self, name: str, device, dtype=torch.float32) -> torch.Tensor:
        key = (name, str(device), dtype)
        t = self._dev_tables.get(key)
        if t is None:
            t = getattr(self, name).to(dtype).to(device).contiguous()
            self._dev_tables[key] = t
        return t

    # -- reference API ------------------------------------------------------------------
    def _extract_into_tensor(self, arr, timesteps, broadcast_shape):  # Model.py:307-312
        res = arr.to(timesteps.device)[timesteps].float()
        while len(res.shape) < len(broadcast_shape):
            res = res[..., None]
        return res.expand(broadcast_shape)

    def q_sample(self, x_start, t, noise=None):  # Model.py:302-305
        if noise is None:
            noise = torch.randn_like(x_start)
        dev, wide = x_start.device, x_start.shape[1]
        if wide < DG_D:      # rows of a narrower latent space: through the 64-wide kernel with zeros behind them
            pad = lambda v: torch.nn.functional.pad(v, (0, DG_D - wide))
            x_start, noise = pad(x_start), pad(noise)
        out = ops.q_sample(x_start.contiguous(), noise.contiguous(), t, self._dev("sqrt_alphas_cumprod", dev),
                           self._dev("sqrt_one_minus_alphas_cumprod", dev))
        return out if wide >= DG_D else out[:, :wide]

    def p_mean_variance(self, model, x, t):  # Model.py:314-325
        model.n_steps = self.steps
        model_output = model(x, t, False)
        model_log_variance = self._extract_into_tensor(self.posterior_log_variance_clipped, t, x.shape)
        model_mean = (self._extract_into_tensor(self.posterior_mean_coef1, t, x.shape) * model_output
                      + self._extract_into_tensor(self.posterior_mean_coef2, t, x.shape) * x)
        return model_mean, model_log_variance

    def p_sample(self, model, x_start, steps, noise=None):  # Model.py:287-300
        """All self.steps reverse iterations in one fused launch (the reference loops in Python)."""
        if steps == 0:
            x_t = x_start
        else:
            t = torch.full((x_start.shape[0],), steps - 1, dtype=torch.long, device=x_start.device)
            x_t = self.q_sample(x_start, t, noise)
        dev = x_start.device
        if not model.fused:
            return self._p_sample_layers(model, x_t)
        p = model.params(bool(model.norm))
        p.time_bias(self.steps, refresh=True)
        return ops.p_sample_loop(p, x_t.contiguous(), self._dev("posterior_mean_coef1", dev),
                                 self._dev("posterior_mean_coef2", dev))

    def _p_sample_layers(self, model, x_t):
        """The reverse loop of Model.py:293-300 step by step (a denoiser outside the fused kernels' shape)."""
        dev, wide = x_t.device, x_t.shape[1]
        c1, c2 = self._dev("posterior_mean_coef1", dev), self._dev("posterior_mean_coef2", dev)
        with torch.no_grad():
            x_t = x_t[:, : model.width]
            for i in range(self.steps - 1, -1, -1):
                t = torch.full((x_t.shape[0],), i, dtype=torch.long, device=dev)
                x_t = c1[i] * model.forward_layers(x_t, t, False) + c2[i] * x_t        # model_mean (Model.py:323)
        return x_t if wide == model.width else torch.nn.functional.pad(x_t, (0, wide - model.width))

    def p_sample_many(self, pairs, steps, noises=None):
        """p_sample for several (model, x_start) pairs in one fused launch (HGDM.predict runs the user and
        the item denoiser back to back, Model.py:114-115)."""
        if any(not model.fused for model, _ in pairs):
            return [self.p_sample(model, x_start, steps, None if noises is None else noises[k])
                    for k, (model, x_start) in enumerate(pairs)]
        prepared = []
        for k, (model, x_start) in enumerate(pairs):
            if steps == 0:
                x_t = x_start
            else:
                t = torch.full((x_start.shape[0],), steps - 1, dtype=torch.long, device=x_start.device)
                x_t = self.q_sample(x_start, t, None if noises is None else noises[k])
            p = model.params(bool(model.norm))
            p.time_bias(self.steps, refresh=True)
            prepared.append((p, x_t.contiguous()))
        dev = pairs[0][1].device
        return ops.p_sample_loop_batched(prepared, self._dev("posterior_mean_coef1", dev),
                                         self._dev("posterior_mean_coef2", dev))

    def training_losses2(self, model, targetEmbeds, x_start, batch, ts=None, noise=None, keep_mask=None,
                         validate=None, drop_seed=None):  # Model.py:348-366
        row_loss, out = self.training_losses_rows(model, targetEmbeds, x_start, ts, noise, keep_mask, validate, drop_seed)
        return row_loss[batch], out

    def training_losses_rows(self, model, targetEmbeds, x_start, ts=None, noise=None, keep_mask=None, validate=None,
                             drop_seed=None):
        """training_losses2 before its final `[batch]` gather (Model.py:348-365): the fp64 SNR-weighted loss of
        EVERY row and the denoiser output.  With `model` in training mode its dropout (Model.py:223-224) is applied
        inside the kernels: to the elements `keep_mask` ([n, 64] bool / uint8) keeps, or to a fresh draw."""
        n = x_start.size(0)
        dev = x_start.device
        if validate is None:
            validate = ts is not None  # caller-supplied timesteps are range-checked (one host sync) unless told otherwise
        if ts is None:
            ts = torch.randint(0, self.steps, (n,), device=dev).long()
        if noise is None and self.noise_scale != 0:
            noise = torch.randn_like(x_start)
        if not model.fused:
            return self._training_losses_layers(model, targetEmbeds, x_start, ts, noise, keep_mask)
        norm = bool(model.norm)
        xs = x_start.contiguous()
        drop = dropout_args([model], keep_mask, dev, drop_seed)
        if torch.is_grad_enabled() and (xs.requires_grad or targetEmbeds.requires_grad
                                        or any(q.requires_grad for q in model.parameters())):
            return _DiffusionLossFn.apply(
                targetEmbeds, xs, noise, ts, model.emb_layer.weight, model.emb_layer.bias, model.in_layers[0].weight,
                model.in_layers[0].bias, model.out_layers[0].weight, model.out_layers[0].bias, self, model, validate, drop)
        p = model.params(norm)
        p.time_bias(self.steps, refresh=True)
        sa = self._dev("sqrt_alphas_cumprod", dev) if self.noise_scale != 0 else None
        sb = self._dev("sqrt_one_minus_alphas_cumprod", dev) if self.noise_scale != 0 else None
        return ops.diffusion_loss_fwd_spans([ops.LossSpan(p, 0, n, drop[0])], targetEmbeds.contiguous(), xs, noise, ts, sa, sb,
                                            self._dev("loss_weight", dev, torch.float64), validate=validate,
                                            keep_mask=drop[1], drop_seed=drop[2])

    def _training_losses_layers(self, model, targetEmbeds, x_start, ts, noise, keep_mask):
        """Model.py:348-365 for a denoiser outside the fused kernels' shape: q_sample, the model layer by layer, the
        SNR-weighted mean square -- differentiable torch ops round the HIP dense layers."""
        dev, wide, L = x_start.device, x_start.shape[1], model.width
        # (rows wider than latdim: everything below is over their first latdim columns -- the mean too, Model.py:368-369)
        x_t = x_start = x_start[:, :L]
        if self.noise_scale != 0:
            sa, sb = self._dev("sqrt_alphas_cumprod", dev), self._dev("sqrt_one_minus_alphas_cumprod", dev)
            x_t = sa[ts][:, None] * x_start + sb[ts][:, None] * noise[:, :L]          # q_sample (Model.py:302-305)
        out = model.forward_layers(x_t, ts, True, None if keep_mask is None else keep_mask[:, :L])
        mse = self.mean_flat((targetEmbeds[:, :L] - out) ** 2)
        if wide != L:
            out = torch.nn.functional.pad(out, (0, wide - L))
        return self._dev("loss_weight", dev, torch.float64)[ts] * mse, out            # (SNR(t-1) - SNR(t), 1 at t = 0)

    def mean_flat(self, tensor):  # Model.py:368-369
        return tensor.mean(dim=list(range(1, len(tensor.shape))))

    def SNR(self, t):  # Model.py:371-373
        ac = self.alphas_cumprod.to(t.device)
        return ac[t] / (1 - ac[t])
