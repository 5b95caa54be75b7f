"""Training-mode dropout inside training_losses2 (nn.Dropout(0.5) between F.normalize and the first Linear of
Denoise, DiffGraph-Rec/Model.py:193,221-224; DiffGraph_NC/Model.py:262-265), fused into the loss kernels.
Golden values come from the reference's own cal_loss in train() mode with the keep masks torch drew recorded
(tools/make_golden.py rec_train / nc_train)."""
import numpy as np
import pytest
import torch

import diffgraph_amd as dg
from diffgraph_amd import data as dgdata, ops
from oracle import ref_cpu as O
from tests.helpers import GRAD_TOL, REC_TABLE, assert_close_to_max, assert_rows_close, load, tiny_fixture, weights_from
from tests.test_gpu_parity import DEV, dev, rec_model, T

pytestmark = pytest.mark.gpu
NAMES = ("w_emb", "b_emb", "w_in", "b_in", "w_out", "b_out")


def _grads(dn):
    return {"w_emb": dn.emb_layer.weight.grad, "b_emb": dn.emb_layer.bias.grad, "w_in": dn.in_layers[0].weight.grad,
            "b_in": dn.in_layers[0].bias.grad, "w_out": dn.out_layers[0].weight.grad, "b_out": dn.out_layers[0].bias.grad}


def test_training_losses2_train_mode_golden():
    fx, nu, *_ = tiny_fixture()
    tr = load("rec_tiny_train.npz")
    gd = dg.GaussianDiffusion(**REC_TABLE)
    m = rec_model(dgdata.rec_workload("tiny"), fx).train()
    with torch.no_grad():
        loss, out = gd.training_losses2(m.usr_denoiser, dev(fx["fwd_tu"]), dev(fx["fwd_hu"]), dev(fx["tl_batch"]),
                                        ts=dev(tr["tl_ts"]), noise=dev(tr["tl_noise"]), keep_mask=dev(tr["tl_keep"]))
    assert_rows_close(out, tr["tl_out"], what="training_losses2 out (train mode)")
    np.testing.assert_allclose(loss.cpu().numpy(), tr["tl_loss"], rtol=1e-5)
    # and the mask matters: eval mode gives other numbers
    m.eval()
    with torch.no_grad():
        _, out_eval = gd.training_losses2(m.usr_denoiser, dev(fx["fwd_tu"]), dev(fx["fwd_hu"]), dev(fx["tl_batch"]),
                                          ts=dev(tr["tl_ts"]), noise=dev(tr["tl_noise"]))
    assert float((out_eval.cpu() - T(tr["tl_out"])).abs().max()) > 1e-3


def test_rec_cal_loss_train_mode_golden():
    """cal_loss + backward in train() mode -- the reference's first epoch -- against the reference's own numbers:
    the four losses, the gradients of both embedding tables and of all twelve denoiser parameters."""
    fx, nu, ni, *_ = tiny_fixture()
    tr = load("rec_tiny_train.npz")
    m = rec_model(dgdata.rec_workload("tiny"), fx).train()
    m.set_embeddings(dev(fx["user_emb"]), dev(fx["item_emb"]))
    from types import SimpleNamespace
    m.args = SimpleNamespace(**{**vars(m.args), "batch": int(fx["cl_batch_div"][0]), "reg": float(fx["cl_batch_div"][1])})
    loss, bpr, reg, diff = m.cal_loss(dev(fx["cl_ancs"]), dev(fx["cl_poss"]), dev(fx["cl_negs"]), dev(tr["ts_u"]),
                                      dev(tr["noise_u"]), dev(tr["ts_i"]), dev(tr["noise_i"]), dev(tr["keep_u"]), dev(tr["keep_i"]))
    np.testing.assert_allclose([float(v.detach()) for v in (loss, bpr, reg, diff)], tr["losses"], rtol=1e-5)
    loss.backward()
    g = m.embedding_table.grad
    assert_rows_close(g[:nu], tr["grad_user"], rtol=1e-5, what="grad user_emb")
    assert_rows_close(g[nu:], tr["grad_item"], rtol=1e-5, what="grad item_emb")
    for pre, dn in (("usr_", m.usr_denoiser), ("item_", m.item_denoiser)):
        for nm, gv in _grads(dn).items():
            want = tr["grad_" + pre + nm]
            assert_close_to_max(gv, want, GRAD_TOL, "grad " + pre + nm)


def test_nc_cal_loss_train_mode_golden():
    """One NC training step with the denoiser's dropout on, as every step of the reference's NC program runs
    (it never calls eval()): losses and all ten parameter gradients, as-coded graphs (3 x APA)."""
    from tests.test_oracle_golden import _dblp_graphs
    fx, tr = load("nc_dblp.npz"), load("nc_dblp_train.npz")
    NA, apa, apcpa, aptpa, features = _dblp_graphs()
    adjs = [dg.CSRGraph.from_scipy(apa) for _ in range(3)]
    model = dg.nc.HGDM(features.shape[1]).train()
    dn = model.user_denoise_model
    with torch.no_grad():
        model.transform_layer.weight.copy_(dev(fx["w_t"])); model.transform_layer.bias.copy_(dev(fx["b_t"]))
        model.dense.weight.copy_(dev(fx["w_dense"])); model.dense.bias.copy_(dev(fx["b_dense"]))
        dn.emb_layer.weight.copy_(dev(fx["dn_w_emb"])); dn.emb_layer.bias.copy_(dev(fx["dn_b_emb"]))
        dn.in_layers[0].weight.copy_(dev(fx["dn_w_in"])); dn.in_layers[0].bias.copy_(dev(fx["dn_b_in"]))
        dn.out_layers[0].weight.copy_(dev(fx["dn_w_out"])); dn.out_layers[0].bias.copy_(dev(fx["dn_b_out"]))
    onehot = torch.zeros(NA, 4)
    onehot[torch.arange(NA), T(load("dblp_graph.npz")["labels"])] = 1.0
    nll, diff = model.cal_loss(dev(fx["cl_ancs"]), dev(onehot), adjs, dev(features), dev(tr["ts"]), dev(tr["noise"]),
                               dev(tr["keep"]))
    np.testing.assert_allclose([float(nll.detach()), float(diff.detach())], tr["losses"], rtol=1e-5)
    (nll + diff).backward()
    got = dict(model.named_parameters())
    checked = 0
    for nm, prm in got.items():
        key = "grad_" + nm.replace(".", "_")
        if key not in tr.files:
            continue
        want = tr[key]
        assert prm.grad is not None, nm
        assert_close_to_max(prm.grad, want, GRAD_TOL, "grad " + nm)
        checked += 1
    assert checked == 10


def test_mask_drawn_in_the_kernel_equals_the_mask_supplied():
    """Without a caller's mask the kernels draw one from a seed word on the device (counter-based generator keyed by
    seed, span and row).  dg_dropout_keep_mask writes that mask out: feeding it back as keep_mask gives the same
    bits, forward and backward -- so the backward redraws exactly what the forward used -- and the draw has the
    statistics of Dropout(0.5)."""
    gen = torch.Generator().manual_seed(5)
    n_u, n_i, n_steps = 777, 333, 200
    n = n_u + n_i
    gd = dg.GaussianDiffusion(**REC_TABLE)
    dns = [dg.Denoise([64, 64], [64, 64], 8, norm=True).to(DEV) for _ in range(2)]
    spans = [ops.LossSpan(dns[0].params(True), 0, n_u, 0.5), ops.LossSpan(dns[1].params(True), n_u, n_i, 0.5)]
    ops.refresh_time_bias([sp.dn for sp in spans], n_steps)
    target, x0, noise = (dev(torch.randn(n, 64, generator=gen)) for _ in range(3))
    ts = dev(torch.randint(0, n_steps, (n,), generator=gen))
    sa, sb = gd._dev("sqrt_alphas_cumprod", DEV), gd._dev("sqrt_one_minus_alphas_cumprod", DEV)
    lw = gd._dev("loss_weight", DEV, torch.float64)
    g_row = dev(torch.rand(n, generator=gen).double())
    g_out = dev(torch.randn(n, 64, generator=gen))
    seed = torch.tensor([123456789012345], dtype=torch.int64, device=DEV)
    rl_a, out_a = ops.diffusion_loss_fwd_spans(spans, target, x0, noise, ts, sa, sb, lw, drop_seed=seed)
    bwd_a = ops.diffusion_loss_bwd_spans(spans, target, x0, noise, ts, sa, sb, lw, g_row, g_out, drop_seed=seed)
    mask = torch.cat([ops.dropout_keep_mask(seed, 0, 0.5, n_u), ops.dropout_keep_mask(seed, 1, 0.5, n_i)])
    rl_b, out_b = ops.diffusion_loss_fwd_spans(spans, target, x0, noise, ts, sa, sb, lw, keep_mask=mask)
    bwd_b = ops.diffusion_loss_bwd_spans(spans, target, x0, noise, ts, sa, sb, lw, g_row, g_out, keep_mask=mask)
    assert torch.equal(rl_a, rl_b) and torch.equal(out_a, out_b)
    for a, b in zip(bwd_a, bwd_b):
        assert torch.equal(a, b)
    m = mask.float()
    assert set(mask.unique().tolist()) == {0, 1}
    assert abs(float(m.mean()) - 0.5) < 0.01
    assert float(m.mean(0).min()) > 0.4 and float(m.mean(0).max()) < 0.6           # every column
    assert abs(float((m[:-1] * m[1:]).mean()) - 0.25) < 0.01                          # neighbouring rows independent
    assert not torch.equal(ops.dropout_keep_mask(seed, 0, 0.5, n_i), ops.dropout_keep_mask(seed, 1, 0.5, n_i))
    assert not torch.equal(ops.dropout_keep_mask(seed, 0, 0.5, n_i), ops.dropout_keep_mask(seed + 1, 0, 0.5, n_i))
    # the dropped inputs reach the first Linear as zeros and the kept ones doubled: xn table of the backward
    xn = bwd_a[5]
    unit = torch.nn.functional.normalize(sa[ts][:, None] * x0 + sb[ts][:, None] * noise)
    assert_rows_close(xn, unit * m * 2.0, what="xn after dropout")
    # a rate that 8 random bits cannot hit: the drawn mask drops round(256 p)/256 of the elements and scales the kept ones
    # by the reciprocal of what is left of THAT rate, so the expectation stays the input's
    sp01 = [ops.LossSpan(sp.dn, sp.begin, sp.n, 0.1) for sp in spans]
    xn01 = ops.diffusion_loss_bwd_spans(sp01, target, x0, noise, ts, sa, sb, lw, g_row, g_out, drop_seed=seed)[5]
    m01 = torch.cat([ops.dropout_keep_mask(seed, 0, 0.1, n_u), ops.dropout_keep_mask(seed, 1, 0.1, n_i)]).float()
    assert abs(float(m01.mean()) - (1 - 26 / 256)) < 0.005
    assert_rows_close(xn01, unit * m01 * (256.0 / 230.0), what="xn after dropout at p = 0.1")
    assert abs(float(m01.mean()) * 256.0 / 230.0 - 1.0) < 0.005      # E[mask * scale] = 1
    for bad in (0.001, 0.999):
        with pytest.raises(dg.DiffGraphHipError):
            ops.dropout_keep_mask(seed, 0, bad, n_u)
    # eval-mode call (drop_p = 0) ignores seed and mask
    plain = [ops.LossSpan(sp.dn, sp.begin, sp.n) for sp in spans]
    rl_c, _ = ops.diffusion_loss_fwd_spans(plain, target, x0, noise, ts, sa, sb, lw)
    assert not torch.equal(rl_c, rl_a)
    with pytest.raises(dg.DiffGraphHipError):
        ops.diffusion_loss_fwd_spans(spans, target, x0, noise, ts, sa, sb, lw)      # dropout on, neither mask nor seed


def test_device_side_draws_of_timesteps_and_noise():
    """ops.diffusion_draw stands in for t.randint(0, steps, (batch,)) and t.randn_like(x_start) (Model.py:352-353): a
    function of (seed, draws made so far) alone, uniform timesteps, standard-normal noise with independent entries;
    the streams are not torch's, so the distribution is what is checked."""
    from scipy import stats
    n, steps = 40000, 200
    st = ops.new_draw_state(DEV, seed=77)
    ts, z = ops.diffusion_draw(st, n, steps)
    torch.cuda.synchronize()
    assert st.tolist()[:2] == [77, 1] and st.tolist()[2] > 0
    key1 = int(st[2])
    ts2, z2 = ops.diffusion_draw(st, n, steps)
    assert st.tolist()[:2] == [77, 2] and int(st[2]) != key1
    assert not torch.equal(ts, ts2) and not torch.equal(z, z2)
    st_b = ops.new_draw_state(DEV, seed=77)
    ts_b, z_b = ops.diffusion_draw(st_b, n, steps)
    assert torch.equal(ts, ts_b) and torch.equal(z, z_b) and int(st_b[2]) == key1       # same state, same draw
    only_t, none_z = ops.diffusion_draw(st_b, n, steps, noise=False)                    # the second draw, timesteps only
    assert none_z is None and torch.equal(only_t, ts2)
    ts_c, _ = ops.diffusion_draw(ops.new_draw_state(DEV, seed=78), n, steps)
    assert not torch.equal(ts, ts_c)
    # timesteps: in range, uniform (chi-square over the 200 values)
    t = ts.cpu().numpy()
    assert t.min() == 0 and t.max() == steps - 1
    chi2 = ((np.bincount(t, minlength=steps) - n / steps) ** 2 / (n / steps)).sum()
    assert chi2 < steps + 5 * np.sqrt(2 * steps)
    # noise: moments, a Kolmogorov-Smirnov test against N(0,1), no correlation between neighbours in a row, a column,
    # the two halves of a Box-Muller pair, or successive draws
    v = z.double().cpu().numpy()
    m = v.size
    assert abs(v.mean()) < 5 / np.sqrt(m) and abs(v.var() - 1) < 5 * np.sqrt(2 / m)
    assert abs((v ** 4).mean() - 3) < 5 * np.sqrt(96 / m) and abs((v ** 3).mean()) < 5 * np.sqrt(15 / m)
    assert stats.kstest(v.ravel()[:: 7], "norm").pvalue > 1e-3
    assert np.abs(v).max() > 4.5          # the tails are there (2.56 M draws) ...
    assert np.abs(v).max() < 5.8          # ... up to the 24-bit uniforms' limit, sqrt(2 ln 2^24) = 5.77
    corr = lambda a, b: abs((a * b).mean()) * np.sqrt(a.size)
    assert corr(v[:, :-1], v[:, 1:]) < 5 and corr(v[:-1], v[1:]) < 5 and corr(v[:, 0::2], v[:, 1::2]) < 5
    assert corr(v, z2.double().cpu().numpy()) < 5
    assert corr(v[:, 0] , t - (steps - 1) / 2) / np.sqrt((steps ** 2 - 1) / 12) < 5


def test_model_in_train_mode_draws_fresh_masks():
    fx, nu, ni, *_ = tiny_fixture()
    m = rec_model(dgdata.rec_workload("tiny"), fx).train()
    m.set_embeddings(dev(fx["user_emb"]), dev(fx["item_emb"]))
    args = (dev(fx["cl_ancs"]), dev(fx["cl_poss"]), dev(fx["cl_negs"]), dev(fx["cl_ts_u"]), dev(fx["cl_noise_u"]),
            dev(fx["cl_ts_i"]), dev(fx["cl_noise_i"]))
    with torch.no_grad():
        a, b = m.cal_loss_vector(*args), m.cal_loss_vector(*args)
        m.eval()
        c, d = m.cal_loss_vector(*args), m.cal_loss_vector(*args)
    assert torch.isfinite(a).all() and not torch.equal(a, b)     # two draws
    assert torch.equal(c, d)                                     # no dropout: deterministic
    np.testing.assert_allclose(c.cpu().numpy(), fx["cl_losses"], rtol=1e-5)


def test_denoise_forward_in_train_mode_runs_the_dropout_inside_the_kernel():
    """Denoise.forward(x, t, mess_dropout=True) in train() mode (Model.py:221-224: normalize, Dropout(0.5), first
    Linear): over a given keep mask it equals the oracle's forward of normalize(x) * mask * 2 without its own
    normalisation; without a mask it draws one (two calls differ, eval mode and mess_dropout=False do not drop)."""
    import torch.nn.functional as F
    fx, *_ = tiny_fixture()
    w = weights_from(fx, "usr_")
    dn = dg.Denoise([64, 64], [64, 64], 8, norm=True).to(DEV)
    with torch.no_grad():
        for prm, val in zip((dn.emb_layer.weight, dn.emb_layer.bias, dn.in_layers[0].weight, dn.in_layers[0].bias,
                             dn.out_layers[0].weight, dn.out_layers[0].bias), (w.w_emb, w.b_emb, w.w_in, w.b_in, w.w_out, w.b_out)):
            prm.copy_(val)
    dn.n_steps = 200
    gen = torch.Generator().manual_seed(11)
    for n in (1, 33, 500):
        x, t = torch.randn(n, 64, generator=gen), torch.randint(0, 200, (n,), generator=gen)
        mask = torch.rand(n, 64, generator=gen) >= 0.5
        dn.train()
        got = dn(dev(x), dev(t), True, keep_mask=dev(mask))
        want = O.denoise_forward(w, F.normalize(x) * mask.float() * 2.0, t, norm=False)
        assert_rows_close(got, want, what=f"train-mode Denoise.forward, n={n}")
        plain = O.denoise_forward(w, x, t, norm=True)
        assert_rows_close(dn(dev(x), dev(t), False), plain, what="mess_dropout=False")
        a, b = dn(dev(x), dev(t)), dn(dev(x), dev(t))
        assert torch.isfinite(a).all() and not torch.equal(a, b)
        dn.eval()
        assert_rows_close(dn(dev(x), dev(t)), plain, what="eval mode")
