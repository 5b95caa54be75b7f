"""A latent space narrower or wider than the kernels' rows -- the reference run with --latdim 32 and with --latdim 128
(DiffGraph-Rec/params.py:14; embeddings [n, latdim], in_layers latdim+8 -> 64, out_layers 64 -> latdim,
Model.py:26-52,177-191): Denoise.forward, training_losses2, p_sample, predict and cal_loss + backward (eval and training
mode) against tensors captured from the reference's own code (tests/golden/rec_tiny_latdim.npz / rec_tiny_latdim128.npz,
tools/make_golden.py rec_latdim / rec_latdim128).
At 32 the tables stay 64 floats wide with zeros from column 32 on: the propagation, the loss head and the evaluation run in
the same HIP kernels as at latdim 64, the denoisers layer by layer in the dense-layer kernels; the zero columns must come
back as exact zeros everywhere.  At 128 every 64-column block goes through the propagation kernel and what couples the
blocks is formed by torch ops (diffgraph_amd/wide.py)."""
import copy
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import diffgraph_amd as dg
from diffgraph_amd import data as dgdata
from diffgraph_amd.model_rec import DEFAULT_ARGS
from tests.helpers import GRAD_TOL, assert_close_to_max, assert_rows_close, load, tiny_fixture
from tests.test_gpu_dims import dev, layers_of

pytestmark = pytest.mark.gpu
DEV = "cuda"
L = 32


def model_and_fixture(L=L):
    fx = load("rec_tiny_latdim.npz" if L == 32 else f"rec_tiny_latdim{L}.npz")
    assert int(fx["latdim"][0]) == L
    _, nu, ni, *_ = tiny_fixture()
    wl = dgdata.rec_workload("tiny")
    graphs = [dg.CSRGraph.from_bipartite(u, i, nu, ni) for (u, i) in [wl.target] + wl.behaviours]
    handler = SimpleNamespace(userNum=nu, itemNum=ni, target_adj=graphs[0], behavior_mats=graphs[1:])
    args = SimpleNamespace(**{**vars(DEFAULT_ARGS), "latdim": L, "batch": int(fx["cl_batch_div"][0]), "reg": float(fx["cl_batch_div"][1])})
    m = dg.rec.HGDM(handler, args).eval()
    assert not m.usr_denoiser.fused and m.usr_denoiser.width == L
    assert [tuple(l.weight.shape) for l in m.usr_denoiser.in_layers] == [(64, L + 8)]
    assert [tuple(l.weight.shape) for l in m.usr_denoiser.out_layers] == [(L, 64)]
    assert tuple(m.embedding_table.shape) == (nu + ni, max(L, 64)) and not bool(m.embedding_table[:, L:].any())
    assert tuple(m.embedding_dict["user_emb"].shape) == (nu, L)
    with torch.no_grad():
        for pre, dn in (("usr_", m.usr_denoiser), ("item_", m.item_denoiser)):
            for nm, layer in layers_of(dn).items():
                layer.weight.copy_(dev(fx[f"{pre}{nm}_w"]))
                layer.bias.copy_(dev(fx[f"{pre}{nm}_b"]))
    m.set_embeddings(dev(fx["user_emb"]), dev(fx["item_emb"]))
    return m, fx, nu, ni


def test_narrow_latent_forward_loss_and_reverse_loop_golden():
    m, fx, nu, ni = model_and_fixture()
    dp = m.diffusion_process
    with torch.no_grad():
        tu, ti, hu, hi = m.forward()
        assert tuple(tu.shape) == (nu, L) and tuple(hi.shape) == (ni, L)
        ui, het = m.propagate(m.embedding_table)          # the 64-wide tables behind those views
        assert not bool(ui[:, L:].any()) and not bool(het[:, L:].any())
        out = m.usr_denoiser(hu, dev(fx["q_t"]), False)
        assert_rows_close(out, fx["dn_out_mixed"], what="Denoise.forward, mixed timesteps")
        wide = m.usr_denoiser(het[:nu], dev(fx["q_t"]), False)      # a 64-wide row in, a 64-wide row out
        assert tuple(wide.shape) == (nu, 64) and torch.equal(wide[:, :L], out) and not bool(wide[:, L:].any())
        loss, out = dp.training_losses2(m.usr_denoiser, tu, hu, dev(fx["tl_batch"]), ts=dev(fx["tl_ts"]), noise=dev(fx["tl_noise"]))
        assert loss.dtype == torch.float64
        np.testing.assert_allclose(loss.cpu().numpy(), fx["tl_loss"], rtol=1e-5)
        assert_rows_close(out, fx["tl_out"], what="training_losses2 output")
        assert_rows_close(dp.p_sample(m.usr_denoiser, hu, 0), fx["ps_usr_x0"], what="p_sample")
        pu, pi = m.predict()
        assert_rows_close(pu, fx["predict_u"], what="predict user")
        assert_rows_close(pi, fx["predict_i"], what="predict item")
        # q_sample on narrow rows: the 64-wide kernel, the same numbers as the reference's expression (Model.py:305)
        t = dev(fx["tl_ts"])
        xq = dp.q_sample(hu, t, dev(fx["tl_noise"]))
        sa, sb = dp.sqrt_alphas_cumprod.to(DEV)[t].float()[:, None], dp.sqrt_one_minus_alphas_cumprod.to(DEV)[t].float()[:, None]
        assert tuple(xq.shape) == (nu, L) and torch.equal(xq, sa * hu + sb * dev(fx["tl_noise"]))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_narrow_latent_cal_loss_and_gradients_golden(mode):
    """The reference's own loss values and gradients; the table's gradient is exactly zero from column 32 on."""
    m, fx, nu, ni = model_and_fixture()
    m.train(mode == "train")
    pre = mode + "_"
    keep = dict(keep_u=dev(fx[pre + "keep_u"]), keep_i=dev(fx[pre + "keep_i"])) if mode == "train" else {}
    vals = m.cal_loss(dev(fx["cl_ancs"]), dev(fx["cl_poss"]), dev(fx["cl_negs"]), dev(fx[pre + "ts_u"]), dev(fx[pre + "noise_u"]),
                      dev(fx[pre + "ts_i"]), dev(fx[pre + "noise_i"]), **keep)
    np.testing.assert_allclose([float(v.detach()) for v in vals], fx[pre + "losses"], rtol=1e-5)
    vals[0].backward()
    g = m.embedding_table.grad
    assert tuple(g.shape) == (nu + ni, 64) and not bool(g[:, L:].any())
    assert_rows_close(g[:nu, :L], fx[pre + "grad_user"], rtol=1e-5, what="grad user_emb")
    assert_rows_close(g[nu:, :L], fx[pre + "grad_item"], rtol=1e-5, what="grad item_emb")
    for who, dn in (("usr_", m.usr_denoiser), ("item_", m.item_denoiser)):
        for nm, layer in layers_of(dn).items():
            assert_close_to_max(layer.weight.grad, fx[f"{pre}grad_{who}{nm}_w"], GRAD_TOL, f"grad {who}{nm} weight")
            assert_close_to_max(layer.bias.grad, fx[f"{pre}grad_{who}{nm}_b"], GRAD_TOL, f"grad {who}{nm} bias")


def test_narrow_latent_trains_and_the_zero_columns_stay_zero():
    """Optimizer steps through training_step (the table's Adam step inside the backward pass): every tensor moves, the
    loss is finite, and the table is still zero from column 32 on -- its own draws of noise and masks included."""
    from diffgraph_amd.optim import Adam
    from diffgraph_amd.train import training_step
    m, fx, nu, ni = model_and_fixture()
    m.train()
    opt = Adam(m, lr=1e-3)
    before = [p.detach().clone() for p in m.parameters()]
    seed = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=DEV)
    for _ in range(3):
        res = training_step(m, opt, dev(fx["cl_ancs"]), dev(fx["cl_poss"]), dev(fx["cl_negs"]), seed)
    assert torch.isfinite(res).all()
    assert all(not torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))
    assert not bool(m.embedding_table[:, L:].any())
    assert bool((m.embedding_table[:, :L] != before[0][:, :L]).any())


@pytest.mark.parametrize("use_graph", [False, True])
def test_narrow_latent_coach_epochs(use_graph):
    """Coach.run over two epochs at latdim 32 (training, all-rank evaluation on the padded tables), eagerly and with
    the training step recorded as a HIP graph."""
    from diffgraph_amd.train import Coach
    from tests.test_gpu_train import _handler
    args = copy.copy(DEFAULT_ARGS)
    args.batch, args.latdim, args.epoch = 64, L, 2
    torch.manual_seed(5)
    c = Coach(_handler(), args, device=DEV, use_graph=use_graph, seed=3)
    lines = []
    res = c.run(log=lines.append)
    assert c.model.usr_denoiser.width == L and not bool(c.model.embedding_table[:, L:].any())
    assert len(lines) == 4 and all("nan" not in ln.lower() for ln in lines), lines
    assert 0.0 <= res["Recall"] <= 1.0 and len(c.metrics["TestRecall"]) == 2




# ------------------------------------------------------------------------------------------------ latdim 128
W = 128


def test_wide_latent_forward_loss_and_reverse_loop_golden():
    m, fx, nu, ni = model_and_fixture(W)
    assert m.wide
    dp = m.diffusion_process
    with torch.no_grad():
        tu, ti, hu, hi = m.forward()
        assert tuple(tu.shape) == (nu, W) and tuple(hi.shape) == (ni, W)
        assert_rows_close(m.usr_denoiser(hu, dev(fx["q_t"]), False), fx["dn_out_mixed"], what="Denoise.forward, mixed timesteps")
        loss, out = dp.training_losses2(m.usr_denoiser, tu, hu, dev(fx["tl_batch"]), ts=dev(fx["tl_ts"]), noise=dev(fx["tl_noise"]))
        np.testing.assert_allclose(loss.cpu().numpy(), fx["tl_loss"], rtol=1e-5)
        assert_rows_close(out, fx["tl_out"], what="training_losses2 output")
        assert_rows_close(dp.p_sample(m.usr_denoiser, hu, 0), fx["ps_usr_x0"], what="p_sample")
        pu, pi = m.predict()
        assert_rows_close(pu, fx["predict_u"], what="predict user")
        assert_rows_close(pi, fx["predict_i"], what="predict item")


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_wide_latent_cal_loss_and_gradients_golden(mode):
    m, fx, nu, ni = model_and_fixture(W)
    m.train(mode == "train")
    pre = mode + "_"
    keep = dict(keep_u=dev(fx[pre + "keep_u"]), keep_i=dev(fx[pre + "keep_i"])) if mode == "train" else {}
    vals = m.cal_loss(dev(fx["cl_ancs"]), dev(fx["cl_poss"]), dev(fx["cl_negs"]), dev(fx[pre + "ts_u"]), dev(fx[pre + "noise_u"]),
                      dev(fx[pre + "ts_i"]), dev(fx[pre + "noise_i"]), **keep)
    np.testing.assert_allclose([float(v.detach()) for v in vals], fx[pre + "losses"], rtol=1e-5)
    vals[0].backward()
    g = m.embedding_table.grad
    assert_rows_close(g[:nu], fx[pre + "grad_user"], rtol=1e-5, what="grad user_emb")
    assert_rows_close(g[nu:], fx[pre + "grad_item"], rtol=1e-5, what="grad item_emb")
    for who, dn in (("usr_", m.usr_denoiser), ("item_", m.item_denoiser)):
        for nm, layer in layers_of(dn).items():
            assert_close_to_max(layer.weight.grad, fx[f"{pre}grad_{who}{nm}_w"], GRAD_TOL, f"grad {who}{nm} weight")
            assert_close_to_max(layer.bias.grad, fx[f"{pre}grad_{who}{nm}_b"], GRAD_TOL, f"grad {who}{nm} bias")


@pytest.mark.parametrize("use_graph", [False, True])
def test_wide_latent_coach_epochs(use_graph):
    """Coach.run over two epochs at latdim 128 (training steps, the evaluation through torch's mm + topk over mask rows
    scattered from the training CSR), eagerly and with the training step recorded as a HIP graph."""
    from diffgraph_amd.train import Coach
    from tests.test_gpu_train import _handler
    args = copy.copy(DEFAULT_ARGS)
    args.batch, args.latdim, args.epoch = 64, W, 2
    torch.manual_seed(5)
    c = Coach(_handler(), args, device=DEV, use_graph=use_graph, seed=3)
    lines = []
    res = c.run(log=lines.append)
    assert c.model.wide and tuple(c.model.embedding_table.shape)[1] == W
    assert len(lines) == 4 and all("nan" not in ln.lower() for ln in lines), lines
    assert 0.0 <= res["Recall"] <= 1.0 and len(c.metrics["TestRecall"]) == 2


def test_wide_evaluation_matches_the_scoring_kernel_on_padded_narrow_tables():
    """The torch form of the all-rank top-k (wide tables) against the HIP kernel: the same 64-column tables, once as they
    are and once with 64 zero columns behind them."""
    import scipy.sparse as sp
    from diffgraph_amd import evaluate as ev
    nu, ni, k = 300, 2000, 20
    gen = torch.Generator().manual_seed(3)
    usr, itm = torch.randn(nu, 64, generator=gen).to(DEV), torch.randn(ni, 64, generator=gen).to(DEV)
    trn_u, trn_i = dgdata.synth_interactions(nu, ni, 9000, 3)
    rng = np.random.default_rng(5)
    coo = lambda u, i: sp.coo_matrix((np.ones(len(u), np.float32), (u, i)), shape=(nu, ni))
    data = ev.AllRankTestData(coo(rng.integers(0, nu, nu).astype(np.int32), rng.integers(0, ni, nu).astype(np.int32)), coo(trn_u, trn_i))
    v0, i0 = ev.all_rank_topk(usr, itm, data.test_users_dev, data, k)
    pad = lambda t: torch.nn.functional.pad(t, (0, 64))
    v1, i1 = ev.all_rank_topk(pad(usr), pad(itm), data.test_users_dev, data, k)
    assert i1.dtype == torch.int32 and tuple(i1.shape) == tuple(i0.shape)
    np.testing.assert_allclose(v1.cpu().numpy(), v0.cpu().numpy(), rtol=2e-5, atol=2e-5)
    assert (i0 != i1).float().mean().item() < 1e-3
