"""A denoiser outside the fused kernels' shape -- the reference run with --dims '[128, 32]' --d_emb_size 10
(DiffGraph-Rec/params.py:32-33; in_layers 74->32->128, out_layers 128->32->64, Model.py:177-191): Denoise.forward,
training_losses2, p_sample, predict and cal_loss + backward (eval and training mode) against tensors captured from the
reference's own code (tests/golden/rec_tiny_dims.npz, tools/make_golden.py rec_dims).  The Linear layers run in the HIP
dense-layer kernels; the model, the graphs and the embeddings are those of rec_tiny.npz."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import diffgraph_amd as dg
from diffgraph_amd import data as dgdata
from diffgraph_amd.model_rec import DEFAULT_ARGS
from tests.helpers import GRAD_TOL, assert_close_to_max, assert_rows_close, load, tiny_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(x):
    return torch.as_tensor(np.asarray(x)).to(DEV)


def layers_of(dn):
    out = {"emb": dn.emb_layer}
    out.update({f"in{k}": l for k, l in enumerate(dn.in_layers)})
    out.update({f"out{k}": l for k, l in enumerate(dn.out_layers)})
    return out


def model_and_fixture():
    fx = load("rec_tiny_dims.npz")
    base, nu, ni, *_ = tiny_fixture()
    wl = dgdata.rec_workload("tiny")
    graphs = [dg.CSRGraph.from_bipartite(u, i, nu, ni) for (u, i) in [wl.target] + wl.behaviours]
    handler = SimpleNamespace(userNum=nu, itemNum=ni, target_adj=graphs[0], behavior_mats=graphs[1:])
    args = SimpleNamespace(**{**vars(DEFAULT_ARGS), "dims": str([int(v) for v in fx["dims"]]), "d_emb_size": int(fx["d_emb_size"][0]),
                              "batch": int(fx["cl_batch_div"][0]), "reg": float(fx["cl_batch_div"][1])})
    m = dg.rec.HGDM(handler, args).eval()
    assert not m.usr_denoiser.fused and [tuple(l.weight.shape) for l in m.usr_denoiser.in_layers] == [(32, 74), (128, 32)]
    with torch.no_grad():
        for pre, dn in (("usr_", m.usr_denoiser), ("item_", m.item_denoiser)):
            for nm, layer in layers_of(dn).items():
                layer.weight.copy_(dev(fx[f"{pre}{nm}_w"]))
                layer.bias.copy_(dev(fx[f"{pre}{nm}_b"]))
    m.set_embeddings(dev(base["user_emb"]), dev(base["item_emb"]))
    return m, fx, nu, ni


def test_general_denoiser_forward_loss_and_reverse_loop_golden():
    m, fx, nu, ni = model_and_fixture()
    dp = m.diffusion_process
    with torch.no_grad():
        tu, ti, hu, hi = m.forward()
        out = m.usr_denoiser(hu, dev(fx["q_t"]), False)
        assert_rows_close(out, fx["dn_out_mixed"], what="Denoise.forward, mixed timesteps")
        loss, out = dp.training_losses2(m.usr_denoiser, tu, hu, dev(fx["tl_batch"]), ts=dev(fx["tl_ts"]), noise=dev(fx["tl_noise"]))
        assert loss.dtype == torch.float64
        np.testing.assert_allclose(loss.cpu().numpy(), fx["tl_loss"], rtol=1e-5)
        assert_rows_close(out, fx["tl_out"], what="training_losses2 output")
        assert_rows_close(dp.p_sample(m.usr_denoiser, hu, 0), fx["ps_usr_x0"], what="p_sample")
        pu, pi = m.predict()
        assert_rows_close(pu, fx["predict_u"], what="predict user")
        assert_rows_close(pi, fx["predict_i"], what="predict item")


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_general_denoiser_cal_loss_and_gradients_golden(mode):
    """The reference's own loss values and gradients (22 denoiser tensors + both embedding tables); in training mode with
    the two dropout masks torch drew for it."""
    m, fx, nu, ni = model_and_fixture()
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


def test_general_denoiser_trains_through_coach_steps():
    """A few optimizer steps with the general denoisers (the embedding table's HIP Adam, one dg_adam_multi launch for the
    22 denoiser tensors): the loss is finite and every tensor moves."""
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


def test_a_denoiser_of_two_widths_is_refused_with_a_clear_message():
    with pytest.raises(NotImplementedError, match="one width"):
        dg.Denoise([128, 64], [64, 32], 8)


@pytest.mark.parametrize("use_graph", [False, True])
def test_general_denoiser_coach_epochs(use_graph):
    """Coach.run over two epochs with the general denoisers, eagerly and with the training step recorded as a HIP graph."""
    import copy
    from diffgraph_amd.train import Coach
    from tests.test_gpu_train import _handler
    args = copy.copy(DEFAULT_ARGS)
    args.batch, args.dims, args.d_emb_size, args.epoch = 64, "[128, 32]", 10, 2
    torch.manual_seed(5)
    c = Coach(_handler(), args, device=DEV, use_graph=use_graph, seed=3)
    lines = []
    res = c.run(log=lines.append)
    assert not c.model.usr_denoiser.fused
    assert len(lines) == 4 and all("nan" not in ln.lower() for ln in lines), lines
    assert 0.0 <= res["Recall"] <= 1.0 and len(c.metrics["TestRecall"]) == 2


def _nc_model_and_fixture(name):
    from tests.test_oracle_golden import _dblp_graphs
    fx = load(name)
    NA, apa, apcpa, aptpa, features = _dblp_graphs()
    adjs = [dg.CSRGraph.from_scipy(apa) for _ in range(3)]
    args = SimpleNamespace(**{**vars(dg.nc.DEFAULT_ARGS), "dims": str([int(v) for v in fx["dims"]]), "d_emb_size": int(fx["d_emb_size"][0]),
                              "latdim": int(fx["latdim"][0]) if "latdim" in fx.files else 64})
    if "layers" in fx.files:
        args.gcn_layer, args.uugcn_layer = int(fx["layers"][0]), int(fx["layers"][1])
    model = dg.nc.HGDM(features.shape[1], args).train()     # (the reference's NC program never leaves training mode)
    assert model.user_denoise_model.fused == (name == "nc_dblp_layers.npz")
    with torch.no_grad():
        for nm, prm in model.named_parameters():
            prm.copy_(dev(fx["param_" + nm.replace(".", "_")]))
    return model, fx, NA, adjs, features


@pytest.mark.parametrize("name, n_tensors", [("nc_dblp_dims.npz", 14), ("nc_dblp_latdim.npz", 10), ("nc_dblp_layers.npz", 10),
                                             ("nc_dblp_latdim128.npz", 10)])
def test_nc_general_denoiser_golden(name, n_tensors):
    """DiffGraph_NC with --dims '[128, 32]' --d_emb_size 10, with --latdim 32 (64-wide zero-padded tables, as in
    the recommendation model), with --gcn_layer 2 --uugcn_layer 3 (the channels then have different layer counts;
    the denoiser is the fused one) and with --latdim 128 (column blocks through the propagation kernel, diffgraph_amd/wide.py),
    against the reference's own runs on DBLP (as-coded graphs, tests/golden/nc_dblp_dims.npz /
    nc_dblp_latdim.npz): get_allembeds (the reverse loop layer by layer), one training step's two losses and the
    gradients of every parameter tensor, with the timesteps, noise and dropout mask torch drew for it."""
    model, fx, NA, adjs, features = _nc_model_and_fixture(name)
    with torch.no_grad():
        emb, scores = model.get_allembeds(adjs, dev(features))
        s1, s2, tgt = model.forward(adjs, dev(features))
    assert tuple(emb.shape) == tuple(tgt.shape) == (NA, int(fx["latdim"][0]) if "latdim" in fx.files else 64)
    if "rows" in fx.files:      # (the file holds a sample of the rows)
        emb, scores = emb[dev(fx["rows"])], scores[dev(fx["rows"])]
    assert_rows_close(emb, fx["all_embeddings"], rtol=1e-5, what="all_embeddings")
    assert_rows_close(scores, fx["scores"], rtol=1e-5, what="scores")
    onehot = torch.zeros(NA, 4)
    onehot[torch.arange(NA), torch.from_numpy(load("dblp_graph.npz")["labels"]).long()] = 1.0
    nll, diff = model.cal_loss(dev(fx["cl_ancs"]), dev(onehot), adjs, dev(features), dev(fx["ts"]), dev(fx["noise"]), dev(fx["keep"]))
    np.testing.assert_allclose([float(nll.detach()), float(diff.detach())], fx["losses"], rtol=1e-5)
    (nll + diff).backward()
    checked = 0
    for nm, prm in model.named_parameters():
        key = "grad_" + nm.replace(".", "_")
        if key in fx.files:
            assert prm.grad is not None, nm
            assert_close_to_max(prm.grad, fx[key], GRAD_TOL, "grad " + nm)
            checked += 1
    assert checked == n_tensors


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("dims, d_emb_size, latdim", [("[128, 32]", 10, 64), ("[64]", 8, 32), ("[64]", 8, 128)])
def test_nc_general_denoiser_coach_epochs(use_graph, dims, d_emb_size, latdim):
    """NCCoach over two epochs with the general denoiser (another --dims / --d_emb_size; --latdim 32), eagerly and with
    the step recorded as a HIP graph."""
    import copy
    from diffgraph_amd.train_nc import NCCoach
    from tests.test_gpu_train import _nc_handler
    args = copy.copy(dg.nc.DEFAULT_ARGS)
    args.dims, args.d_emb_size, args.latdim, args.batch = dims, d_emb_size, latdim, 8
    c = NCCoach(_nc_handler(), args, device=DEV, use_graph=use_graph, seed=3)
    lines = []
    res = c.run(epochs=2, log=lines.append)
    assert not c.model.user_denoise_model.fused
    assert len(lines) == 2 and all("nan" not in ln.lower() for ln in lines), lines
    assert 0.0 <= res["macro"] <= 1.0
