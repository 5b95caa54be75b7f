"""Pins the CPU oracle (oracle/ref_cpu.py) to golden vectors captured from the reference's
own Model.py (tools/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest
import torch

from diffgraph_amd import data as dgdata
from oracle import ref_cpu as O
from tests.helpers import (NC_TABLE, REC_TABLE, assert_rows_close, edges_t, load, tiny_fixture, weights_from)

T = torch.from_numpy


def test_tables_bit_exact():
    fx = load("rec_tiny.npz")
    tab = O.DiffusionTables(**REC_TABLE)
    for nm in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
               "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
               "posterior_mean_coef1", "posterior_mean_coef2"):
        got = getattr(tab, nm).numpy()
        assert got.dtype == np.float64
        np.testing.assert_allclose(got, fx["tab_" + nm], rtol=4e-16, atol=0, err_msg=nm)  # last-bit host differences
        np.testing.assert_array_equal(got.astype(np.float32), fx["tab_" + nm].astype(np.float32), err_msg=nm)
    nc = load("nc_dblp.npz")
    tab = O.DiffusionTables(**NC_TABLE)
    for nm in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
               "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_allclose(getattr(tab, nm).numpy(), nc["tab_" + nm], rtol=4e-16, atol=0, err_msg=nm)
    # facts the kernels rely on: the last reverse step is out*1 + x*0 in fp32
    tab = O.DiffusionTables(**REC_TABLE)
    assert np.float32(tab.posterior_mean_coef1[0].item()) == np.float32(1.0)
    assert tab.posterior_mean_coef2[0].item() == 0.0


def test_layer_and_forward_tiny():
    fx, nu, ni, target, behs = tiny_fixture()
    ue, ie = T(fx["user_emb"]), T(fx["item_emb"])
    y = O.gcn_layer(target[0], target[1], nu + ni, torch.cat([ue, ie]))
    np.testing.assert_array_equal(y.numpy(), fx["layer_y"])  # same ops, same order: bit-exact
    outs = O.rec_forward(target, behs, ue, ie, 2)
    for o, nm in zip(outs, ("fwd_tu", "fwd_ti", "fwd_hu", "fwd_hi")):
        np.testing.assert_array_equal(o.numpy(), fx[nm], err_msg=nm)
    # edge cases present in the tiny graph: isolated user/item rows give exactly e0 (+0 per layer)
    np.testing.assert_array_equal(outs[0][nu - 1].numpy(), fx["user_emb"][nu - 1])
    np.testing.assert_array_equal(outs[1][ni - 1].numpy(), fx["item_emb"][ni - 1])


def test_forward_retail_rocket_shape():
    fx = load("rec_retail_rocket.npz")
    wl = dgdata.rec_workload("retail_rocket")
    ue = T(dgdata.xavier_uniform(wl.n_user, 64, seed=0))
    ie = T(dgdata.xavier_uniform(wl.n_item, 64, seed=1))
    target = edges_t(*wl.target, wl.n_user)
    behs = [edges_t(*b, wl.n_user) for b in wl.behaviours]
    outs = O.rec_forward(target, behs, ue, ie, 2)
    for o, nm, rows in zip(outs, ("tu", "ti", "hu", "hi"), ("rows_u", "rows_i", "rows_u", "rows_i")):
        assert_rows_close(o[fx[rows]], fx[f"fwd_{nm}_rows"], what=nm)
        colsum = o.double().sum(0).numpy()
        np.testing.assert_allclose(colsum, fx[f"fwd_{nm}_colsum"], rtol=0, atol=1e-5 * fx[f"fwd_{nm}_abssum"].max())


def test_q_sample_and_denoise():
    fx, nu, ni, target, behs = tiny_fixture()
    tab = O.DiffusionTables(**REC_TABLE)
    hu = T(fx["fwd_hu"])
    q = O.q_sample(tab, hu, T(fx["q_t"]), T(fx["q_noise"]))
    np.testing.assert_array_equal(q.numpy(), fx["q_out"])
    w = weights_from(fx, "usr_")
    for tv in (0, 1, 50, 199):
        out = O.denoise_forward(w, hu, torch.full((nu,), tv, dtype=torch.long))
        assert_rows_close(out, fx[f"dn_out_t{tv}"], what=f"denoise t={tv}")
    out = O.denoise_forward(w, hu, T(fx["q_t"]))
    assert_rows_close(out, fx["dn_out_mixed"], what="denoise mixed t")


def test_p_sample():
    fx, nu, ni, target, behs = tiny_fixture()
    tab = O.DiffusionTables(**REC_TABLE)
    hu, hi = T(fx["fwd_hu"]), T(fx["fwd_hi"])
    x0, kept = O.p_sample(tab, weights_from(fx, "usr_"), hu, 0, keep=(199, 100, 1))
    assert_rows_close(x0, fx["ps_usr_x0"], what="p_sample x0")
    for k in (199, 100, 1):
        assert_rows_close(kept[k], fx[f"ps_usr_after{k}"], what=f"p_sample after step {k}")
    assert_rows_close(O.p_sample(tab, weights_from(fx, "item_"), hi, 0), fx["ps_item_x0"], what="p_sample item")
    x0 = O.p_sample(tab, weights_from(fx, "usr_"), hu, 5, noise=T(fx["ps5_noise"]))
    assert_rows_close(x0, fx["ps5_usr_x0"], what="p_sample steps=5")
    # predict() glue: emb + denoised
    assert_rows_close(T(fx["fwd_tu"]) + T(fx["ps_usr_x0"]), fx["predict_u"], what="predict")


def test_training_losses2():
    fx, nu, ni, target, behs = tiny_fixture()
    tab = O.DiffusionTables(**REC_TABLE)
    loss, out = O.training_losses2(tab, weights_from(fx, "usr_"), T(fx["fwd_tu"]), T(fx["fwd_hu"]), T(fx["tl_batch"]),
                                   T(fx["tl_ts"]), T(fx["tl_noise"]))
    assert loss.dtype == torch.float64
    assert_rows_close(out, fx["tl_out"], what="tl out")
    np.testing.assert_allclose(loss.numpy(), fx["tl_loss"], rtol=1e-5)
    # measured loss weights quoted for the reference defaults
    w = tab.loss_weight(torch.tensor([0, 1, 199]))
    assert w[0].item() == 1.0
    assert abs(w[1].item() - 0.04522140) < 1e-7 and abs(w[2].item() - 0.04514052) < 1e-7


def test_cal_loss_and_gradients():
    """a11: the oracle's forward under torch autograd reproduces the reference's loss values and
    its gradients wrt the embeddings and denoiser weights."""
    fx, nu, ni, target, behs = tiny_fixture()
    tab = O.DiffusionTables(**REC_TABLE)
    ue = T(fx["user_emb"]).clone().requires_grad_(True)
    ie = T(fx["item_emb"]).clone().requires_grad_(True)
    wu, wi = weights_from(fx, "usr_"), weights_from(fx, "item_")
    for w in (wu, wi):
        for nm in ("w_emb", "b_emb", "w_in", "b_in", "w_out", "b_out"):
            setattr(w, nm, getattr(w, nm).clone().requires_grad_(True))
    ancs, poss, negs = T(fx["cl_ancs"]), T(fx["cl_poss"]), T(fx["cl_negs"])
    batch, reg = fx["cl_batch_div"]
    loss, bpr, regl, diff = O.rec_cal_loss(tab, wu, wi, target, behs, ue, ie, ancs, poss, negs, T(fx["cl_ts_u"]),
                                           T(fx["cl_noise_u"]), T(fx["cl_ts_i"]), T(fx["cl_noise_i"]), batch, reg)
    np.testing.assert_allclose([loss.item(), bpr.item(), regl.item(), diff.item()], fx["cl_losses"], rtol=1e-6)
    loss.backward()
    assert_rows_close(ue.grad, fx["cl_grad_user"], rtol=1e-4, what="grad user_emb")
    assert_rows_close(ie.grad, fx["cl_grad_item"], rtol=1e-4, what="grad item_emb")
    for pre, w in (("usr_", wu), ("item_", wi)):
        for nm in ("w_emb", "b_emb", "w_in", "b_in", "w_out", "b_out"):
            assert_rows_close(getattr(w, nm).grad, fx["cl_grad_" + pre + nm], rtol=1e-4, what="grad " + pre + nm)


def test_training_mode_dropout_losses_and_gradients():
    """The reference's first epoch (train() mode): Dropout(0.5) after F.normalize in both denoisers.  The oracle with
    the recorded keep masks reproduces training_losses2, cal_loss and the gradients of the reference's own run."""
    fx, nu, ni, target, behs = tiny_fixture()
    tr = load("rec_tiny_train.npz")
    tab = O.DiffusionTables(**REC_TABLE)
    loss, out = O.training_losses2(tab, weights_from(fx, "usr_"), T(fx["fwd_tu"]), T(fx["fwd_hu"]), T(fx["tl_batch"]),
                                   T(tr["tl_ts"]), T(tr["tl_noise"]), T(tr["tl_keep"]))
    assert_rows_close(out, tr["tl_out"], what="tl out (train mode)")
    np.testing.assert_allclose(loss.numpy(), tr["tl_loss"], rtol=1e-5)
    ue = T(fx["user_emb"]).clone().requires_grad_(True)
    ie = T(fx["item_emb"]).clone().requires_grad_(True)
    wu, wi = weights_from(fx, "usr_"), weights_from(fx, "item_")
    for w in (wu, wi):
        for nm in ("w_emb", "b_emb", "w_in", "b_in", "w_out", "b_out"):
            setattr(w, nm, getattr(w, nm).clone().requires_grad_(True))
    ancs, poss, negs = T(fx["cl_ancs"]), T(fx["cl_poss"]), T(fx["cl_negs"])
    batch, reg = fx["cl_batch_div"]
    loss, bpr, regl, diff = O.rec_cal_loss(tab, wu, wi, target, behs, ue, ie, ancs, poss, negs, T(tr["ts_u"]),
                                           T(tr["noise_u"]), T(tr["ts_i"]), T(tr["noise_i"]), batch, reg,
                                           T(tr["keep_u"]), T(tr["keep_i"]))
    np.testing.assert_allclose([loss.item(), bpr.item(), regl.item(), diff.item()], tr["losses"], rtol=1e-6)
    loss.backward()
    assert_rows_close(ue.grad, tr["grad_user"], rtol=1e-4, what="grad user_emb")
    assert_rows_close(ie.grad, tr["grad_item"], rtol=1e-4, what="grad item_emb")
    for pre, w in (("usr_", wu), ("item_", wi)):
        for nm in ("w_emb", "b_emb", "w_in", "b_in", "w_out", "b_out"):
            assert_rows_close(getattr(w, nm).grad, tr["grad_" + pre + nm], rtol=1e-4, what="grad " + pre + nm)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_narrow_latent_space(mode):
    """--latdim 32 (params.py:14): the oracle is shape-generic over the one-layer denoiser, so the same restatement must
    reproduce the reference's run at embeddings [n, 32], in_layers 40->64, out_layers 64->32 (rec_tiny_latdim.npz)."""
    fx = load("rec_tiny_latdim.npz")
    _, nu, ni, target, behs = tiny_fixture()
    tab = O.DiffusionTables(**REC_TABLE)

    def weights(pre):
        return O.DenoiseWeights(*[T(fx[pre + k]).clone().requires_grad_(True)
                                  for k in ("emb_w", "emb_b", "in0_w", "in0_b", "out0_w", "out0_b")])
    ue, ie = T(fx["user_emb"]).clone().requires_grad_(True), T(fx["item_emb"]).clone().requires_grad_(True)
    wu, wi = weights("usr_"), weights("item_")
    if mode == "eval":
        with torch.no_grad():
            tu, ti, hu, hi = O.rec_forward(target, behs, ue, ie, 2)
            assert tuple(hu.shape) == (nu, 32)
            assert_rows_close(O.denoise_forward(wu, hu, T(fx["q_t"])), fx["dn_out_mixed"], what="denoise mixed t")
            assert_rows_close(O.p_sample(tab, wu, hu, 0), fx["ps_usr_x0"], what="p_sample")
            loss, out = O.training_losses2(tab, wu, tu, hu, T(fx["tl_batch"]), T(fx["tl_ts"]), T(fx["tl_noise"]))
            np.testing.assert_allclose(loss.numpy(), fx["tl_loss"], rtol=1e-5)
            assert_rows_close(out, fx["tl_out"], what="tl out")
            assert_rows_close(tu + O.p_sample(tab, wu, hu, 0), fx["predict_u"], what="predict")
    pre = mode + "_"
    keep = (T(fx[pre + "keep_u"]), T(fx[pre + "keep_i"])) if mode == "train" else (None, None)
    batch, reg = fx["cl_batch_div"]
    loss, bpr, regl, diff = O.rec_cal_loss(tab, wu, wi, target, behs, ue, ie, T(fx["cl_ancs"]), T(fx["cl_poss"]), T(fx["cl_negs"]),
                                           T(fx[pre + "ts_u"]), T(fx[pre + "noise_u"]), T(fx[pre + "ts_i"]), T(fx[pre + "noise_i"]),
                                           batch, reg, *keep)
    np.testing.assert_allclose([loss.item(), bpr.item(), regl.item(), diff.item()], fx[pre + "losses"], rtol=1e-6)
    loss.backward()
    assert_rows_close(ue.grad, fx[pre + "grad_user"], rtol=1e-4, what="grad user_emb")
    assert_rows_close(ie.grad, fx[pre + "grad_item"], rtol=1e-4, what="grad item_emb")
    for who, w in (("usr_", wu), ("item_", wi)):
        for attr, nm in (("w_emb", "emb_w"), ("b_emb", "emb_b"), ("w_in", "in0_w"), ("b_in", "in0_b"), ("w_out", "out0_w"),
                         ("b_out", "out0_b")):
            assert_rows_close(getattr(w, attr).grad, fx[f"{pre}grad_{who}{nm}"], rtol=1e-4, what=f"grad {who}{nm}")


def _dblp_graphs():
    import scipy.sparse as sp
    g = load("dblp_graph.npz")
    NA, NP, NC, NT = (int(v) for v in g["counts"])
    mk = lambda e, shape: sp.coo_matrix((np.ones(len(e), dtype=np.float32), (e[:, 0], e[:, 1])), shape=shape).tocsr()
    A_P, P_C, P_T = mk(g["ap"], (NA, NP)), mk(g["pc"], (NP, NC)), mk(g["pt"], (NP, NT))
    binz = lambda m: (m != 0).astype(np.float32)
    apa = binz(A_P @ A_P.T)
    apc, apt = binz(A_P @ P_C), binz(A_P @ P_T)
    feats = sp.csr_matrix((g["feat_data"], g["feat_indices"], g["feat_indptr"]), shape=tuple(g["feat_shape"]))
    rowsum = np.array(feats.sum(1)).flatten()
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    features = torch.from_numpy(np.asarray(sp.diags(r_inv).dot(feats).todense(), dtype=np.float32))
    return NA, apa, binz(apc @ apc.T), binz(apt @ apt.T), features


def _coo_edges(m):
    c = m.tocoo()
    return torch.from_numpy(c.row.astype(np.int64)), torch.from_numpy(c.col.astype(np.int64))


@pytest.mark.parametrize("tag", ["coded", "named"])
def test_nc_forward_dblp(tag):
    fx = load("nc_dblp.npz")
    NA, apa, apcpa, aptpa, features = _dblp_graphs()
    assert (apa.nnz, apcpa.nnz, aptpa.nnz) == (11113, 5000495, 7043571)
    adjs = [_coo_edges(apa)] * 3 if tag == "coded" else [_coo_edges(apa), _coo_edges(apcpa), _coo_edges(aptpa)]
    s1, s2, tgt = O.nc_forward(adjs, features, T(fx["w_t"]), T(fx["b_t"]), 3)
    rows = fx["rows"]
    for nm, o in (("s1", s1), ("s2", s2), ("target", tgt)):
        assert_rows_close(o[rows], fx[f"{tag}_{nm}_rows"], what=f"{tag} {nm}")
    tab = O.DiffusionTables(**NC_TABLE)
    den = O.p_sample(tab, weights_from(fx, "dn_"), s1 + s2, 0)
    emb = tgt + den
    assert_rows_close(emb[rows], fx[f"{tag}_allembeds_rows"], what=f"{tag} allembeds")
    scores = torch.nn.functional.linear(emb, T(fx["w_dense"]), T(fx["b_dense"]))
    assert_rows_close(scores[rows], fx[f"{tag}_scores_rows"], rtol=2e-5, what=f"{tag} scores")


# ------------------------------------------------------------------------------ N1: all-rank evaluation
def test_eval_oracle_matches_reference_test_epoch():
    """oracle.test_epoch against the reference's own Coach.testEpoch + AllRankTestData
    (tools/make_golden.py gen_eval): test-user order, top-20 lists, per-user and epoch metrics."""
    fx = load("rec_eval.npz")
    r = O.test_epoch(T(fx["usr_emb"]), T(fx["item_emb"]), fx["trn_u"], fx["trn_i"], fx["tst_u"], fx["tst_i"],
                     topk=int(fx["topk"][0]), tst_bat=64)
    np.testing.assert_array_equal(r["users"], fx["users"])
    np.testing.assert_array_equal(r["top_locs"], fx["top_locs"])
    np.testing.assert_array_equal(r["recall_per_user"], fx["recall_per_user"])
    np.testing.assert_array_equal(r["ndcg_per_user"], fx["ndcg_per_user"])
    np.testing.assert_array_equal(np.array([r["Recall"], r["NDCG"]]), fx["metrics"])


# ------------------------------------------------------------------------------ N2: meta-path composition
def _sig(m):
    m = m.tocsr()
    m.sort_indices()
    deg = np.diff(m.indptr).astype(np.int64)
    cs = np.add.reduceat(np.append(m.indices.astype(np.int64), 0), m.indptr[:-1]) * (deg > 0)
    cx = np.array([np.bitwise_xor.reduce(m.indices[m.indptr[r]: m.indptr[r + 1]].astype(np.int64)) if deg[r] else 0
                   for r in range(m.shape[0])])
    return deg, cs, cx


def test_metapath_oracle_matches_shipped_matrices():
    """The oracle's composition (DiffGraph_NC/DataHandler.py:147-165) reproduces the author x author matrices the
    reference ships: APA as the as-coded 2-hop product of the mixed relations, APCPA as the chain AP.PC.CP.PA."""
    import scipy.sparse as sp
    g, fx = load("dblp_graph.npz"), load("dblp_metapaths.npz")
    NA = int(g["counts"][0])
    mats = O.dblp_global_mats(g["ap"], g["pc"], g["pt"], g["counts"])
    alpha = np.full((3, 6), 1.0 / 6, dtype=np.float32)          # any positive mixture gives the same pattern
    for h in O.metapath_he_adjs(mats, alpha, alpha, NA):
        deg, cs, cx = _sig(h)
        assert h.nnz == int(fx["apa_nnz"][0])
        np.testing.assert_array_equal(deg, fx["apa_deg"]); np.testing.assert_array_equal(cs, fx["apa_colsum"])
        np.testing.assert_array_equal(cx, fx["apa_colxor"])
    _, _, apcpa, _, _ = _dblp_graphs()
    deg, cs, cx = _sig(apcpa)
    np.testing.assert_array_equal(deg, fx["apcpa_deg"]); np.testing.assert_array_equal(cs, fx["apcpa_colsum"])
    np.testing.assert_array_equal(cx, fx["apcpa_colxor"])
