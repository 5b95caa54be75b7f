#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the reference's OWN Model.py.

Run in the build container only (it needs /root/reference, which never travels):

    python tools/make_golden.py            # both programs, one subprocess each
    python tools/make_golden.py rec|nc|eval     # one of them, in this process

How the reference is run.  Its Model.py files import `dgl` (not installed, no network) for
exactly three things: dgl.function.copy_u / dgl.function.sum message-passing tags, a graph
object offering local_scope / out_degrees / in_degrees / ndata / update_all, and (NC only,
imported but never called) dgl.nn.pytorch.GraphConv.  An in-memory stand-in for those names
is installed in sys.modules before the import; update_all(copy_u, sum) is
zeros.index_add_(0, dst, x[src]) -- DGL's documented semantics.  Everything else
(LeakyReLU, normalize, the diffusion tables, q_sample, Denoise, p_sample, training_losses2,
cal_loss + backward) is the reference's code, unmodified, on torch CPU.  `params.py` parses
sys.argv at import, so argv is reset first.  Only OUTPUT tensors (and the synthetic inputs
that produced them) are written; no reference source is copied.
"""
from __future__ import annotations

import contextlib
import os
import subprocess
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


# --------------------------------------------------------------------------------------
# dgl stand-in
# --------------------------------------------------------------------------------------
class FakeGraph:
    def __init__(self, src, dst, num_nodes):
        self.src = torch.as_tensor(np.asarray(src), dtype=torch.long)
        self.dst = torch.as_tensor(np.asarray(dst), dtype=torch.long)
        self.n = int(num_nodes)
        self.ndata = {}

    @contextlib.contextmanager
    def local_scope(self):
        saved = dict(self.ndata)
        try:
            yield
        finally:
            self.ndata = saved

    def out_degrees(self):
        return torch.bincount(self.src, minlength=self.n)

    def in_degrees(self):
        return torch.bincount(self.dst, minlength=self.n)

    def update_all(self, message_func, reduce_func):
        assert message_func[0] == "copy_u" and reduce_func[0] == "sum"
        x = self.ndata[message_func[1]]
        self.ndata[reduce_func[2]] = torch.zeros_like(x).index_add_(0, self.dst, x[self.src])


def install_dgl_stub():
    dgl = types.ModuleType("dgl")
    fn = types.ModuleType("dgl.function")
    fn.copy_u = lambda u, out: ("copy_u", u, out)
    fn.sum = lambda msg, out: ("sum", msg, out)
    nn_mod = types.ModuleType("dgl.nn")
    nn_pt = types.ModuleType("dgl.nn.pytorch")
    nn_pt.GraphConv = object
    dgl.function, dgl.nn, nn_mod.pytorch = fn, nn_mod, nn_pt
    dgl.graph = lambda data, idtype=None, num_nodes=None: FakeGraph(data[0], data[1], num_nodes)
    sys.modules.update({"dgl": dgl, "dgl.function": fn, "dgl.nn": nn_mod, "dgl.nn.pytorch": nn_pt})


def import_reference(subdir):
    install_dgl_stub()
    sys.argv = ["make_golden"]
    sys.path.insert(0, os.path.join(REF, subdir))
    import Model  # noqa: the reference's module
    return Model


def np64(t):
    return t.detach().cpu().numpy()


class Recorder:
    """Captures the tensors torch.randint / torch.randn_like hand to the reference."""

    def __init__(self):
        self.randint, self.randn, self.dropout = [], [], []

    def __enter__(self):
        self._ri, self._rn = torch.randint, torch.randn_like
        self._dr = torch.nn.functional.dropout

        def dr(input, p=0.5, training=True, inplace=False):
            # torch's own dropout; the keep mask is read off its output (no input element is exactly zero)
            out = self._dr(input, p, training, False)
            if training and p > 0:
                assert bool((input != 0).all())
                self.dropout.append((out != 0).clone())
            return out
        torch.nn.functional.dropout = dr

        def ri(*a, **k):
            out = self._ri(*a, **k)
            self.randint.append(out.clone())
            return out

        def rn(*a, **k):
            out = self._rn(*a, **k)
            self.randn.append(out.clone())
            return out

        torch.randint, torch.randn_like = ri, rn
        return self

    def __exit__(self, *exc):
        torch.randint, torch.randn_like = self._ri, self._rn
        torch.nn.functional.dropout = self._dr


# --------------------------------------------------------------------------------------
# Rec
# --------------------------------------------------------------------------------------
def denoiser_state(dn, prefix):
    return {prefix + "w_emb": np64(dn.emb_layer.weight), prefix + "b_emb": np64(dn.emb_layer.bias),
            prefix + "w_in": np64(dn.in_layers[0].weight), prefix + "b_in": np64(dn.in_layers[0].bias),
            prefix + "w_out": np64(dn.out_layers[0].weight), prefix + "b_out": np64(dn.out_layers[0].bias)}


def build_rec_model(Model, wl, seed, latdim=64):
    from diffgraph_amd import data as dgdata
    nu, ni = wl.n_user, wl.n_item
    mk = lambda ui: FakeGraph(*dgdata.bipartite_edges(ui[0], ui[1], nu), nu + ni)
    handler = types.SimpleNamespace(userNum=nu, itemNum=ni, target_adj=mk(wl.target),
                                    behavior_mats=[mk(b) for b in wl.behaviours])
    torch.manual_seed(seed)
    model = Model.HGDM(handler)
    model.eval()  # dropout off: parity runs are deterministic
    ue = torch.from_numpy(dgdata.xavier_uniform(nu, latdim, seed=0))
    ie = torch.from_numpy(dgdata.xavier_uniform(ni, latdim, seed=1))
    with torch.no_grad():
        model.embedding_dict["user_emb"].copy_(ue)
        model.embedding_dict["item_emb"].copy_(ie)
    return model, handler


def gen_rec():
    Model = import_reference("DiffGraph-Rec")
    from diffgraph_amd import data as dgdata

    # ---------- tiny graph: full tensors for every function on the path ----------
    wl = dgdata.rec_workload("tiny")
    model, handler = build_rec_model(Model, wl, seed=7)
    nu, ni = wl.n_user, wl.n_item
    fx = {"n_user": nu, "n_item": ni,
          "target_u": wl.target[0], "target_i": wl.target[1]}
    for k, b in enumerate(wl.behaviours):
        fx[f"beh{k}_u"], fx[f"beh{k}_i"] = b
    ue, ie = model.embedding_dict["user_emb"], model.embedding_dict["item_emb"]
    fx["user_emb"], fx["item_emb"] = np64(ue), np64(ie)
    with torch.no_grad():
        fx["layer_y"] = np64(model.layers[0](handler.target_adj, ue, ie))                 # a1
        tu, ti, hu, hi = model.forward()                                                    # a2
        fx["fwd_tu"], fx["fwd_ti"], fx["fwd_hu"], fx["fwd_hi"] = map(np64, (tu, ti, hu, hi))
        dp = model.diffusion_process
        for nm in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",        # a5
                   "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                   "posterior_mean_coef1", "posterior_mean_coef2"):
            fx["tab_" + nm] = np64(getattr(dp, nm))
        fx.update(denoiser_state(model.usr_denoiser, "usr_"))
        fx.update(denoiser_state(model.item_denoiser, "item_"))
        g = torch.Generator().manual_seed(11)
        t_q = torch.randint(0, 200, (nu,), generator=g)
        t_q[:4] = torch.tensor([0, 1, 199, 100])
        noise_q = torch.randn(nu, 64, generator=g)
        fx["q_t"], fx["q_noise"] = np64(t_q), np64(noise_q)
        fx["q_out"] = np64(dp.q_sample(hu, t_q, noise_q))                                   # a6
        for tv in (0, 1, 50, 199):                                                          # a7
            fx[f"dn_out_t{tv}"] = np64(model.usr_denoiser(hu, torch.full((nu,), tv), False))
        fx["dn_out_mixed"] = np64(model.usr_denoiser(hu, t_q, False))
        # a8: p_sample(steps=0) with intermediates, by wrapping p_mean_variance
        kept = {}
        orig = dp.p_mean_variance

        def spy(m, x, t):
            mean, lv = orig(m, x, t)
            if int(t[0]) in (199, 100, 1):
                kept[int(t[0])] = mean.clone()
            return mean, lv

        dp.p_mean_variance = spy
        fx["ps_usr_x0"] = np64(dp.p_sample(model.usr_denoiser, hu, 0))
        for k, v in kept.items():
            fx[f"ps_usr_after{k}"] = np64(v)
        dp.p_mean_variance = orig
        fx["ps_item_x0"] = np64(dp.p_sample(model.item_denoiser, hi, 0))
        # a8 with sampling_steps > 0: the start point is q_sample(x, steps-1, fresh noise)
        with Recorder() as rec:
            torch.manual_seed(5)
            fx["ps5_usr_x0"] = np64(dp.p_sample(model.usr_denoiser, hu, 5))
        fx["ps5_noise"] = np64(rec.randn[0])
        # a9: training_losses2 with recorded draws
        batch = torch.tensor([0, 3, 3, 17, nu - 1, 5, 9, 9], dtype=torch.long)
        with Recorder() as rec:
            torch.manual_seed(13)
            loss, out = dp.training_losses2(model.usr_denoiser, tu, hu, batch)
        fx["tl_ts"], fx["tl_noise"], fx["tl_batch"] = np64(rec.randint[0]), np64(rec.randn[0]), np64(batch)
        fx["tl_loss"], fx["tl_out"] = np64(loss), np64(out)
        fx["predict_u"], fx["predict_i"] = map(np64, model.predict())                       # a10 glue
    # a11: cal_loss + backward
    g = torch.Generator().manual_seed(17)
    B = 64
    ancs = torch.randint(0, nu, (B,), generator=g)
    poss = torch.randint(0, ni, (B,), generator=g)
    negs = torch.randint(0, ni, (B,), generator=g)
    model.zero_grad()
    with Recorder() as rec:
        torch.manual_seed(19)
        loss, bpr, reg, diff = model.cal_loss(ancs, poss, negs)
    loss.backward()
    fx["cl_ancs"], fx["cl_poss"], fx["cl_negs"] = np64(ancs), np64(poss), np64(negs)
    fx["cl_ts_u"], fx["cl_ts_i"] = np64(rec.randint[0]), np64(rec.randint[1])
    fx["cl_noise_u"], fx["cl_noise_i"] = np64(rec.randn[0]), np64(rec.randn[1])
    fx["cl_losses"] = np.array([loss.item(), bpr.item(), reg.item(), diff.item()], dtype=np.float64)
    fx["cl_batch_div"] = np.array([Model.args.batch, Model.args.reg], dtype=np.float64)
    fx["cl_grad_user"], fx["cl_grad_item"] = np64(ue.grad), np64(ie.grad)
    for pre, dn in (("usr_", model.usr_denoiser), ("item_", model.item_denoiser)):
        fx["cl_grad_" + pre + "w_emb"] = np64(dn.emb_layer.weight.grad)
        fx["cl_grad_" + pre + "b_emb"] = np64(dn.emb_layer.bias.grad)
        fx["cl_grad_" + pre + "w_in"] = np64(dn.in_layers[0].weight.grad)
        fx["cl_grad_" + pre + "b_in"] = np64(dn.in_layers[0].bias.grad)
        fx["cl_grad_" + pre + "w_out"] = np64(dn.out_layers[0].weight.grad)
        fx["cl_grad_" + pre + "b_out"] = np64(dn.out_layers[0].bias.grad)
    np.savez_compressed(os.path.join(OUT, "rec_tiny.npz"), **fx)
    print("rec_tiny.npz:", len(fx), "arrays")

    # ---------- Retail-Rocket-shaped graph: sampled rows + fp64 column sums ----------
    wl = dgdata.rec_workload("retail_rocket")
    model, handler = build_rec_model(Model, wl, seed=7)
    rows_u = np.sort(np.random.default_rng(3).choice(wl.n_user, 256, replace=False))
    rows_i = np.sort(np.random.default_rng(4).choice(wl.n_item, 256, replace=False))
    fx = {"rows_u": rows_u, "rows_i": rows_i}
    with torch.no_grad():
        outs = model.forward()
        for nm, o, rows in zip(("tu", "ti", "hu", "hi"), outs, (rows_u, rows_i, rows_u, rows_i)):
            fx["fwd_" + nm + "_rows"] = np64(o)[rows]
            fx["fwd_" + nm + "_colsum"] = np64(o).astype(np.float64).sum(0)
            fx["fwd_" + nm + "_abssum"] = np.abs(np64(o).astype(np.float64)).sum(0)
        fx.update(denoiser_state(model.usr_denoiser, "usr_"))
        fx.update(denoiser_state(model.item_denoiser, "item_"))
        pu, pi = model.predict()
        fx["predict_u_rows"], fx["predict_i_rows"] = np64(pu)[rows_u], np64(pi)[rows_i]
        fx["predict_u_colsum"] = np64(pu).astype(np.float64).sum(0)
        fx["predict_i_colsum"] = np64(pi).astype(np.float64).sum(0)
    np.savez_compressed(os.path.join(OUT, "rec_retail_rocket.npz"), **fx)
    print("rec_retail_rocket.npz:", len(fx), "arrays")


def gen_rec_train():
    """cal_loss + backward of the tiny Rec model in TRAINING mode -- what the reference's first epoch runs
    (main.py:144 is its only eval()): nn.Dropout(0.5) between F.normalize and the first Linear of both denoisers
    (Model.py:193,221-224).  Same model and inputs as the eval-mode `cl_*` entries of rec_tiny.npz; the two keep
    masks torch drew are recorded beside the timesteps and the noise."""
    Model = import_reference("DiffGraph-Rec")
    from diffgraph_amd import data as dgdata
    wl = dgdata.rec_workload("tiny")
    model, handler = build_rec_model(Model, wl, seed=7)
    nu, ni = wl.n_user, wl.n_item
    ref = np.load(os.path.join(OUT, "rec_tiny.npz"))
    for k, v in denoiser_state(model.usr_denoiser, "usr_").items():
        assert np.array_equal(v, ref[k]), k          # the same model as rec_tiny.npz describes
    model.train()
    ancs, poss, negs = (torch.from_numpy(ref[k]) for k in ("cl_ancs", "cl_poss", "cl_negs"))
    model.zero_grad()
    with Recorder() as rec:
        torch.manual_seed(37)
        loss, bpr, reg, diff = model.cal_loss(ancs, poss, negs)
    loss.backward()
    fx = {"ts_u": np64(rec.randint[0]), "ts_i": np64(rec.randint[1]), "noise_u": np64(rec.randn[0]),
          "noise_i": np64(rec.randn[1]), "keep_u": np64(rec.dropout[0]).astype(np.uint8),
          "keep_i": np64(rec.dropout[1]).astype(np.uint8),
          "losses": np.array([loss.item(), bpr.item(), reg.item(), diff.item()], dtype=np.float64)}
    assert fx["keep_u"].shape == (nu, 64) and fx["keep_i"].shape == (ni, 64) and len(rec.dropout) == 2
    ue, ie = model.embedding_dict["user_emb"], model.embedding_dict["item_emb"]
    fx["grad_user"], fx["grad_item"] = np64(ue.grad), np64(ie.grad)
    for pre, dn in (("usr_", model.usr_denoiser), ("item_", model.item_denoiser)):
        for nm, q in (("w_emb", dn.emb_layer.weight), ("b_emb", dn.emb_layer.bias), ("w_in", dn.in_layers[0].weight),
                      ("b_in", dn.in_layers[0].bias), ("w_out", dn.out_layers[0].weight), ("b_out", dn.out_layers[0].bias)):
            fx["grad_" + pre + nm] = np64(q.grad)
    # training_losses2 alone, in training mode (forward values)
    with torch.no_grad():
        tu, ti, hu, hi = model.forward()
        batch = torch.from_numpy(ref["tl_batch"])
        with Recorder() as rec:
            torch.manual_seed(41)
            l2, out = model.diffusion_process.training_losses2(model.usr_denoiser, tu, hu, batch)
        fx["tl_ts"], fx["tl_noise"], fx["tl_keep"] = np64(rec.randint[0]), np64(rec.randn[0]), np64(rec.dropout[0]).astype(np.uint8)
        fx["tl_loss"], fx["tl_out"] = np64(l2), np64(out)
    np.savez_compressed(os.path.join(OUT, "rec_tiny_train.npz"), **fx)
    print("rec_tiny_train.npz:", len(fx), "arrays; keep fractions", fx["keep_u"].mean(), fx["keep_i"].mean())


def denoiser_layers(dn, prefix):
    """Every tensor of a Denoise with any number of layers: emb, in0.., out0.."""
    out = {prefix + "emb_w": np64(dn.emb_layer.weight), prefix + "emb_b": np64(dn.emb_layer.bias)}
    for kind, layers in (("in", dn.in_layers), ("out", dn.out_layers)):
        for k, layer in enumerate(layers):
            out[f"{prefix}{kind}{k}_w"], out[f"{prefix}{kind}{k}_b"] = np64(layer.weight), np64(layer.bias)
    return out


def gen_rec_dims():
    """The tiny Rec model with a second denoiser configuration, --dims '[128, 32]' --d_emb_size 10 (in_layers
    74->32->128, out_layers 128->32->64; params.py:32-33, Model.py:177-191,226-232): Denoise.forward, training_losses2,
    p_sample, predict, and cal_loss + backward in eval and in training mode, with the draws torch made recorded."""
    _gen_rec_variant("rec_tiny_dims.npz", "[128, 32]", 10, 64, [(32, 74), (128, 32)], [(32, 128), (64, 32)])


def gen_rec_latdim():
    """... and with a narrower latent space, --latdim 32 (params.py:14; embeddings [n, 32], in_layers 40->64,
    out_layers 64->32): the same calls; the embeddings are the first 32 columns' worth of a fresh xavier draw."""
    _gen_rec_variant("rec_tiny_latdim.npz", "[64]", 8, 32, [(64, 40)], [(32, 64)])


def gen_rec_latdim128():
    """... and with a wider one, --latdim 128 (embeddings [n, 128], in_layers 136->64, out_layers 64->128)."""
    _gen_rec_variant("rec_tiny_latdim128.npz", "[64]", 8, 128, [(64, 136)], [(128, 64)])


def _gen_rec_variant(out_name, dims, d_emb_size, latdim, want_in, want_out):
    Model = import_reference("DiffGraph-Rec")
    from diffgraph_amd import data as dgdata
    Model.args.dims, Model.args.d_emb_size, Model.args.latdim = dims, d_emb_size, latdim
    wl = dgdata.rec_workload("tiny")
    model, handler = build_rec_model(Model, wl, seed=7, latdim=latdim)
    nu, ni = wl.n_user, wl.n_item
    assert [tuple(l.weight.shape) for l in model.usr_denoiser.in_layers] == want_in
    assert [tuple(l.weight.shape) for l in model.usr_denoiser.out_layers] == want_out
    fx = {"dims": np.array(eval(dims)), "d_emb_size": np.array([d_emb_size]), "latdim": np.array([latdim])}
    fx["user_emb"], fx["item_emb"] = np64(model.embedding_dict["user_emb"]), np64(model.embedding_dict["item_emb"])
    fx.update(denoiser_layers(model.usr_denoiser, "usr_"))
    fx.update(denoiser_layers(model.item_denoiser, "item_"))
    dp = model.diffusion_process
    with torch.no_grad():
        tu, ti, hu, hi = model.forward()
        g = torch.Generator().manual_seed(11)
        t_q = torch.randint(0, 200, (nu,), generator=g)
        t_q[:4] = torch.tensor([0, 1, 199, 100])
        fx["q_t"] = np64(t_q)
        fx["dn_out_mixed"] = np64(model.usr_denoiser(hu, t_q, False))
        fx["ps_usr_x0"] = np64(dp.p_sample(model.usr_denoiser, hu, 0))
        batch = torch.tensor([0, 3, 3, 17, nu - 1, 5, 9, 9], dtype=torch.long)
        with Recorder() as rec:
            torch.manual_seed(13)
            loss, out = dp.training_losses2(model.usr_denoiser, tu, hu, batch)
        fx["tl_ts"], fx["tl_noise"], fx["tl_batch"] = np64(rec.randint[0]), np64(rec.randn[0]), np64(batch)
        fx["tl_loss"], fx["tl_out"] = np64(loss), np64(out)
        fx["predict_u"], fx["predict_i"] = map(np64, model.predict())
    g = torch.Generator().manual_seed(17)
    B = 64
    ancs, poss, negs = torch.randint(0, nu, (B,), generator=g), torch.randint(0, ni, (B,), generator=g), torch.randint(0, ni, (B,), generator=g)
    fx["cl_ancs"], fx["cl_poss"], fx["cl_negs"] = np64(ancs), np64(poss), np64(negs)
    ue, ie = model.embedding_dict["user_emb"], model.embedding_dict["item_emb"]
    for mode, seed in (("eval", 19), ("train", 37)):
        model.train(mode == "train")
        model.zero_grad()
        with Recorder() as rec:
            torch.manual_seed(seed)
            loss, bpr, reg, diff = model.cal_loss(ancs, poss, negs)
        loss.backward()
        pre = mode + "_"
        fx[pre + "ts_u"], fx[pre + "ts_i"] = np64(rec.randint[0]), np64(rec.randint[1])
        fx[pre + "noise_u"], fx[pre + "noise_i"] = np64(rec.randn[0]), np64(rec.randn[1])
        if mode == "train":
            assert len(rec.dropout) == 2
            fx[pre + "keep_u"], fx[pre + "keep_i"] = (np64(m).astype(np.uint8) for m in rec.dropout)
        fx[pre + "losses"] = np.array([loss.item(), bpr.item(), reg.item(), diff.item()], dtype=np.float64)
        fx[pre + "grad_user"], fx[pre + "grad_item"] = np64(ue.grad), np64(ie.grad)
        for who, dn in (("usr_", model.usr_denoiser), ("item_", model.item_denoiser)):
            grads = {"emb_w": dn.emb_layer.weight.grad, "emb_b": dn.emb_layer.bias.grad}
            for kind, layers in (("in", dn.in_layers), ("out", dn.out_layers)):
                for k, layer in enumerate(layers):
                    grads[f"{kind}{k}_w"], grads[f"{kind}{k}_b"] = layer.weight.grad, layer.bias.grad
            for nm, q in grads.items():
                fx[pre + "grad_" + who + nm] = np64(q)
    fx["cl_batch_div"] = np.array([Model.args.batch, Model.args.reg], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, out_name), **fx)
    print(out_name + ":", len(fx), "arrays")


# --------------------------------------------------------------------------------------
# NC (DBLP)
# --------------------------------------------------------------------------------------
def load_dblp():
    """Restates the ~30 data-loading lines of DiffGraph_NC/DataHandler.py:208-259 that the
    model needs (features, six raw relations); sklearn's removed OneHotEncoder(sparse=...)
    argument keeps the reference's own loader from running."""
    import scipy.sparse as sp
    d = os.path.join(REF, "DiffGraph_NC", "data", "DBLP")
    feats = sp.load_npz(os.path.join(d, "a_feat.npz")).astype("float32")
    edges = {k: np.loadtxt(os.path.join(d, k + ".txt"), dtype=int) for k in ("ap", "pa", "cp", "pc", "pt", "tp")}
    labels = np.load(os.path.join(d, "labels.npy"))
    return feats, edges, labels


def gen_nc():
    Model = import_reference("DiffGraph_NC")
    import scipy.sparse as sp
    feats, edges, labels = load_dblp()
    NA = feats.shape[0]
    ap, pa, cp, pc, pt, tp = (edges[k] for k in ("ap", "pa", "cp", "pc", "pt", "tp"))
    NP = max(ap[:, 1].max(), pa[:, 0].max()) + 1
    NC = max(cp[:, 0].max(), pc[:, 1].max()) + 1
    NT = max(pt[:, 1].max(), tp[:, 0].max()) + 1
    # graph inputs shipped to the GPU box as plain arrays (public DBLP edge lists / features)
    fcsr = feats.tocsr()
    np.savez_compressed(os.path.join(OUT, "dblp_graph.npz"), ap=ap.astype(np.int32), pc=pc.astype(np.int32),
                        pt=pt.astype(np.int32), feat_data=fcsr.data.astype(np.float32),
                        feat_indices=fcsr.indices.astype(np.int32), feat_indptr=fcsr.indptr.astype(np.int32),
                        feat_shape=np.array(fcsr.shape), labels=labels.astype(np.int64),
                        counts=np.array([NA, NP, NC, NT]))
    # row-normalised features (preprocess_features, DataHandler.py:254-259)
    rowsum = np.array(feats.sum(1)).flatten()
    r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    features = torch.FloatTensor(np.asarray(sp.diags(r_inv).dot(feats).todense()))

    A_P = sp.coo_matrix((np.ones(len(ap)), (ap[:, 0], ap[:, 1])), shape=(NA, NP)).tocsr()
    P_C = sp.coo_matrix((np.ones(len(pc)), (pc[:, 0], pc[:, 1])), shape=(NP, NC)).tocsr()
    P_T = sp.coo_matrix((np.ones(len(pt)), (pt[:, 0], pt[:, 1])), shape=(NP, NT)).tocsr()
    binz = lambda m: (m != 0).astype(np.float32)
    apa = binz(A_P @ A_P.T)
    apc = binz(A_P @ P_C)
    apcpa = binz(apc @ apc.T)
    apt = binz(A_P @ P_T)
    aptpa = binz(apt @ apt.T)
    print("APA/APCPA/APTPA nnz:", apa.nnz, apcpa.nnz, aptpa.nnz)

    def fg(m):
        coo = m.tocoo()
        return FakeGraph(coo.row, coo.col, NA)

    torch.manual_seed(23)
    model = Model.HGDM(features.shape[1])
    model.eval()
    rows = np.sort(np.random.default_rng(6).choice(NA, 256, replace=False))
    fx = {"rows": rows, "w_t": np64(model.transform_layer.weight), "b_t": np64(model.transform_layer.bias),
          "w_dense": np64(model.dense.weight), "b_dense": np64(model.dense.bias)}
    fx.update(denoiser_state(model.user_denoise_model, "dn_"))
    dm = model.diffusion_model
    for nm in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
               "posterior_mean_coef1", "posterior_mean_coef2"):
        fx["tab_" + nm] = np64(getattr(dm, nm))
    with torch.no_grad():
        # "as coded": the three channels are the identical APA graph (all softmax weights > 0)
        for tag, adjs in (("coded", [fg(apa), fg(apa), fg(apa)]), ("named", [fg(apa), fg(apcpa), fg(aptpa)])):
            s1, s2, tgt = model.forward(adjs, features)
            for nm, o in (("s1", s1), ("s2", s2), ("target", tgt)):
                fx[f"{tag}_{nm}_rows"] = np64(o)[rows]
                fx[f"{tag}_{nm}_colsum"] = np64(o).astype(np.float64).sum(0)
            emb, scores = model.get_allembeds(adjs, features)
            fx[f"{tag}_allembeds_rows"] = np64(emb)[rows]
            fx[f"{tag}_scores_rows"] = np64(scores)[rows]
            fx[f"{tag}_allembeds_colsum"] = np64(emb).astype(np.float64).sum(0)
        # cal_loss forward values with recorded draws (as-coded graphs)
        adjs = [fg(apa), fg(apa), fg(apa)]
        onehot = torch.zeros(NA, 4)
        onehot[torch.arange(NA), torch.from_numpy(labels).long()] = 1.0
        ancs = torch.tensor([3, 17, 200, 1024, 4056, 5, 8, 13])
        with Recorder() as rec:
            torch.manual_seed(29)
            nll, diff = model.cal_loss(ancs, onehot, adjs, features)
        fx["cl_ancs"], fx["cl_ts"], fx["cl_noise"] = np64(ancs), np64(rec.randint[0]), np64(rec.randn[0])
        fx["cl_losses"] = np.array([nll.item(), diff.item()], dtype=np.float64)
    # one training step's gradients (main.py:378-386: loss = nll + diff_loss; backward) on the as-named
    # graphs, dropout off, same recorded draws
    adjs = [fg(apa), fg(apcpa), fg(aptpa)]
    model.zero_grad()
    with Recorder() as rec:
        torch.manual_seed(31)
        nll, diff = model.cal_loss(ancs, onehot, adjs, features)
    (nll + diff).backward()
    fx["tr_ts"], fx["tr_noise"] = np64(rec.randint[0]), np64(rec.randn[0])
    fx["tr_losses"] = np.array([nll.item(), diff.item()], dtype=np.float64)
    for nm, p in model.named_parameters():
        if p.grad is not None:
            fx["tr_grad_" + nm.replace(".", "_")] = np64(p.grad)
    np.savez_compressed(os.path.join(OUT, "nc_dblp.npz"), **fx)
    print("nc_dblp.npz:", len(fx), "arrays")


def gen_nc_train():
    """One NC training step in TRAINING mode on the as-coded graphs (3 x APA).  The reference's NC program never
    calls eval(), so every one of its training steps runs the denoiser's Dropout(0.5) (DiffGraph_NC/Model.py:262-265;
    main.py:378-386: loss = nll + diff_loss; backward).  Same model as nc_dblp.npz; the keep mask is recorded."""
    Model = import_reference("DiffGraph_NC")
    import scipy.sparse as sp
    feats, edges, labels = load_dblp()
    NA = feats.shape[0]
    ap = edges["ap"]
    NP = max(ap[:, 1].max(), edges["pa"][:, 0].max()) + 1
    rowsum = np.array(feats.sum(1)).flatten()
    r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    features = torch.FloatTensor(np.asarray(sp.diags(r_inv).dot(feats).todense()))
    A_P = sp.coo_matrix((np.ones(len(ap)), (ap[:, 0], ap[:, 1])), shape=(NA, NP)).tocsr()
    apa = ((A_P @ A_P.T) != 0).astype(np.float32).tocoo()
    adjs = [FakeGraph(apa.row, apa.col, NA) for _ in range(3)]
    torch.manual_seed(23)
    model = Model.HGDM(features.shape[1])
    ref = np.load(os.path.join(OUT, "nc_dblp.npz"))
    assert np.array_equal(np64(model.transform_layer.weight), ref["w_t"])      # the model nc_dblp.npz describes
    assert np.array_equal(np64(model.user_denoise_model.in_layers[0].weight), ref["dn_w_in"])
    model.train()
    onehot = torch.zeros(NA, 4)
    onehot[torch.arange(NA), torch.from_numpy(labels).long()] = 1.0
    ancs = torch.from_numpy(ref["cl_ancs"])
    model.zero_grad()
    with Recorder() as rec:
        torch.manual_seed(43)
        nll, diff = model.cal_loss(ancs, onehot, adjs, features)
    (nll + diff).backward()
    fx = {"ts": np64(rec.randint[0]), "noise": np64(rec.randn[0]), "keep": np64(rec.dropout[0]).astype(np.uint8),
          "losses": np.array([nll.item(), diff.item()], dtype=np.float64)}
    assert len(rec.dropout) == 1 and fx["keep"].shape == (NA, 64)
    for nm, q in model.named_parameters():
        if q.grad is not None:
            fx["grad_" + nm.replace(".", "_")] = np64(q.grad)
    np.savez_compressed(os.path.join(OUT, "nc_dblp_train.npz"), **fx)
    print("nc_dblp_train.npz:", len(fx), "arrays; keep fraction", fx["keep"].mean())


def gen_nc_dims():
    """The NC model with a second denoiser configuration, --dims '[128, 32]' --d_emb_size 10 (DiffGraph_NC/params.py;
    in_layers 74->32->128, out_layers 128->32->64), on the as-coded graphs (3 x APA): one training step in training mode
    (losses and every parameter's gradient, the draws recorded) and get_allembeds (the reverse loop, dropout off)."""
    _gen_nc_variant("nc_dblp_dims.npz", "[128, 32]", 10, 64, [(32, 74), (128, 32)], [(32, 128), (64, 32)])


def gen_nc_latdim():
    """... and with --latdim 32 (transform_layer 334->32, in_layers 40->64, out_layers 64->32, dense 32->4)."""
    _gen_nc_variant("nc_dblp_latdim.npz", "[64]", 8, 32, [(64, 40)], [(32, 64)])


def gen_nc_latdim128():
    """... and with --latdim 128 (transform_layer 334->128, in_layers 136->64, out_layers 64->128, dense 128->4)."""
    _gen_nc_variant("nc_dblp_latdim128.npz", "[64]", 8, 128, [(64, 136)], [(128, 64)])


def gen_nc_layers():
    """... and with different layer counts per channel, --gcn_layer 2 --uugcn_layer 3 (the two source channels two
    layers, the target channel three: Model.py:31-36), otherwise the defaults."""
    _gen_nc_variant("nc_dblp_layers.npz", "[64]", 8, 64, [(64, 72)], [(64, 64)], gcn_layer=2, uugcn_layer=3)


def _gen_nc_variant(out_name, dims, d_emb_size, latdim, want_in, want_out, gcn_layer=None, uugcn_layer=None):
    Model = import_reference("DiffGraph_NC")
    import scipy.sparse as sp
    Model.args.dims, Model.args.d_emb_size, Model.args.latdim = dims, d_emb_size, latdim
    if gcn_layer is not None:
        Model.args.gcn_layer, Model.args.uugcn_layer = gcn_layer, uugcn_layer
    feats, edges, labels = load_dblp()
    NA = feats.shape[0]
    ap = edges["ap"]
    NP = max(ap[:, 1].max(), edges["pa"][:, 0].max()) + 1
    rowsum = np.array(feats.sum(1)).flatten()
    r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    features = torch.FloatTensor(np.asarray(sp.diags(r_inv).dot(feats).todense()))
    A_P = sp.coo_matrix((np.ones(len(ap)), (ap[:, 0], ap[:, 1])), shape=(NA, NP)).tocsr()
    apa = ((A_P @ A_P.T) != 0).astype(np.float32).tocoo()
    adjs = [FakeGraph(apa.row, apa.col, NA) for _ in range(3)]
    torch.manual_seed(29)
    model = Model.HGDM(features.shape[1])
    dn = model.user_denoise_model
    assert [tuple(l.weight.shape) for l in dn.in_layers] == want_in
    assert [tuple(l.weight.shape) for l in dn.out_layers] == want_out
    model.train()
    fx = {"dims": np.array(eval(dims)), "d_emb_size": np.array([d_emb_size]), "latdim": np.array([latdim]),
          "layers": np.array([Model.args.gcn_layer, Model.args.uugcn_layer])}
    for nm, q in model.named_parameters():
        fx["param_" + nm.replace(".", "_")] = np64(q)
    onehot = torch.zeros(NA, 4)
    onehot[torch.arange(NA), torch.from_numpy(labels).long()] = 1.0
    ancs = torch.from_numpy(np.load(os.path.join(OUT, "nc_dblp.npz"))["cl_ancs"])
    fx["cl_ancs"] = np64(ancs)
    with torch.no_grad():
        emb, scores = model.get_allembeds(adjs, features)
        if latdim > 64:      # (a sample of the rows keeps the file small)
            rows = torch.sort(torch.randperm(NA, generator=torch.Generator().manual_seed(3))[:512]).values
            fx["rows"] = np64(rows)
            emb, scores = emb[rows], scores[rows]
        fx["all_embeddings"], fx["scores"] = np64(emb), np64(scores)
    model.zero_grad()
    with Recorder() as rec:
        torch.manual_seed(47)
        nll, diff = model.cal_loss(ancs, onehot, adjs, features)
    (nll + diff).backward()
    fx.update({"ts": np64(rec.randint[0]), "noise": np64(rec.randn[0]), "keep": np64(rec.dropout[0]).astype(np.uint8),
               "losses": np.array([nll.item(), diff.item()], dtype=np.float64)})
    assert len(rec.dropout) == 1 and fx["keep"].shape == (NA, latdim)
    for nm, q in model.named_parameters():
        if q.grad is not None:
            fx["grad_" + nm.replace(".", "_")] = np64(q.grad)
    np.savez_compressed(os.path.join(OUT, out_name), **fx)
    print(out_name + ":", len(fx), "arrays; keep fraction", fx["keep"].mean())


# --------------------------------------------------------------------------------------
# Rec all-rank evaluation (Coach.testEpoch / calcRes, AllRankTestData)
# --------------------------------------------------------------------------------------
def gen_eval():
    """Runs the reference's own Coach.testEpoch (main.py:136-167) with its AllRankTestData
    (DataHandler.py:250-272) on a small synthetic train/test split and given embeddings.  The model is
    a stand-in whose predict() returns those embeddings; Tensor.cuda() is made the identity for the
    run (the reference hard-codes .cuda(), main.py:154-155); the run happens in a scratch directory
    because testEpoch pickles the embeddings into the working directory (main.py:150-152)."""
    import tempfile
    import scipy.sparse as sp
    from diffgraph_amd import data as dgdata
    install_dgl_stub()
    sys.argv = ["make_golden"]
    sys.path.insert(0, os.path.join(REF, "DiffGraph-Rec"))
    import main as ref_main          # noqa: the reference's modules
    import DataHandler as ref_dh     # noqa
    nu, ni, k = 300, 500, 20
    trn_u, trn_i = dgdata.synth_interactions(nu, ni, 3000, 7)
    cand_u, cand_i = dgdata.synth_interactions(nu, ni, 900, 8)
    trn_keys = set((trn_u.astype(np.int64) * ni + trn_i).tolist())
    keep = np.array([(int(u) * ni + int(i)) not in trn_keys for u, i in zip(cand_u, cand_i)])
    tst_u, tst_i = cand_u[keep], cand_i[keep]
    # a seeded shuffle, so the per-user test lists are not sorted by item (calcRes walks them in stored order)
    perm = np.random.default_rng(9).permutation(tst_u.size)
    tst_u, tst_i = tst_u[perm], tst_i[perm]
    gen = torch.Generator().manual_seed(5)
    usr = torch.randn(nu, 64, generator=gen) * 0.3
    itm = torch.randn(ni, 64, generator=gen) * 0.3
    trn = sp.coo_matrix((np.ones(trn_u.size, np.float32), (trn_u, trn_i)), shape=(nu, ni))
    tst = sp.coo_matrix((np.ones(tst_u.size, np.float32), (tst_u, tst_i)), shape=(nu, ni))
    ds = ref_dh.AllRankTestData(tst, trn)
    ref_main.args.tstBat, ref_main.args.topk = 64, k
    loader = torch.utils.data.DataLoader(ds, batch_size=ref_main.args.tstBat, shuffle=False, num_workers=0)
    coach = ref_main.Coach.__new__(ref_main.Coach)
    coach.handler = types.SimpleNamespace(test_dataloader=loader, userNum=nu, itemNum=ni)
    coach.model = types.SimpleNamespace(eval=lambda: None, predict=lambda: (usr, itm))
    captured = []
    real_calc = coach.calcRes

    def calc(topLocs, tstLocs, batIds):
        captured.append((np.array(topLocs), np64(batIds)))
        return real_calc(topLocs, tstLocs, batIds)

    coach.calcRes = calc
    cwd, real_cuda = os.getcwd(), torch.Tensor.cuda
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        torch.Tensor.cuda = lambda self, *a, **kw: self
        try:
            ret = coach.testEpoch()
        finally:
            torch.Tensor.cuda = real_cuda
            os.chdir(cwd)
    top = np.concatenate([c[0] for c in captured])
    users = np.concatenate([c[1] for c in captured])
    per = [real_calc(top[j: j + 1], ds.user_pos_lists, users[j: j + 1]) for j in range(len(users))]
    fx = {"n_user": np.array([nu]), "n_item": np.array([ni]), "topk": np.array([k]), "usr_emb": np64(usr), "item_emb": np64(itm),
          "trn_u": trn_u, "trn_i": trn_i, "tst_u": tst_u, "tst_i": tst_i, "users": users, "top_locs": top,
          "recall_per_user": np.array([p[0] for p in per], dtype=np.float64),
          "ndcg_per_user": np.array([p[1] for p in per], dtype=np.float64),
          "metrics": np.array([ret["Recall"], ret["NDCG"]], dtype=np.float64)}
    np.savez_compressed(os.path.join(OUT, "rec_eval.npz"), **fx)
    print("rec_eval.npz:", len(fx), "arrays; test users", len(users), "Recall/NDCG", ret)


# --------------------------------------------------------------------------------------
# NC meta-path matrices the reference ships (data files, not code)
# --------------------------------------------------------------------------------------
def pattern_signature(m):
    """Per-row entry count, sum and xor of the column ids of a CSR pattern: a compact stand-in for the
    matrix (a 4057 x 4057 pattern with 5 M entries becomes three 4057-vectors)."""
    m = m.tocsr()
    m.sort_indices()
    deg = np.diff(m.indptr).astype(np.int64)
    colsum = np.zeros(m.shape[0], np.int64)
    colxor = np.zeros(m.shape[0], np.int64)
    for r in range(m.shape[0]):
        c = m.indices[m.indptr[r]: m.indptr[r + 1]].astype(np.int64)
        colsum[r] = c.sum()
        colxor[r] = np.bitwise_xor.reduce(c) if len(c) else 0
    return deg, colsum, colxor


def gen_metapath():
    """Signatures of the author x author meta-path matrices the reference ships beside its DBLP edge lists
    (DiffGraph_NC/data/DBLP/apa.npz, apcpa.npz; scipy .npz, loaded without pickle)."""
    import scipy.sparse as sp
    fx = {}
    for name in ("apa", "apcpa"):
        m = sp.load_npz(os.path.join(REF, "DiffGraph_NC", "data", "DBLP", name + ".npz"))
        deg, cs, cx = pattern_signature(m != 0)
        fx[name + "_deg"], fx[name + "_colsum"], fx[name + "_colxor"] = deg, cs, cx
        fx[name + "_nnz"] = np.array([int(deg.sum())])
    np.savez_compressed(os.path.join(OUT, "dblp_metapaths.npz"), **fx)
    print("dblp_metapaths.npz:", {k: int(v[0]) for k, v in fx.items() if k.endswith("_nnz")})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["rec", "rec_train", "rec_dims", "rec_latdim", "rec_latdim128", "nc", "nc_train", "nc_dims", "nc_latdim", "nc_latdim128", "nc_layers", "eval", "metapath"]
    if len(which) == 1:
        {"rec": gen_rec, "rec_train": gen_rec_train, "rec_dims": gen_rec_dims, "rec_latdim": gen_rec_latdim, "rec_latdim128": gen_rec_latdim128, "nc": gen_nc, "nc_train": gen_nc_train, "nc_dims": gen_nc_dims, "nc_latdim": gen_nc_latdim, "nc_latdim128": gen_nc_latdim128, "nc_layers": gen_nc_layers, "eval": gen_eval,
         "metapath": gen_metapath}[which[0]]()
    else:
        for w in which:  # the two programs share module names (Model, params, Utils): one process each
            subprocess.run([sys.executable, os.path.abspath(__file__), w], check=True)
