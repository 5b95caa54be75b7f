"""Coach (diffgraph_amd.train) against the reference's training loop semantics (DiffGraph-Rec/main.py:107-167)."""
import copy
from types import SimpleNamespace

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import diffgraph_amd as dg
from diffgraph_amd import data as dgdata
from diffgraph_amd.evaluate import AllRankTestData
from diffgraph_amd.model_rec import DEFAULT_ARGS
from diffgraph_amd.sampling import PairwiseTrnData
from diffgraph_amd.train import Coach

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _coo(u, i, nu, ni):
    return sp.coo_matrix((np.ones(len(u), np.float32), (u, i)), shape=(nu, ni))


def _handler(name="tiny", seed=1):
    wl = dgdata.rec_workload(name)
    nu, ni = wl.n_user, wl.n_item
    graphs = [dg.CSRGraph.from_bipartite(u, i, nu, ni) for (u, i) in [wl.target] + wl.behaviours]
    rng = np.random.default_rng(7)
    tst_u, tst_i = np.arange(nu, dtype=np.int32), rng.integers(0, ni, nu).astype(np.int32)
    trn = _coo(*wl.target, nu, ni)
    return SimpleNamespace(userNum=nu, itemNum=ni, target_adj=graphs[0], behavior_mats=graphs[1:],
                           trn_data=PairwiseTrnData(trn, device=DEV, seed=seed), tst_data=AllRankTestData(_coo(tst_u, tst_i, nu, ni), trn, device=DEV))


def _args(batch):
    a = copy.copy(DEFAULT_ARGS)
    a.batch = batch
    return a


def test_train_epoch_equals_the_reference_loop():
    """Eager Coach.trainEpoch against the reference's loop written out (negSampling, shuffled batches incl. the short
    last one, cal_loss / backward / Adam, `.item()` sums divided by len // batch): identical numbers."""
    args = _args(64)
    torch.manual_seed(5)
    c = Coach(_handler(), args, device=DEV, use_graph=False, seed=3)
    c.prepareModel()
    got = [c.trainEpoch() for _ in range(2)]

    torch.manual_seed(5)
    h = _handler()
    model = dg.rec.HGDM(h, args, device=DEV)       # train() mode, like Coach's until its first testEpoch: dropout on
    opt = dg.optim.Adam(model, lr=args.lr, weight_decay=0)
    gen = torch.Generator(device=DEV)
    gen.manual_seed(3)
    for ep in range(2):
        h.trn_data.negSampling()
        epLoss = epRecLoss = eDiffLoss = 0
        steps = len(h.trn_data) // args.batch
        n_batches = 0
        for ancs, poss, negs in h.trn_data.batches(args.batch, shuffle=True, generator=gen):
            opt.zero_grad()
            loss, bprLoss, regLoss, diff_loss = model.cal_loss(ancs, poss, negs)
            epLoss += loss.item(); epRecLoss += bprLoss.item(); eDiffLoss += diff_loss.item()
            loss.backward()
            opt.step()
            n_batches += 1
        assert n_batches == steps + 1      # 186 pairs, batch 64: the short last batch is trained on, as in the reference
        want = {"Loss": epLoss / steps, "recLoss": epRecLoss / steps, "diffLoss": eDiffLoss / steps}
        for k in want:
            assert got[ep][k] == pytest.approx(want[k], rel=1e-12), (ep, k)
    for p, q in zip(c.model.parameters(), model.parameters()):
        assert torch.equal(p, q)


def _fixed_draws(coach, n_steps, seed, keep):
    n = coach.model.n_user + coach.model.n_item
    gen = torch.Generator().manual_seed(seed)
    ts = torch.randint(0, coach.args.steps, (n_steps, n), generator=gen)
    noise = torch.randn(n_steps, n, 64, generator=gen)
    mask = (torch.rand(n_steps, n, 64, generator=gen) >= 0.5).to(torch.uint8) if keep else None
    coach.set_draws(ts, noise, mask)


def _optimizer_state(o):
    # the table's moments and step count, then the denoiser tensors' (in parameter order, the same for every model instance)
    return [o.exp_avg, o.exp_avg_sq, o.state] + o.rest.exp_avg + o.rest.exp_avg_sq + [o.rest.state]


@pytest.mark.parametrize("draws", ["given", "drawn"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(mode, draws):
    """The step Coach replays as a HIP graph (full batch and the epoch's short last batch) against the same step
    launched eagerly: the epoch losses, every parameter and the optimizer's moments are the same bits after two
    epochs -- with the timesteps, the noise and (train mode) the dropout masks of every step handed in, and with
    the step drawing them itself from the model's device-side draw state (same seed: same draws, replayed or not)."""
    args = _args(64)
    coaches = {}
    for use_graph in (True, False):
        torch.manual_seed(9)
        c = Coach(_handler(), args, device=DEV, use_graph=use_graph, seed=3)
        c.prepareModel()
        c.model.train(mode == "train")
        if draws == "given":
            _fixed_draws(c, 16, seed=21, keep=mode == "train")
        assert c.model.draw_state.tolist() == [9, 0, 0, 0]
        if use_graph:
            before = [p.detach().clone() for p in c.model.parameters()]
            c._capture(64)
            for p, q in zip(c.model.parameters(), before):
                assert torch.equal(p, q)
            o = c.opt
            assert not o.exp_avg.any() and not o.exp_avg_sq.any() and not o.state.any() and not c._sums.any()
            assert int(c._batch_no) == 0 and (draws == "drawn" or int(c._draw_no) == 0)
            assert c.model.draw_state.tolist() == [9, 0, 0, 0]
        coaches[use_graph] = (c, [c.trainEpoch() for _ in range(2)])
    (g, g_res), (e, e_res) = coaches[True], coaches[False]
    assert sorted(g._graphs) == [(58, mode == "train"), (64, mode == "train")]      # 186 pairs: 64 + 64 + 58
    if draws == "given":
        assert int(g._draw_no) == int(e._draw_no) == 6
    assert int(g.model.draw_state[1]) == int(e.model.draw_state[1]) == (6 if draws == "drawn" else 0)
    assert torch.equal(g.model.draw_state, e.model.draw_state)
    for a, b in zip(g_res, e_res):
        assert np.isfinite(a["Loss"])
        assert a == b, (a, b)
    assert g_res[-1]["Loss"] < g_res[0]["Loss"]
    for (nm, p), q in zip(g.model.named_parameters(), e.model.parameters()):
        assert torch.isfinite(p).all(), nm
        assert torch.equal(p, q), nm
    for a, b in zip(_optimizer_state(g.opt), _optimizer_state(e.opt)):
        assert torch.equal(a, b)


def test_replayed_cal_loss_reproduces_the_reference_golden():
    """cal_loss + backward recorded as a HIP graph and replayed, with the reference's recorded draws in static device
    buffers: the four losses and all gradients match the reference's own numbers (the bar of the eager tests)."""
    from tests.helpers import GRAD_TOL, assert_close_to_max, assert_rows_close, tiny_fixture
    from tests.test_gpu_parity import dev, rec_model
    fx, nu, ni, *_ = tiny_fixture()
    m = rec_model(dgdata.rec_workload("tiny"), fx)
    m.set_embeddings(dev(fx["user_emb"]), dev(fx["item_emb"]))
    m.args = SimpleNamespace(**{**vars(m.args), "batch": int(fx["cl_batch_div"][0]), "reg": float(fx["cl_batch_div"][1])})
    inputs = [dev(fx[k]) for k in ("cl_ancs", "cl_poss", "cl_negs", "cl_ts_u", "cl_noise_u", "cl_ts_i", "cl_noise_i")]
    seed = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=DEV)

    def step():
        res = m.cal_loss_vector(*inputs, validate=False)
        m.zero_grad(set_to_none=True)
        res.backward(seed)
        return res
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    first = res.detach().clone()
    grads = [p.grad.clone() for p in m.parameters()]
    for _ in range(20):      # replays separated by host synchronisation and eager reductions (fresh allocations between replays)
        graph.replay()
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(p).all()) and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
        assert torch.equal(res.detach(), first)
        assert all(torch.equal(p.grad, g) for p, g in zip(m.parameters(), grads))
    np.testing.assert_allclose(res.detach().cpu().numpy(), fx["cl_losses"], rtol=1e-5)
    g = m.embedding_table.grad
    assert_rows_close(g[:nu], fx["cl_grad_user"], rtol=1e-5, what="grad user_emb")
    assert_rows_close(g[nu:], fx["cl_grad_item"], rtol=1e-5, what="grad item_emb")
    for pre, dn in (("usr_", m.usr_denoiser), ("item_", m.item_denoiser)):
        got = {"w_emb": dn.emb_layer.weight.grad, "b_emb": dn.emb_layer.bias.grad, "w_in": dn.in_layers[0].weight.grad,
               "b_in": dn.in_layers[0].bias.grad, "w_out": dn.out_layers[0].weight.grad, "b_out": dn.out_layers[0].bias.grad}
        for nm, gv in got.items():
            want = fx["cl_grad_" + pre + nm]
            assert_close_to_max(gv, want, GRAD_TOL, "grad " + pre + nm)
    eager = step()
    torch.cuda.synchronize()
    assert torch.equal(eager.detach(), res.detach())


def test_run_tests_every_epoch_and_checkpoints(tmp_path):
    args = _args(64)
    torch.manual_seed(2)
    c = Coach(_handler(), args, device=DEV, seed=1)
    path = str(tmp_path / "best.pt")
    out = c.run(epochs=3, save_path=path, log=lambda s: None)
    assert len(c.metrics["TestRecall"]) == 3 and 0.0 <= out["Recall"] <= 1.0 and 0.0 <= out["NDCG"] <= 1.0
    res = c.testEpoch()
    want = dg.evaluate.testEpoch(c.model, c.handler.tst_data, args.topk, args.tstBat)
    assert res["Recall"] == want["Recall"] and res["NDCG"] == want["NDCG"]
    d = Coach(_handler(), args, device=DEV, seed=1)
    d.loadModel(path)
    assert d.testEpoch()["Recall"] == pytest.approx(out["Recall"], abs=1e-12)


@pytest.mark.parametrize("batch", [62, 256])
def test_batch_sizes_that_divide_the_data_or_exceed_it(batch):
    """186 training pairs: batch 62 leaves no short last batch, batch 256 leaves only one (len // batch == 0: the
    reference would divide by zero there; Coach divides by 1).  Replayed and eager epochs see the same batches."""
    args = _args(batch)
    res = {}
    for use_graph in (True, False):
        torch.manual_seed(4)
        c = Coach(_handler(), args, device=DEV, use_graph=use_graph, seed=2)
        c.prepareModel()
        c.model.eval()
        _fixed_draws(c, 12, seed=22, keep=False)
        before = c.model.embedding_table.detach().clone()
        res[use_graph] = [c.trainEpoch() for _ in range(2)]
        assert not torch.equal(before, c.model.embedding_table)
        if use_graph:
            assert sorted(c._graphs) == [(min(batch, 186), False)]
    for a, b in zip(res[True], res[False]):
        assert np.isfinite(a["Loss"]) and a == b


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_nc_step_replayed_as_a_graph_equals_the_eager_step_bit_for_bit(mode):
    """One DiffGraph_NC training step (DiffGraph_NC/main.py:378-386: cal_loss, backward, Adam) over one sparse and two
    dense author graphs, recorded as a HIP graph and replayed five times, against five eager steps from the same
    start with the same timesteps / noise / dropout mask: the losses, every parameter and Adam's moments are the same
    bits."""
    n, f = 700, 40
    rng = np.random.default_rng(5)
    mats = []
    for dens in (0.01, 0.3, 0.45):
        a = rng.random((n, n)) < dens
        mats.append(sp.csr_matrix((a | a.T).astype(np.float32)))
    gen = torch.Generator().manual_seed(13)
    feats = torch.rand(n, f, generator=gen).to(DEV)
    onehot = torch.zeros(n, 4, device=DEV)
    onehot[torch.arange(n), torch.randint(0, 4, (n,), generator=gen)] = 1.0
    ancs = torch.randperm(n, generator=gen)[:8].to(DEV)
    ts = torch.randint(0, 200, (n,), generator=gen).to(DEV)
    noise = torch.randn(n, 64, generator=gen).to(DEV)
    keep = (torch.rand(n, 64, generator=gen) >= 0.5).to(torch.uint8).to(DEV) if mode == "train" else None
    out = {}
    for use_graph in (True, False):
        adjs = [dg.CSRGraph.from_scipy(m) for m in mats]
        torch.manual_seed(17)
        model = dg.nc.HGDM(f).train(mode == "train")
        opt = torch.optim.Adam(model.parameters(), lr=3e-3, weight_decay=0, capturable=True)
        start = [p.detach().clone() for p in model.parameters()]

        def step():
            nll, diff = model.cal_loss(ancs, onehot, adjs, feats, ts, noise, keep, validate=False)
            opt.zero_grad(set_to_none=True)
            (nll + diff).backward()
            opt.step()
            return torch.stack([nll.detach().double(), diff.detach().double()])
        losses = []
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                res = step()
            torch.cuda.synchronize()
            with torch.no_grad():       # back to the start: parameters, moments and Adam's device-side step counter
                for p, q in zip(model.parameters(), start):
                    p.copy_(q)
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            for _ in range(5):
                graph.replay()
                torch.cuda.synchronize()
                losses.append(res.clone())
        else:
            for _ in range(5):
                losses.append(step().clone())
        out[use_graph] = (losses, [p.detach().clone() for p in model.parameters()],
                          [st[k].clone() for p in opt.param_groups[0]["params"] for st in [opt.state[p]] for k in sorted(st)])
    (gl, gp, gs), (el, ep, es) = out[True], out[False]
    assert all(bool(torch.isfinite(a).all()) for a in gl) and float(gl[-1].sum()) != float(gl[0].sum())
    for a, b in zip(gl, el):
        assert torch.equal(a, b), (a, b)
    for a, b in zip(gp + gs, ep + es):
        assert torch.equal(a, b)


def _nc_handler(n=700, f=40, seed=5):
    rng = np.random.default_rng(seed)
    mats = []
    for dens in (0.01, 0.3, 0.45):
        a = rng.random((n, n)) < dens
        mats.append(sp.csr_matrix((a | a.T).astype(np.float32)))
    gen = torch.Generator().manual_seed(13)
    labels = torch.zeros(n, 4)
    labels[torch.arange(n), torch.randint(0, 4, (n,), generator=gen)] = 1.0
    # features that carry the label, so that a few epochs of training show in the accuracy
    feats = torch.rand(n, f, generator=gen) * 0.2 + labels.repeat(1, f // 4)
    perm = rng.permutation(n)
    return SimpleNamespace(feature_list=feats, labels=labels, he_adjs=[dg.CSRGraph.from_scipy(m) for m in mats],
                           train_idx=[perm[:60], perm[:101]], val_idx=[perm[200:300]] * 2, test_idx=[perm[300:]] * 2)


def test_nc_coach_replayed_epochs_equal_eager_epochs_bit_for_bit():
    """NCCoach (DiffGraph_NC/main.py:364-449): two epochs over 101 labelled authors in batches of 8 (12 full batches
    and a short one of 5 -- two recordings), dropout on as in the reference, replayed against eagerly launched with the
    same seed: the same epoch losses, parameters, Adam moments and test metrics; and the model has learnt something."""
    from diffgraph_amd.train_nc import NCCoach
    out = {}
    for use_graph in (True, False):
        torch.manual_seed(3)
        c = NCCoach(_nc_handler(), device=DEV, use_graph=use_graph, seed=7)
        c.prepareModel()
        assert c.model.training
        eps = [c.trainEpoch(1) for _ in range(2)]
        val, test = c.testEpoch(1)
        if use_graph:
            assert sorted(c._graphs) == [(5, True), (8, True)]
        out[use_graph] = (eps, [p.detach().clone() for p in c.model.parameters()],
                          [t.clone() for t in c.opt.exp_avg + c.opt.exp_avg_sq + [c.opt.state]],
                          (float(val["acc"]), val["macro"], val["micro"], float(test["acc"]), test["macro"], test["micro"]),
                          test["logits"].clone())
    (g_eps, g_p, g_s, g_m, g_l), (e_eps, e_p, e_s, e_m, e_l) = out[True], out[False]
    assert all(np.isfinite(v) for ep in g_eps for v in ep.values()) and g_eps == e_eps, (g_eps, e_eps)
    assert g_eps[1]["bceLoss"] < g_eps[0]["bceLoss"]
    for a, b in zip(g_p + g_s + [g_l], e_p + e_s + [e_l]):
        assert torch.equal(a, b)
    assert g_m == e_m and g_m[3] > 0.5 and g_l.shape == (400, 4)      # four balanced classes: chance is 0.25
    assert g_s[-1].tolist() == e_s[-1].tolist() == [26.0, 0.0, 0.0, 0.0]       # Adam's step count: 2 x 13 steps, warm-ups undone


def test_nc_coach_run_and_checkpoint(tmp_path):
    from diffgraph_amd.train_nc import NCCoach
    torch.manual_seed(3)
    c = NCCoach(_nc_handler(), device=DEV, seed=1)
    res = c.run(ratio=0, epochs=3, log=lambda s: None)
    assert set(res) == {"acc", "macro", "macro_val", "micro"} and len(c.metrics["bceLoss"]) == 3 and 0.25 < res["acc"] <= 1.0
    path = str(tmp_path / "nc.pt")
    c.saveModel(path)
    d = NCCoach(_nc_handler(), device=DEV, seed=1)
    d.loadModel(path)
    a, b = c.testEpoch(0)[1], d.testEpoch(0)[1]
    assert torch.equal(a["logits"], b["logits"]) and a["macro"] == b["macro"]
    # the generators of the draws and of the batch order travel too: the next epoch is the same, number for number
    assert c.trainEpoch(0) == d.trainEpoch(0)
    for p, q in zip(c.model.parameters(), d.model.parameters()):
        assert torch.equal(p, q)


@pytest.mark.parametrize("nu,ni,B", [(40, 60, 64), (31882, 31232, 4096), (1025, 2048, 8192), (5, 3000, 1)])
def test_active_rows_are_the_distinct_batch_rows_in_order(nu, ni, B):
    """dg_active_rows: the distinct ancs, and the distinct poss and negs, ascending, with their counts; indices outside
    their table are left out."""
    from diffgraph_amd import ops
    gen = torch.Generator().manual_seed(nu + B)
    ancs = torch.randint(0, nu, (B,), generator=gen)
    poss, negs = torch.randint(0, ni, (B,), generator=gen), torch.randint(0, ni, (B,), generator=gen)
    if B > 8:
        ancs[3], poss[5], negs[7] = -1, ni, 2 ** 40
    rows_u, rows_i, counts = ops.active_rows(ancs.to(DEV), poss.to(DEV), negs.to(DEV), nu, ni)
    cu, ci = (int(c) for c in counts.cpu())
    want_u = torch.unique(ancs[(ancs >= 0) & (ancs < nu)])
    both = torch.cat([poss, negs])
    want_i = torch.unique(both[(both >= 0) & (both < ni)])
    assert cu == want_u.numel() and ci == want_i.numel()
    assert torch.equal(rows_u[:cu].cpu().long(), want_u) and torch.equal(rows_i[:ci].cpu().long(), want_i)


def test_loss_on_the_batch_rows_equals_the_loss_over_every_row():
    """HGDM.cal_loss with the denoisers run over the rows the batch names only (the default) against the same call over
    every row, as the reference runs it (loss_on_batch_rows = False): the same four values bit for bit, the same
    gradient of the table bit for bit, the denoisers' weight gradients to the last bits (their row sums are grouped
    differently)."""
    h = _handler()
    torch.manual_seed(11)
    m = dg.rec.HGDM(h, _args(64), device=DEV).eval()
    nu, ni = h.userNum, h.itemNum
    gen = torch.Generator().manual_seed(2)
    ancs, poss, negs = (torch.randint(0, hi, (64,), generator=gen).to(DEV) for hi in (nu, ni, ni))
    ts = torch.randint(0, 200, (nu + ni,), generator=gen).to(DEV)
    noise = torch.randn(nu + ni, 64, generator=gen).to(DEV)
    got = {}
    for listed in (True, False):
        m.loss_on_batch_rows = listed
        m.zero_grad()
        res = m.cal_loss_vector(ancs, poss, negs, ts[:nu], noise[:nu], ts[nu:], noise[nu:])
        res.backward(torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=DEV))
        got[listed] = (res.detach().clone(), m.embedding_table.grad.clone(),
                       [p.grad.clone() for p in m.usr_denoiser.parameters()] + [p.grad.clone() for p in m.item_denoiser.parameters()])
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1])
    for a, b in zip(got[True][2], got[False][2]):
        assert torch.allclose(a, b, rtol=0, atol=2e-6 * float(b.abs().max()))
