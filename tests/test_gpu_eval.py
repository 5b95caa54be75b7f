"""GPU parity of the all-rank evaluation (Coach.testEpoch / calcRes, DiffGraph-Rec/main.py:136-187)
against the reference's golden run and the CPU oracle."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import diffgraph_amd as dg
from diffgraph_amd import data as dgdata, evaluate as ev
from oracle import ref_cpu as O
from tests.helpers import load

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = torch.from_numpy


def coo(u, i, nu, ni):
    return sp.coo_matrix((np.ones(len(u), np.float32), (u, i)), shape=(nu, ni))


def check_top_lists(got_idx, got_val, want_idx, scores, what):
    """Top-k lists agree with the oracle's except where the reference leaves the order open: positions whose
    scores tie (exactly, or within fp32 rounding of the differently ordered dot products)."""
    got_idx, want_idx = np.asarray(got_idx), np.asarray(want_idx)
    assert got_idx.shape == want_idx.shape, what
    for r in np.nonzero((got_idx != want_idx).any(axis=1))[0]:
        for c in np.nonzero(got_idx[r] != want_idx[r])[0]:
            a, b = scores[r, got_idx[r, c]], scores[r, want_idx[r, c]]
            assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), f"{what}: user row {r} position {c}: item {got_idx[r, c]} ({a}) vs {want_idx[r, c]} ({b})"
    np.testing.assert_allclose(got_val, np.take_along_axis(scores, want_idx.astype(np.int64), axis=1), rtol=2e-5, atol=2e-5,
                               err_msg=what)


def test_scores_on_split_bf16_match_the_fp32_matrix_instruction(monkeypatch):
    """The scoring kernel's two forms -- the bf16 matrix instruction over fp32 values cut into three bf16 numbers (the
    default up to k = 20) and the exact-fp32 instruction (DG_EVAL_FP32=1) -- against the fp64 scores and each other:
    the same lists except at ties within fp32 rounding, values within two fp32 roundings of each other."""
    nu, ni, k = 700, 5003, 20
    gen = torch.Generator().manual_seed(77)
    usr, itm = torch.randn(nu, 64, generator=gen), torch.randn(ni, 64, generator=gen)
    trn_u, trn_i = dgdata.synth_interactions(nu, ni, 40000, 3)
    rng = np.random.default_rng(6)
    data = ev.AllRankTestData(coo(rng.integers(0, nu, nu).astype(np.int32), rng.integers(0, ni, nu).astype(np.int32), nu, ni),
                              coo(trn_u, trn_i, nu, ni))
    users = data.test_users_dev
    mask = np.zeros((nu, ni), np.float32)
    mask[trn_u, trn_i] = 1.0
    m = T(mask[data.test_users])
    scores = (torch.mm(usr[T(data.test_users).long()].double(), itm.double().t()) * (1 - m) - m * 1e8).numpy()
    _, want = torch.topk(T(scores), k)
    got = {}
    for form in ("0", "1"):
        monkeypatch.setenv("DG_EVAL_FP32", form)
        vals, top = ev.all_rank_topk(usr.to(DEV), itm.to(DEV), users, data, k)
        got[form] = (top.cpu().numpy(), vals.cpu().numpy())
        check_top_lists(got[form][0], got[form][1], want.numpy(), scores, f"DG_EVAL_FP32={form}")
    np.testing.assert_allclose(np.sort(got["0"][1], axis=1), np.sort(got["1"][1], axis=1), rtol=0, atol=2e-6 * np.abs(scores[scores > -1e7]).max())
    assert (got["0"][0] != got["1"][0]).mean() < 1e-3


def test_eval_golden_reference_test_epoch():
    """The reference's own testEpoch run (tools/make_golden.py gen_eval): same test-user order, top-20
    lists, per-user Recall/NDCG bit for bit and the epoch metrics."""
    fx = load("rec_eval.npz")
    nu, ni, k = int(fx["n_user"][0]), int(fx["n_item"][0]), int(fx["topk"][0])
    data = ev.AllRankTestData(coo(fx["tst_u"], fx["tst_i"], nu, ni), coo(fx["trn_u"], fx["trn_i"], nu, ni))
    np.testing.assert_array_equal(data.test_users, fx["users"])
    res = ev.rank_metrics(T(fx["usr_emb"]).to(DEV), T(fx["item_emb"]).to(DEV), data, topk=k, tstBat=64)
    np.testing.assert_array_equal(res["topLocs"].cpu().numpy(), fx["top_locs"])
    np.testing.assert_array_equal(res["recall_per_user"], fx["recall_per_user"])
    np.testing.assert_array_equal(res["ndcg_per_user"], fx["ndcg_per_user"])
    np.testing.assert_array_equal(np.array([res["Recall"], res["NDCG"]]), fx["metrics"])


@pytest.mark.parametrize("nu,ni,k,n_trn", [(1, 5, 3, 2), (130, 33, 20, 1500), (257, 1000, 1, 0), (500, 4099, 32, 60000),
                                            (64, 40, 20, 2200),
                                            # every list size the kernel is built for (k rounded up to 4, 8, ..., 32)
                                            (300, 2000, 5, 9000), (300, 2000, 9, 9000), (300, 2000, 13, 9000),
                                            (300, 2000, 24, 9000), (300, 2000, 25, 9000), (300, 2000, 29, 9000)])
def test_eval_ragged_sizes_against_oracle(nu, ni, k, n_trn):
    """Ragged user / item counts, k = 1 and 32, no training data, and users whose training row leaves
    fewer than k items (the masked ones then fill the list at -1e8, lowest item id first)."""
    gen = torch.Generator().manual_seed(nu * 7 + ni)
    usr, itm = torch.randn(nu, 64, generator=gen), torch.randn(ni, 64, generator=gen)
    trn_u, trn_i = dgdata.synth_interactions(nu, ni, n_trn, 3) if n_trn else (np.zeros(0, np.int32), np.zeros(0, np.int32))
    rng = np.random.default_rng(5)
    tst_u, tst_i = rng.integers(0, nu, 3 * nu), rng.integers(0, ni, 3 * nu)
    key = np.unique(tst_u.astype(np.int64) * ni + tst_i)
    tst_u, tst_i = (key // ni).astype(np.int32), (key % ni).astype(np.int32)
    data = ev.AllRankTestData(coo(tst_u, tst_i, nu, ni), coo(trn_u, trn_i, nu, ni))
    users = data.test_users_dev
    vals, top = ev.all_rank_topk(usr.to(DEV), itm.to(DEV), users, data, k)
    mask = np.zeros((nu, ni), np.float32)
    mask[trn_u, trn_i] = 1.0
    m = T(mask[data.test_users])
    scores = (torch.mm(usr[T(data.test_users).long()].double(), itm.double().t()) * (1 - m) - m * 1e8).numpy()
    _, want = torch.topk(T(scores), min(k, ni))
    got_idx, got_val = top.cpu().numpy(), vals.cpu().numpy()
    if k > ni:
        assert (got_idx[:, ni:] == -1).all()
        got_idx, got_val = got_idx[:, :ni], got_val[:, :ni]
    check_top_lists(got_idx, got_val, want.numpy(), scores, f"{nu}x{ni} k={k}")
    # masked items only ever appear behind every unmasked one, in ascending id order
    for r in range(got_idx.shape[0]):
        masked = scores[r, got_idx[r]] <= -1e8
        if masked.any():
            first = int(np.argmax(masked))
            assert masked[first:].all() and (np.diff(got_idx[r, first:]) > 0).all()
    # metrics from the kernel's lists == the oracle's calcRes on the same lists
    _, lists = O.test_users_and_lists(nu, tst_u, tst_i)
    kk = min(k, ni)
    r_all, n_all, r_u, n_u = ev.calcRes(top[:, :kk].contiguous(), data, users)
    wr, wn, wr_u, wn_u = O.calc_res(got_idx, lists, data.test_users, topk=kk)
    np.testing.assert_array_equal(r_u, wr_u)
    np.testing.assert_array_equal(n_u, wn_u)
    assert r_all == wr and n_all == wn


def test_eval_tmall_size_sample_against_oracle():
    """Full Tmall-shaped item table, every user a test user; 512 sampled users checked against the oracle's
    torch.mm + topk, and size-independent properties for all: sorted scores, no training item in any list."""
    wl = dgdata.rec_workload("tmall")
    nu, ni, k = wl.n_user, wl.n_item, 20
    usr = T(dgdata.xavier_uniform(nu, 64, seed=0)) * 3
    itm = T(dgdata.xavier_uniform(ni, 64, seed=1)) * 3
    trn_u, trn_i = wl.target
    rng = np.random.default_rng(11)
    tst_u = np.arange(nu, dtype=np.int32)
    tst_i = rng.integers(0, ni, nu).astype(np.int32)
    data = ev.AllRankTestData(coo(tst_u, tst_i, nu, ni), coo(trn_u, trn_i, nu, ni))
    vals, top = ev.all_rank_topk(usr.to(DEV), itm.to(DEV), data.test_users_dev, data, k)
    top_h, vals_h = top.cpu().numpy(), vals.cpu().numpy()
    assert (np.diff(vals_h, axis=1) <= 0).all() and (top_h >= 0).all() and (top_h < ni).all()
    trn_keys = set((trn_u.astype(np.int64) * ni + trn_i).tolist())
    sample = rng.choice(len(data), 512, replace=False)
    su = data.test_users[sample]
    mask = np.zeros((512, ni), np.float32)
    for r, u in enumerate(su):
        mask[r, trn_i[trn_u == u]] = 1.0
        assert not any((int(u) * ni + int(i)) in trn_keys for i in top_h[sample[r]])
    m = T(mask)
    scores = (torch.mm(usr[T(su).long()].double(), itm.double().t()) * (1 - m) - m * 1e8).numpy()
    _, want = torch.topk(T(scores), k)
    check_top_lists(top_h[sample], vals_h[sample], want.numpy(), scores, "tmall sample")


# ------------------------------------------------------------------------------ N3: negative sampling
def test_neg_sampling_properties_and_distribution():
    """PairwiseTrnData.negSampling (DiffGraph-Rec/DataHandler.py:282-289): every negative is in range and not a
    training item of its user; the same seed gives the same draws, another seed others; per user the negatives
    are uniform over the items the user has not interacted with, like the oracle's (the reference draws from
    numpy's global generator, so the streams themselves cannot be compared)."""
    from diffgraph_amd.sampling import PairwiseTrnData
    nu, ni = 40, 64
    u, i = dgdata.synth_interactions(nu, ni, 900, 3)
    keep = u >= 2
    # user 1 has interacted with every item: there is nothing to draw, and the reference's loop would not end
    full = PairwiseTrnData(coo(np.concatenate([u[keep], np.ones(ni, np.int32)]),
                               np.concatenate([i[keep], np.arange(ni, dtype=np.int32)]), nu, ni), seed=7)
    with pytest.raises(ValueError, match="user 1 has interacted with all"):
        full.negSampling()
    # user 0 has interacted with all items but three, and five of those pairs are stored twice (more stored pairs
    # than there are items: only distinct items count)
    u = np.concatenate([u[keep], np.zeros(ni - 3, np.int32), np.zeros(5, np.int32)])
    i = np.concatenate([i[keep], np.arange(3, ni, dtype=np.int32), np.arange(3, 8, dtype=np.int32)])
    train = set(zip(u.tolist(), i.tolist()))
    ds = PairwiseTrnData(coo(u, i, nu, ni), seed=7)
    assert len(ds) == len(u)
    reps = 300
    draws = []
    for _ in range(reps):                                        # one launch per "epoch", a new stream each
        ds.negSampling()
        draws.append(ds.negs)
    negs, uu = np.concatenate(draws), np.tile(u, reps)
    ok = np.ones(len(uu), bool)
    assert ((negs >= 0) & (negs < ni)).all()
    assert not any((int(a), int(b)) in train for a, b in zip(uu[ok][:20000], negs[ok][:20000]))
    assert set(negs[uu == 0].tolist()) == {0, 1, 2}
    ds2 = PairwiseTrnData(coo(u, i, nu, ni), seed=7)
    ds2.negSampling()
    np.testing.assert_array_equal(ds2.negs, draws[0])             # same seed, same epoch: same draws
    assert (draws[1][u != 0] != draws[0][u != 0]).mean() > 0.5    # the next epoch draws afresh
    # distribution: chi-square of each user's negatives against uniform over its free items, and the same
    # statistic for the oracle's sampler on the same pairs
    sub = np.isin(uu, [2, 3, 4, 5])
    on = O.neg_sampling(uu[sub], np.tile(i, reps)[sub], ni, np.random.default_rng(1))
    for user in (2, 3, 4, 5):
        free = np.array([it for it in range(ni) if (user, it) not in train])
        for name, sample in (("kernel", negs[uu == user]), ("oracle", on[uu[sub] == user])):
            counts = np.array([(sample == it).sum() for it in free], dtype=np.float64)
            exp = counts.sum() / len(free)
            chi2 = ((counts - exp) ** 2 / exp).sum()
            assert chi2 < len(free) + 6 * np.sqrt(2 * len(free)), f"{name} user {user}: chi2 {chi2} over {len(free)} items"
    # batches: every pair once, shuffled, as int64 triples
    seen = 0
    for ancs, poss, ngs in ds.batches(4096):
        assert ancs.dtype == torch.int64 and ancs.is_cuda and len(ancs) == len(poss) == len(ngs)
        seen += len(ancs)
    assert seen == len(ds)


# ------------------------------------------------------------------------------ N2: meta-path composition
def _pattern_sig(m):
    m = m.tocsr()
    m.sort_indices()
    deg = np.diff(m.indptr).astype(np.int64)
    cs = np.add.reduceat(np.append(m.indices.astype(np.int64), 0), m.indptr[:-1]) * (deg > 0)
    cx = np.array([np.bitwise_xor.reduce(m.indices[m.indptr[r]: m.indptr[r + 1]].astype(np.int64)) if deg[r] else 0
                   for r in range(m.shape[0])])
    return deg, cs, cx


def test_metapath_composition_dblp_golden():
    """Device boolean SpGEMM against the matrices the reference ships (apa.npz, apcpa.npz signatures), the oracle's
    scipy composition (APTPA, 7 M entries, 43 % dense) and the as-coded channel graphs of LoadData."""
    from diffgraph_amd import metapath as mp
    from tests.test_oracle_golden import _dblp_graphs
    g, fx = load("dblp_graph.npz"), load("dblp_metapaths.npz")
    NA, NP, NC, NT = (int(v) for v in g["counts"])
    mk = lambda e, shape: sp.coo_matrix((np.ones(len(e), np.float32), (e[:, 0], e[:, 1])), shape=shape).tocsr()
    AP, PC, PT = mk(g["ap"], (NA, NP)), mk(g["pc"], (NP, NC)), mk(g["pt"], (NP, NT))
    D = mp.DevicePattern.from_scipy
    apa = mp.compose([D(AP), D(AP.T)])
    apcpa = mp.compose([D(AP), D(PC), D(PC.T), D(AP.T)])
    aptpa = mp.compose([D(AP), D(PT), D(PT.T), D(AP.T)])
    for name, pat in (("apa", apa), ("apcpa", apcpa)):
        deg, cs, cx = _pattern_sig(pat.to_scipy())
        assert pat.nnz == int(fx[name + "_nnz"][0])
        np.testing.assert_array_equal(deg, fx[name + "_deg"])
        np.testing.assert_array_equal(cs, fx[name + "_colsum"])
        np.testing.assert_array_equal(cx, fx[name + "_colxor"])
    _, _, _, want_aptpa, _ = _dblp_graphs()
    got = aptpa.to_scipy()
    assert got.nnz == want_aptpa.nnz == 7043571 and (got != want_aptpa).nnz == 0
    # LoadData option B: three channels of mixed relations -> author x author graphs
    mats = O.dblp_global_mats(g["ap"], g["pc"], g["pt"], g["counts"])
    rng = np.random.default_rng(0)
    a1 = torch.softmax(T(rng.standard_normal((3, 6)).astype(np.float32)), 1).numpy()
    a2 = torch.softmax(T(rng.standard_normal((3, 6)).astype(np.float32)), 1).numpy()
    graphs = mp.compose_he_adjs(mats, a1, a2, NA)
    want = O.metapath_he_adjs(mats, a1, a2, NA)
    for gr, w in zip(graphs, want):
        rowptr, col = gr.csr()           # CSR by destination of the edges row -> col
        wt = w.T.tocsr(); wt.sort_indices()
        np.testing.assert_array_equal(rowptr, wt.indptr)
        np.testing.assert_array_equal(col, wt.indices)


def test_spgemm_pattern_ragged_against_scipy():
    from diffgraph_amd import metapath as mp
    rng = np.random.default_rng(3)
    for (m, k, n, da, db) in [(1, 1, 1, 1.0, 1.0), (37, 50, 33, 0.1, 0.2), (300, 5, 4100, 0.5, 0.3), (129, 700, 65, 0.02, 0.0),
                              (64, 64, 262144, 0.05, 0.0002)]:
        A = sp.random(m, k, density=da, random_state=rng, format="csr", dtype=np.float32)
        B = sp.random(k, n, density=db, random_state=rng, format="csr", dtype=np.float32)
        got = mp.spgemm_pattern(mp.DevicePattern.from_scipy(A), mp.DevicePattern.from_scipy(B)).to_scipy()
        want = ((A != 0).astype(np.float32) @ (B != 0).astype(np.float32) != 0).astype(np.float32).tocsr()
        want.sort_indices()
        np.testing.assert_array_equal(got.indptr, want.indptr)
        np.testing.assert_array_equal(got.indices, want.indices)
    with pytest.raises(dg.DiffGraphHipError):
        mp.spgemm_pattern(mp.DevicePattern.from_scipy(sp.eye(4, format="csr")), mp.DevicePattern.from_scipy(sp.eye(4, 300000, format="csr")))


@pytest.mark.parametrize("k", [1, 7, 20])
def test_eval_exact_ties_go_to_the_lower_item_id(k):
    """Item tables with a handful of distinct rows repeated hundreds of times: most scores of a user tie EXACTLY.
    The kernel's lists must then be the stable order -- by falling score, equal scores by rising item id -- across
    tile, half-wavefront and item-range boundaries."""
    nu, ni = 200, 3000
    gen = torch.Generator().manual_seed(k)
    usr = torch.randn(nu, 64, generator=gen)
    base = torch.randn(9, 64, generator=gen)
    itm = base[torch.randint(0, 9, (ni,), generator=gen)]
    trn_u, trn_i = dgdata.synth_interactions(nu, ni, 5000, 8)
    tst_u, tst_i = np.arange(nu, dtype=np.int32), np.zeros(nu, np.int32)
    data = ev.AllRankTestData(coo(tst_u, tst_i, nu, ni), coo(trn_u, trn_i, nu, ni))
    vals, top = ev.all_rank_topk(usr.to(DEV), itm.to(DEV), data.test_users_dev, data, k)
    # the kernel's own fp32 scores decide what ties: recompute them on the device, item by distinct row
    sc = (usr.to(DEV) @ base.to(DEV).t()).cpu().double().numpy()          # [nu, 9] (any rounding is shared by equal rows)
    which = np.array([int(np.argmin(((base - itm[j]) ** 2).sum(1).numpy())) for j in range(ni)])
    got = top.cpu().numpy()
    mask = np.zeros((nu, ni), bool)
    mask[trn_u, trn_i] = True
    for r, u in enumerate(data.test_users):
        items = got[r]
        assert len(set(items.tolist())) == k and not mask[u, items].any()
        cls = which[items]
        # classes appear by falling score, and inside a class the ids ascend and are the class's lowest unmasked ids
        order = [c for j, c in enumerate(cls) if j == 0 or c != cls[j - 1]]
        assert len(order) == len(set(order)) and all(sc[u, a] >= sc[u, b] for a, b in zip(order, order[1:]))
        for c in set(cls.tolist()):
            mine = items[cls == c]
            avail = np.nonzero((which == c) & ~mask[u])[0]
            np.testing.assert_array_equal(mine, avail[: len(mine)])
            if c != order[-1]:
                assert len(mine) == len(avail)      # a better class is used up before the next one starts
