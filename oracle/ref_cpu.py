"""CPU restatement of the DiffGraph hot path (TEST INFRASTRUCTURE ONLY).

This module is the parity oracle: a plain torch-CPU restatement of the reference's
normalised-adjacency propagation, layer-sum aggregation, Gaussian diffusion tables,
q_sample / p_sample / training_losses2 and the denoiser MLP.  It is imported only by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the
product package (diffgraph_amd), which must fail loudly when the HIP library is absent.

Pinning: every function here is checked against golden vectors captured by running the
reference's own Model.py in the build container (tools/make_golden.py ->
tests/golden/*.npz, tests/test_oracle_golden.py).  The one piece of arithmetic that lives
in an absent third-party dependency is DGL's update_all(copy_u, sum) (dgl==1.0.2,
reference README.md:36); it is restated from DGL's documented semantics
(out[dst] += x[src] over all edges; out_degrees/in_degrees = edge counts by src/dst) and
the reference holds no test of its own at that boundary, so SpMM parity is "unpinned" in
that strict sense and pinned everywhere else.

Every function cites the reference file:line it follows (paths relative to the reference
checkout, DiffGraph-Rec/ unless stated).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# a1: one propagation layer -- DGLLayer.forward Model.py:142-174,
#     UUGCNLayer.forward DiffGraph_NC/Model.py:184-214
# --------------------------------------------------------------------------------------
def degrees(src: torch.Tensor, dst: torch.Tensor, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """out_degrees (edge count by source) and in_degrees (by destination), int64."""
    out_deg = torch.bincount(src.long(), minlength=n)
    in_deg = torch.bincount(dst.long(), minlength=n)
    return out_deg, in_deg


def gcn_layer(src: torch.Tensor, dst: torch.Tensor, n: int, node_f: torch.Tensor,
              slope: Optional[float] = 0.5) -> torch.Tensor:
    """Y = act( in_norm * scatter_add_dst( (node_f * out_norm)[src] ) ).

    Model.py:151-152 out-degree norm, :157 pre-scale, :161 copy_u/sum message passing,
    :165-169 in-degree norm + post-scale, :171-172 LeakyReLU(0.5).  weight=False always
    (Model.py:30) so the u_w/v_w branch is not restated.
    """
    out_deg, in_deg = degrees(src, dst, n)
    dt = node_f.dtype
    norm_out = torch.pow(out_deg.to(torch.float32).clamp(min=1), -0.5).to(dt).view(-1, 1)
    norm_in = torch.pow(in_deg.to(torch.float32).clamp(min=1), -0.5).to(dt).view(-1, 1)
    if dt == torch.float64:  # fp64 mode recomputes the norms in fp64 (error-bound oracle)
        norm_out = torch.pow(out_deg.to(dt).clamp(min=1), -0.5).view(-1, 1)
        norm_in = torch.pow(in_deg.to(dt).clamp(min=1), -0.5).view(-1, 1)
    h = node_f * norm_out
    rst = torch.zeros_like(h).index_add_(0, dst.long(), h[src.long()])
    rst = rst * norm_in
    if slope is not None:
        rst = F.leaky_relu(rst, slope)
    return rst


# --------------------------------------------------------------------------------------
# a2: Rec HGDM.forward Model.py:53-92
# --------------------------------------------------------------------------------------
def rec_forward(target: Tuple[torch.Tensor, torch.Tensor],
                behaviours: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                user_emb: torch.Tensor, item_emb: torch.Tensor, n_layers: int = 2):
    """Returns (target_user, target_item, heter_user, heter_item).

    The raw (un-normalised) activation feeds the next layer (Model.py:65,79); sums are
    Python left-to-right sums starting from int 0 (Model.py:70,84,87).
    """
    n_user = user_emb.shape[0]
    n = n_user + item_emb.shape[0]
    e0 = torch.cat([user_emb, item_emb], dim=0)

    def one_graph(g):
        acc = [e0]
        emb = e0
        for _ in range(n_layers):
            emb = gcn_layer(g[0], g[1], n, emb)
            acc.append(F.normalize(emb, p=2, dim=1))
        return sum(acc)

    ui = one_graph(target)
    het = sum(one_graph(g) for g in behaviours)
    return ui[:n_user], ui[n_user:], het[:n_user], het[n_user:]


# --------------------------------------------------------------------------------------
# a3: NC HGDM.forward DiffGraph_NC/Model.py:44-73
# --------------------------------------------------------------------------------------
def nc_forward(he_adjs: Sequence[Tuple[torch.Tensor, torch.Tensor]], features: torch.Tensor,
               w: torch.Tensor, b: torch.Tensor, n_layers: int = 3):
    """Returns (source1, source2, target): he_adjs[1], he_adjs[2], he_adjs[0] channels.

    The NORMALISED activation feeds the next layer (DiffGraph_NC/Model.py:53,60,67).
    """
    n = features.shape[0]
    embed = F.linear(features, w, b)

    def one_graph(g):
        acc = [embed]
        for _ in range(n_layers):
            emb = gcn_layer(g[0], g[1], n, acc[-1])
            acc.append(F.normalize(emb, p=2, dim=1))
        return sum(acc)

    target = one_graph(he_adjs[0])
    s1 = one_graph(he_adjs[1])
    s2 = one_graph(he_adjs[2])
    return s1, s2, target


# --------------------------------------------------------------------------------------
# a5: diffusion tables -- GaussianDiffusion.__init__/get_betas/calculate_for_diffusion
#     Model.py:237-285 (fp64 on host)
# --------------------------------------------------------------------------------------
class DiffusionTables:
    def __init__(self, noise_scale: float, noise_min: float, noise_max: float, steps: int,
                 beta_fixed: bool = True):
        self.steps = steps
        self.noise_scale = noise_scale
        start = noise_scale * noise_min
        end = noise_scale * noise_max
        variance = np.linspace(start, end, steps, dtype=np.float64)       # Model.py:260
        alpha_bar = 1 - variance
        betas = [1 - alpha_bar[0]]
        for i in range(1, steps):
            betas.append(min(1 - alpha_bar[i] / alpha_bar[i - 1], 0.999))  # Model.py:265
        betas = torch.tensor(np.array(betas), dtype=torch.float64)
        if beta_fixed:
            betas[0] = 0.0001                                               # Model.py:252-253
        self.betas = betas
        alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(alphas, dim=0)
        self.alphas_cumprod_prev = torch.cat(
            [torch.tensor([1.0], dtype=torch.float64), self.alphas_cumprod[:-1]])
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = torch.log(
            torch.cat([self.posterior_variance[1].unsqueeze(0), self.posterior_variance[1:]]))
        self.posterior_mean_coef1 = betas * torch.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * torch.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    def snr(self, t: torch.Tensor) -> torch.Tensor:                        # Model.py:371-373
        return self.alphas_cumprod[t] / (1 - self.alphas_cumprod[t])

    def loss_weight(self, ts: torch.Tensor) -> torch.Tensor:
        """SNR(ts-1)-SNR(ts) in fp64, 1.0 where ts==0 (index -1 wraps, then masked) Model.py:362-363."""
        w = self.snr(ts - 1) - self.snr(ts)
        return torch.where(ts == 0, 1.0, w)


def extract(arr: torch.Tensor, t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """_extract_into_tensor Model.py:307-312: index the fp64 table, THEN cast to fp32."""
    return arr[t].to(dtype).view(-1, 1)


# a6: q_sample Model.py:302-305
def q_sample(tab: DiffusionTables, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor):
    dt = x_start.dtype
    return extract(tab.sqrt_alphas_cumprod, t, dt) * x_start + \
        extract(tab.sqrt_one_minus_alphas_cumprod, t, dt) * noise


# --------------------------------------------------------------------------------------
# a7: Denoise.forward Model.py:214-234 (eval mode: dropout is the identity)
# --------------------------------------------------------------------------------------
class DenoiseWeights:
    """emb_layer (8x8), in_layers[0] (64x72), out_layers[0] (64x64), torch [out,in] layout."""

    def __init__(self, w_emb, b_emb, w_in, b_in, w_out, b_out):
        self.w_emb, self.b_emb = w_emb, b_emb
        self.w_in, self.b_in = w_in, b_in
        self.w_out, self.b_out = w_out, b_out

    @staticmethod
    def init(latdim: int = 64, emb: int = 8, seed: int = 3, dtype=torch.float32) -> "DenoiseWeights":
        """Reference init (Model.py:196-212): normal(0, sqrt(2/(fan_in+fan_out))), bias normal(0,1e-3)."""
        g = torch.Generator().manual_seed(seed)

        def lin(o, i):
            std = math.sqrt(2.0 / (o + i))
            return (torch.randn(o, i, generator=g) * std).to(dtype), (torch.randn(o, generator=g) * 1e-3).to(dtype)

        w_in, b_in = lin(latdim, latdim + emb)
        w_out, b_out = lin(latdim, latdim)
        w_emb, b_emb = lin(emb, emb)
        return DenoiseWeights(w_emb, b_emb, w_in, b_in, w_out, b_out)

    def to(self, dtype):
        return DenoiseWeights(*[x.to(dtype) for x in
                                (self.w_emb, self.b_emb, self.w_in, self.b_in, self.w_out, self.b_out)])


def time_embedding(timesteps: torch.Tensor, dim: int = 8, dtype=torch.float32) -> torch.Tensor:
    """Model.py:215-217: freqs = exp(-ln(1e4) * arange(dim/2) / (dim/2)); [cos, sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    temp = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(temp), torch.sin(temp)], dim=-1).to(dtype)


def denoise_forward(wts: DenoiseWeights, x: torch.Tensor, timesteps: torch.Tensor, norm: bool = True,
                    keep_mask: Optional[torch.Tensor] = None, drop_p: float = 0.5):
    """keep_mask ([n, 64], nonzero = kept): the module in training mode with mess_dropout (Model.py:223-224) --
    nn.Dropout multiplies the kept elements by 1/(1-p) and zeroes the others; the mask is passed in (torch draws it)."""
    dt = x.dtype
    te = time_embedding(timesteps, wts.w_emb.shape[1], dt)
    emb = F.linear(te, wts.w_emb, wts.b_emb)                  # Model.py:220
    if norm:
        x = F.normalize(x)                                    # Model.py:221-222
    if keep_mask is not None:
        x = x * (keep_mask != 0).to(dt).div(1.0 - drop_p)     # Model.py:223-224 (the noise tensor of at::dropout)
    h = torch.cat([x, emb], dim=-1)                           # Model.py:225
    h = torch.tanh(F.linear(h, wts.w_in, wts.b_in))           # Model.py:226-228
    return F.linear(h, wts.w_out, wts.b_out)                  # Model.py:229-232 (single out layer: no tanh)


# a8: p_mean_variance Model.py:314-325, p_sample Model.py:287-300
def p_sample(tab: DiffusionTables, wts: DenoiseWeights, x_start: torch.Tensor, steps: int = 0,
             noise: Optional[torch.Tensor] = None, keep: Sequence[int] = ()):
    """Always runs all tab.steps reverse iterations (Model.py:294).  `keep` lists step
    indices whose post-update x is also returned (for golden intermediates)."""
    dt = x_start.dtype
    n = x_start.shape[0]
    if steps == 0:
        x_t = x_start
    else:
        t = torch.full((n,), steps - 1, dtype=torch.long)
        x_t = q_sample(tab, x_start, t, noise)
    kept = {}
    for i in range(tab.steps - 1, -1, -1):
        t = torch.full((n,), i, dtype=torch.long)
        out = denoise_forward(wts, x_t, t)
        x_t = extract(tab.posterior_mean_coef1, t, dt) * out + extract(tab.posterior_mean_coef2, t, dt) * x_t
        if i in keep:
            kept[i] = x_t.clone()
    return (x_t, kept) if keep else x_t


# a9: training_losses2 Model.py:348-366
def training_losses2(tab: DiffusionTables, wts: DenoiseWeights, target: torch.Tensor, x_start: torch.Tensor,
                     batch: torch.Tensor, ts: torch.Tensor, noise: torch.Tensor, keep_mask: Optional[torch.Tensor] = None):
    """ts / noise are passed in (the reference draws them at Model.py:352-353); keep_mask: training mode."""
    x_t = q_sample(tab, x_start, ts, noise) if tab.noise_scale != 0 else x_start
    out = denoise_forward(wts, x_t, ts, keep_mask=keep_mask)
    mse = ((target - out) ** 2).mean(dim=1)
    loss = tab.loss_weight(ts) * mse                          # fp64 weight * fp32 mse -> fp64
    return loss[batch], out


# a11: HGDM.cal_loss DiffGraph-Rec/Model.py:94-111 (pairPredict: Utils/Utils.py, (anc*pos).sum(-1) - (anc*neg).sum(-1))
def rec_cal_loss(tab: DiffusionTables, w_usr: DenoiseWeights, w_item: DenoiseWeights, target, behaviours,
                 user_emb: torch.Tensor, item_emb: torch.Tensor, ancs, poss, negs, ts_u, noise_u, ts_i, noise_i,
                 batch: float, reg: float, keep_u: Optional[torch.Tensor] = None, keep_i: Optional[torch.Tensor] = None,
                 n_layers: int = 2):
    """Returns (loss, bpr, reg, diffusion loss); the draws of both training_losses2 calls (and, in training mode,
    both dropout keep masks) are arguments.  batch / reg are args.batch / args.reg of the reference."""
    tu, ti, hu, hi = rec_forward(target, behaviours, user_emb, item_emb, n_layers)
    lu, du = training_losses2(tab, w_usr, tu, hu, ancs, ts_u, noise_u, keep_u)
    li, di = training_losses2(tab, w_item, ti, hi, poss, ts_i, noise_i, keep_i)
    diff = lu.mean() + li.mean()
    usr, itm = tu + du, ti + di
    a, p, n = usr[ancs], itm[poss], itm[negs]
    score = (a * p).sum(-1) - (a * n).sum(-1)
    bpr = -score.sigmoid().log().sum() / batch
    regl = ((torch.norm(a) ** 2 + torch.norm(p) ** 2 + torch.norm(n) ** 2) * reg) / batch
    return bpr + regl + diff, bpr, regl, diff


# --------------------------------------------------------------------------------------
# Dense normalised-adjacency formulation (used for the cpu_baseline leg): Y = lrelu(A_hat X)
# --------------------------------------------------------------------------------------
def normalised_adj_csr(src: torch.Tensor, dst: torch.Tensor, n: int) -> torch.Tensor:
    """fp32 CSR of A_hat[dst, src] = in_norm[dst]*out_norm[src] (what DataHandler.py:103-118 builds as COO)."""
    out_deg, in_deg = degrees(src, dst, n)
    no = torch.pow(out_deg.float().clamp(min=1), -0.5)
    ni = torch.pow(in_deg.float().clamp(min=1), -0.5)
    vals = ni[dst.long()] * no[src.long()]
    coo = torch.sparse_coo_tensor(torch.stack([dst.long(), src.long()]), vals, (n, n)).coalesce()
    return coo.to_sparse_csr()


# --------------------------------------------------------------------------------------
# N1: all-rank evaluation -- Coach.testEpoch main.py:147-167, Coach.calcRes main.py:168-187,
#     AllRankTestData DataHandler.py:250-272
# --------------------------------------------------------------------------------------
def test_users_and_lists(n_user: int, tst_u: np.ndarray, tst_i: np.ndarray) -> Tuple[np.ndarray, List[List[int]]]:
    """AllRankTestData.__init__ (DataHandler.py:251-263): per-user lists of test items in the order
    the test matrix stores them, and the test users as np.array(list(set(...)))."""
    user_pos_lists: List[List[int]] = [list() for _ in range(n_user)]
    test_users = set()
    for row, col in zip(tst_u.tolist(), tst_i.tolist()):
        user_pos_lists[row].append(col)
        test_users.add(row)
    return np.array(list(test_users)), user_pos_lists


def all_rank_topk(usr_emb: torch.Tensor, item_emb: torch.Tensor, users: torch.Tensor, trn_mask: torch.Tensor, topk: int = 20):
    """main.py:157-158: scores of a batch of users against every item, training interactions pushed
    to -1e8, then torch.topk (sorted).  trn_mask: dense {0,1} float [len(users), n_item]."""
    all_preds = torch.mm(usr_emb[users], torch.transpose(item_emb, 1, 0)) * (1 - trn_mask) - trn_mask * 1e8
    vals, top_locs = torch.topk(all_preds, topk)
    return vals, top_locs


def calc_res(top_locs: np.ndarray, tst_locs: List[List[int]], bat_ids: Sequence[int], topk: int = 20):
    """Coach.calcRes main.py:168-187: per-user recall@k and NDCG@k (numpy float64), and their
    sums accumulated in user order exactly as the reference does.  Returns (allRecall, allNdcg,
    per-user recall, per-user ndcg)."""
    all_recall = all_ndcg = 0
    recalls, ndcgs = [], []
    for i in range(len(bat_ids)):
        tem_top = list(top_locs[i])
        tem_tst = tst_locs[bat_ids[i]]
        tst_num = len(tem_tst)
        max_dcg = np.sum([np.reciprocal(np.log2(loc + 2)) for loc in range(min(tst_num, topk))])
        recall = dcg = 0
        for val in tem_tst:
            if val in tem_top:
                recall += 1
                dcg += np.reciprocal(np.log2(tem_top.index(val) + 2))
        recall = recall / tst_num
        ndcg = dcg / max_dcg
        all_recall += recall
        all_ndcg += ndcg
        recalls.append(recall)
        ndcgs.append(ndcg)
    return all_recall, all_ndcg, np.asarray(recalls, dtype=np.float64), np.asarray(ndcgs, dtype=np.float64)


def test_epoch(usr_emb: torch.Tensor, item_emb: torch.Tensor, trn_u: np.ndarray, trn_i: np.ndarray, tst_u: np.ndarray,
               tst_i: np.ndarray, topk: int = 20, tst_bat: int = 1024):
    """Coach.testEpoch main.py:147-167 after model.predict(): batches of tst_bat test users in
    AllRankTestData order, dense train-mask rows (DataHandler.py:268-272), top-k, calcRes; the epoch
    metrics are the per-batch sums added up and divided by the number of test users.
    Returns dict(Recall, NDCG, users, top_locs, recall_per_user, ndcg_per_user)."""
    n_user, n_item = usr_emb.shape[0], item_emb.shape[0]
    users, lists = test_users_and_lists(n_user, tst_u, tst_i)
    mask = np.zeros((n_user, n_item), dtype=np.float32)   # (trn_mat.tocsr() != 0) * 1.0, DataHandler.py:252
    mask[trn_u, trn_i] = 1.0
    ep_recall = ep_ndcg = 0
    tops, recs, nds = [], [], []
    for b in range(0, len(users), tst_bat):
        usr = users[b: b + tst_bat]
        _, top_locs = all_rank_topk(usr_emb, item_emb, torch.from_numpy(usr).long(), torch.from_numpy(mask[usr]), topk)
        r, n, rr, nn = calc_res(top_locs.numpy(), lists, usr, topk)
        ep_recall += r
        ep_ndcg += n
        tops.append(top_locs.numpy()); recs.append(rr); nds.append(nn)
    num = len(users)
    return {"Recall": ep_recall / num, "NDCG": ep_ndcg / num, "users": users, "top_locs": np.concatenate(tops),
            "recall_per_user": np.concatenate(recs), "ndcg_per_user": np.concatenate(nds)}


# --------------------------------------------------------------------------------------
# N3: negative sampling -- PairwiseTrnData.negSampling DataHandler.py:282-289
# --------------------------------------------------------------------------------------
def neg_sampling(rows: np.ndarray, cols: np.ndarray, n_item: int, rng: np.random.Generator) -> np.ndarray:
    """One negative per training pair: draw item ids uniformly until the (user, item) pair is not a
    training interaction.  The reference draws from numpy's global generator (np.random.randint);
    a Generator is passed here so that tests are repeatable -- only the distribution is comparable."""
    seen = set(zip(rows.tolist(), cols.tolist()))      # coomat.todok() membership
    negs = np.zeros(len(rows), dtype=np.int32)
    for i in range(len(rows)):
        u = int(rows[i])
        while True:
            i_neg = int(rng.integers(n_item))
            if (u, i_neg) not in seen:
                break
        negs[i] = i_neg
    return negs


# --------------------------------------------------------------------------------------
# N2: meta-path composition -- DiffGraph_NC/DataHandler.py:91-165 (LoadData, option B)
# --------------------------------------------------------------------------------------
def dblp_global_mats(ap: np.ndarray, pc: np.ndarray, pt: np.ndarray, counts):
    """The six global relation matrices of DataHandler.py:91-116 (A->P, P->A, C->P, P->C, P->T, T->P on a
    common node numbering authors | papers | conferences | terms), all values 1."""
    import scipy.sparse as sp
    NA, NP, NC, NT = (int(v) for v in counts)
    off_A, off_P, off_C, off_T = 0, NA, NA + NP, NA + NP + NC
    n_tot = NA + NP + NC + NT
    def make_global(rows, cols, src, dst):
        return sp.coo_matrix((np.ones(len(rows), np.float32), (rows + src, cols + dst)), shape=(n_tot, n_tot))
    return [make_global(ap[:, 0], ap[:, 1], off_A, off_P), make_global(ap[:, 1], ap[:, 0], off_P, off_A),
            make_global(pc[:, 1], pc[:, 0], off_C, off_P), make_global(pc[:, 0], pc[:, 1], off_P, off_C),
            make_global(pt[:, 0], pt[:, 1], off_P, off_T), make_global(pt[:, 1], pt[:, 0], off_T, off_P)]


def metapath_he_adjs(global_mats, alpha1: np.ndarray, alpha2: np.ndarray, n_target: int):
    """DataHandler.py:150-165: per channel Q1 = sum_e alpha1[c,e] G_e, Q2 likewise, Hc = Q1.dot(Q2),
    binarised, sliced to the first n_target rows and columns (scipy CSR, values 1)."""
    out = []
    for c in range(alpha1.shape[0]):
        Q1 = sum(global_mats[e] * alpha1[c, e] for e in range(len(global_mats)))
        Q2 = sum(global_mats[e] * alpha2[c, e] for e in range(len(global_mats)))
        Hc = Q1.dot(Q2)
        Hc = (Hc != 0).astype(np.float32)
        Haa = Hc.tocsr()[:n_target, :n_target]
        Haa.data[:] = 1.0
        Haa.eliminate_zeros()
        Haa.sort_indices()
        out.append(Haa)
    return out
