"""All-rank evaluation with the reference's names (DiffGraph-Rec/main.py:136-187,
DiffGraph-Rec/DataHandler.py:250-272), on the GPU.

AllRankTestData keeps the reference's attributes (test_users, user_pos_lists, __len__) but holds the
training and test interactions as device CSR arrays instead of producing dense mask rows; testEpoch
scores every test user against every item, masks the training interactions, takes the top-k and
computes Recall@k / NDCG@k without the score matrix ever existing in memory.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import _lib
from ._lib import DG_D, DiffGraphHipError


def _csr(rows: np.ndarray, cols: np.ndarray, n_rows: int, sort_cols: bool):
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    order = np.lexsort((cols, rows)) if sort_cols else np.argsort(rows, kind="stable")
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n_rows), out=rowptr[1:])
    return rowptr.astype(np.int32), cols[order].astype(np.int32)


class AllRankTestData:
    """AllRankTestData(coomat, trn_mat) of DataHandler.py:250-272.  `coomat` / `trn_mat`: scipy sparse
    matrices [n_user, n_item] of test / training interactions (any nonzero counts)."""

    def __init__(self, coomat, trn_mat, device="cuda"):
        coomat = coomat.tocoo()
        trn = trn_mat.tocoo()
        n_user, n_item = coomat.shape
        keep = trn.data != 0                                   # (trn_mat.tocsr() != 0) * 1.0
        self.n_user, self.n_item = int(n_user), int(n_item)
        user_pos_lists: List[List[int]] = [list() for _ in range(n_user)]
        test_users = set()
        for row, col in zip(coomat.row.tolist(), coomat.col.tolist()):
            user_pos_lists[row].append(col)
            test_users.add(row)
        self.test_users = np.array(list(test_users))
        self.user_pos_lists = user_pos_lists
        trn_rowptr, trn_col = _csr(trn.row[keep], trn.col[keep], n_user, sort_cols=True)
        tst_rowptr, tst_col = _csr(coomat.row, coomat.col, n_user, sort_cols=False)   # stored order inside a row
        dev = torch.device(device)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.trn_rowptr, self.trn_col = T(trn_rowptr), T(trn_col if trn_col.size else np.zeros(1, np.int32))
        self.tst_rowptr, self.tst_col = T(tst_rowptr), T(tst_col if tst_col.size else np.zeros(1, np.int32))
        self.test_users_dev = T(self.test_users.astype(np.int64))

    def __len__(self):
        return len(self.test_users)


def all_rank_topk(usrEmbeds: torch.Tensor, itmEmbeds: torch.Tensor, users: torch.Tensor, data: AllRankTestData, topk: int = 20):
    """(values, topLocs) of t.topk(t.mm(usrEmbeds[users], itmEmbeds.T) * (1 - trnMask) - trnMask * 1e8, topk)
    (main.py:157-158) for int64 `users` on the GPU; topLocs int32 [len(users), topk]."""
    for nm, t in (("usrEmbeds", usrEmbeds), ("itmEmbeds", itmEmbeds)):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] < 1 or t.shape[1] != usrEmbeds.shape[1]:
            raise DiffGraphHipError(f"{nm}: expected a float32 GPU tensor [rows, latdim], both of one width")
    if usrEmbeds.shape[1] > DG_D:    # wider than the scoring kernel's rows: torch's mm + topk over mask rows built on the device
        from . import wide
        if usrEmbeds.shape[0] != data.n_user or itmEmbeds.shape[0] != data.n_item:
            raise DiffGraphHipError("embedding tables do not match the interaction matrices")
        return wide.all_rank_topk(usrEmbeds, itmEmbeds, users, data.trn_rowptr, data.trn_col, topk)
    if usrEmbeds.shape[1] < DG_D:    # a narrower latent space: zeros behind the rows add nothing to a score
        pad = (0, DG_D - usrEmbeds.shape[1])
        usrEmbeds, itmEmbeds = torch.nn.functional.pad(usrEmbeds, pad), torch.nn.functional.pad(itmEmbeds, pad)
    if usrEmbeds.shape[0] != data.n_user or itmEmbeds.shape[0] != data.n_item:
        raise DiffGraphHipError("embedding tables do not match the interaction matrices")
    if users.dtype != torch.int64 or not users.is_cuda or users.dim() != 1:
        raise DiffGraphHipError("users: expected a 1-D int64 GPU tensor")
    if users.numel() and (int(users.min()) < 0 or int(users.max()) >= data.n_user):
        raise DiffGraphHipError(f"users must lie in [0,{data.n_user})")
    usrEmbeds, itmEmbeds, users = usrEmbeds.contiguous(), itmEmbeds.contiguous(), users.contiguous()
    lib = _lib.load()
    dev = usrEmbeds.device
    n = users.numel()
    ws_bytes = int(lib.dg_eval_topk_workspace(n, data.n_item, topk))
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=dev)
    top_idx = torch.empty((n, topk), dtype=torch.int32, device=dev)
    top_val = torch.empty((n, topk), dtype=torch.float32, device=dev)
    rc = lib.dg_eval_topk(usrEmbeds.data_ptr(), itmEmbeds.data_ptr(), data.n_user, data.n_item, users.data_ptr(), n,
                          data.trn_rowptr.data_ptr(), data.trn_col.data_ptr(), int(topk), DG_D, ws.data_ptr(), ws_bytes,
                          top_idx.data_ptr(), top_val.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_eval_topk")
    return top_val, top_idx


def _dcg_tables(topk: int, device):
    # evaluated exactly as Coach.calcRes does (main.py:177,182): numpy float64
    inv = np.array([np.reciprocal(np.log2(loc + 2)) for loc in range(topk)], dtype=np.float64)
    mx = np.array([0.0] + [np.sum([np.reciprocal(np.log2(loc + 2)) for loc in range(t)]) for t in range(1, topk + 1)],
                  dtype=np.float64)
    return torch.from_numpy(inv).to(device), torch.from_numpy(mx).to(device)


def calcRes(topLocs: torch.Tensor, data: AllRankTestData, batIds: torch.Tensor):
    """Coach.calcRes(topLocs, tstLocs, batIds) (main.py:168-187) with tstLocs = data.user_pos_lists on the
    device.  Returns (allRecall, allNdcg, recall per user, ndcg per user); the two sums are added on the
    host in user order, as the reference adds them."""
    if topLocs.dtype != torch.int32 or not topLocs.is_cuda or topLocs.dim() != 2 or not topLocs.is_contiguous():
        raise DiffGraphHipError("topLocs: expected a contiguous int32 GPU tensor [users, k]")
    n, k = topLocs.shape
    if batIds.dtype != torch.int64 or batIds.numel() != n or not batIds.is_cuda:
        raise DiffGraphHipError("batIds: expected an int64 GPU tensor with one entry per row of topLocs")
    dev = topLocs.device
    inv, mx = _dcg_tables(k, dev)
    recall = torch.empty(n, dtype=torch.float64, device=dev)
    ndcg = torch.empty(n, dtype=torch.float64, device=dev)
    rc = _lib.load().dg_eval_metrics(topLocs.data_ptr(), n, k, batIds.contiguous().data_ptr(), data.tst_rowptr.data_ptr(),
                                     data.tst_col.data_ptr(), inv.data_ptr(), mx.data_ptr(), recall.data_ptr(), ndcg.data_ptr(),
                                     _lib.stream_ptr(dev))
    _lib.check(rc, "dg_eval_metrics")
    r, d = recall.cpu().numpy(), ndcg.cpu().numpy()
    return _seq_sum(r), _seq_sum(d), r, d


def _seq_sum(v: np.ndarray) -> float:
    """v[0] + v[1] + ... added one after the other in float64, as the reference's `+=` loops add them (np.cumsum is
    sequential; np.sum adds pairwise and can differ in the last bit)."""
    return float(np.cumsum(v, dtype=np.float64)[-1]) if v.size else 0


def testEpoch(model, data: AllRankTestData, topk: int = 20, tstBat: int = 1024) -> Dict[str, float]:
    """Coach.testEpoch (main.py:136-167): model.predict(), then batches of tstBat test users."""
    with torch.no_grad():
        usrEmbeds, itmEmbeds = model.predict()
    return rank_metrics(usrEmbeds, itmEmbeds, data, topk, tstBat)


def rank_metrics(usrEmbeds, itmEmbeds, data: AllRankTestData, topk: int = 20, tstBat: int = 1024) -> Dict[str, float]:
    """The evaluation loop of Coach.testEpoch (main.py:153-166) for given embeddings.  All test users go
    through one launch; the sums are formed per batch of tstBat users, as the reference forms them."""
    num = len(data)
    _, topLocs = all_rank_topk(usrEmbeds, itmEmbeds, data.test_users_dev, data, topk)
    _, _, r, d = calcRes(topLocs, data, data.test_users_dev)
    epRecall = epNdcg = 0
    for b in range(0, num, tstBat):
        epRecall += _seq_sum(r[b: b + tstBat])
        epNdcg += _seq_sum(d[b: b + tstBat])
    return {"Recall": epRecall / num, "NDCG": epNdcg / num, "topLocs": topLocs, "recall_per_user": r, "ndcg_per_user": d}
