"""Meta-path graph construction on the GPU (DiffGraph_NC/DataHandler.py:147-174).

The reference mixes its six global relation matrices with softmax weights into Q1 and Q2, multiplies
them with scipy, binarises the product and keeps the author x author block (`Hc = Q1.dot(Q2); Hc = (Hc
!= 0); Haa = Hc[:NA, :NA]`).  Softmax weights are positive, so the block is the boolean product of the
relations' union pattern with itself; spgemm_pattern computes such products on the device, and
compose_he_adjs strings them into the graphs HGDM.forward takes.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import DiffGraphHipError
from .graph import CSRGraph


class DevicePattern:
    """A CSR pattern [n_rows x n_cols] on the GPU (int32 rowptr / col, ascending inside a row)."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, shape: Tuple[int, int]):
        self.rowptr, self.col, self.shape = rowptr, col, (int(shape[0]), int(shape[1]))

    @classmethod
    def from_scipy(cls, mat, device="cuda") -> "DevicePattern":
        m = mat.tocsr()
        m.sort_indices()
        dev = torch.device(device)
        col = m.indices.astype(np.int32) if m.nnz else np.zeros(1, np.int32)
        return cls(torch.from_numpy(m.indptr.astype(np.int32)).to(dev), torch.from_numpy(col).to(dev), m.shape)

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        rp = self.rowptr.cpu().numpy()
        col = self.col.cpu().numpy()[: rp[-1]]
        return sp.csr_matrix((np.ones(len(col), np.float32), col, rp), shape=self.shape)


def spgemm_pattern(a: DevicePattern, b: DevicePattern) -> DevicePattern:
    """Pattern of a . b  ((Q1.dot(Q2) != 0), DataHandler.py:158-161)."""
    if a.shape[1] != b.shape[0]:
        raise DiffGraphHipError(f"inner sizes differ: {a.shape} x {b.shape}")
    lib = _lib.load()
    dev = a.rowptr.device
    n_rows, n_mid, n_cols = a.shape[0], a.shape[1], b.shape[1]
    words = int(lib.dg_spgemm_bitmap_words(n_cols))
    bitmap = torch.empty((max(n_rows, 1), max(words, 1)), dtype=torch.int32, device=dev)
    row_nnz = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    rc = lib.dg_spgemm_pattern_rows(a.rowptr.data_ptr(), a.col.data_ptr(), n_rows, b.rowptr.data_ptr(), b.col.data_ptr(), n_mid,
                                    n_cols, bitmap.data_ptr(), row_nnz.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_spgemm_pattern_rows")
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(row_nnz, 0, dtype=torch.int64).to(torch.int32)
    total = int(rowptr[-1])
    col = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    rc = lib.dg_spgemm_pattern_fill(bitmap.data_ptr(), n_rows, n_cols, rowptr.data_ptr(), col.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "dg_spgemm_pattern_fill")
    return DevicePattern(rowptr, col, (n_rows, n_cols))


def compose(chain: Sequence[DevicePattern]) -> DevicePattern:
    """Pattern of chain[0] . chain[1] . ... (left to right), e.g. [AP, PC, CP, PA] for the APCPA meta-path."""
    out = chain[0]
    for m in chain[1:]:
        out = spgemm_pattern(out, m)
    return out


def compose_he_adjs(global_mats, alpha1: np.ndarray, alpha2: np.ndarray, n_target: int, device="cuda") -> List[CSRGraph]:
    """The per-channel graphs of LoadData option B (DataHandler.py:147-174): for channel c, Q1 / Q2 are the
    relations mixed with weights alpha1[c] / alpha2[c]; the graph is the pattern of Q1.Q2 restricted to the
    first n_target rows and columns.  Relations with a zero weight drop out of the mixture."""
    import scipy.sparse as sp
    graphs = []
    for c in range(alpha1.shape[0]):
        pats = []
        for alpha in (alpha1, alpha2):
            u = None
            for e, g in enumerate(global_mats):
                if alpha[c, e] != 0:
                    p = (g != 0).astype(np.float32).tocsr()
                    u = p if u is None else u + p
            pats.append(u)
        q1 = pats[0][:n_target, :]                       # only the first n_target rows of the product are kept
        q2 = pats[1][:, :n_target]                       # ... and its first n_target columns
        h = spgemm_pattern(DevicePattern.from_scipy(q1, device), DevicePattern.from_scipy(q2, device))
        graphs.append(CSRGraph.from_scipy(h.to_scipy()))
    return graphs
