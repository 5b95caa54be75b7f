"""PairwiseTrnData with the reference's interface (DiffGraph-Rec/DataHandler.py:274-296), on the GPU.

The reference keeps the training pairs in numpy, draws one negative per pair in a Python loop once per
epoch (negSampling) and feeds (user, positive, negative) triples through a shuffling DataLoader with
batch size args.batch (DataHandler.py:150-151).  Here the pairs, the training CSR and the negatives live
on the GPU; negSampling() is one kernel launch and batches() yields the shuffled triples as int64 GPU
tensors, the form HGDM.cal_loss takes.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .evaluate import _csr


class PairwiseTrnData:
    def __init__(self, coomat, device="cuda", seed: int = 0):
        coomat = coomat.tocoo()
        self.rows = np.asarray(coomat.row, dtype=np.int32)
        self.cols = np.asarray(coomat.col, dtype=np.int32)
        self.n_user, self.n_item = (int(v) for v in coomat.shape)
        dev = torch.device(device)
        # the look-up rows hold every (user, item) once: a pair stored twice must not make a user look saturated
        key = np.unique(self.rows.astype(np.int64) * self.n_item + self.cols)
        rowptr, col = _csr(key // self.n_item, key % self.n_item, self.n_user, sort_cols=True)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.trn_rowptr, self.trn_col = T(rowptr), T(col if col.size else np.zeros(1, np.int32))
        self.rows_dev, self.cols_dev = T(self.rows), T(self.cols)
        self.negs_dev = torch.zeros(len(self.rows), dtype=torch.int32, device=dev)
        self.seed, self.epoch = int(seed), 0

    @property
    def negs(self) -> np.ndarray:
        return self.negs_dev.cpu().numpy()

    def negSampling(self, seed: Optional[int] = None) -> None:
        """One negative per training pair (DataHandler.py:282-289); a new stream of draws on every call."""
        if seed is None:
            seed = (self.seed << 32) + self.epoch
            self.epoch += 1
        dev = self.rows_dev.device
        rc = _lib.load().dg_neg_sampling(self.trn_rowptr.data_ptr(), self.trn_col.data_ptr(), self.rows_dev.data_ptr(),
                                         len(self.rows), self.n_user, self.n_item, int(seed) & (2 ** 64 - 1),
                                         self.negs_dev.data_ptr(), _lib.stream_ptr(dev))
        _lib.check(rc, "dg_neg_sampling")
        if len(self.rows) and int(self.negs_dev.min()) < 0:
            u = int(self.rows_dev[self.negs_dev < 0][0])
            raise ValueError(f"negSampling: user {u} has interacted with all {self.n_item} items, there is no negative to draw "
                             "(the reference's loop, DataHandler.py:284-288, would not end)")

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        return self.rows[idx], self.cols[idx], int(self.negs_dev[idx])

    def batches(self, batch: int, shuffle: bool = True, generator: Optional[torch.Generator] = None
                ) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """(ancs, poss, negs) int64 GPU tensors of up to `batch` pairs each, in a fresh random order --
        what the reference's DataLoader(batch_size=args.batch, shuffle=True) delivers (DataHandler.py:151)."""
        n = len(self)
        dev = self.rows_dev.device
        order = torch.randperm(n, device=dev, generator=generator) if shuffle else torch.arange(n, device=dev)
        for b in range(0, n, batch):
            ids = order[b: b + batch]
            yield self.rows_dev[ids].long(), self.cols_dev[ids].long(), self.negs_dev[ids].long()
