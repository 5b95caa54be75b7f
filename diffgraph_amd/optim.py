"""Adam for the recommendation model (DiffGraph-Rec/main.py:117-119: t.optim.Adam(model.parameters(), lr=args.lr,
weight_decay=0)).

The embedding table -- 16 MB of the model's 16.1 MB -- is stepped by one HIP kernel that reads the gradient as
the tables the backward pass produced (their sum is never written); the denoiser weights go through torch's fused
multi-tensor Adam.  state_dict() / load_state_dict() carry both, so a run can be resumed (the reference's
loadModel re-creates Adam from scratch, main.py:205-211).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib


class Adam:
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.model = model
        self.table = model.embedding_table
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        rest = [p for p in model.parameters() if p is not self.table]
        self.rest = torch.optim.Adam(rest, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True, capturable=True)
        self.exp_avg = torch.zeros_like(self.table)
        self.exp_avg_sq = torch.zeros_like(self.table)
        self.state = torch.zeros(4, dtype=torch.float32, device=self.table.device)
        model.defer_embedding_grad = True   # the backward pass leaves the table's gradient as parts (see HGDM.propagate)
        model.embedding_grad_parts = None

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.rest.zero_grad(set_to_none=set_to_none)
        self.table.grad = None
        self.model.embedding_grad_parts = None

    def step(self) -> None:
        parts = self.model.embedding_grad_parts
        if parts is None and self.table.grad is not None:
            parts = [self.table.grad]
        lib = _lib.load()

        def small():    # the denoisers' update, and the table's step count and bias corrections
            self.rest.step()
            if parts is not None:
                _lib.check(lib.dg_adam_prepare(self.state.data_ptr(), self.betas[0], self.betas[1], _lib.stream_ptr(self.table.device)),
                           "dg_adam_prepare")

        late = getattr(self.model, "run_late_side_work", None)
        if late is not None:    # the denoisers' gradients may be left to be formed beside the backward pass: update them there
            late(then=small)
        else:
            small()
        if parts is None:
            return
        parts = [p.contiguous() for p in parts]
        arr = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        t = self.table.data
        rc = lib.dg_adam_step_prepared(t.data_ptr(), arr, len(parts), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                       self.state.data_ptr(), t.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                       self.weight_decay, _lib.stream_ptr(t.device))
        _lib.check(rc, "dg_adam_step_prepared")

    def state_dict(self) -> Dict:
        return {"table": {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "state": self.state}, "rest": self.rest.state_dict(),
                "hyper": {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}}

    def load_state_dict(self, sd: Dict) -> None:
        with torch.no_grad():
            self.exp_avg.copy_(sd["table"]["exp_avg"])
            self.exp_avg_sq.copy_(sd["table"]["exp_avg_sq"])
            self.state.copy_(sd["table"]["state"])
        self.rest.load_state_dict(sd["rest"])
        h = sd["hyper"]
        self.lr, self.betas, self.eps, self.weight_decay = h["lr"], tuple(h["betas"]), h["eps"], h["weight_decay"]
