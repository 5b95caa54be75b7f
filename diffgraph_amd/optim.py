"""Adam for the recommendation model (DiffGraph-Rec/main.py:117-119: t.optim.Adam(model.parameters(), lr=args.lr,
weight_decay=0)).

The embedding table -- 16 MB of the model's 16.1 MB -- is stepped by one HIP kernel that reads the gradient as
the tables the backward pass produced (their sum is never written); the twelve denoiser tensors by one launch of
dg_adam_multi (SmallAdam) with its step count on the device -- no torch kernel is left in a recorded step.
state_dict() / load_state_dict() carry both, so a run can be resumed (the reference's loadModel re-creates Adam
from scratch, main.py:205-211).
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
        self.rest = SmallAdam(rest, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.exp_avg = torch.zeros_like(self.table)
        self.exp_avg_sq = torch.zeros_like(self.table)
        self.state = torch.zeros(4, dtype=torch.float32, device=self.table.device)
        model.defer_embedding_grad = True   # the backward pass leaves the table's gradient as parts (see HGDM.propagate)
        model.embedding_grad_parts = None

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.rest.zero_grad(set_to_none=set_to_none)
        self.table.grad = None
        self.model.embedding_grad_parts = None

    def step_table_in(self, launch) -> None:
        """For a backward pass that forms the table's gradient row by row and can update the table on the spot
        (HGDM under `table_step_in_backward`): `launch(param, exp_avg, exp_avg_sq, state, lr, beta1, beta2, eps,
        weight_decay)` runs it and moves the step count on; step() then leaves the table alone."""
        launch(self.table.data, self.exp_avg, self.exp_avg_sq, self.state, self.lr, self.betas[0], self.betas[1], self.eps,
               self.weight_decay)

    def step(self) -> None:
        parts = self.model.embedding_grad_parts
        if parts is None and self.table.grad is not None:
            parts = [self.table.grad]
        lib = _lib.load()

        def small():    # the denoisers' update, and the table's step count and bias corrections
            self.rest.step()
            if parts is not None and not isinstance(parts, str):    # (a backward pass that stepped the table moved the count on too)
                _lib.check(lib.dg_adam_prepare(self.state.data_ptr(), self.betas[0], self.betas[1], _lib.stream_ptr(self.table.device)),
                           "dg_adam_prepare")

        late = getattr(self.model, "run_late_side_work", None)
        if late is not None:    # the denoisers' gradients may be left to be formed beside the backward pass: update them there
            late(then=small)
        else:
            small()
        if parts is None or isinstance(parts, str):     # no gradient, or the backward pass has stepped the table already
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


class SmallAdam:
    """torch.optim.Adam(params, lr, weight_decay=0) (DiffGraph_NC/main.py:356-357) for a handful of small parameter
    tensors: ONE launch per step for all of them (`dg_adam_multi`), the step count on the device, so a recorded
    training step can be replayed.  Parameters without a gradient are left alone, as torch does."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        if any(p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() for p in self.params):
            raise _lib.DiffGraphHipError("SmallAdam: parameters must be contiguous float32 GPU tensors")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.state = torch.zeros(4, dtype=torch.float32, device=self.params[0].device)
        self._live = None       # which parameters had a gradient in the first step (see step)

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self) -> None:
        mask = tuple(p.grad is not None for p in self.params)
        if not any(mask):
            return
        # ONE step count for all tensors: a tensor that joined or left between steps would get another tensor's bias
        # correction (torch keeps a count per tensor), so the set of tensors with a gradient is fixed by the first step
        if self._live is None:
            self._live = mask
        elif mask != self._live:
            k = next(i for i, (a, b) in enumerate(zip(mask, self._live)) if a != b)
            raise _lib.DiffGraphHipError(f"SmallAdam: parameter {k} {'has' if mask[k] else 'lacks'} a gradient, unlike in the steps "
                                         "before: the tensors share one step count")
        live = [(p, p.grad.contiguous(), m, v) for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq) if p.grad is not None]
        n = len(live)
        if n > 32:
            raise _lib.DiffGraphHipError("SmallAdam: at most 32 tensors per step (one launch, one step count)")
        arrs = [(C.c_void_p * n)(*[t[k].data_ptr() for t in live]) for k in range(4)]
        counts = (C.c_int64 * n)(*[t[0].numel() for t in live])
        rc = _lib.load().dg_adam_multi(arrs[0], arrs[1], arrs[2], arrs[3], counts, n, self.state.data_ptr(), self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                       _lib.stream_ptr(self.state.device))
        _lib.check(rc, "dg_adam_multi")

    def state_dict(self) -> Dict:
        return {"exp_avg": list(self.exp_avg), "exp_avg_sq": list(self.exp_avg_sq), "state": self.state,
                "hyper": {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}}

    def load_state_dict(self, sd: Dict) -> None:
        if "param_groups" in sd:      # a torch.optim.Adam state_dict (checkpoints written before this class stepped the tensors)
            st = sd["state"]
            steps = {float(v["step"]) for v in st.values()}
            if len(st) not in (0, len(self.params)) or len(steps) > 1:
                raise _lib.DiffGraphHipError("SmallAdam: a torch Adam state loads only with one step count over all tensors")
            with torch.no_grad():
                for k in st:
                    self.exp_avg[int(k)].copy_(st[k]["exp_avg"])
                    self.exp_avg_sq[int(k)].copy_(st[k]["exp_avg_sq"])
                self.state.zero_()
                self.state[0] = steps.pop() if steps else 0.0
            return
        with torch.no_grad():
            for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd["exp_avg"]) + list(sd["exp_avg_sq"])):
                dst.copy_(src)
            self.state.copy_(sd["state"])
        h = sd["hyper"]
        self.lr, self.betas, self.eps, self.weight_decay = h["lr"], tuple(h["betas"]), h["eps"], h["weight_decay"]
