"""Coach of the reference's DiffGraph-Rec/main.py:34-215 -- trainEpoch / testEpoch / run -- over the HIP path.

What differs from the reference's loop, and why the results do not:
* the training triples never leave the GPU: negSampling() is one kernel, the epoch's shuffled order one randperm,
  a step's (ancs, poss, negs) a slice of the shuffled table taken inside the recorded step;
* a step (cal_loss, backward, Adam) is captured once per batch size as a HIP graph and replayed -- two recordings:
  the full batch and the epoch's short last batch;
* the reference reads three losses back per step (`.item()`, main.py:122-124); here they are added up on the device
  and read once per epoch -- the same sums in the same order, in the losses' own precision (float64).
The handler is any object with userNum, itemNum, target_adj, behavior_mats (CSRGraph), trn_data (PairwiseTrnData)
and tst_data (AllRankTestData): the fields Coach and HGDM read from the reference's DataHandler.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import evaluate, io
from .model_rec import DEFAULT_ARGS, HGDM
from .optim import Adam


def training_step(model, opt, ancs, poss, negs, seed: torch.Tensor, draws: Optional[dict] = None) -> torch.Tensor:
    """One step of Coach.trainEpoch's loop (main.py:120-128): cal_loss, backward, Adam.  `seed` is the float64 device
    vector [1, 0, 0, 0] = d loss / d (loss, bprLoss, regLoss, diff_loss).  `draws`: the step's random inputs as
    keyword arguments of HGDM.cal_loss_vector (ts_u, noise_u, ts_i, noise_i, keep_u, keep_i) when they are not to
    be drawn inside.  Returns the four values (float64[4])."""
    # What nothing on the way to the table's update waits for -- the sums of the four loss values, the denoisers'
    # weight gradients and their Adam step -- may run beside the propagation: opt.step() joins before it returns.
    model.side_join_deferred = isinstance(opt, Adam)
    model._late_side = []            # (nothing of an abandoned step)
    # ... and backward() is followed by step() at once: the last launch of the backward pass may update the embedding
    # table itself, row by row as it forms the gradient (HGDM's propagation over a graph set does)
    model.table_step_in_backward = opt if isinstance(opt, Adam) else None
    try:
        res = model.cal_loss_vector(ancs, poss, negs, **(draws or {}))
        opt.zero_grad(set_to_none=True)
        res.backward(seed)                                       # = loss.backward()
    finally:
        model.side_join_deferred = False
        model.table_step_in_backward = None
    opt.step()
    return res


class Coach:
    def __init__(self, handler, args=None, device="cuda", use_graph: bool = True, seed: int = 0):
        self.handler, self.args, self.device = handler, args or DEFAULT_ARGS, torch.device(device)
        self.use_graph = use_graph
        self.metrics: Dict[str, list] = dict()
        self.model: Optional[HGDM] = None
        self.opt: Optional[Adam] = None
        self._graphs = {}
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self._draws = None
        self._draw_no = None

    # ------------------------------------------------------------------ main.py:102-105
    def prepareModel(self) -> None:
        # nn.Module's default mode is train(): like the reference's, the model runs its first epoch with the
        # denoisers' dropout on, and testEpoch() switches it off for good (main.py:144)
        self.model = HGDM(self.handler, self.args, device=self.device)
        self.opt = Adam(self.model, lr=self.args.lr, weight_decay=0)
        self._sums = torch.zeros(4, dtype=torch.float64, device=self.device)        # Loss, recLoss, regLoss, diffLoss
        self._seed = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=self.device)   # d loss / d (the four)
        n, B = len(self.handler.trn_data), self.args.batch
        self._n_batches = -(-n // B)
        # the epoch's shuffled triples, one row of B per batch (the last one padded)
        self._epoch = torch.zeros((self._n_batches, 3, B), dtype=torch.int64, device=self.device)
        self._flat = torch.zeros((3, self._n_batches * B), dtype=torch.int64, device=self.device)
        self._batch_no = torch.zeros(1, dtype=torch.int64, device=self.device)       # advanced by every recorded step
        self._graphs = {}     # (batch size, training mode) -> recorded step

    def set_draws(self, ts: torch.Tensor, noise: torch.Tensor, keep: Optional[torch.Tensor] = None) -> None:
        """Fixes the random inputs of the training steps to come: step k (counted from this call, over epochs) takes
        timesteps ts[k] (int64 [n_user + n_item]), noise noise[k] ([n_user + n_item, 64]) and, in training mode, the
        dropout keep mask keep[k] (uint8, same shape) instead of drawing them -- so that two Coaches, or a replayed and
        an eagerly launched step, can be compared number for number.  The step reads them through a device-side
        counter: recorded steps stay valid."""
        dev = self.device
        self._draws = (ts.to(dev).long().contiguous(), noise.to(dev).float().contiguous(),
                       None if keep is None else keep.to(dev).to(torch.uint8).contiguous())
        self._draw_no = torch.zeros(1, dtype=torch.int64, device=dev)
        self._graphs = {}

    def _step(self, ancs, poss, negs) -> None:
        draws = None
        if self._draws is not None:
            ts, noise, keep = (None if t is None else t.index_select(0, self._draw_no)[0] for t in self._draws)
            self._draw_no.add_(1)
            nu = self.model.n_user
            draws = dict(ts_u=ts[:nu], ts_i=ts[nu:], noise_u=noise[:nu], noise_i=noise[nu:], validate=False)
            if keep is not None:
                draws.update(keep_u=keep[:nu], keep_i=keep[nu:])
        self._sums += training_step(self.model, self.opt, ancs, poss, negs, self._seed, draws).detach()

    def _snapshot(self):
        o = self.opt
        return ([p.detach().clone() for p in self.model.parameters()], o.exp_avg.clone(), o.exp_avg_sq.clone(), o.state.clone(),
                ([t.clone() for t in o.rest.exp_avg + o.rest.exp_avg_sq], o.rest.state.clone()), self._sums.clone(),
                self._batch_no.clone(), None if self._draw_no is None else self._draw_no.clone(), self.model.draw_state.clone())

    def _restore(self, snap) -> None:
        """Puts the values back IN PLACE: the recorded graph holds the addresses of these tensors."""
        params, m, v, state, rest, sums, batch_no, draw_no, draw_state = snap
        o = self.opt
        with torch.no_grad():
            for p, q in zip(self.model.parameters(), params):
                p.copy_(q)
            o.exp_avg.copy_(m); o.exp_avg_sq.copy_(v); o.state.copy_(state); self._sums.copy_(sums); self._batch_no.copy_(batch_no)
            self.model.draw_state.copy_(draw_state)
            if draw_no is not None:
                self._draw_no.copy_(draw_no)
            for t, q in zip(o.rest.exp_avg + o.rest.exp_avg_sq, rest[0]):
                t.copy_(q)
            o.rest.state.copy_(rest[1])

    def _capture(self, size: int):
        """Records one step over a batch of `size` pairs as a HIP graph.  The recording takes its triples out of the
        epoch's table at the position a device-side batch counter names, and advances the counter: an epoch is then
        replay after replay with nothing in between.  The warm-up steps torch asks for before a capture are real
        steps; their effect on the model, the optimizer, the counter and the loss sums is undone afterwards."""
        def step():
            tri = self._epoch.index_select(0, self._batch_no)[0]      # [3, B]: the batch the counter names
            self._batch_no.add_(1)
            self._step(tri[0, :size], tri[1, :size], tri[2, :size])
        keep = self._snapshot()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._batch_no.zero_()
                step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        self._batch_no.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        self._restore(keep)
        self._graphs[(size, self.model.training)] = graph
        return graph

    # ------------------------------------------------------------------ main.py:107-136
    def trainEpoch(self) -> Dict[str, float]:
        trn, B = self.handler.trn_data, self.args.batch
        trn.negSampling()
        n = len(trn)
        steps = n // B
        order = torch.randperm(n, device=self.device, generator=self._gen)        # DataLoader(shuffle=True)
        epoch = torch.stack([trn.rows_dev[order].long(), trn.cols_dev[order].long(), trn.negs_dev[order].long()])
        self._sums.zero_()
        if self.use_graph:          # two recordings at most: the full batch and the epoch's short last one
            self._flat[:, :n] = epoch                                 # [3, n_batches * B]; the padding stays zero
            self._epoch.copy_(self._flat.view(3, self._n_batches, B).permute(1, 0, 2))
            for b in range(0, n, B):
                size = min(B, n - b)
                graph = self._graphs.get((size, self.model.training)) or self._capture(size)   # a recording holds its mode
                if b == 0:
                    self._batch_no.zero_()
                graph.replay()
        else:
            for b in range(0, n, B):
                part = epoch[:, b: b + B].contiguous()
                self._step(part[0], part[1], part[2])
        ep = self._sums.cpu().tolist()                                             # the epoch's one read-back
        div = max(steps, 1)
        return {"Loss": ep[0] / div, "recLoss": ep[1] / div, "diffLoss": ep[3] / div}

    # ------------------------------------------------------------------ main.py:138-167
    def testEpoch(self) -> Dict[str, float]:
        self.model.eval()      # main.py:144 -- and nothing switches back: dropout is on during the first epoch only
        res = evaluate.testEpoch(self.model, self.handler.tst_data, self.args.topk, self.args.tstBat)
        return {"Recall": res["Recall"], "NDCG": res["NDCG"]}

    # ------------------------------------------------------------------ main.py:40-99
    def run(self, epochs: Optional[int] = None, save_path: Optional[str] = None, log=print) -> Dict[str, float]:
        if self.model is None:
            self.prepareModel()
        recallMax = ndcgMax = 0.0
        bestEpoch = wait = 0
        for ep in range(self.args.epoch if epochs is None else epochs):
            reses = self.trainEpoch()
            log("Epoch %d Train: " % ep + ", ".join("%s = %.4f" % kv for kv in reses.items()))
            reses = self.testEpoch()
            for k, v in reses.items():
                self.metrics.setdefault("Test" + k, []).append(v)
            if reses["Recall"] > recallMax:
                recallMax, ndcgMax, bestEpoch, wait = reses["Recall"], reses["NDCG"], ep, 0
                if save_path:
                    self.saveModel(save_path)
            else:
                wait += 1
            log("Epoch %d Test: " % ep + ", ".join("%s = %.4f" % kv for kv in reses.items()))
            if wait >= self.args.patience:
                log(f"Early stop at epoch {ep}, best epoch {bestEpoch}")
                break
        return {"bestEpoch": bestEpoch, "Recall": recallMax, "NDCG": ndcgMax}

    # ------------------------------------------------------------------ main.py:196-211 (state dicts, no pickled modules)
    def saveModel(self, path: str) -> None:
        # (the shuffle's generator too: the model's own draw state travels in its state_dict)
        io.save_checkpoint(path, self.model, self.opt, extra={"shuffle_generator": self._gen.get_state()})

    def loadModel(self, path: str) -> None:
        if self.model is None:
            self.prepareModel()
        info = io.load_checkpoint(path, self.model, self.opt)
        if "shuffle_generator" in info["extra"]:
            self._gen.set_state(info["extra"]["shuffle_generator"])
        self._graphs = {}
