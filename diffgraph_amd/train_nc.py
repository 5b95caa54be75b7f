"""Coach of the reference's DiffGraph_NC/main.py:333-449 -- prepareModel / trainEpoch(i) / testEpoch(i) / run -- over
the HIP path.

What differs from the reference's loop, and why the results do not:
* a step (cal_loss, backward, Adam -- all ten parameter tensors in one launch; main.py:378-387) is recorded once per
  batch size as a HIP graph and replayed: two recordings, the full batch and the epoch's short last one.  The step's random inputs (timesteps, noise, the key of
  the dropout mask) are drawn BEFORE the step into fixed device buffers, from a generator of the coach's own, so a
  replayed and an eagerly launched run with the same seed train on the same numbers;
* the reference reads two losses back per step (`.item()`, main.py:381-383); here they are added up on the device and
  read once per epoch;
* macro / micro F1 (Utils/Utils.py:108-136, sklearn's f1_score there) come from the confusion matrix on the device.
The handler is any object with feature_list ([n, f] float32), labels ([n, classes] one-hot), he_adjs (three CSRGraph)
and train_idx / val_idx / test_idx (one index array per labelled-ratio setting): the fields the reference's Coach
reads.  The reference also hands its handler's metapath_learner parameters to Adam (main.py:356); nothing in the step
gives them a gradient (the graphs are fixed while training, main.py:373 is commented out), so Adam never moves them.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import io
from .model_nc import DEFAULT_ARGS, HGDM
from .optim import SmallAdam


def f1_scores(y_true: torch.Tensor, y_pred: torch.Tensor, n_classes: int) -> Tuple[float, float]:
    """(macro, micro) F1 of single-label predictions, as sklearn.metrics.f1_score(average='macro' | 'micro') gives
    them: per-class F1 = 2 TP / (2 TP + FP + FN) (0 where a class has neither), averaged over the classes that occur
    in y_true or y_pred; micro = the share of correct predictions."""
    cm = torch.bincount(y_true.long() * n_classes + y_pred.long(), minlength=n_classes * n_classes)
    cm = cm.view(n_classes, n_classes).double()
    tp = cm.diag()
    denom = cm.sum(0) + cm.sum(1)           # 2 TP + FP + FN
    f1 = torch.where(denom > 0, 2 * tp / denom.clamp_min(1), torch.zeros_like(tp))
    present = denom > 0
    macro = float(f1[present].mean()) if bool(present.any()) else 0.0
    return macro, float(tp.sum() / cm.sum().clamp_min(1))


def evaluate(scores: torch.Tensor, idx_val, idx_test, label: torch.Tensor, nb_classes: int):
    """Utils/Utils.py:108-136: accuracy, macro-F1 and micro-F1 of argmax(scores) on the validation and the test
    authors, and the test authors' scores."""
    out = []
    for idx in (idx_val, idx_test):
        idx = torch.as_tensor(np.asarray(idx), dtype=torch.int64, device=scores.device)
        lbls = torch.argmax(label[idx], dim=-1)
        preds = torch.argmax(scores[idx], dim=1)
        macro, micro = f1_scores(lbls, preds, nb_classes)
        out.append({"acc": torch.sum(preds == lbls).float() / lbls.shape[0], "macro": macro, "micro": micro})
    out[1]["logits"] = scores[torch.as_tensor(np.asarray(idx_test), dtype=torch.int64, device=scores.device)]
    return out[0], out[1]


class NCCoach:
    def __init__(self, handler, args=None, device="cuda", use_graph: bool = True, seed: int = 0):
        self.handler, self.args, self.device = handler, args or DEFAULT_ARGS, torch.device(device)
        self.use_graph = use_graph
        self.lr = float(getattr(self.args, "lr", 3e-3))
        self.metrics: Dict[str, list] = dict()
        self.model: Optional[HGDM] = None
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self._order = np.random.default_rng(seed)       # index_generator's shuffle (DataHandler.py:267-295)
        self._graphs = {}

    # ------------------------------------------------------------------ main.py:333-357
    def prepareModel(self) -> None:
        h, dev = self.handler, self.device
        self.initial_feature = h.feature_list.to(dev).float().contiguous()
        self.label = h.labels.to(dev)
        self.nbclasses = int(self.label.shape[1])
        self.model = HGDM(int(self.initial_feature.shape[1]), self.args, device=dev)      # stays in train(): main.py never calls eval()
        # one launch for all parameter tensors, the step count on the device (a recorded step advances it per replay)
        self.opt = SmallAdam(self.model.parameters(), lr=self.lr, weight_decay=0)
        n, B = int(self.initial_feature.shape[0]), int(self.args.batch)
        self._ancs = torch.zeros(B, dtype=torch.int64, device=dev)
        self._ts = torch.zeros(n, dtype=torch.int64, device=dev)
        self._noise = torch.zeros(n, self.args.latdim, dtype=torch.float32, device=dev)
        self._drop_seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self._sums = torch.zeros(2, dtype=torch.float64, device=dev)      # bceLoss, diffLoss
        self._graphs = {}

    def _draw(self) -> None:
        """The next step's timesteps, noise and dropout key (Model.py:352-353,223-224 draw them inside the step)."""
        self._ts.random_(0, self.args.steps, generator=self._gen)
        self._noise.normal_(generator=self._gen)
        self._drop_seed.random_(0, 2 ** 62, generator=self._gen)

    def _step(self, size: int) -> None:
        nll, diff = self.model.cal_loss(self._ancs[:size], self.label, self.handler.he_adjs, self.initial_feature,
                                        self._ts, self._noise, None, validate=False, drop_seed=self._drop_seed)
        self.opt.zero_grad(set_to_none=True)
        (nll + diff).backward()
        self.opt.step()
        self._sums += torch.stack([nll.detach().double(), diff.detach().double()])

    def _capture(self, size: int):
        """Records one step over `size` authors.  The warm-up steps torch asks for before a capture are real steps on
        the draws already in the buffers; their effect on the model, Adam and the loss sums is undone afterwards (in
        place: the recording holds these tensors' addresses)."""
        o = self.opt
        held = [t.detach().clone() for t in list(self.model.parameters()) + o.exp_avg + o.exp_avg_sq + [o.state, self._sums]]
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._step(size)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._step(size)
        with torch.no_grad():
            for t, q in zip(list(self.model.parameters()) + o.exp_avg + o.exp_avg_sq + [o.state, self._sums], held):
                t.copy_(q)
        self._graphs[(size, self.model.training)] = graph
        return graph

    # ------------------------------------------------------------------ main.py:364-395
    def trainEpoch(self, i: int) -> Dict[str, float]:
        B = int(self.args.batch)
        idx = np.array(self.handler.train_idx[i], dtype=np.int64)
        self._order.shuffle(idx)
        steps = -(-len(idx) // B)                      # index_generator.num_iterations: the short last batch counts
        table = np.zeros((steps, B), dtype=np.int64)
        sizes = []
        for k in range(steps):
            part = np.sort(idx[k * B: (k + 1) * B])    # main.py:375
            table[k, : len(part)] = part
            sizes.append(len(part))
        table_dev = torch.from_numpy(table).to(self.device)
        self._sums.zero_()
        for k, size in enumerate(sizes):
            self._ancs.copy_(table_dev[k])
            self._draw()
            if self.use_graph:
                (self._graphs.get((size, self.model.training)) or self._capture(size)).replay()
            else:
                self._step(size)
        ep = self._sums.cpu().tolist()                 # the epoch's one read-back
        return {"bceLoss": ep[0] / max(steps, 1), "diffLoss": ep[1] / max(steps, 1)}

    # ------------------------------------------------------------------ main.py:432-449
    def testEpoch(self, i: int):
        with torch.no_grad():
            _, scores = self.model.get_allembeds(self.handler.he_adjs, self.initial_feature)
            return evaluate(scores, self.handler.val_idx[i], self.handler.test_idx[i], self.label, self.nbclasses)

    # ------------------------------------------------------------------ main.py:101-222, one labelled-ratio setting, one repeat
    def run(self, ratio: int = 0, epochs: Optional[int] = None, log=print) -> Dict[str, float]:
        """Trains `epochs` epochs on train_idx[ratio], testing after every one, and reports what the reference keeps of
        a repeat (main.py:192-199): the best test accuracy, the best test macro-F1 with the validation macro-F1 of
        that epoch, and the best test micro-F1."""
        self.prepareModel()
        hist = {k: [] for k in ("val_acc", "val_macro", "val_micro", "test_acc", "test_macro", "test_micro")}
        for ep in range(int(getattr(self.args, "epoch", 1)) if epochs is None else epochs):
            reses = self.trainEpoch(ratio)
            self.metrics.setdefault("bceLoss", []).append(reses["bceLoss"])
            self.metrics.setdefault("diffLoss", []).append(reses["diffLoss"])
            val, test = self.testEpoch(ratio)
            for pre, r in (("val_", val), ("test_", test)):
                hist[pre + "acc"].append(float(r["acc"])); hist[pre + "macro"].append(r["macro"]); hist[pre + "micro"].append(r["micro"])
            log("Epoch %d: bceLoss = %.4f, diffLoss = %.4f, val macro-F1 = %.4f, test macro-F1 = %.4f, test micro-F1 = %.4f"
                % (ep, reses["bceLoss"], reses["diffLoss"], val["macro"], test["macro"], test["micro"]))
        self.metrics.update(hist)
        if not hist["test_acc"]:
            return {}
        k = int(np.argmax(hist["test_macro"]))
        return {"acc": max(hist["test_acc"]), "macro": hist["test_macro"][k], "macro_val": hist["val_macro"][k],
                "micro": max(hist["test_micro"])}

    def saveModel(self, path: str) -> None:
        # with the generators of the steps' draws and of the batch order: a resumed run draws what comes next
        st = self._order.bit_generator.state
        io.save_checkpoint(path, self.model, self.opt, extra={
            "draw_generator": self._gen.get_state(),
            "order_generator": {"bit_generator": st["bit_generator"], "state": str(st["state"]["state"]), "inc": str(st["state"]["inc"]),
                                "has_uint32": int(st["has_uint32"]), "uinteger": int(st["uinteger"])}})

    def loadModel(self, path: str) -> None:
        if self.model is None:
            self.prepareModel()
        extra = io.load_checkpoint(path, self.model, self.opt)["extra"]
        if "draw_generator" in extra:
            self._gen.set_state(extra["draw_generator"])
        if "order_generator" in extra:
            o = extra["order_generator"]
            self._order.bit_generator.state = {"bit_generator": o["bit_generator"], "state": {"state": int(o["state"]), "inc": int(o["inc"])},
                                               "has_uint32": int(o["has_uint32"]), "uinteger": int(o["uinteger"])}
        self._graphs = {}
