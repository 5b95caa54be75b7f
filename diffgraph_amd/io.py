"""Pickle-free files for graphs, checkpoints and metric histories.

The reference reads its interaction matrices from pickled scipy objects (DiffGraph-Rec/DataHandler.py:45-47,63),
saves the whole pickled module with torch.save and re-creates Adam from scratch on load, and pickles its metric
history (DiffGraph-Rec/main.py:189-211).  Here a behaviour matrix is an .npz of three int32/float arrays, a
checkpoint is a state_dict file (model + optimiser + epoch + metrics) that torch.load(weights_only=True) accepts,
and the history is JSON -- nothing that executes code on load.
"""
from __future__ import annotations

import json
from typing import Dict, Optional

import numpy as np
import torch


def save_interactions(path: str, mat) -> None:
    """A sparse [n_user, n_item] interaction matrix (any scipy format) as row / col / data arrays + shape."""
    coo = mat.tocoo()
    np.savez_compressed(path, row=coo.row.astype(np.int32), col=coo.col.astype(np.int32), data=np.asarray(coo.data),
                        shape=np.asarray(coo.shape, dtype=np.int64))


def load_interactions(path: str):
    """The scipy COO matrix written by save_interactions (np.load without pickle)."""
    import scipy.sparse as sp
    with np.load(path, allow_pickle=False) as f:
        return sp.coo_matrix((f["data"], (f["row"], f["col"])), shape=tuple(int(v) for v in f["shape"]))


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def save_checkpoint(path: str, model, optimizer=None, epoch: int = 0, metrics: Optional[Dict] = None,
                    extra: Optional[Dict] = None) -> None:
    """Coach.saveModel (main.py:196-201) as tensors only: model.state_dict() (with the Rec model's device-side draw
    state: seed and draws made so far), the optimiser's state (moments and step counts, so training resumes where it
    stopped), the epoch, the metric lists and `extra` (tensors / numbers / nested dicts and lists of them: a coach's
    random-generator states)."""
    torch.save({"model": _to_cpu(model.state_dict()), "optimizer": None if optimizer is None else _to_cpu(optimizer.state_dict()),
                "epoch": int(epoch), "metrics": metrics or {}, "extra": _to_cpu(extra or {})}, path)


def load_checkpoint(path: str, model, optimizer=None) -> Dict:
    """Coach.loadModel (main.py:203-211): restores the model in place and -- unlike the reference, which starts Adam
    afresh -- the optimiser.  Returns {'epoch': ..., 'metrics': ..., 'extra': ...}.  A file written before the draw
    state was part of the model's state loads too (the draws then start over)."""
    ckp = torch.load(path, map_location="cpu", weights_only=True)
    res = model.load_state_dict(ckp["model"], strict=False)
    bad = [k for k in list(res.missing_keys) + list(res.unexpected_keys) if k != "draw_state"]
    if bad:
        raise RuntimeError(f"checkpoint {path}: state_dict keys do not match the model: {bad}")
    if optimizer is not None and ckp.get("optimizer") is not None:
        optimizer.load_state_dict(ckp["optimizer"])
    return {"epoch": ckp["epoch"], "metrics": ckp["metrics"], "extra": ckp.get("extra", {})}


def save_history(path: str, metrics: Dict) -> None:
    """Coach.saveHistory (main.py:189-193) as JSON."""
    with open(path, "w") as f:
        json.dump({k: [float(x) for x in v] for k, v in metrics.items()}, f)


def load_history(path: str) -> Dict:
    with open(path) as f:
        return json.load(f)
