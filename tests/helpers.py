"""Shared test helpers: fixture loading, oracle graph inputs, the parity tolerance."""
from __future__ import annotations

import os

import numpy as np
import torch

from diffgraph_amd import data as dgdata
from oracle import ref_cpu as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 parity bar of the hot path: |a-b| <= RTOL * max(|b|, largest |b| in the same row).
# (Pure relative error is ill-posed for entries that cancel to ~0.)
RTOL = 1e-5


# Parameter gradients against the reference's own backward (golden files): |a-b| <= GRAD_TOL * largest |entry| of the
# tensor.  Both sides sum fp32 products over the batch in different orders; the largest difference measured over all
# twelve Rec and ten NC parameter tensors, tiny golden and full-size Tmall batch, is 1.9e-6.
GRAD_TOL = 1e-5


def load(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


# Every comparison made through assert_rows_close, for the parity report tests/conftest.py writes at the end of a
# session: the worst error in units of the row scale (what the assertion bounds) and the worst PURE relative error
# |a-b| / |b| over the entries that are at least 1e-3 of their row's largest (entries that cancel to ~0 excepted).
PARITY_LOG = []


def _record(what, rtol, err, b, scale):
    big = (np.abs(b) >= 1e-3 * scale) & (b != 0)
    rel = float((err[big] / np.abs(b[big])).max()) if big.any() else 0.0
    PARITY_LOG.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "what": what[:72], "rtol": rtol,
                       "entries": int(err.size), "worst_over_row_scale": float((err / (scale + 1e-300)).max()),
                       "worst_pure_relative": rel})


def assert_rows_close(a, b, rtol: float = RTOL, what: str = ""):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.size == 0:
        return
    if b.ndim == 1:
        scale = np.maximum(np.abs(b), np.abs(b).max())
    else:
        scale = np.maximum(np.abs(b), np.abs(b).max(axis=-1, keepdims=True))
    err = np.abs(a - b)
    _record(what, rtol, err, b, scale)
    bad = err > rtol * scale + 1e-30
    if bad.any():
        k = np.unravel_index(np.argmax(err / (scale + 1e-30)), err.shape)
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} entries exceed rtol={rtol}; worst at {k}: "
                             f"got {a[k]!r}, want {b[k]!r}, row scale {scale[k]!r}")


def assert_close_to_max(a, b, tol: float, what: str = ""):
    """Whole-tensor bound used for parameter gradients (sums over every row of a batch, so the natural scale is the
    tensor's largest entry): |a-b| <= tol * max|b|.  Recorded in PARITY_LOG like the row-wise comparisons."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = np.full(b.shape, np.abs(b).max())
    err = np.abs(a - b)
    _record(what + " (tensor scale)", tol, err, b, scale)
    if (err > tol * scale + 1e-30).any():
        k = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError(f"{what}: worst at {k}: got {a[k]!r}, want {b[k]!r}, |diff| {err[k]:.3e} > {tol} * {scale[k]:.3e}")


def edges_t(u, i, n_user):
    s, d = dgdata.bipartite_edges(np.asarray(u), np.asarray(i), n_user)
    return torch.from_numpy(s.astype(np.int64)), torch.from_numpy(d.astype(np.int64))


def tiny_fixture():
    fx = load("rec_tiny.npz")
    nu, ni = int(fx["n_user"]), int(fx["n_item"])
    target = edges_t(fx["target_u"], fx["target_i"], nu)
    behs = [edges_t(fx[f"beh{k}_u"], fx[f"beh{k}_i"], nu) for k in range(2)]
    return fx, nu, ni, target, behs


def weights_from(fx, prefix, dtype=torch.float32) -> O.DenoiseWeights:
    g = lambda k: torch.from_numpy(fx[prefix + k]).to(dtype)
    return O.DenoiseWeights(g("w_emb"), g("b_emb"), g("w_in"), g("b_in"), g("w_out"), g("b_out"))


REC_TABLE = dict(noise_scale=1e-4, noise_min=1e-4, noise_max=1e-3, steps=200)
NC_TABLE = dict(noise_scale=1e-5, noise_min=1e-4, noise_max=1e-3, steps=200)
