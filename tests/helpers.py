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


def load(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


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
    bad = err > rtol * scale + 1e-30
    if bad.any():
        k = np.unravel_index(np.argmax(err / (scale + 1e-30)), err.shape)
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} entries exceed rtol={rtol}; worst at {k}: "
                             f"got {a[k]!r}, want {b[k]!r}, row scale {scale[k]!r}")


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
