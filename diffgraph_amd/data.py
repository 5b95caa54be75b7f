"""Host-side workload builders: synthetic interaction graphs shaped like the reference's
datasets, and loaders for the pickle-free graph fixtures shipped with this repo.

Nothing here touches the GPU.  All generators are pure numpy with explicit seeds so the
same graph is rebuilt bit-for-bit in tests, in the golden-vector generator and in bench.py.

Shapes follow the reference's dataset table (README.md:68-80) and the files it ships
(DiffGraph-Rec/data/*, DiffGraph_NC/data/DBLP/*).  The reference's pickled scipy matrices
are never unpickled; Rec workloads are synthetic graphs with the published node/link counts.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np


def synth_interactions(n_user: int, n_item: int, n_links: int, seed: int, zipf_alpha: float = 1.0,
                       user_zipf_alpha: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """`n_links` DISTINCT (user, item) pairs: user uniform (or ~ Zipf(user_zipf_alpha) over a seeded permutation
    of the users when that is > 0), item ~ Zipf(alpha) over a seeded permutation of the items; duplicate pairs
    are re-drawn.  Returned sorted by (user, item) as int32 arrays."""
    if n_links > n_user * n_item:
        raise ValueError("more links requested than user x item cells")
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n_item)
    w = 1.0 / np.power(np.arange(1, n_item + 1, dtype=np.float64), zipf_alpha)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    if user_zipf_alpha > 0:      # (drawn after the item permutation: the uniform-user graphs stay what they were)
        uperm = rng.permutation(n_user)
        uw = 1.0 / np.power(np.arange(1, n_user + 1, dtype=np.float64), user_zipf_alpha)
        ucdf = np.cumsum(uw)
        ucdf /= ucdf[-1]
    keys = np.empty(0, dtype=np.int64)
    while keys.size < n_links:
        need = n_links - keys.size
        m = int(need * 1.15) + 1024
        if user_zipf_alpha > 0:
            ur = np.searchsorted(ucdf, rng.random(m), side="right")
            u = uperm[np.minimum(ur, n_user - 1)].astype(np.int64)
        else:
            u = rng.integers(0, n_user, size=m, dtype=np.int64)
        r = np.searchsorted(cdf, rng.random(m), side="right")
        np.minimum(r, n_item - 1, out=r)
        i = perm[r].astype(np.int64)
        keys = np.unique(np.concatenate([keys, u * n_item + i]))
        if keys.size > n_links:
            # drop a seeded random subset of the surplus so the kept set is still Zipf-shaped
            drop = rng.choice(keys.size, size=keys.size - n_links, replace=False)
            keys = np.delete(keys, drop)
    return (keys // n_item).astype(np.int32), (keys % n_item).astype(np.int32)


@dataclass
class RecWorkload:
    """A DiffGraph-Rec shaped workload: target behaviour ('buy') + auxiliary behaviours."""
    name: str
    n_user: int
    n_item: int
    target: Tuple[np.ndarray, np.ndarray]              # (user, item) int32
    behaviours: List[Tuple[np.ndarray, np.ndarray]]    # auxiliary behaviours, reference order
    behaviour_names: List[str]
    note: str

    @property
    def n(self) -> int:
        return self.n_user + self.n_item


# name -> (n_user, n_item, target links, [(aux name, links, seed)], note)
# Link counts: Retail Rocket from the shipped matrices (view 75 374 / cart 12 456 / buy 9 551,
# README.md:71); Tmall / IJCAI shipped buy/cart/fav counts plus a synthesised largest
# behaviour (pv / click are missing blobs); Industry from README.md:73 split 50/30/20 %.
_REC_SHAPES: Dict[str, tuple] = {
    "tiny": (40, 60, 150, [("view", 420, 101), ("cart", 90, 102)],
             "tiny synthetic bipartite graph for unit tests"),
    "retail_rocket": (2174, 30113, 9551, [("view", 75374, 21), ("cart", 12456, 22)],
                      "synthetic graph with Retail Rocket's shape and per-behaviour link counts"),
    "tmall": (31882, 31232, 167862, [("pv", 1093807, 4), ("fav", 49482, 31), ("cart", 140068, 32)],
              "synthetic graph with Tmall's shape; buy/cart/fav link counts as shipped, pv synthesised"),
    "tmall_skewed": (31882, 31232, 167862, [("pv", 1093807, 4), ("fav", 49482, 31), ("cart", 140068, 32)],
                     "Tmall's shape and link counts with heavy-tailed users as well (users ~ Zipf(0.8), items ~ Zipf(1)): "
                     "hub rows on both sides"),
    "ijcai": (22438, 35573, 199654, [("click", 440265, 5), ("fav", 158119, 41), ("cart", 1330, 42)],
              "synthetic graph with the shipped IJCAI shape (22 438 x 35 573); click synthesised"),
    "ijcai_readme": (17435, 35920, 198243, [("click", 439652, 6), ("fav", 159874, 43), ("cart", 1599, 44)],
                     "synthetic graph with the IJCAI shape the README states (17 435 x 35 920, 799 368 links split "
                     "24.8 / 55 / 20 / 0.2 % over buy / click / fav / cart, the shipped matrices' proportions)"),
    "industry": (1000000, 361, 11945223, [("rel1", 7167133, 11), ("rel2", 4778089, 12)],
                 "synthetic Industry-shaped graph (1M x 361, 23 890 445 links over 3 relations)"),
}
_TARGET_SEED = {"tiny": 100, "retail_rocket": 20, "tmall": 30, "tmall_skewed": 30, "ijcai": 40, "ijcai_readme": 45,
                "industry": 10}
_USER_ALPHA = {"tmall_skewed": 0.8}     # every other workload draws its users uniformly


def rec_workload(name: str) -> RecWorkload:
    n_user, n_item, n_target, aux, note = _REC_SHAPES[name]
    ua = _USER_ALPHA.get(name, 0.0)
    target = synth_interactions(n_user, n_item, n_target, _TARGET_SEED[name], user_zipf_alpha=ua)
    if name == "tiny":
        target = _tiny_edge_cases(*target, n_user, n_item)
    behaviours = [synth_interactions(n_user, n_item, m, s, user_zipf_alpha=ua) for (_, m, s) in aux]
    if name == "tiny":
        behaviours = [_tiny_edge_cases(u, i, n_user, n_item) for (u, i) in behaviours]
    return RecWorkload(name, n_user, n_item, target, behaviours, [a[0] for a in aux], note)


def _tiny_edge_cases(u: np.ndarray, i: np.ndarray, n_user: int, n_item: int):
    """Force the edge cases the layer must handle: an isolated user (last), an isolated
    item (last), and one hub item (item 0) linked to every non-isolated user."""
    keep = (u != n_user - 1) & (i != n_item - 1)
    u, i = u[keep], i[keep]
    hub_u = np.arange(n_user - 1, dtype=np.int32)
    keys = np.unique(np.concatenate([u.astype(np.int64) * n_item + i, hub_u.astype(np.int64) * n_item]))
    return (keys // n_item).astype(np.int32), (keys % n_item).astype(np.int32)


def bipartite_edges(u: np.ndarray, i: np.ndarray, n_user: int) -> Tuple[np.ndarray, np.ndarray]:
    """Directed (src, dst) edge list of the symmetric bipartite block adjacency
    [[0, R], [R^T, 0]] (reference makeBiAdj, DiffGraph-Rec/DataHandler.py:120-132)."""
    src = np.concatenate([u, i + n_user]).astype(np.int32)
    dst = np.concatenate([i + n_user, u]).astype(np.int32)
    return src, dst


def xavier_uniform(rows: int, cols: int, seed: int) -> np.ndarray:
    """xavier_uniform_ bound sqrt(6/(fan_in+fan_out)) (reference Model.py:45-52), fp32."""
    bound = np.sqrt(6.0 / (rows + cols))
    rng = np.random.default_rng(seed)
    return rng.uniform(-bound, bound, size=(rows, cols)).astype(np.float32)
