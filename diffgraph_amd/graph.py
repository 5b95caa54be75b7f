"""CSRGraph: the graph handle the propagation layers take in place of a dgl.DGLGraph.

The reference builds its graphs with dgl.graph((src, dst), num_nodes=...) from the
bipartite block adjacency (DiffGraph-Rec/DataHandler.py:120-132) or dgl.from_scipy
(DiffGraph_NC/DataHandler.py:174).  CSRGraph takes the same directed edge list and owns a
device-side CSR + degree norms + row schedule created by dg_graph_create.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib


class CSRGraph:
    def __init__(self, src, dst, num_nodes: int, group_max: int = 0, seg_len: int = 0, out_degrees=None,
                 in_degrees=None, device=None):
        """device: the GPU the graph's arrays live on (default: the current one); layers refuse a graph of another
        device than the one they are launched on.
        out_degrees / in_degrees (int32[num_nodes]) override the degrees counted from the given
        edges; a rank holding an edge shard passes the whole graph's degrees.  in_degrees[i] < 0 marks
        node i as not owned: the forward propagation skips that row (row-block partitions)."""
        src = np.ascontiguousarray(np.asarray(src), dtype=np.int32)
        dst = np.ascontiguousarray(np.asarray(dst), dtype=np.int32)
        if src.shape != dst.shape or src.ndim != 1:
            raise ValueError("src and dst must be 1-D arrays of equal length")
        degs = []
        for d in (out_degrees, in_degrees):
            if d is not None:
                d = np.ascontiguousarray(np.asarray(d), dtype=np.int32)
                if d.shape != (int(num_nodes),):
                    raise ValueError("degree overrides must have one entry per node")
            degs.append(d)
        lib = _lib.load()
        h = C.c_void_p()
        with _lib.device_guard(device):
            rc = lib.dg_graph_create_ex(int(num_nodes), int(src.size), src.ctypes.data, dst.ctypes.data,
                                        None if degs[0] is None else degs[0].ctypes.data,
                                        None if degs[1] is None else degs[1].ctypes.data,
                                        int(group_max), int(seg_len), C.byref(h))
        _lib.check(rc, "dg_graph_create_ex")
        self._h = h
        self._n = self._rows = int(num_nodes)
        self._nnz = int(src.size)

    # -- constructors mirroring the reference's DataHandler helpers --------------------
    @classmethod
    def from_bipartite(cls, user, item, n_user: int, n_item: int, **kw) -> "CSRGraph":
        """makeBiAdj (DiffGraph-Rec/DataHandler.py:120-132): [[0, R], [R^T, 0]] over n_user+n_item nodes."""
        user = np.asarray(user, dtype=np.int32)
        item = np.asarray(item, dtype=np.int32)
        src = np.concatenate([user, item + n_user])
        dst = np.concatenate([item + n_user, user])
        return cls(src, dst, n_user + n_item, **kw)

    @classmethod
    def from_scipy(cls, mat, **kw) -> "CSRGraph":
        """dgl.from_scipy: edge (row -> col) for every stored entry; weights are ignored, as copy_u does."""
        coo = mat.tocoo()
        return cls(coo.row, coo.col, coo.shape[0], **kw)

    @classmethod
    def weighted(cls, rows, cols, weights, n_rows: int, n_cols: int, group_max: int = 0, seg_len: int = 0,
                 device=None) -> "CSRGraph":
        """A rectangular operator with a weight per entry on the propagation kernels: a layer over it with slope 1
        and no normalisation gives out[r] = sum_(r,c) weight * x[c] -- x of max(n_rows, n_cols) rows, out of n_rows.
        Forward only.  (HGDM's backward uses one to add up the last propagation step of all its graphs.)"""
        rows = np.ascontiguousarray(np.asarray(rows), dtype=np.int32)
        cols = np.ascontiguousarray(np.asarray(cols), dtype=np.int32)
        weights = np.ascontiguousarray(np.asarray(weights), dtype=np.float32)
        if not (rows.shape == cols.shape == weights.shape) or rows.ndim != 1:
            raise ValueError("rows, cols and weights must be 1-D arrays of equal length")
        self = cls.__new__(cls)
        h = C.c_void_p()
        with _lib.device_guard(device):
            rc = _lib.load().dg_graph_create_weighted(int(n_rows), int(n_cols), int(rows.size), rows.ctypes.data,
                                                      cols.ctypes.data, weights.ctypes.data, int(group_max), int(seg_len),
                                                      C.byref(h))
        _lib.check(rc, "dg_graph_create_weighted")
        self._h = h
        self._n, self._rows, self._nnz = max(int(n_rows), int(n_cols)), int(n_rows), int(rows.size)
        return self

    # -- the few DGLGraph attributes callers use --------------------------------------
    @property
    def handle(self) -> C.c_void_p:
        if self._h is None:
            raise _lib.DiffGraphHipError("graph handle already destroyed")
        return self._h

    def num_nodes(self) -> int:
        return self._n

    def num_rows(self) -> int:
        """Rows a layer over this graph computes and writes (= num_nodes() except for weighted operators)."""
        return self._rows

    def num_edges(self) -> int:
        return self._nnz

    @property
    def is_symmetric(self) -> bool:
        return bool(_lib.load().dg_graph_is_symmetric(self.handle))

    def _export(self, which: int, count: int, dtype) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        rc = _lib.load().dg_graph_export(self.handle, which, out.ctypes.data, out.nbytes)
        _lib.check(rc, "dg_graph_export")
        return out

    def csr(self) -> Tuple[np.ndarray, np.ndarray]:
        return self._export(0, self._n + 1, np.int32), self._export(1, self._nnz, np.int32)

    def norms(self) -> Tuple[np.ndarray, np.ndarray]:
        """(out-degree norm, in-degree norm) = clamp(deg,1)^-1/2, fp32."""
        return self._export(2, self._n, np.float32), self._export(3, self._n, np.float32)

    def schedule(self) -> dict:
        c = self._export(5, 6, np.int32)
        order = np.zeros(0, np.int32) if c[5] > 0 else self._export(4, int(c[0]) + int(c[1]) + int(c[2]), np.int32)
        return {"row_order": order, "n_hub": int(c[0]), "n_wave": int(c[1]), "n_group": int(c[2]), "n_seg": int(c[3]),
                "lds_split": int(c[4]), "dense_splits": int(c[5])}

    def close(self) -> None:
        if self._h is not None:
            _lib.load().dg_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GraphSet:
    """2..4 symmetric graphs over the same nodes behind ONE row schedule (dg_graphset): a propagation layer over the
    set sums row r's neighbours in every graph back to back and writes the first graph's layer sum and the sum of the
    others' once each -- HGDM.forward's target graph and behaviour graphs (DiffGraph-Rec/Model.py:61-87).  The set
    copies what it needs from the graphs."""

    def __init__(self, graphs, group_max: int = 0, seg_len: int = 0, device=None):
        graphs = list(graphs)
        arr = (C.c_void_p * len(graphs))(*[g.handle for g in graphs])
        h = C.c_void_p()
        with _lib.device_guard(device):
            rc = _lib.load().dg_graphset_create(arr, len(graphs), int(group_max), int(seg_len), C.byref(h))
        _lib.check(rc, "dg_graphset_create")
        self._h = h
        self._n, self._g = graphs[0].num_nodes(), len(graphs)
        self._nnz = [g.num_edges() for g in graphs]

    @property
    def handle(self) -> C.c_void_p:
        if self._h is None:
            raise _lib.DiffGraphHipError("graph set already destroyed")
        return self._h

    def num_nodes(self) -> int:
        return self._n

    def num_graphs(self) -> int:
        return self._g

    def num_edges(self):
        """Directed edges per graph."""
        return list(self._nnz)

    def counts(self) -> dict:
        c = (C.c_int32 * 7)()
        _lib.check(_lib.load().dg_graphset_counts(self.handle, c), "dg_graphset_counts")
        return {"n": c[0], "graphs": c[1], "n_hub": c[2], "n_seg": c[3], "quad_max": c[4], "seg_len": c[5], "side_split": c[6]}

    def close(self) -> None:
        if self._h is not None:
            _lib.load().dg_graphset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
