"""A torch-CPU stand-in for diffgraph_amd.dist.HipBackend, used ONLY by the gloo tests of the
sharding orchestration.  It implements the documented semantics of dg_gcn_layer_fwd (with
partial_out / partial_from) and dg_gcn_partial_finalize with the oracle's arithmetic."""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from diffgraph_amd._lib import (DG_ACC_REBUILD, DG_IN_PRESCALED, DG_OUT_NORM, DG_OUT_PRESCALE, DG_ROWS_FULL_ONLY,
                                DG_ROWS_PARTIAL_ONLY)


class CpuBackend:
    def can_split_rows(self, graphs, split):
        return True

    def make_graph(self, lg, n_nodes):
        deg = torch.from_numpy(lg.degrees).float().clamp(min=1)
        owned = None
        if getattr(lg, "in_degrees", None) is not None:   # row-block shard: rows with in-degree -1 are not computed
            owned = torch.nonzero(torch.from_numpy(lg.in_degrees) >= 0).flatten()
        return SimpleNamespace(src=torch.from_numpy(lg.src).long(), dst=torch.from_numpy(lg.dst).long(), n=n_nodes,
                               dinv=torch.pow(deg, -0.5).view(-1, 1), owned=owned)

    def empty(self, rows):
        return torch.full((rows, 64), float("nan"))

    @staticmethod
    def _epilogue(g, rows, s, e_out, acc_in, acc_out, flags, slope, x=None):
        y = F.leaky_relu(s * g.dinv[rows], slope)
        yn = F.normalize(y, p=2, dim=1)
        if acc_out is not None:
            a = acc_in[rows]
            if flags & DG_ACC_REBUILD:  # e0 + normalize(E_1), E_1 recovered from the stored dinv * E_1
                a = a + F.normalize(x[rows] / g.dinv[rows], p=2, dim=1)
            acc_out[rows] = a + yn
        if e_out is not None:
            e = yn if flags & DG_OUT_NORM else y
            e_out[rows] = e * g.dinv[rows] if flags & DG_OUT_PRESCALE else e

    def layer(self, tasks):
        for t in tasks:
            g, x = t["graph"], t["x"]
            h = x[: g.n] if t["flags"] & DG_IN_PRESCALED else x[: g.n] * g.dinv
            s = torch.zeros_like(h).index_add_(0, g.dst, h[g.src])
            pf = t["partial_from"] if t.get("partial_out") is not None else g.n
            if t.get("partial_out") is not None and not t["flags"] & DG_ROWS_FULL_ONLY:
                t["partial_out"].copy_(s[pf:])
            if t["flags"] & DG_ROWS_PARTIAL_ONLY:
                continue
            rows = slice(0, pf) if g.owned is None else g.owned
            self._epilogue(g, rows, s[rows], t["e_out"], t["acc_in"], t["acc_out"], t["flags"], t["slope"], x)

    def prepare_layer(self, tasks):
        return lambda: self.layer(tasks)

    def prepare_finalize(self, tasks):
        return lambda: self.finalize(tasks)

    def prepare_sum(self, tables, out):
        return lambda: out.copy_(self.sum_tables(tables))

    def finalize(self, tasks):
        for t in tasks:
            partials = t["partials"]
            s = torch.zeros_like(partials[0])
            for p in range(partials.shape[0]):
                s = s + partials[p]
            rows = slice(t["row_begin"], t["row_begin"] + partials.shape[1])
            self._epilogue(t["graph"], rows, s, t["e_out"], t["acc_in"], t["acc_out"], t["flags"], t["slope"], t["x"])

    def sum_tables(self, tables):
        out = tables[0]
        for t in tables[1:]:
            out = out + t
        return out
