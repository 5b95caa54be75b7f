"""Propagation layers with the reference's call signatures.

DGLLayer.forward(graph, u_f, v_f)  -- DiffGraph-Rec/Model.py:142-174
UUGCNLayer.forward(graph, u_f)     -- DiffGraph_NC/Model.py:184-214

`graph` is a diffgraph_amd.CSRGraph instead of a dgl.DGLGraph.  The reference only ever
instantiates these with weight=False (Model.py:30), so the u_w/v_w projections are not
provided; asking for them raises.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .graph import CSRGraph


def _slope_of(activation) -> float:
    if activation is None:
        return 1.0
    if isinstance(activation, nn.LeakyReLU):
        return float(activation.negative_slope)
    raise NotImplementedError("only LeakyReLU (or no activation) is fused into the propagation kernel")


class DGLLayer(nn.Module):
    def __init__(self, in_feats, out_feats, weight=False, bias=False, activation=None):
        super().__init__()
        if weight or bias:
            raise NotImplementedError("the reference never enables weight/bias on its propagation layers")
        self._in_feats, self._out_feats = in_feats, out_feats
        self.weight, self.bias = weight, bias
        self._activation = activation

    def forward(self, graph: CSRGraph, u_f: torch.Tensor, v_f: torch.Tensor) -> torch.Tensor:
        node_f = torch.cat([u_f, v_f], dim=0)
        return ops.spmm_norm_act(graph, node_f, _slope_of(self._activation))


class UUGCNLayer(nn.Module):
    def __init__(self, in_feats, out_feats, weight=False, bias=False, activation=None):
        super().__init__()
        if weight or bias:
            raise NotImplementedError("the reference never enables weight/bias on its propagation layers")
        self._in_feats, self._out_feats = in_feats, out_feats
        self.weight, self.bias = weight, bias
        self._activation = activation

    def forward(self, graph: CSRGraph, u_f: torch.Tensor) -> torch.Tensor:
        return ops.spmm_norm_act(graph, u_f.contiguous(), _slope_of(self._activation))
