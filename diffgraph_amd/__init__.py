"""diffgraph_amd: MI355X (gfx950) implementation of DiffGraph's per-epoch compute hot path.

Hand-written HIP kernels (libdiffgraph_hip.so, C ABI in include/diffgraph_hip.h) behind
Python classes that keep the reference's call signatures:

    CSRGraph                          in place of dgl.DGLGraph
    DGLLayer, UUGCNLayer              DiffGraph-Rec/Model.py:120-174, DiffGraph_NC/Model.py:166-214
    Denoise, GaussianDiffusion        DiffGraph-Rec/Model.py:176-399
    rec.HGDM, nc.HGDM                 DiffGraph-Rec/Model.py:16-118, DiffGraph_NC/Model.py:18-108
    evaluate.AllRankTestData, testEpoch, calcRes   DiffGraph-Rec/DataHandler.py:250-272, main.py:136-187
    sampling.PairwiseTrnData          DiffGraph-Rec/DataHandler.py:274-296
    train.Coach                       DiffGraph-Rec/main.py:34-215 (trainEpoch, testEpoch, run)
    train_nc.NCCoach                  DiffGraph_NC/main.py:333-449 (prepareModel, trainEpoch, testEpoch, run)
    metapath.compose_he_adjs          DiffGraph_NC/DataHandler.py:147-174

Importing the package does not load the shared library; the first op does, and raises
DiffGraphHipError if it is missing (there is no CPU fallback).
"""
from ._lib import DiffGraphHipError  # noqa: F401
from .graph import CSRGraph, GraphSet  # noqa: F401
from .layers import DGLLayer, UUGCNLayer  # noqa: F401
from .diffusion import Denoise, GaussianDiffusion  # noqa: F401
from . import model_rec as rec  # noqa: F401
from . import model_nc as nc  # noqa: F401
from . import data, evaluate, io, metapath, ops, optim, sampling, train, train_nc  # noqa: F401

__all__ = ["CSRGraph", "GraphSet", "DGLLayer", "UUGCNLayer", "Denoise", "GaussianDiffusion", "rec", "nc", "ops", "data",
           "DiffGraphHipError"]
