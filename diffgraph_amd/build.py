"""Builds libdiffgraph_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m diffgraph_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdiffgraph_hip.so")
SOURCES = ["graph.hip", "spmm.hip", "fused.hip", "diffusion.hip", "wgrad.hip", "losshead.hip", "eval.hip", "sampling.hip", "spgemm.hip", "optim.hip", "linear.hip"]
HEADERS = [os.path.join(CSRC, "dg_internal.h"), os.path.join(CSRC, "dg_gather.h"), os.path.join(CSRC, "dg_philox.h"), os.path.join(CSRC, "dg_wgrad.h"), os.path.join(HERE, "..", "include", "diffgraph_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libdiffgraph_hip.so cannot be built")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
