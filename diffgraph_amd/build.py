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
SOURCES = ["graph.hip", "spmm.hip", "fused.hip", "diffusion.hip", "wgrad.hip", "losshead.hip", "eval.hip", "sampling.hip", "spgemm.hip", "optim.hip", "linear.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "dg_internal.h"), os.path.join(CSRC, "dg_gather.h"), os.path.join(CSRC, "dg_philox.h"), os.path.join(CSRC, "dg_wgrad.h"), os.path.join(HERE, "..", "include", "diffgraph_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def source_digest() -> str:
    """Digest of the kernel sources (csrc/*.hip, csrc/*.h): compiled into the library (dg_source_digest) so that a
    library built from other sources than the ones on disk is refused at load, and quoted by the profile files."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


DIGEST_MARK = b"DG_SOURCE_DIGEST="


def library_digest(path: str = LIB) -> str:
    """The digest a built library carries (read from the file's bytes, without loading it); '' if none."""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return ""
    at = data.find(DIGEST_MARK)
    return data[at + len(DIGEST_MARK): at + len(DIGEST_MARK) + 16].decode("ascii", "replace") if at >= 0 else ""


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libdiffgraph_hip.so cannot be built")
    return exe


def needs_build() -> bool:
    """The library is missing, was built from other kernel sources (digest), or is older than the public header."""
    if not os.path.exists(LIB) or library_digest() != source_digest():
        return True
    return os.path.getmtime(os.path.join(HERE, "..", "include", "diffgraph_hip.h")) > os.path.getmtime(LIB)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    digest = source_digest()
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", obj]
        if s == "graph.hip":      # dg_source_digest() lives there
            cmd.insert(-4, f'-DDG_SOURCE_DIGEST="{digest}"')
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
