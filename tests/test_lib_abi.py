"""The C-ABI library loads and exports every symbol include/diffgraph_hip.h declares, and the
host-side argument checks reject bad input before anything reaches a GPU.  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from diffgraph_amd import _lib, data as dgdata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "diffgraph_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert lib.dg_version() == 1


def test_argument_errors_are_reported_not_launched():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.dg_graph_create(0, 0, None, None, 0, 0, C.byref(h)) < 0
    assert b"n_nodes" in lib.dg_last_error()
    src = np.array([0, 5], dtype=np.int32)
    dst = np.array([1, 0], dtype=np.int32)
    assert lib.dg_graph_create(3, 2, src.ctypes.data, dst.ctypes.data, 0, 0, C.byref(h)) < 0
    assert b"out of range" in lib.dg_last_error()
    assert lib.dg_graph_create(3, 2, src.ctypes.data, dst.ctypes.data, 0, 100, C.byref(h)) < 0  # seg_len % 64
    # wrong embedding width is refused by every data entry point
    assert lib.dg_spmm_norm_act_fwd(None, None, None, 32, 0.5, 0, None) == -2
    assert lib.dg_q_sample(None, None, None, None, None, 200, 4, 128, None, None) == -2
    assert lib.dg_gcn_layer_fwd(None, 0, 64, None) < 0
    assert lib.dg_sum_tables(None, 0, None, 0, None) < 0
    # the collective entry point checks its arguments before it looks for RCCL
    assert lib.dg_allgather_rows(None, None, None, 8, 32, None) == -2
    assert lib.dg_allgather_rows(None, None, None, 8, 64, None) == -1 and b"comm is NULL" in lib.dg_last_error()
    fake = C.c_void_p(1)
    assert lib.dg_allgather_rows(fake, None, None, -1, 64, None) == -1 and b"negative" in lib.dg_last_error()
    assert lib.dg_allgather_rows(fake, None, None, 0, 64, None) == 0            # nothing to move
    assert lib.dg_allgather_rows(fake, None, None, 8, 64, None) == -1 and b"NULL buffer" in lib.dg_last_error()


def test_synthetic_workloads_have_the_published_shapes():
    wl = dgdata.rec_workload("retail_rocket")
    assert (wl.n_user, wl.n_item) == (2174, 30113)
    assert [len(wl.target[0])] + [len(b[0]) for b in wl.behaviours] == [9551, 75374, 12456]
    for u, i in [wl.target] + wl.behaviours:
        keys = u.astype(np.int64) * wl.n_item + i
        assert len(np.unique(keys)) == len(keys) and u.max() < wl.n_user and i.max() < wl.n_item
    tiny = dgdata.rec_workload("tiny")
    deg_u = np.bincount(tiny.target[0], minlength=tiny.n_user)
    deg_i = np.bincount(tiny.target[1], minlength=tiny.n_item)
    assert deg_u[-1] == 0 and deg_i[-1] == 0 and deg_i[0] == tiny.n_user - 1  # isolated nodes + hub


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DiffGraphHipError):
        _lib.load()


def test_library_carries_the_digest_of_its_sources_and_a_stale_one_is_refused(monkeypatch):
    """The library names the kernel sources it was built from (dg_source_digest, compiled in by diffgraph_amd.build);
    _lib.load() refuses a library whose digest is not the one of the sources on disk."""
    from diffgraph_amd import build
    lib = _lib.load()
    assert lib.dg_source_digest().decode() == build.source_digest() == build.library_digest()
    assert not build.needs_build()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "source_digest", lambda: "0123456789abcdef")      # as if a source file had changed
    with pytest.raises(_lib.DiffGraphHipError, match="other kernel sources"):
        _lib.load()
    assert build.needs_build()
