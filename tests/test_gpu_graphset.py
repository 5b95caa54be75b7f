"""The row walk over a set of graphs (dg_graphset: one 16-lane group takes a row of ALL graphs back to back) against
the task-per-graph path and the CPU oracle, forward and backward.  Bar: tests.helpers.RTOL (1e-5 of the row scale)
against the oracle; the two HIP paths differ only in how the rows summed in pieces are cut."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import diffgraph_amd as dg
from diffgraph_amd import data as dgdata, ops
from oracle import ref_cpu as O
from tests.helpers import assert_rows_close, edges_t

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"


def rec_model(name, n_graphs=None, **set_kw):
    wl = dgdata.rec_workload(name)
    pairs = ([wl.target] + wl.behaviours)[:n_graphs]
    graphs = [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item) for (u, i) in pairs]
    handler = SimpleNamespace(userNum=wl.n_user, itemNum=wl.n_item, target_adj=graphs[0], behavior_mats=graphs[1:])
    model = dg.rec.HGDM(handler, device=DEV).eval()
    ue, ie = T(dgdata.xavier_uniform(wl.n_user, 64, seed=0)), T(dgdata.xavier_uniform(wl.n_item, 64, seed=1))
    model.set_embeddings(ue.to(DEV), ie.to(DEV))
    if set_kw:
        model._gset, model._gset_graphs = dg.GraphSet(graphs, **set_kw), list(graphs)
    return wl, pairs, model, ue, ie


@pytest.mark.parametrize("name,n_graphs,set_kw", [
    ("tiny", 3, {}), ("tiny", 3, dict(group_max=12, seg_len=64)), ("tiny", 2, dict(group_max=12, seg_len=64)),
    ("retail_rocket", 3, {}), ("retail_rocket", 3, dict(group_max=16, seg_len=64)), ("ijcai", 4, {})])
def test_graph_set_forward_against_oracle_and_per_graph_path(name, n_graphs, set_kw):
    wl, pairs, model, ue, ie = rec_model(name, n_graphs, **set_kw)
    assert model.graph_set() is not None
    with torch.no_grad():
        ui, het = model.propagate(model.embedding_table)
        model.fuse_graphs = False
        ui_p, het_p = model.propagate(model.embedding_table)
    assert model.graph_set() is None
    mk = lambda ui_: edges_t(ui_[0], ui_[1], wl.n_user)
    tu, ti, hu, hi = O.rec_forward(mk(pairs[0]), [mk(b) for b in pairs[1:]], ue, ie, 2)
    assert_rows_close(ui, torch.cat([tu, ti]), what=f"{name} set: target sums vs oracle")
    assert_rows_close(het, torch.cat([hu, hi]), what=f"{name} set: behaviour sums vs oracle")
    assert_rows_close(ui, ui_p, rtol=2e-6, what=f"{name} set vs per-graph: target sums")
    assert_rows_close(het, het_p, rtol=2e-6, what=f"{name} set vs per-graph: behaviour sums")


def test_graph_set_rows_summed_whole_equal_the_per_graph_path_bit_for_bit():
    """No cut rows on either path: both add a row's neighbours in ascending order, the epilogues are the same
    arithmetic -- the same bits."""
    wl, pairs, model, ue, ie = rec_model("tiny", 3, group_max=1016, seg_len=1024)
    assert model.graph_set().counts()["n_hub"] == 0
    with torch.no_grad():
        ui, het = model.propagate(model.embedding_table)
    wl, pairs, m2, _, _ = rec_model("tiny", 3)
    m2.fuse_graphs = False
    m2.target_adj, m2.behavior_mats = (lambda g: (g[0], g[1:]))(
        [dg.CSRGraph.from_bipartite(u, i, wl.n_user, wl.n_item, group_max=1000, seg_len=1024) for (u, i) in pairs])
    with torch.no_grad():
        ui_p, het_p = m2.propagate(m2.embedding_table)
    assert torch.equal(ui, ui_p) and torch.equal(het, het_p)


@pytest.mark.parametrize("name,n_graphs,set_kw", [("tiny", 3, {}), ("tiny", 3, dict(group_max=12, seg_len=64)),
                                                  ("retail_rocket", 3, {}), ("tmall", 4, {})])
def test_graph_set_backward_matches_per_graph_path(name, n_graphs, set_kw):
    wl, pairs, model, ue, ie = rec_model(name, n_graphs, **set_kw)
    gen = torch.Generator().manual_seed(7)
    d_ui, d_het = (torch.randn(wl.n, 64, generator=gen).to(DEV) for _ in range(2))
    grads = []
    for fuse in (True, False):
        model.fuse_graphs = fuse
        model.embedding_table.grad = None
        ui, het = model.propagate(model.embedding_table)
        assert (model.graph_set() is not None) == fuse
        ((ui * d_ui).sum() + (het * d_het).sum()).backward()
        grads.append(model.embedding_table.grad.clone())
    # rows with a few thousand neighbours are added in a different order of pieces on the two paths
    assert_rows_close(grads[0], grads[1], rtol=5e-6, what=f"{name}: d e0, graph set vs per-graph path")


def test_graph_set_backward_against_oracle_autograd():
    wl, pairs, model, ue, ie = rec_model("tiny", 3, group_max=12, seg_len=64)
    gen = torch.Generator().manual_seed(8)
    d_ui, d_het = (torch.randn(wl.n, 64, generator=gen) for _ in range(2))
    ui, het = model.propagate(model.embedding_table)
    ((ui * d_ui.to(DEV)).sum() + (het * d_het.to(DEV)).sum()).backward()
    mk = lambda ui_: edges_t(ui_[0], ui_[1], wl.n_user)
    ue64, ie64 = ue.double().requires_grad_(), ie.double().requires_grad_()
    tu, ti, hu, hi = O.rec_forward(mk(pairs[0]), [mk(b) for b in pairs[1:]], ue64, ie64, 2)
    ((torch.cat([tu, ti]) * d_ui.double()).sum() + (torch.cat([hu, hi]) * d_het.double()).sum()).backward()
    assert_rows_close(model.embedding_table.grad, torch.cat([ue64.grad, ie64.grad]), what="tiny set: d e0 vs fp64 oracle autograd")


@pytest.mark.parametrize("n_graphs,seed", [(2, 0), (3, 1), (4, 2), (4, 3)])
def test_graph_set_random_symmetric_graphs(n_graphs, seed):
    """General symmetric graphs (not bipartite), degrees from 0 to a few hundred, small group limits so that every row
    class and every empty-share case occurs; each stacked output against the single-graph layer."""
    rng = np.random.default_rng(seed)
    n = 700 + 37 * seed
    graphs = []
    for g in range(n_graphs):
        m = [30, 4000, 900, 12000][(g + seed) % 4]
        a = np.minimum((rng.pareto(1.2, m) * 3).astype(np.int64), n - 1)
        b = rng.integers(0, n, m)
        src, dst = np.concatenate([a, b]).astype(np.int32), np.concatenate([b, a]).astype(np.int32)
        graphs.append(dg.CSRGraph(src, dst, n))
        assert graphs[-1].is_symmetric
    gset = dg.GraphSet(graphs, group_max=[16, 64, 128, 40][seed], seg_len=64)
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(seed)).to(DEV)
    e1 = torch.empty(n_graphs, n, 64, device=DEV)
    ops.set_layer_first(gset, x, e1, 0.5)
    for g, graph in enumerate(graphs):
        y = ops.spmm_norm_act(graph, x, 0.5) * T(graph.norms()[0]).to(DEV)[:, None]
        assert_rows_close(e1[g], y, rtol=2e-6, what=f"random set: first layer, graph {g}")
    ui, het, e2 = torch.empty(n, 64, device=DEV), torch.empty(n, 64, device=DEV), torch.empty_like(e1)
    ops.set_layer_last(gset, e1, x, ui, het, e2, 0.5)
    norm = torch.nn.functional.normalize
    sums = []
    for g, graph in enumerate(graphs):
        dsrc = T(graph.norms()[0]).to(DEV)[:, None]
        y2 = ops.spmm_norm_act(graph, e1[g] / dsrc, 0.5)      # (the single-graph layer scales its input itself)
        assert_rows_close(e2[g], y2, rtol=3e-6, what=f"random set: last layer, graph {g}")
        sums.append(x + norm(e1[g] / dsrc, dim=1) + norm(y2, dim=1))
    assert_rows_close(ui, sums[0], rtol=3e-6, what="random set: first graph's layer sum")
    assert_rows_close(het, sum(sums[1:]), rtol=3e-6, what="random set: sum of the other graphs' layer sums")


def test_graph_set_refuses_what_it_cannot_walk():
    n = 64
    a = np.arange(n, dtype=np.int32)
    sym = dg.CSRGraph(np.concatenate([a, (a + 1) % n]), np.concatenate([(a + 1) % n, a]), n)
    directed = dg.CSRGraph(a, (a + 1) % n, n)
    other = dg.CSRGraph(np.concatenate([a, (a + 1) % n]), np.concatenate([(a + 1) % n, a]), n + 1)
    for graphs in ([sym], [sym] * 5, [sym, directed], [sym, other]):
        with pytest.raises(dg.DiffGraphHipError):
            dg.GraphSet(graphs)
    gset = dg.GraphSet([sym, sym])
    x = torch.zeros(n, 64, device=DEV)
    with pytest.raises(dg.DiffGraphHipError):
        ops.set_layer_first(gset, x, torch.empty(3, n, 64, device=DEV))
    with pytest.raises(dg.DiffGraphHipError):
        ops.set_layer_first(gset, x[:-1], torch.empty(2, n, 64, device=DEV))


@pytest.mark.parametrize("name,n_graphs,set_kw", [("tiny", 3, dict(group_max=12, seg_len=64)), ("tiny", 2, {}), ("ijcai", 4, {})])
def test_last_backward_step_with_the_optimizer_inside(name, n_graphs, set_kw):
    """dg_set_layer_bwd_last_adam against dg_set_layer_bwd_last followed by the table's Adam kernels over the parts
    [d_first, d_rest x (graphs - 1), g0]: the same bits in the table, both moments and the step count, three steps in a
    row (weight decay on in the last one)."""
    import ctypes as C
    from diffgraph_amd import _lib
    wl, pairs, model, ue, ie = rec_model(name, n_graphs, **set_kw)
    gset = model.graph_set()
    n, G = wl.n_user + wl.n_item, n_graphs
    gen = torch.Generator().manual_seed(3)
    lib = _lib.load()
    state = {k: [torch.randn(n, 64, generator=gen).to(DEV) * 0.1, torch.zeros(n, 64, device=DEV), torch.zeros(n, 64, device=DEV),
                 torch.zeros(4, device=DEV)] for k in ("fused", "parts")}
    state["parts"][0].copy_(state["fused"][0])
    for step, wd in enumerate((0.0, 0.0, 0.01)):
        w_in = torch.randn(G, n, 64, generator=gen).to(DEV)
        d_first, d_rest = torch.randn(n, 64, generator=gen).to(DEV), torch.randn(n, 64, generator=gen).to(DEV)
        hyper = (1e-3, 0.9, 0.999, 1e-8, wd)
        p, m, v, st = state["fused"]
        ops.set_layer_bwd_last_adam(gset, w_in, d_first, d_rest, p, m, v, st, *hyper)
        p, m, v, st = state["parts"]
        g0 = torch.empty(n, 64, device=DEV)
        ops.set_layer_bwd_last(gset, w_in, g0)
        parts = [d_first] + [d_rest] * (G - 1) + [g0]
        arr = (C.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
        _lib.check(lib.dg_adam_prepare(st.data_ptr(), 0.9, 0.999, _lib.stream_ptr(p.device)), "dg_adam_prepare")
        _lib.check(lib.dg_adam_step_prepared(p.data_ptr(), arr, len(parts), m.data_ptr(), v.data_ptr(), st.data_ptr(), p.numel(),
                                             *hyper, _lib.stream_ptr(p.device)), "dg_adam_step_prepared")
        torch.cuda.synchronize()
        for a, b, what in zip(state["fused"][:3], state["parts"][:3], ("table", "exp_avg", "exp_avg_sq")):
            assert torch.equal(a, b), (step, what)
        assert float(state["fused"][3][0]) == float(state["parts"][3][0]) == step + 1
