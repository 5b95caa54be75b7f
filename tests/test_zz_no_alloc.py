"""dg_denoiser_wgrad_static -- the entry point that allocates nothing (its scratch is a device array inside the
library) -- against dg_denoiser_wgrad with a caller-supplied workspace: same kernels, same summation order, so the
six gradients must be the same bits.  Called through the C ABI, twice in a row, the second time on a smaller
input (the scratch keeps the first call's contents), then once more past the built-in limit."""
import ctypes as C

import pytest
import torch

import diffgraph_amd as dg
from diffgraph_amd import _lib, ops

pytestmark = pytest.mark.gpu
SHAPES = ((8, 8), (8,), (64, 72), (64,), (64, 64), (64,))   # d_w_emb, d_b_emb, d_w_in, d_b_in, d_w_out, d_b_out


def _inputs(n, n_steps, seed, dev):
    gen = torch.Generator().manual_seed(seed)
    ts = torch.randint(0, n_steps, (n,), generator=gen).to(dev)
    tabs = [torch.randn(n, 64, generator=gen).to(dev) for _ in range(4)]      # dout_tot, hid, dpre, xn
    return ts, tabs


def _static(lib, p, n_steps, ts, tabs, dev):
    outs = [torch.full(s, float("nan"), dtype=torch.float32, device=dev) for s in SHAPES]
    rc = lib.dg_denoiser_wgrad_static(C.byref(p.c), n_steps, ts.data_ptr(), *[t.data_ptr() for t in tabs], ts.numel(),
                                      _lib.DG_D, *[o.data_ptr() for o in outs], _lib.stream_ptr(dev))
    return rc, outs


def test_static_scratch_entry_point_matches_workspace_entry_point():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    torch.manual_seed(11)
    dn = dg.Denoise([64, 64], [64, 64], 8, norm=True).to(dev)
    p = dn.params(True)
    n_steps = 200
    assert lib.dg_denoiser_wgrad_static_max_steps() >= n_steps
    for n, seed in ((40001, 1), (1234, 2)):           # the second call is smaller and finds the first one's scratch
        ts, tabs = _inputs(n, n_steps, seed, dev)
        want = ops.denoiser_wgrad(p, n_steps, ts, *tabs)       # dg_denoiser_wgrad_batched with a torch-allocated workspace
        rc, got = _static(lib, p, n_steps, ts, tabs, dev)
        _lib.check(rc, "dg_denoiser_wgrad_static")
        torch.cuda.synchronize()
        for nm, g, w in zip(("w_emb", "b_emb", "w_in", "b_in", "w_out", "b_out"), got, want):
            assert not torch.isnan(g).any(), nm
            assert torch.equal(g, w), f"n={n} grad {nm}: max diff {(g - w).abs().max().item():.3e}"
    # beyond the built-in scratch: refused with a clear message, nothing launched
    too_many = lib.dg_denoiser_wgrad_static_max_steps() + 1
    ts, tabs = _inputs(64, too_many, 3, dev)
    rc, _ = _static(lib, p, too_many, ts, tabs, dev)
    assert rc == -3
    assert b"exceed the built-in scratch" in lib.dg_last_error()
