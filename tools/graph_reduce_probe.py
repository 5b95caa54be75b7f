"""Does a torch column reduction captured in a HIP graph return wrong values when the host synchronises and runs
eager reductions between replays?  (The pattern under which a replayed training step once produced NaN in a bias
gradient that was computed as dout.sum(0).)  Prints the number of bad replays per variant."""
import sys
import torch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 31232
x = torch.randn(n, 64, device=dev)
others = [torch.randn(257, 64, device=dev) for _ in range(12)]


def run(label, fn, sync, reps=300):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fn()
    torch.cuda.synchronize()
    want = fn().clone()
    bad = 0
    for it in range(reps):
        g.replay()
        if sync:
            torch.cuda.synchronize()
            if "no eager" not in label:
                junk = [bool(torch.isfinite(o).all()) for o in others]       # eager reductions between replays
            if not torch.isfinite(y).all() or not torch.allclose(y, want, rtol=1e-4, atol=1e-4):
                bad += 1
    torch.cuda.synchronize()
    if not sync and (not torch.isfinite(y).all() or not torch.allclose(y, want, rtol=1e-4, atol=1e-4)):
        bad += 1
    print(f"{label:50s} sync={sync!s:5s} bad replays: {bad}", flush=True)


w = torch.randn(64, 64, device=dev)
for sync in (True, False):
    run("x.sum(0)", lambda: x.sum(0), sync)
    run("(x * 1.5).sum(0) (temporary inside the graph)", lambda: (x * 1.5).sum(0), sync)
    run("x.t() @ x  and  x.sum(0)", lambda: (x.t() @ x).sum(0) + x.sum(0), sync)
    run("x.t() @ x alone (K = n rows)", lambda: x.t() @ x, sync)
    run("x @ w alone (K = 64)", lambda: x @ w, sync)
    run("x.t() @ x, no eager work between replays", lambda: x.t() @ x, sync) if sync else None
