"""What one rank of a W-rank Industry run spends per forward, measured on ONE GPU: the rank-0 shard is built as for
a W-rank job, the exchange is replaced by a local copy of the rank's own partial sums (same bytes, no peers), and
the script reports the device time per forward, the host time to enqueue it, and the per-kernel split.
Usage: python tools/shard_probe.py [W ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffgraph_amd as dg  # noqa: E402
from diffgraph_amd import data as dgdata  # noqa: E402
from diffgraph_amd.dist import HipBackend, ShardedRecPropagation, _Pending  # noqa: E402


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    dev = torch.device("cuda:0")
    wl = dgdata.rec_workload("industry")
    inter = [wl.target] + wl.behaviours
    ie = torch.from_numpy(dgdata.xavier_uniform(wl.n_item, 64, seed=1))
    ue_all = dgdata.xavier_uniform(wl.n_user, 64, seed=0)
    for w in worlds:
        p = ShardedRecPropagation(inter, wl.n_user, wl.n_item, 0, w, HipBackend(dev), reuse_buffers=True)
        side = torch.cuda.Stream()

        class _Done:
            def __init__(self, ev):
                self.ev = ev

            def wait(self):
                torch.cuda.current_stream().wait_event(self.ev)

        def exchange(partials, out):
            # stand-in for the collective: a copy on another stream that starts after what is enqueued so far
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out.copy_(partials.unsqueeze(0).expand_as(out))
                ev = torch.cuda.Event()
                ev.record(side)
            return _Pending(out, _Done(ev))
        p._exchange = exchange
        e0 = torch.cat([torch.from_numpy(ue_all[p.u0:p.u1]), ie]).to(dev)
        for _ in range(5):
            p.forward(e0)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            p.forward(e0)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        # device time alone: events around one forward, queue kept short
        evs = []
        for _ in range(10):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            p.forward(e0)
            b.record()
            torch.cuda.synchronize()
            evs.append(a.elapsed_time(b))
        # the same pass recorded as a HIP graph (stand-in exchange included) and replayed
        graph, _ = p.capture(e0)
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        t_graph = time.perf_counter() - t0
        print(f"W={w}: users/rank {p.n_loc}  wall/forward {t_all / n * 1e3:.3f} ms  host enqueue {t_enq / n * 1e3:.3f} ms  "
              f"events (sync before each) min {min(evs):.3f} med {sorted(evs)[5]:.3f} ms   ideal {1.446 / w:.3f} ms   "
              f"replayed as a graph: wall/forward {t_graph / n * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
