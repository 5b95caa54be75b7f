"""Run by tests/test_zy_rccl.py in a process of its own: the two RCCL call sites of the row-sharded propagation on a
ONE-rank communicator (all a one-GPU box allows) -- the calls, their arguments and the stream hand-over are those of a
multi-rank run; with one rank the gathered tensor equals the input.  Prints RCCL-OK <what> per check; exit code 3 =
no communicator could be made here (the caller skips)."""
import ctypes as C
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29581")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffgraph_amd import _lib  # noqa: E402
from diffgraph_amd import dist as dgdist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

# 1. torch.distributed's "nccl" backend (RCCL): the asynchronous all_gather_into_tensor of dist.all_gather_start
try:
    dist.init_process_group("nccl", rank=0, world_size=1)
except Exception as exc:  # no interface to bootstrap on, ...
    print("no process group:", exc)
    sys.exit(3)
part = torch.randn(3, 361, 64, device=dev)
out = torch.full((1, 3, 361, 64), float("nan"), device=dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side):          # (the propagation runs on a stream of its own)
    pending = dgdist._gather_async(part, 1, None, out)
    other = part * 2.0                 # work enqueued behind the collective
    got = pending.wait()
side.synchronize()
assert got.data_ptr() == out.data_ptr() and torch.equal(got[0], part) and torch.equal(other, part * 2.0)
print("RCCL-OK torch.distributed all_gather_into_tensor(async_op=True)")

# ... and a whole sharded forward pass with the collective in it (buffers reused over three passes) against the same
# pass without: the exchange is enqueued between the partial-sum launch and the user rows' launch, waited for before the
# finalize launch reads what it gathered
from diffgraph_amd import data as dgdata  # noqa: E402
nu, ni = 9000, 37
inter = [dgdata.synth_interactions(nu, ni, m, seed) for m, seed in ((30000, 1), (80000, 2), (12000, 3))]
e0 = torch.randn(nu + ni, 64, device=dev)
plain = dgdist.ShardedRecPropagation(inter, nu, ni, 0, 1, dgdist.HipBackend(dev))
want = [t.clone() for t in plain.forward(e0)]
prop = dgdist.ShardedRecPropagation(inter, nu, ni, 0, 1, dgdist.HipBackend(dev), reuse_buffers=True)
prop.always_collective = True
assert prop.split_rows
for _ in range(3):
    ui, het = prop.forward(e0)
torch.cuda.synchronize()
assert torch.equal(ui, want[0]) and torch.equal(het, want[1])
print("RCCL-OK sharded forward with the collective")

# ... and the same pass, collective included, recorded as a HIP graph and replayed over a changed table
graph, (gui, ghet) = prop.capture(e0)
e0.mul_(1.5)
want = [t.clone() for t in plain.forward(e0)]
for _ in range(3):
    gui.fill_(float("nan"))
    ghet.fill_(float("nan"))
    graph.replay()
torch.cuda.synchronize()
assert torch.equal(gui, want[0]) and torch.equal(ghet, want[1])
del graph
print("RCCL-OK sharded forward recorded as a graph")
dist.destroy_process_group()

# 2. dg_allgather_rows on a communicator made with the RCCL the library resolves (the process's, else librccl.so)
glob = C.CDLL(None)
if not hasattr(glob, "ncclCommInitRank"):
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    glob = C.CDLL(None)
if not hasattr(glob, "ncclCommInitRank"):
    print("no RCCL library")
    sys.exit(3)


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


uid, comm = UniqueId(), C.c_void_p()
glob.ncclGetUniqueId.argtypes, glob.ncclGetUniqueId.restype = [C.POINTER(UniqueId)], C.c_int
glob.ncclCommInitRank.argtypes, glob.ncclCommInitRank.restype = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int], C.c_int
glob.ncclCommDestroy.argtypes, glob.ncclCommDestroy.restype = [C.c_void_p], C.c_int
if glob.ncclGetUniqueId(C.byref(uid)) != 0 or glob.ncclCommInitRank(C.byref(comm), 1, uid, 0) != 0:
    print("no communicator")
    sys.exit(3)
lib = _lib.load()
local = torch.randn(1083, 64, device=dev)
full = torch.full((1083, 64), float("nan"), device=dev)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    _lib.check(lib.dg_allgather_rows(comm, local.data_ptr(), full.data_ptr(), 1083, 64, _lib.stream_ptr(dev)), "dg_allgather_rows")
stream.synchronize()
assert torch.equal(full, local)
print("RCCL-OK dg_allgather_rows")
glob.ncclCommDestroy(comm)
