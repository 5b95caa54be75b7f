// The one collective of the row-sharded propagation as a C entry point, for callers that hold an RCCL communicator
// of their own (diffgraph_amd.dist goes through torch.distributed, whose backend "nccl" is RCCL on ROCm).
// RCCL is resolved at the first call -- from the process if it is already loaded there (torch brings its own copy),
// else from librccl.so -- so the library carries no link-time dependency on it.
#include "dg_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

using all_gather_fn = decltype(&ncclAllGather);
using error_string_fn = decltype(&ncclGetErrorString);

all_gather_fn g_all_gather = nullptr;
error_string_fn g_error_string = nullptr;

int resolve() {
  static std::atomic<int> state{0};   // 0: not tried, 1: found, -1: missing
  int st = state.load(std::memory_order_acquire);
  if (st == 0) {
    void *sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
    void *lib = nullptr;
    if (!sym) {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
      }
      if (lib) sym = dlsym(lib, "ncclAllGather");
    }
    if (sym) {
      g_all_gather = reinterpret_cast<all_gather_fn>(sym);
      g_error_string = reinterpret_cast<error_string_fn>(lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    }
    st = sym ? 1 : -1;
    state.store(st, std::memory_order_release);
  }
  return st;
}

}  // namespace

extern "C" int dg_allgather_rows(void *comm, const float *local, float *full, int64_t rows_per_rank, int32_t d, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_allgather_rows: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!comm) return dg_fail(DG_ERR_ARG, "dg_allgather_rows: comm is NULL (an ncclComm_t of the caller's RCCL)");
  if (rows_per_rank < 0) return dg_fail(DG_ERR_ARG, "dg_allgather_rows: negative row count");
  if (rows_per_rank == 0) return 0;
  if (!local || !full) return dg_fail(DG_ERR_ARG, "dg_allgather_rows: NULL buffer");
  if (resolve() < 0) return dg_fail(DG_ERR_ARG, "dg_allgather_rows: RCCL (ncclAllGather) is neither loaded in this process nor found as librccl.so");
  const ncclResult_t rc = g_all_gather(local, full, (size_t)rows_per_rank * DG_D, ncclFloat32, static_cast<ncclComm_t>(comm),
                                       static_cast<hipStream_t>(stream));
  if (rc != ncclSuccess)
    return dg_fail(DG_ERR_ARG, "dg_allgather_rows: ncclAllGather failed: %s", g_error_string ? g_error_string(rc) : "(no error text)");
  return 0;
}
