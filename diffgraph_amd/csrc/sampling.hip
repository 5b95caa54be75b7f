// Negative sampling for gfx950.
//
// Replaces PairwiseTrnData.negSampling (DiffGraph-Rec/DataHandler.py:282-289): for every training
// pair (u, i) draw item ids uniformly from [0, n_item) until one is found that u has no training
// interaction with.  The reference does this once per epoch in a Python loop with dictionary
// look-ups; here one thread serves one pair, the look-up is a binary search in the user's sorted
// training row, and the draws come from a counter-based generator (Philox-4x32-10 keyed by the
// seed, counter = pair index and attempt), so a given (seed, pair) always yields the same negative
// whatever the launch shape.
#include "dg_internal.h"
#include "dg_philox.h"

namespace {

// Uniform integer in [0, n) from 32 random bits: multiply-shift with rejection of the biased low range.
__device__ __forceinline__ bool bounded(uint32_t r, uint32_t n, uint32_t &out) {
  const uint64_t m = (uint64_t)r * n;
  if ((uint32_t)m < n) {
    const uint32_t t = (0u - n) % n;
    if ((uint32_t)m < t) return false;
  }
  out = (uint32_t)(m >> 32);
  return true;
}

__global__ __launch_bounds__(256) void neg_sampling_kernel(const int32_t *__restrict__ trn_rowptr,
                                                           const int32_t *__restrict__ trn_col,
                                                           const int32_t *__restrict__ rows, int64_t n_pairs, int32_t n_item,
                                                           uint64_t seed, int32_t *__restrict__ negs) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pairs) return;
  const int u = rows[p];
  const int b = trn_rowptr[u], e = trn_rowptr[u + 1];
  if (e - b >= n_item) {  // nothing to draw: the reference would loop for ever
    negs[p] = -1;
    return;
  }
  for (uint32_t attempt = 0;; ++attempt) {
    const dg_u32x4 r = dg_philox4x32_10((uint32_t)p, (uint32_t)(p >> 32), attempt, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t draws[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t cand;
      if (!bounded(draws[k], (uint32_t)n_item, cand)) continue;
      int lo = b, hi = e;  // is cand in the ascending row trn_col[b, e)?
      bool found = false;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1, c = trn_col[mid];
        if (c == (int)cand) { found = true; break; }
        if (c < (int)cand) lo = mid + 1; else hi = mid;
      }
      if (!found) {
        negs[p] = (int32_t)cand;
        return;
      }
    }
  }
}

}  // namespace

extern "C" int dg_neg_sampling(const int32_t *trn_rowptr, const int32_t *trn_col, const int32_t *rows, int64_t n_pairs,
                               int32_t n_user, int32_t n_item, uint64_t seed, int32_t *negs, void *stream) {
  if (!trn_rowptr || !trn_col || !rows || !negs) return dg_fail(DG_ERR_ARG, "dg_neg_sampling: NULL argument");
  if (n_pairs < 0 || n_user <= 0 || n_item <= 0) return dg_fail(DG_ERR_ARG, "dg_neg_sampling: bad sizes");
  if (n_pairs == 0) return 0;
  const int64_t blocks = (n_pairs + 255) / 256;
  if (blocks > 0x7fffffff) return dg_fail(DG_ERR_SIZE, "dg_neg_sampling: too many pairs");
  hipLaunchKernelGGL(neg_sampling_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, trn_rowptr, trn_col, rows,
                     n_pairs, n_item, seed, negs);
  DG_HIP(hipGetLastError());
  return 0;
}
