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

namespace {

struct u32x4_t { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4_t philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return {c0, c1, c2, c3};
}

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
    const u32x4_t r = philox4x32_10((uint32_t)p, (uint32_t)(p >> 32), attempt, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
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
