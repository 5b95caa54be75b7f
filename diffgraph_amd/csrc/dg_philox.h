// Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): a counter-based generator, so a
// draw depends only on (key, counter) and can be regenerated anywhere -- the sampling kernel keys it by pair and
// attempt, the dropout of the diffusion loss by row, so that its backward pass redraws the forward's mask.
#pragma once
#include <stdint.h>

struct dg_u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ dg_u32x4 dg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
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
