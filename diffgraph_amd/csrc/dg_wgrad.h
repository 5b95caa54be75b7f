// Pieces shared by the kernels that form the denoiser's weight gradients: the layout of one partial result (the sums
// over one strip of table rows, added up by wgrad_finalize_kernel in strip order) and the exact-fp32 16x16x4 matrix
// tiles that contract over table rows.
#pragma once
#include <hip/hip_runtime.h>

namespace dg_wg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTabCols = 16;  // per-timestep operand row: emb[t] (8) | te[t] (8)
// floats of one partial result: dW_out, dW_in[:, :64], dpre^T.tab (64x16), db_out, db_in
constexpr int kPartWout = 0, kPartWin = 4096, kPartTab = 8192, kPartBout = 9216, kPartBin = 9280;
constexpr int kPartFloats = 9344;

// acc[mb][nb] += A-tile mb (features 4j+mb of the `a` rows) ^T . B-tile nb (features 4j+nb of the `b` rows):
// lane (j = lane & 15, k = lane >> 4) brings float4s of features 4j .. 4j+3 of table row base + k.
__device__ __forceinline__ void mma64(const f32x4 a, const f32x4 b, f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
      acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
}

// Element (o, i) of a 64x64 product sits in acc[mb][nb][r] of lane l with
//   o = 16 (l>>4) + 4 r + mb,  i = 4 (l&15) + nb:   the four nb of one (mb, r) are a contiguous float4.
__device__ __forceinline__ void store64_mb(float *dst, int lane, int mb, const f32x4 (&acc)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 16 * (lane >> 4) + 4 * r + mb;
    *reinterpret_cast<f32x4 *>(dst + o * 64 + 4 * (lane & 15)) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
  }
}

__device__ __forceinline__ void store64(float *dst, int lane, const f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) store64_mb(dst, lane, mb, acc[mb]);
}

}  // namespace dg_wg
