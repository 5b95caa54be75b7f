// Device helpers of the row-gather kernels: 16-lane row groups, DPP broadcasts and sums, buffer-descriptor
// row loads (an id of -1 gives an out-of-range offset, answered with zeros and no memory access).
#pragma once

#include "dg_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <int K>
__device__ __forceinline__ int bcast16(int v) {
  // lane K of each row of 16 lanes -> every lane of that row
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false);
}
template <int K>
__device__ __forceinline__ float bcast16f(float v) {
  return __builtin_bit_cast(float, bcast16<K>(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const float *base, int32_t rows) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, rows * (DG_D * 4), 0x00020000);
}

__device__ __forceinline__ f32x4 load_row16(__amdgpu_buffer_rsrc_t r, int row, unsigned lane_off) {
  unsigned off = (unsigned)row * (unsigned)(DG_D * 4) + lane_off;  // row == -1 -> out of range -> zeros
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// Tables the epilogue streams through once (running sums, stored activations).  Plain accesses: the
// non-temporal hint made the layer 2-7 % slower on the Tmall-shaped graphs.
__device__ __forceinline__ f32x4 load_stream(const float *base, size_t o4) {
  return reinterpret_cast<const f32x4 *>(base)[o4];
}
__device__ __forceinline__ void store_stream(float *base, size_t o4, f32x4 v) {
  reinterpret_cast<f32x4 *>(base)[o4] = v;
}

// Four neighbours K0..K0+3 of the current chunk: issue the four row loads, then add.
template <int K0, bool SCALE>
__device__ __forceinline__ void gather4(__amdgpu_buffer_rsrc_t xr, int c, float s, unsigned lane_off, f32x4 &acc) {
  const int c0 = bcast16<K0>(c), c1 = bcast16<K0 + 1>(c), c2 = bcast16<K0 + 2>(c), c3 = bcast16<K0 + 3>(c);
  f32x4 v0 = load_row16(xr, c0, lane_off);
  f32x4 v1 = load_row16(xr, c1, lane_off);
  f32x4 v2 = load_row16(xr, c2, lane_off);
  f32x4 v3 = load_row16(xr, c3, lane_off);
  if (SCALE) {  // node_f * norm, rounded, before the sum (Model.py:157)
    v0 *= bcast16f<K0>(s); v1 *= bcast16f<K0 + 1>(s); v2 *= bcast16f<K0 + 2>(s); v3 *= bcast16f<K0 + 3>(s);
  }
  acc += v0; acc += v1; acc += v2; acc += v3;
}

// Adds the table rows listed in colidx[p .. end), taken 16 at a time with stride `stride`
// between chunks (16: the group owns the whole list; 64: four groups interleave).
// INDEXED = false reads rows p, p+1, ... themselves (slab rows of a hub row).
template <bool INDEXED, bool SCALE>
__device__ __forceinline__ void gather_rows(__amdgpu_buffer_rsrc_t xr, const int32_t *__restrict__ colidx,
                                            const float *__restrict__ colscale, int p, int end, int stride, int l,
                                            f32x4 &acc) {
  const unsigned lane_off = (unsigned)l * 16u;
  int c = -1;
  float s = 0.f;
  if (p + l < end) {
    c = INDEXED ? colidx[p + l] : p + l;
    if (SCALE) s = colscale[p + l];
  }
  while (p < end) {
    const int pn = p + stride;
    // The next chunk's ids (and norms) are in flight while this chunk's rows load.  The load is UNCONDITIONAL (its
    // address clamped into the list, the value dropped afterwards): a load under a branch leaves the number of
    // requests in flight unknown to the compiler, which then waits for ALL of them before the first use of this
    // chunk's ids -- the prefetch included.
    int cn = pn + l;
    float sn = 0.f;
    if (INDEXED) {
      const int at = min(pn + l, end - 1);
      cn = colidx[at];
      if (SCALE) sn = colscale[at];
    }
    if (pn + l >= end) {
      cn = -1;
      sn = 0.f;
    }
    const int cnt = end - p;
    gather4<0, SCALE>(xr, c, s, lane_off, acc);
    if (cnt > 4) gather4<4, SCALE>(xr, c, s, lane_off, acc);
    if (cnt > 8) gather4<8, SCALE>(xr, c, s, lane_off, acc);
    if (cnt > 12) gather4<12, SCALE>(xr, c, s, lane_off, acc);
    c = cn; s = sn; p = pn;
  }
}

__device__ __forceinline__ f32x4 add_groups(f32x4 v) {  // (g0+g1)+(g2+g3), same value in every group
  f32x4 o;
  o.x = __shfl_xor(v.x, 16); o.y = __shfl_xor(v.y, 16); o.z = __shfl_xor(v.z, 16); o.w = __shfl_xor(v.w, 16);
  v += o;
  o.x = __shfl_xor(v.x, 32); o.y = __shfl_xor(v.y, 32); o.z = __shfl_xor(v.z, 32); o.w = __shfl_xor(v.w, 32);
  return v + o;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// Sum over the 16 lanes of a group, left in every lane: two quad permutes, then the 8-lane and
// 16-lane mirrors (after the quad steps all lanes of a quad agree, so mirroring pairs the right
// partial sums).  Four VALU instructions; no LDS crossbar traffic.
__device__ __forceinline__ float sum16(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);  // row_half_mirror
  v += dpp_f<0x140>(v);  // row_mirror
  return v;
}

// 1 / max(sqrt(ss), 1e-12): the factor F.normalize divides by, from the hardware reciprocal
// square root (1 ulp), so the normalised row is within ~2 ulp of the reference's division.
__device__ __forceinline__ float inv_norm(float ss) { return fminf(__builtin_amdgcn_rsqf(ss), 1e12f); }

}  // namespace
