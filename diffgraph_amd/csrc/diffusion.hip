// Latent diffusion kernels for gfx950: q_sample, the timestep-conditioned denoiser MLP,
// the SNR-weighted training loss and the T-step reverse (p_sample) loop.
//
// Replaces GaussianDiffusion.q_sample / p_sample / p_mean_variance / training_losses2
// (DiffGraph-Rec/Model.py:287-366) and Denoise.forward (DiffGraph-Rec/Model.py:214-234).
//
// The denoiser is two 64x64 fp32 contractions per row (the 8 time-embedding columns of
// in_layers[0] depend only on t and collapse into a per-timestep bias table).  They run on
// the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 in TRANSPOSED orientation:
//     H^T[out][row] = W[out][in] . X^T[in][row]
// A wavefront owns 32 embedding rows.  The weight matrix is the A operand (read from an
// LDS image laid out lane-linear, so the reads are conflict-free); the activations are the
// B operand.  In this orientation the accumulator layout (row of the tile on the lane,
// feature index  8*(r>>2) + 4*(lane>>5) + (r&3)  in register r) is exactly the B-operand
// layout of the next contraction, so x, tanh(.) and the output never leave registers:
// the whole 200-step reverse loop is one launch with no memory traffic inside the loop
// beyond 256 B of bias per step.
#include "dg_internal.h"
#include "dg_philox.h"
#include "dg_wgrad.h"

#include <math.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kDnMaxTasks = 4;
constexpr int kRowsPerWave = 32;
constexpr int kWavesPerBlock = 4;
constexpr int kRowsPerBlock = kRowsPerWave * kWavesPerBlock;
constexpr int kThreads = kWavesPerBlock * 64;

enum { MODE_FWD = 0, MODE_LOSS = 1, MODE_PSAMPLE = 2 };

struct dn_task {
  const float *w_in, *w_out, *b_out;
  const float *time_bias;             // [n_steps][64]
  const float *x;                     // FWD: x; LOSS: x_start; PSAMPLE: x_T
  const float *noise, *target;        // LOSS
  const int64_t *t;                   // FWD, LOSS: per-row timestep
  const float *sqrt_ac, *sqrt_1mac;   // LOSS (NULL: x_t = x_start)
  const float *coef1, *coef2;         // PSAMPLE
  const double *loss_w;               // LOSS
  float *out;
  double *row_loss;
  float *x_t_out;
  // backward of the loss (loss_bwd_kernel)
  const double *g_row;                // dL/d row_loss [n]
  const float *g_out;                 // dL/d out [n][64] or NULL
  float *d_x_start, *d_target;        // [n][64]
  const float *d_target_add;          // [n][64] added into d_target (a gradient that reached the target elsewhere), or NULL
  float *dout_tot, *hid, *dpre, *xn;  // [n][64] each: operands of the weight-gradient GEMMs (not written when wg_partials is set)
  const float *wg_tab;                // [n_steps][16] operand table of the time-embedding gradients
  float *wg_partials;                 // one partial result of the weight gradients per workgroup, or NULL
  // dropout between F.normalize and the first Linear (LOSS and its backward; Model.py:221-224)
  const uint8_t *keep_mask;           // [n][64], nonzero = kept; NULL: drawn from drop_seed
  const int64_t *drop_seed;           // one device word: the key of the draw
  float drop_scale;                   // 1 / (1 - rate applied); 0: no dropout
  uint32_t drop_thr, drop_stream;     // kept iff the element's random byte >= drop_thr; stream = task number
  int64_t n;
  int32_t n_steps, norm, blk_begin;
  // rows != NULL: the task's workgroups take the table rows rows[0 .. *n_rows) (at most rows_cap of them) instead of
  // rows 0 .. n: slot s of the launch is table row rows[s]
  const int32_t *rows, *n_rows;
  int32_t rows_cap;
};

// The table row of a launch slot, and whether the slot holds one.  *none: the workgroup's first slot is already past
// the listed rows (the same for every thread of the workgroup).
__device__ __forceinline__ int64_t slot_row(const dn_task &T, int64_t slot, int64_t block_first, bool &valid, bool &none) {
  none = false;
  if (T.rows == nullptr) {
    valid = slot < T.n;
    return slot;
  }
  const int64_t cnt = min((int64_t)T.n_rows[0], (int64_t)T.rows_cap);
  none = block_first >= cnt;
  valid = slot < cnt;
  return valid ? (int64_t)T.rows[slot] : 0;
}

struct dn_batch {
  dn_task t[kDnMaxTasks];
  int32_t n_tasks;
};

// Tile register r of lane half h holds feature 8*(r>>2) + 4*h + (r&3) of a 32-feature block;
// the four registers 4q..4q+3 are the contiguous floats [8q+4h, 8q+4h+4).
__device__ __forceinline__ void load_tile(const float *__restrict__ rowp, int h, bool valid, f32x16 (&v)[2]) {
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 f = {0.f, 0.f, 0.f, 0.f};
      if (valid) f = *reinterpret_cast<const f32x4 *>(rowp + 32 * tau + 8 * q + 4 * h);
      v[tau][4 * q + 0] = f.x; v[tau][4 * q + 1] = f.y; v[tau][4 * q + 2] = f.z; v[tau][4 * q + 3] = f.w;
    }
}

__device__ __forceinline__ void store_tile(float *__restrict__ rowp, int h, bool valid, const f32x16 (&v)[2]) {
  if (!valid) return;
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 f = {v[tau][4 * q + 0], v[tau][4 * q + 1], v[tau][4 * q + 2], v[tau][4 * q + 3]};
      *reinterpret_cast<f32x4 *>(rowp + 32 * tau + 8 * q + 4 * h) = f;
    }
}

// Weight-gradient staging in LDS: a 32-row tile with rows padded to 68 floats (the 16-byte column pieces of
// consecutive rows then start in different banks), two tiles per wavefront.
constexpr int kWgRowFloats = 68, kWgTileFloats = kRowsPerWave * kWgRowFloats;
constexpr int kWgLdsF4 = kWavesPerBlock * 2 * kWgTileFloats / 4;   // 4352 float4 = 68 KiB (the weight images take 4096)
static_assert(kWgLdsF4 >= 4 * 2 * 2 * 4 * 64, "the staging area also holds the four weight images");

// Writes the lane's tile (row j, lane half h) row-major into a staged tile; rows outside the table as zeros.
__device__ __forceinline__ void stage_tile(float *__restrict__ tile, int j, int h, bool valid, const f32x16 (&v)[2]) {
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 f = {v[tau][4 * q + 0], v[tau][4 * q + 1], v[tau][4 * q + 2], v[tau][4 * q + 3]};
      if (!valid) f = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4 *>(tile + j * kWgRowFloats + 32 * tau + 8 * q + 4 * h) = f;
    }
}

// acc[to] += W[32*to .. +32][:] . act  for one 64x64 weight image in LDS.
__device__ __forceinline__ void contract64(const f32x4 *__restrict__ wl, int lane, const f32x16 (&act)[2],
                                           f32x16 (&acc)[2]) {
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 a0 = wl[((0 * 2 + ti) * 4 + rq) * 64 + lane];
      const f32x4 a1 = wl[((1 * 2 + ti) * 4 + rq) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], act[ti][4 * rq + e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], act[ti][4 * rq + e], acc[1], 0, 0, 0);
      }
    }
}

// The 32 random bytes of one (row, lane half): element tau*16 + 4q + e of the lane's tile uses byte number
// 4 (4 tau + q) + e.  Two Philox calls keyed by the seed word; the counter names the row, the half and the stream.
__device__ __forceinline__ void drop_bytes(uint64_t seed, uint32_t stream, int64_t row, int h, uint32_t (&w)[8]) {
  const dg_u32x4 a = dg_philox4x32_10((uint32_t)row, (uint32_t)((uint64_t)row >> 32), 2u * h, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
  const dg_u32x4 b = dg_philox4x32_10((uint32_t)row, (uint32_t)((uint64_t)row >> 32), 2u * h + 1u, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}

// nn.Dropout's multiplier for the lane's tile: 1/(1-p) where the element is kept, 0 where it is dropped.
__device__ __forceinline__ void drop_tile(const dn_task &T, int64_t row, int h, bool valid, f32x16 (&m)[2]) {
  uint32_t w[8];
  if (T.keep_mask != nullptr) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      w[c] = 0;
      if (valid) w[c] = *reinterpret_cast<const uint32_t *>(T.keep_mask + row * DG_D + 32 * (c >> 2) + 8 * (c & 3) + 4 * h);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) m[c >> 2][4 * (c & 3) + e] = ((w[c] >> (8 * e)) & 0xffu) ? T.drop_scale : 0.f;
    return;
  }
  drop_bytes((uint64_t)T.drop_seed[0], T.drop_stream, row, h, w);
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) m[c >> 2][4 * (c & 3) + e] = ((w[c] >> (8 * e)) & 0xffu) >= T.drop_thr ? T.drop_scale : 0.f;
}

// tanh on the hardware's exp2 and reciprocal (1 ulp each): 1 - 2 / (e^{2|x|} + 1) from |x| = 1/4 on, the odd series
// below (where the first form would cancel); within 5e-7 of tanhf, relative, at a fifth of its instructions.  For
// the single-step kernels, whose wavefronts wait on it, and for the reverse loop on the bf16 instruction, whose time
// is the vector unit's; the reverse loop on the fp32 instruction keeps tanhf (its time is the matrix instructions'
// chain: the short form measured no faster there).
__device__ __forceinline__ float tanh_short(float x) {
  const float ax = fabsf(x), x2 = ax * ax;
  const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);      // e^{2|x|}
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float small = ax * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  return copysignf(ax < 0.25f ? small : big, x);
}

// One Denoise.forward on the 32-row tile held in xs: F.normalize, (dropout,) in-layer + tanh, out-layer.
template <bool SHORT_TANH = false>
__device__ __forceinline__ void denoise_tile(const f32x4 *__restrict__ wl, int lane, int h, bool norm,
                                             const float *__restrict__ tb_row, const float *__restrict__ b_out,
                                             const f32x16 (&xs)[2], f32x16 (&out)[2], const f32x16 *drop = nullptr) {
  f32x16 xn[2] = {xs[0], xs[1]};
  if (norm) {
    float ss = 0.f;
#pragma unroll
    for (int tau = 0; tau < 2; ++tau)
#pragma unroll
      for (int r = 0; r < 16; ++r) ss += xs[tau][r] * xs[tau][r];
    ss += __shfl_xor(ss, 32);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    xn[0] = xs[0] * inv;
    xn[1] = xs[1] * inv;
  }
  if (drop != nullptr) {
    xn[0] *= drop[0];
    xn[1] *= drop[1];
  }
  f32x16 hid[2];
  load_tile(tb_row, h, true, hid);  // accumulator starts at b_in + W_in[:,64:72].emb(t)
  contract64(wl, lane, xn, hid);
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[tau][r] = SHORT_TANH ? tanh_short(hid[tau][r]) : tanhf(hid[tau][r]);
  load_tile(b_out, h, true, out);
  contract64(wl + 2 * 2 * 4 * 64, lane, hid, out);
}

template <int MODE>
__global__ __launch_bounds__(kThreads, 2) void denoise_kernel(const dn_batch batch) {
  // [matrix][to][ti][rq][lane] float4: lane reads W[32*to + (lane&31)][32*ti + 8*rq + 4*(lane>>5) .. +4)
  __shared__ f32x4 wl[2 * 2 * 2 * 4 * 64];
  const int b = blockIdx.x;
  int ti_ = 0;
  for (int k = 1; k < batch.n_tasks; ++k)
    if (b >= batch.t[k].blk_begin) ti_ = k;
  const dn_task &T = batch.t[ti_];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  bool valid, none;
  const int64_t first = (int64_t)(b - T.blk_begin) * kRowsPerBlock;
  const int64_t row = slot_row(T, first + wave * kRowsPerWave + j, first, valid, none);
  if (none) return;   // (the whole workgroup: a row list shorter than its capacity)
  const bool norm = T.norm != 0;

  // the row tile is asked for before the weight images are staged: its way from HBM overlaps the staging
  f32x16 xs[2], out[2];
  load_tile(T.x + row * DG_D, h, valid, xs);

  for (int idx = threadIdx.x; idx < 2 * 2 * 2 * 4 * 64; idx += blockDim.x) {
    const int ln = idx & 63, rq = (idx >> 6) & 3, ti = (idx >> 8) & 1, to = (idx >> 9) & 1, m = idx >> 10;
    const int wrow = 32 * to + (ln & 31), col = 32 * ti + 8 * rq + 4 * (ln >> 5);
    const float *src = m == 0 ? T.w_in + wrow * 72 + col : T.w_out + wrow * 64 + col;
    wl[idx] = *reinterpret_cast<const f32x4 *>(src);
  }
  __syncthreads();

  if (MODE == MODE_PSAMPLE) {
    for (int step = T.n_steps - 1; step >= 0; --step) {
      // Opaque copy of the lane id: keeps the compiler from hoisting the 128 weight
      // registers out of the loop (that would halve the waves per SIMD).
      int ln = lane;
      asm volatile("" : "+v"(ln));
      denoise_tile(wl, ln, h, norm, T.time_bias + step * DG_D, T.b_out, xs, out);
      const float c1 = T.coef1[step], c2 = T.coef2[step];
      xs[0] = c1 * out[0] + c2 * xs[0];  // model_mean (Model.py:323)
      xs[1] = c1 * out[1] + c2 * xs[1];
    }
    store_tile(T.out + row * DG_D, h, valid, xs);
    return;
  }

  int t = 0;
  if (valid) t = min(max((int)T.t[row], 0), T.n_steps - 1);  // callers validate; never index outside the tables
  if (MODE == MODE_LOSS && T.sqrt_ac != nullptr) {  // q_sample (Model.py:302-305)
    f32x16 nz[2];
    load_tile(T.noise + row * DG_D, h, valid, nz);
    const float a = T.sqrt_ac[t], s = T.sqrt_1mac[t];
    xs[0] = a * xs[0] + s * nz[0];
    xs[1] = a * xs[1] + s * nz[1];
    if (T.x_t_out) store_tile(T.x_t_out + row * DG_D, h, valid, xs);
  }
  if ((MODE == MODE_LOSS || MODE == MODE_FWD) && T.drop_scale != 0.f) {
    f32x16 m[2];
    drop_tile(T, row, h, valid, m);
    denoise_tile<true>(wl, lane, h, norm, T.time_bias + t * DG_D, T.b_out, xs, out, m);
  } else {
    denoise_tile<true>(wl, lane, h, norm, T.time_bias + t * DG_D, T.b_out, xs, out);
  }
  store_tile(T.out + row * DG_D, h, valid, out);
  if (MODE == MODE_LOSS) {  // mean_flat((target-out)^2) * (SNR(t-1)-SNR(t)) (Model.py:360-364)
    f32x16 tg[2];
    load_tile(T.target + row * DG_D, h, valid, tg);
    float ss = 0.f;
#pragma unroll
    for (int tau = 0; tau < 2; ++tau)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dlt = tg[tau][r] - out[tau][r];
        ss += dlt * dlt;
      }
    ss += __shfl_xor(ss, 32);
    if (valid && h == 0) T.row_loss[row] = T.loss_w[t] * (double)(ss * (1.0f / DG_D));
  }
}

// ---- the reverse loop on the bf16 matrix instruction, fp32 values cut three ways -------------------------------
// v_mfma_f32_32x32x16_bf16 runs at sixteen times the rate of the exact-fp32 instruction.  An fp32 value is the sum of
// three bf16 numbers EXACTLY (its 24 significand bits in three runs of eight: hi, mid, lo), a product of two bf16
// numbers is exact in fp32, and the accumulator is fp32: with w = wh + wm + wl and x = xh + xm + xl the six products
//     wh xh + wh xm + wm xh + wm xm + wh xl + wl xh
// leave out only terms below 2^-24 of |w x| -- the size of ONE fp32 rounding -- so a contraction carried out this way
// is as close to the exact sum as the fp32 instruction's own result (not the same bits: the roundings fall
// elsewhere), at 6/16 of its matrix cycles.  The loop's 400 contractions per row are where that pays.
// Layout: the accumulator layout of the 32x32 instructions is the same for both types (row of the tile on the lane,
// feature 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of a 32-feature block in register r), and the B operand of the bf16
// instruction takes eight consecutive k per lane half: registers 8p .. 8p+7 of a block ARE such a run -- chunk
// (block tau, pair p) carries the sixteen features 32 tau + 16 p .. + 15, lane half h holding
// 32 tau + 8 (2p + e / 4) + 4 h + e % 4 as its e-th -- so the activations still never leave the lane; the weight
// images in LDS are laid out to match (bf16x8 per lane, plane, chunk and output half).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct bf3 {
  u32x4_t hi, mid, lo;   // eight bf16 each, two per dword (element 2k in the low half)
};

// eight fp32 -> their three bf16 planes (hi + mid + lo == x up to the bits below the 24th)
// (v_cvt_pk_bf16_f32 rounds two values to nearest-even bf16 in one instruction; the remainders x - hi and
// (x - hi) - mid are exact in fp32, and rounding to nearest leaves them at most eight significant bits each)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ bf3 split8(const float (&x)[8]) {
  bf3 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = x[2 * k], b = x[2 * k + 1];
    const unsigned h = pack_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = pack_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    o.hi[k] = h;
    o.mid[k] = m;
    o.lo[k] = pack_bf16(sa, sb);
  }
  return o;
}

constexpr int kB3Image = 2 * 4 * 3 * 64;   // u32x4 per 64x64 matrix: [to][chunk][plane][lane]

// acc[to] += W[32 to .. + 32][:] . act  with W as a three-plane image and act in the accumulator layout
__device__ __forceinline__ void contract64_b3(const u32x4_t *__restrict__ img, int lane, const f32x16 (&act)[2], f32x16 (&acc)[2]) {
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = act[tau][8 * p + e];
      const bf3 b = split8(x);
      const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, b.hi), bm = __builtin_bit_cast(bf16x8_t, b.mid),
                     bl = __builtin_bit_cast(bf16x8_t, b.lo);
      const int chunk = 2 * tau + p;
#pragma unroll
      for (int to = 0; to < 2; ++to) {
        const u32x4_t *a = img + ((to * 4 + chunk) * 3) * 64 + lane;
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, a[0]), am = __builtin_bit_cast(bf16x8_t, a[64]),
                       al = __builtin_bit_cast(bf16x8_t, a[128]);
        // smallest terms first
        acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[to], 0, 0, 0);
      }
    }
}

// Stages the two weight matrices of a denoiser as three-plane images: [matrix][to][chunk][plane][lane].
__device__ __forceinline__ void stage_b3(u32x4_t *__restrict__ img, const float *__restrict__ w_in, const float *__restrict__ w_out) {
  for (int idx = threadIdx.x; idx < 2 * 2 * 4 * 64; idx += blockDim.x) {
    const int ln = idx & 63, chunk = (idx >> 6) & 3, to = (idx >> 8) & 1, m = idx >> 9;
    const int tau = chunk >> 1, p = chunk & 1, i = ln & 31, kh = ln >> 5;
    const float *row = m == 0 ? w_in + (32 * to + i) * 72 : w_out + (32 * to + i) * 64;
    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(row + 32 * tau + 8 * (2 * p) + 4 * kh);
    const f32x4 c1 = *reinterpret_cast<const f32x4 *>(row + 32 * tau + 8 * (2 * p + 1) + 4 * kh);
    const float x[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const bf3 s3 = split8(x);
    u32x4_t *dst = img + m * kB3Image + ((to * 4 + chunk) * 3) * 64 + ln;
    dst[0] = s3.hi; dst[64] = s3.mid; dst[128] = s3.lo;
  }
}

__global__ __launch_bounds__(kThreads, 2) void psample_b3_kernel(const dn_batch batch) {
  __shared__ u32x4_t img[2 * kB3Image];   // 48 KiB
  const int b = blockIdx.x;
  int ti_ = 0;
  for (int k = 1; k < batch.n_tasks; ++k)
    if (b >= batch.t[k].blk_begin) ti_ = k;
  const dn_task &T = batch.t[ti_];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int64_t row = ((int64_t)(b - T.blk_begin) * kWavesPerBlock + wave) * kRowsPerWave + j;
  const bool valid = row < T.n;
  const bool norm = T.norm != 0;
  f32x16 xs[2];
  load_tile(T.x + row * DG_D, h, valid, xs);
  stage_b3(img, T.w_in, T.w_out);
  __syncthreads();
  for (int step = T.n_steps - 1; step >= 0; --step) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // (keeps the weight operands out of loop-invariant registers)
    f32x16 xn[2] = {xs[0], xs[1]};
    if (norm) {
      float ss = 0.f;
#pragma unroll
      for (int tau = 0; tau < 2; ++tau)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss += xs[tau][r] * xs[tau][r];
      ss += __shfl_xor(ss, 32);
      const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
      xn[0] = xs[0] * inv;
      xn[1] = xs[1] * inv;
    }
    f32x16 hid[2], out[2];
    load_tile(T.time_bias + step * DG_D, h, true, hid);
    contract64_b3(img, ln, xn, hid);
#pragma unroll
    for (int tau = 0; tau < 2; ++tau)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[tau][r] = tanh_short(hid[tau][r]);
    load_tile(T.b_out, h, true, out);
    contract64_b3(img + kB3Image, ln, hid, out);
    const float c1 = T.coef1[step], c2 = T.coef2[step];
    xs[0] = c1 * out[0] + c2 * xs[0];  // model_mean (Model.py:323)
    xs[1] = c1 * out[1] + c2 * xs[1];
  }
  store_tile(T.out + row * DG_D, h, valid, xs);
}

// Backward of training_losses2 for a 32-row tile per wavefront: recomputes the forward
// (q_sample, normalise, two contractions, tanh), then runs the two transposed contractions.
// The weight gradients are formed by dg_denoiser_wgrad from the dout_tot / hid / dpre / xn tables stored here.
__global__ __launch_bounds__(kThreads, 2) void loss_bwd_kernel(const dn_batch batch) {
  // images 0,1: W_in[:, :64], W_out (as in denoise_kernel); 2: W_out^T; 3: W_in[:, :64]^T; afterwards the staging
  // area of the weight-gradient partials (two padded 32-row tiles per wavefront)
  __shared__ f32x4 wl[kWgLdsF4];
  const int b = blockIdx.x;
  int ti_ = 0;
  for (int k = 1; k < batch.n_tasks; ++k)
    if (b >= batch.t[k].blk_begin) ti_ = k;
  const dn_task &T = batch.t[ti_];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  bool valid, none;
  const int64_t first = (int64_t)(b - T.blk_begin) * kRowsPerBlock;
  const int64_t row = slot_row(T, first + wave * kRowsPerWave + j, first, valid, none);
  if (none) {   // (the whole workgroup) no rows: its partial result of the weight gradients is zero
    if (T.wg_partials != nullptr) {
      float *P = T.wg_partials + (size_t)(b - T.blk_begin) * dg_wg::kPartFloats;
      for (int i = threadIdx.x; i < dg_wg::kPartFloats; i += kThreads) P[i] = 0.f;
    }
    return;
  }
  const int64_t ro = row * DG_D;
  int t = 0;
  if (valid) t = min(max((int)T.t[row], 0), T.n_steps - 1);
  // the row tiles are asked for before the weight images are staged: their way from HBM overlaps the staging
  f32x16 xs[2], nz[2];
  load_tile(T.x + ro, h, valid, xs);
  if (T.sqrt_ac != nullptr) load_tile(T.noise + ro, h, valid, nz);
  for (int idx = threadIdx.x; idx < 4 * 2 * 2 * 4 * 64; idx += blockDim.x) {
    const int ln = idx & 63, rq = (idx >> 6) & 3, ti = (idx >> 8) & 1, to = (idx >> 9) & 1, m = idx >> 10;
    const int row = 32 * to + (ln & 31), col = 32 * ti + 8 * rq + 4 * (ln >> 5);
    f32x4 v;
    if (m == 0) v = *reinterpret_cast<const f32x4 *>(T.w_in + row * 72 + col);
    else if (m == 1) v = *reinterpret_cast<const f32x4 *>(T.w_out + row * 64 + col);
    else {
      const float *w = m == 2 ? T.w_out : T.w_in;
      const int ld = m == 2 ? 64 : 72;
      v = f32x4{w[(col + 0) * ld + row], w[(col + 1) * ld + row], w[(col + 2) * ld + row], w[(col + 3) * ld + row]};
    }
    wl[idx] = v;
  }
  __syncthreads();
  float a_t = 1.0f;
  if (T.sqrt_ac != nullptr) {
    a_t = T.sqrt_ac[t];
    const float s_t = T.sqrt_1mac[t];
    xs[0] = a_t * xs[0] + s_t * nz[0];
    xs[1] = a_t * xs[1] + s_t * nz[1];
  }
  // forward recompute
  float nrm = 1.0f;
  f32x16 xn[2] = {xs[0], xs[1]};
  if (T.norm) {
    float ss = 0.f;
#pragma unroll
    for (int tau = 0; tau < 2; ++tau)
#pragma unroll
      for (int r = 0; r < 16; ++r) ss += xs[tau][r] * xs[tau][r];
    ss += __shfl_xor(ss, 32);
    nrm = fmaxf(sqrtf(ss), 1e-12f);
    const float inv = 1.0f / nrm;
    xn[0] = xs[0] * inv;
    xn[1] = xs[1] * inv;
  }
  const bool drop = T.drop_scale != 0.f;
  f32x16 xd[2] = {xn[0], xn[1]};   // what the first Linear sees: the normalised row after dropout
  f32x16 dm[2];
  if (drop) {
    drop_tile(T, row, h, valid, dm);
    xd[0] *= dm[0];
    xd[1] *= dm[1];
  }
  const bool fused = T.wg_partials != nullptr;   // the weight gradients' partial sums are formed here, from registers
  if (!fused) store_tile(T.xn + ro, h, valid, xd);
  f32x16 hid[2], out[2];
  load_tile(T.time_bias + t * DG_D, h, true, hid);
  contract64(wl, lane, xd, hid);
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[tau][r] = tanh_short(hid[tau][r]);   // (the forward kernel's form: the same bits)
  if (!fused) store_tile(T.hid + ro, h, valid, hid);
  load_tile(T.b_out, h, true, out);
  contract64(wl + 1 * 1024, lane, hid, out);
  // d out = g_out + g_row * w_t * 2 (out - target) / 64 ;  d target = -(second term)
  {
    f32x16 tg[2];
    load_tile(T.target + ro, h, valid, tg);
    float coef = 0.f;
    if (valid) coef = (float)(T.g_row[row] * T.loss_w[t] * (2.0 / DG_D));
    f32x16 dl[2] = {coef * (out[0] - tg[0]), coef * (out[1] - tg[1])};
    f32x16 ng[2] = {-dl[0], -dl[1]};
    if (T.d_target_add != nullptr) {
      f32x16 base[2];
      load_tile(T.d_target_add + ro, h, valid, base);
      ng[0] += base[0];
      ng[1] += base[1];
    }
    store_tile(T.d_target + ro, h, valid, ng);
    out[0] = dl[0];
    out[1] = dl[1];
    if (T.g_out != nullptr) {
      load_tile(T.g_out + ro, h, valid, tg);
      out[0] += tg[0];
      out[1] += tg[1];
    }
  }
  if (!fused) store_tile(T.dout_tot + ro, h, valid, out);
  // dh = W_out^T dout ; dpre = (1 - h^2) dh
  f32x16 dh[2];
  dh[0] = dh[1] = f32x16{0.f};
  contract64(wl + 2 * 1024, lane, out, dh);
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[tau][r] = (1.0f - hid[tau][r] * hid[tau][r]) * dh[tau][r];
  if (!fused) store_tile(T.dpre + ro, h, valid, dh);
  // dxn = W_in[:, :64]^T dpre ; through F.normalize and q_sample
  f32x16 dx[2];
  dx[0] = dx[1] = f32x16{0.f};
  contract64(wl + 3 * 1024, lane, dh, dx);
  if (drop) {
    dx[0] *= dm[0];
    dx[1] *= dm[1];
  }
  if (T.norm) {
    float ng = 0.f;
#pragma unroll
    for (int tau = 0; tau < 2; ++tau)
#pragma unroll
      for (int r = 0; r < 16; ++r) ng += xn[tau][r] * dx[tau][r];
    ng += __shfl_xor(ng, 32);
    const float inv = 1.0f / nrm;
    if (nrm > 1e-12f) {
      dx[0] = (dx[0] - xn[0] * ng) * inv;
      dx[1] = (dx[1] - xn[1] * ng) * inv;
    } else {
      dx[0] = dx[0] * inv;
      dx[1] = dx[1] * inv;
    }
  }
  dx[0] = dx[0] * a_t;
  dx[1] = dx[1] * a_t;
  store_tile(T.d_x_start + ro, h, valid, dx);
  if (!fused) return;   // (the whole workgroup)

  // ---- weight-gradient partial of this workgroup's 128 rows:  dout^T.hid, dpre^T.xd, dpre^T.tab[ts], column sums
  // The contraction runs over table rows, so the tiles (row on the lane) go through LDS once to come back with
  // four rows per wavefront load, 16 lanes per row (the operand layout of dg_wg::mma64).  Each wavefront multiplies
  // its own 32 rows; the four results are added in wavefront order, wavefront w finishing output rows 4r + w.
  __syncthreads();   // every wavefront is done with the weight images
  float *stage_a = reinterpret_cast<float *>(wl) + wave * (2 * kWgTileFloats), *stage_b = stage_a + kWgTileFloats;
  f32x4 *red = wl;   // [wave][mb * 4 + nb][lane], the four accumulator sets side by side (64 KiB)
  float *P = T.wg_partials + (size_t)(b - T.blk_begin) * dg_wg::kPartFloats;
  const int jj = lane & 15, kk = lane >> 4;
  f32x4 acc[4][4], acc_t[4], cs_out, cs_in;
  for (int prod = 0; prod < 2; ++prod) {
    if (prod == 1) __syncthreads();   // the first product's sums have been read
    stage_tile(stage_a, j, h, valid, prod == 0 ? out : dh);
    stage_tile(stage_b, j, h, true, prod == 0 ? hid : xd);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      acc_t[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kRowsPerWave / 4; ++q) {
      const f32x4 av = *reinterpret_cast<const f32x4 *>(stage_a + (4 * q + kk) * kWgRowFloats + 4 * jj);
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(stage_b + (4 * q + kk) * kWgRowFloats + 4 * jj);
      dg_wg::mma64(av, bv, acc);
      cs += av;
      if (prod == 1) {   // the time-embedding columns: operand row tab[ts] of table row 4q + kk
        const int tr = __shfl(t, 4 * q + kk);
        const bool vr = __shfl((int)valid, 4 * q + kk) != 0;
        const float tv = vr ? T.wg_tab[tr * dg_wg::kTabCols + jj] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc_t[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], tv, acc_t[mb], 0, 0, 0);
      }
    }
    for (int c = 0; c < 4; ++c) {   // column sums: add the four k-lanes that share jj (fixed order)
      float v = cs[c];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      cs[c] = v;
    }
    if (prod == 0) cs_out = cs; else cs_in = cs;
    __syncthreads();   // all staged tiles have been read: the area becomes the exchange buffer
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) red[(wave * 16 + mb * 4 + nb) * 64 + lane] = acc[mb][nb];
    __syncthreads();
    f32x4 sum[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      sum[nb] = red[(0 * 16 + wave * 4 + nb) * 64 + lane];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) sum[nb] += red[(w * 16 + wave * 4 + nb) * 64 + lane];
    }
    dg_wg::store64_mb(P + (prod == 0 ? dg_wg::kPartWout : dg_wg::kPartWin), lane, wave, sum);
  }
  __syncthreads();
  // the small pieces: [wave][0..3] acc_t, [wave][4] column sums of dout, [wave][5] of dpre
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) red[(wave * 6 + mb) * 64 + lane] = acc_t[mb];
  red[(wave * 6 + 4) * 64 + lane] = cs_out;
  red[(wave * 6 + 5) * 64 + lane] = cs_in;
  __syncthreads();
  {
    // wavefront w: (dpre^T . tab) rows 4r + w -- acc_t[mb][r] of lane l is o = 16 (l>>4) + 4 r + mb, c = l & 15
    f32x4 s = red[(0 * 6 + wave) * 64 + lane];
    for (int w = 1; w < kWavesPerBlock; ++w) s += red[(w * 6 + wave) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) P[dg_wg::kPartTab + (16 * kk + 4 * r + wave) * dg_wg::kTabCols + jj] = s[r];
    if (wave < 2 && kk == 0) {   // wavefront 0: db_out, wavefront 1: db_in
      f32x4 c = red[(0 * 6 + 4 + wave) * 64 + lane];
      for (int w = 1; w < kWavesPerBlock; ++w) c += red[(w * 6 + 4 + wave) * 64 + lane];
      *reinterpret_cast<f32x4 *>(P + (wave == 0 ? dg_wg::kPartBout : dg_wg::kPartBin) + 4 * jj) = c;
    }
  }
}

__global__ __launch_bounds__(256) void q_sample_kernel(const f32x4 *__restrict__ x0, const f32x4 *__restrict__ noise,
                                                       const int64_t *__restrict__ t, const float *__restrict__ sa,
                                                       const float *__restrict__ sb, f32x4 *__restrict__ xt, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int ts = (int)t[i >> 4];
    xt[i] = sa[ts] * x0[i] + sb[ts] * noise[i];
  }
}

struct tb_args {
  const float *w_emb, *b_emb, *w_in, *b_in;
  float freqs[4];
};

// table[t][f] = b_in[f] + sum_e W_in[f][64+e] * (b_emb[e] + sum_k W_emb[e][k] * te_t[k]),
// te_t = [cos(t*freqs), sin(t*freqs)]  (Model.py:215-220 and the last 8 input columns of :226-227).
constexpr int kMaxBiasTables = 4;
struct tb_batch {
  tb_args a[kMaxBiasTables];
  float *table[kMaxBiasTables];
};

__global__ __launch_bounds__(64) void time_bias_kernel(const tb_batch all) {
  const tb_args &a = all.a[blockIdx.y];
  float *__restrict__ table = all.table[blockIdx.y];
  const int t = blockIdx.x, f = threadIdx.x;
  float te[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float arg = (float)t * a.freqs[k];
    te[k] = cosf(arg);
    te[k + 4] = sinf(arg);
  }
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float emb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) emb += te[k] * a.w_emb[e * 8 + k];
    emb += a.b_emb[e];
    acc += emb * a.w_in[f * 72 + 64 + e];
  }
  table[t * DG_D + f] = acc + a.b_in[f];
}

int check_denoiser(const char *who, const dg_denoiser *dn, const float *time_bias, int32_t n_steps, int32_t d) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "%s: embedding width %d unsupported (built for %d)", who, d, DG_D);
  if (!dn || !dn->w_in || !dn->w_out || !dn->b_out || !time_bias) return dg_fail(DG_ERR_ARG, "%s: NULL denoiser argument", who);
  if (n_steps <= 0) return dg_fail(DG_ERR_ARG, "%s: n_steps must be positive", who);
  if (((uintptr_t)dn->w_in | (uintptr_t)dn->w_out | (uintptr_t)dn->b_out | (uintptr_t)time_bias) & 15)
    return dg_fail(DG_ERR_ARG, "%s: weight / table pointers must be 16-byte aligned", who);
  return 0;
}

// DG_PSAMPLE_FP32=1: the reverse loop on the exact-fp32 matrix instruction (the form before the bf16 three-way split; kept
// for A/B runs and as the reference the split form is tested against)
bool psample_fp32() {
  const char *e = getenv("DG_PSAMPLE_FP32");   // (read at every launch: a test compares the two forms in one process)
  return e && e[0] == '1';
}

template <int MODE>
int launch_one(dn_task &t, hipStream_t stream) {
  if (t.n <= 0) return 0;
  dn_batch batch;
  t.blk_begin = 0;
  for (int k = 0; k < kDnMaxTasks; ++k) batch.t[k] = t;
  batch.n_tasks = 1;
  const int64_t blocks = (t.n + kRowsPerBlock - 1) / kRowsPerBlock;
  if (MODE == MODE_PSAMPLE && !psample_fp32()) hipLaunchKernelGGL(psample_b3_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, batch);
  else hipLaunchKernelGGL(denoise_kernel<MODE>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, batch);
  DG_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" int dg_time_bias_tables(const dg_denoiser *const *dns, float *const *tables, int32_t n_tables, int32_t n_steps,
                                   void *stream) {
  if (!dns || !tables || n_tables <= 0 || n_tables > kMaxBiasTables)
    return dg_fail(DG_ERR_ARG, "dg_time_bias_tables: n_tables must be in [1,%d] (got %d)", kMaxBiasTables, n_tables);
  if (n_steps <= 0) return dg_fail(DG_ERR_ARG, "dg_time_bias_table: n_steps must be positive");
  tb_batch b;
  for (int i = 0; i < n_tables; ++i) {
    const dg_denoiser *dn = dns[i];
    if (!dn || !dn->w_emb || !dn->b_emb || !dn->w_in || !dn->b_in || !dn->freqs_host || !tables[i])
      return dg_fail(DG_ERR_ARG, "dg_time_bias_table: NULL argument");
    tb_args &a = b.a[i];
    a.w_emb = dn->w_emb; a.b_emb = dn->b_emb; a.w_in = dn->w_in; a.b_in = dn->b_in;
    for (int k = 0; k < 4; ++k) a.freqs[k] = dn->freqs_host[k];
    b.table[i] = tables[i];
  }
  for (int i = n_tables; i < kMaxBiasTables; ++i) { b.a[i] = b.a[0]; b.table[i] = b.table[0]; }
  hipLaunchKernelGGL(time_bias_kernel, dim3(n_steps, n_tables), dim3(64), 0, (hipStream_t)stream, b);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_time_bias_table(const dg_denoiser *dn, int32_t n_steps, float *table, void *stream) {
  return dg_time_bias_tables(&dn, &table, 1, n_steps, stream);
}

extern "C" int dg_q_sample(const float *x0, const float *noise, const int64_t *t, const float *sqrt_ac,
                           const float *sqrt_1mac, int32_t n_steps, int64_t n, int32_t d, float *x_t, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_q_sample: embedding width %d unsupported (built for %d)", d, DG_D);
  if (n == 0) return 0;
  if (!x0 || !noise || !t || !sqrt_ac || !sqrt_1mac || !x_t) return dg_fail(DG_ERR_ARG, "dg_q_sample: NULL argument");
  if (n_steps <= 0 || n < 0) return dg_fail(DG_ERR_ARG, "dg_q_sample: bad sizes");
  const int64_t n4 = n * (DG_D / 4);
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4 *>(x0), reinterpret_cast<const f32x4 *>(noise), t, sqrt_ac, sqrt_1mac,
                     reinterpret_cast<f32x4 *>(x_t), n4);
  DG_HIP(hipGetLastError());
  return 0;
}

// Training-mode dropout of a task (dn_task::keep_mask ...): off for drop_p == 0.
static int set_dropout(const char *who, const uint8_t *keep_mask, const int64_t *drop_seed, float drop_p, dn_task &k) {
  if (drop_p == 0.f) return 0;
  if (!(drop_p > 0.f && drop_p < 1.f)) return dg_fail(DG_ERR_ARG, "%s: drop_p must lie in [0,1) (got %g)", who, (double)drop_p);
  if (!keep_mask && !drop_seed) return dg_fail(DG_ERR_ARG, "%s: dropout needs keep_mask or drop_seed", who);
  if (keep_mask && ((uintptr_t)keep_mask & 3)) return dg_fail(DG_ERR_ARG, "%s: keep_mask must be 4-byte aligned", who);
  k.keep_mask = keep_mask; k.drop_seed = drop_seed;
  k.drop_scale = 1.0f / (float)(1.0 - (double)drop_p);
  k.drop_thr = (uint32_t)lrint((double)drop_p * 256.0);
  if (!keep_mask) {
    // a mask drawn here has 8 random bits per element: the rate actually applied is drop_thr / 256, and the kept
    // elements are scaled by ITS reciprocal complement, so that the expectation stays the input's (p = 0.5: exact)
    if (k.drop_thr == 0 || k.drop_thr >= 256)
      return dg_fail(DG_ERR_ARG, "%s: a drawn dropout mask takes rates of k/256, 0 < k < 256 (got %g); hand in a keep_mask", who,
                     (double)drop_p);
    k.drop_scale = 256.0f / (float)(256 - (int)k.drop_thr);
  }
  return 0;
}

extern "C" int dg_denoise_fwd(const dg_denoiser *dn, const float *time_bias, int32_t n_steps, const float *x,
                              const int64_t *t, int64_t n, int32_t d, float *out, void *stream) {
  int rc = check_denoiser("dg_denoise_fwd", dn, time_bias, n_steps, d);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!x || !t || !out || n < 0) return dg_fail(DG_ERR_ARG, "dg_denoise_fwd: NULL argument");
  dn_task k = {};
  k.w_in = dn->w_in; k.w_out = dn->w_out; k.b_out = dn->b_out; k.time_bias = time_bias;
  k.x = x; k.t = t; k.out = out; k.n = n; k.n_steps = n_steps; k.norm = dn->norm;
  return launch_one<MODE_FWD>(k, (hipStream_t)stream);
}

extern "C" int dg_denoise_fwd_train(const dg_denoiser *dn, const float *time_bias, int32_t n_steps, const float *x,
                                    const int64_t *t, int64_t n, int32_t d, const uint8_t *keep_mask,
                                    const int64_t *drop_seed, float drop_p, float *out, void *stream) {
  int rc = check_denoiser("dg_denoise_fwd_train", dn, time_bias, n_steps, d);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!x || !t || !out || n < 0) return dg_fail(DG_ERR_ARG, "dg_denoise_fwd_train: NULL argument");
  dn_task k = {};
  k.w_in = dn->w_in; k.w_out = dn->w_out; k.b_out = dn->b_out; k.time_bias = time_bias;
  k.x = x; k.t = t; k.out = out; k.n = n; k.n_steps = n_steps; k.norm = dn->norm;
  if ((rc = set_dropout("dg_denoise_fwd_train", keep_mask, drop_seed, drop_p, k))) return rc;
  return launch_one<MODE_FWD>(k, (hipStream_t)stream);
}

extern "C" int dg_p_sample_loop(const dg_denoiser *dn, const float *time_bias, const float *coef1, const float *coef2,
                                int32_t n_steps, const float *x_T, int64_t n, int32_t d, float *x_0, void *stream) {
  int rc = check_denoiser("dg_p_sample_loop", dn, time_bias, n_steps, d);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!coef1 || !coef2 || !x_T || !x_0 || n < 0) return dg_fail(DG_ERR_ARG, "dg_p_sample_loop: NULL argument");
  dn_task k = {};
  k.w_in = dn->w_in; k.w_out = dn->w_out; k.b_out = dn->b_out; k.time_bias = time_bias;
  k.x = x_T; k.coef1 = coef1; k.coef2 = coef2; k.out = x_0; k.n = n; k.n_steps = n_steps; k.norm = dn->norm;
  return launch_one<MODE_PSAMPLE>(k, (hipStream_t)stream);
}

extern "C" int dg_p_sample_loop_batched(const dg_psample_task *tasks, int32_t n_tasks, const float *coef1,
                                        const float *coef2, int32_t n_steps, int32_t d, void *stream) {
  if (!tasks || n_tasks <= 0 || n_tasks > kDnMaxTasks)
    return dg_fail(DG_ERR_ARG, "dg_p_sample_loop_batched: n_tasks must be in [1,%d]", kDnMaxTasks);
  if (!coef1 || !coef2) return dg_fail(DG_ERR_ARG, "dg_p_sample_loop_batched: NULL coefficient table");
  dn_batch batch;
  batch.n_tasks = 0;
  int64_t blocks = 0;
  for (int k = 0; k < n_tasks; ++k) {
    const dg_psample_task &t = tasks[k];
    int rc = check_denoiser("dg_p_sample_loop_batched", t.dn, t.time_bias, n_steps, d);
    if (rc) return rc;
    if (t.n < 0 || (t.n > 0 && (!t.x_T || !t.x_0))) return dg_fail(DG_ERR_ARG, "dg_p_sample_loop_batched: task %d: bad table", k);
    if (t.n == 0) continue;
    dn_task q = {};
    q.w_in = t.dn->w_in; q.w_out = t.dn->w_out; q.b_out = t.dn->b_out; q.time_bias = t.time_bias;
    q.x = t.x_T; q.coef1 = coef1; q.coef2 = coef2; q.out = t.x_0; q.n = t.n; q.n_steps = n_steps; q.norm = t.dn->norm;
    q.blk_begin = (int32_t)blocks;
    blocks += (t.n + kRowsPerBlock - 1) / kRowsPerBlock;
    batch.t[batch.n_tasks++] = q;
  }
  if (batch.n_tasks == 0) return 0;
  for (int k = batch.n_tasks; k < kDnMaxTasks; ++k) batch.t[k] = batch.t[0];
  if (psample_fp32()) hipLaunchKernelGGL(denoise_kernel<MODE_PSAMPLE>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, batch);
  else hipLaunchKernelGGL(psample_b3_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, batch);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_diffusion_loss_fwd(const dg_denoiser *dn, const float *time_bias, int32_t n_steps, const float *target,
                                     const float *x_start, const float *noise, const int64_t *ts, const float *sqrt_ac,
                                     const float *sqrt_1mac, const double *loss_w, int64_t n, int32_t d, float *out,
                                     double *row_loss, float *x_t_out, void *stream) {
  int rc = check_denoiser("dg_diffusion_loss_fwd", dn, time_bias, n_steps, d);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!target || !x_start || !ts || !loss_w || !out || !row_loss || n < 0)
    return dg_fail(DG_ERR_ARG, "dg_diffusion_loss_fwd: NULL argument");
  if ((sqrt_ac != nullptr) != (sqrt_1mac != nullptr) || (sqrt_ac && !noise))
    return dg_fail(DG_ERR_ARG, "dg_diffusion_loss_fwd: sqrt_ac, sqrt_1mac and noise go together");
  dn_task k = {};
  k.w_in = dn->w_in; k.w_out = dn->w_out; k.b_out = dn->b_out; k.time_bias = time_bias;
  k.x = x_start; k.noise = noise; k.target = target; k.t = ts; k.sqrt_ac = sqrt_ac; k.sqrt_1mac = sqrt_1mac;
  k.loss_w = loss_w; k.out = out; k.row_loss = row_loss; k.x_t_out = x_t_out; k.n = n; k.n_steps = n_steps;
  k.norm = dn->norm;
  return launch_one<MODE_LOSS>(k, (hipStream_t)stream);
}

extern "C" int dg_diffusion_loss_bwd(const dg_denoiser *dn, const float *time_bias, int32_t n_steps, const float *target,
                                     const float *x_start, const float *noise, const int64_t *ts, const float *sqrt_ac,
                                     const float *sqrt_1mac, const double *loss_w, const double *g_row_loss,
                                     const float *g_out, int64_t n, int32_t d, float *d_x_start, float *d_target,
                                     float *dout_tot, float *hid, float *dpre, float *xn, void *stream) {
  int rc = check_denoiser("dg_diffusion_loss_bwd", dn, time_bias, n_steps, d);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!target || !x_start || !ts || !loss_w || !g_row_loss || !d_x_start || !d_target || !dout_tot || !hid || !dpre ||
      !xn || n < 0)
    return dg_fail(DG_ERR_ARG, "dg_diffusion_loss_bwd: NULL argument");
  if ((sqrt_ac != nullptr) != (sqrt_1mac != nullptr) || (sqrt_ac && !noise))
    return dg_fail(DG_ERR_ARG, "dg_diffusion_loss_bwd: sqrt_ac, sqrt_1mac and noise go together");
  dn_task k = {};
  k.w_in = dn->w_in; k.w_out = dn->w_out; k.b_out = dn->b_out; k.time_bias = time_bias;
  k.x = x_start; k.noise = noise; k.target = target; k.t = ts; k.sqrt_ac = sqrt_ac; k.sqrt_1mac = sqrt_1mac;
  k.loss_w = loss_w; k.g_row = g_row_loss; k.g_out = g_out; k.d_x_start = d_x_start; k.d_target = d_target;
  k.dout_tot = dout_tot; k.hid = hid; k.dpre = dpre; k.xn = xn; k.n = n; k.n_steps = n_steps; k.norm = dn->norm;
  dn_batch batch;
  k.blk_begin = 0;
  for (int i = 0; i < kDnMaxTasks; ++i) batch.t[i] = k;
  batch.n_tasks = 1;
  const int64_t blocks = (n + kRowsPerBlock - 1) / kRowsPerBlock;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, batch);
  DG_HIP(hipGetLastError());
  return 0;
}

// -- both denoisers of the recommendation model (users, items) in one launch each way ----------------
namespace {
int fill_loss_task(const char *who, const dg_loss_task &t, int32_t n_steps, int32_t d, const float *sqrt_ac,
                   const float *sqrt_1mac, const double *loss_w, bool bwd, dn_task &k) {
  int rc = check_denoiser(who, t.dn, t.time_bias, n_steps, d);
  if (rc) return rc;
  if (t.n < 0) return dg_fail(DG_ERR_ARG, "%s: negative row count", who);
  if (t.n == 0) return 0;
  if (!t.target || !t.x_start || !t.ts || (sqrt_ac && !t.noise)) return dg_fail(DG_ERR_ARG, "%s: NULL table", who);
  k = dn_task{};
  k.w_in = t.dn->w_in; k.w_out = t.dn->w_out; k.b_out = t.dn->b_out; k.time_bias = t.time_bias;
  k.x = t.x_start; k.noise = t.noise; k.target = t.target; k.t = t.ts; k.sqrt_ac = sqrt_ac; k.sqrt_1mac = sqrt_1mac;
  k.loss_w = loss_w; k.n = t.n; k.n_steps = n_steps; k.norm = t.dn->norm;
  if (t.rows != nullptr) {
    if (!t.n_rows || t.rows_cap <= 0 || t.rows_cap > t.n)
      return dg_fail(DG_ERR_ARG, "%s: a row list needs its DEVICE count and a capacity in [1, n] (got %d, n = %lld)", who,
                     t.rows_cap, (long long)t.n);
    k.rows = t.rows; k.n_rows = t.n_rows; k.rows_cap = t.rows_cap;
  }
  if ((rc = set_dropout(who, t.keep_mask, t.drop_seed, t.drop_p, k))) return rc;
  if (!bwd) {
    if (!t.out || !t.row_loss) return dg_fail(DG_ERR_ARG, "%s: NULL output", who);
    k.out = t.out; k.row_loss = t.row_loss;
  } else {
    if (!t.g_row_loss || !t.d_x_start || !t.d_target) return dg_fail(DG_ERR_ARG, "%s: NULL backward argument", who);
    if (t.wg_partials) {
      if (!t.wg_tab || ((uintptr_t)t.wg_partials & 15)) return dg_fail(DG_ERR_ARG, "%s: wg_partials needs wg_tab and 16-byte alignment", who);
    } else if (!t.dout_tot || !t.hid || !t.dpre || !t.xn) {
      return dg_fail(DG_ERR_ARG, "%s: NULL operand table (dout_tot / hid / dpre / xn) and no wg_partials", who);
    }
    k.g_row = t.g_row_loss; k.g_out = t.g_out; k.d_target_add = t.d_target_add; k.d_x_start = t.d_x_start;
    k.d_target = t.d_target; k.dout_tot = t.dout_tot; k.hid = t.hid; k.dpre = t.dpre; k.xn = t.xn;
    k.wg_tab = t.wg_tab; k.wg_partials = t.wg_partials;
  }
  return 0;
}

int loss_batched(const char *who, const dg_loss_task *tasks, int32_t n_tasks, int32_t n_steps, const float *sqrt_ac,
                 const float *sqrt_1mac, const double *loss_w, int32_t d, bool bwd, hipStream_t stream) {
  if (!tasks || n_tasks <= 0 || n_tasks > kDnMaxTasks) return dg_fail(DG_ERR_ARG, "%s: n_tasks must be in [1,%d]", who, kDnMaxTasks);
  if (!loss_w || (sqrt_ac != nullptr) != (sqrt_1mac != nullptr)) return dg_fail(DG_ERR_ARG, "%s: bad schedule tables", who);
  dn_batch batch;
  batch.n_tasks = 0;
  int64_t blocks = 0;
  for (int i = 0; i < n_tasks; ++i) {
    dn_task k;
    int rc = fill_loss_task(who, tasks[i], n_steps, d, sqrt_ac, sqrt_1mac, loss_w, bwd, k);
    if (rc) return rc;
    if (tasks[i].n == 0) continue;
    k.blk_begin = (int32_t)blocks;
    k.drop_stream = (uint32_t)i;
    blocks += ((tasks[i].rows ? (int64_t)tasks[i].rows_cap : tasks[i].n) + kRowsPerBlock - 1) / kRowsPerBlock;
    batch.t[batch.n_tasks++] = k;
  }
  if (batch.n_tasks == 0) return 0;
  for (int i = batch.n_tasks; i < kDnMaxTasks; ++i) batch.t[i] = batch.t[0];
  if (bwd) hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, batch);
  else hipLaunchKernelGGL(denoise_kernel<MODE_LOSS>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, batch);
  DG_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int dg_diffusion_loss_fwd_batched(const dg_loss_task *tasks, int32_t n_tasks, int32_t n_steps,
                                             const float *sqrt_ac, const float *sqrt_1mac, const double *loss_w, int32_t d,
                                             void *stream) {
  return loss_batched("dg_diffusion_loss_fwd_batched", tasks, n_tasks, n_steps, sqrt_ac, sqrt_1mac, loss_w, d, false,
                      (hipStream_t)stream);
}

extern "C" int dg_diffusion_loss_bwd_batched(const dg_loss_task *tasks, int32_t n_tasks, int32_t n_steps,
                                             const float *sqrt_ac, const float *sqrt_1mac, const double *loss_w, int32_t d,
                                             void *stream) {
  return loss_batched("dg_diffusion_loss_bwd_batched", tasks, n_tasks, n_steps, sqrt_ac, sqrt_1mac, loss_w, d, true,
                      (hipStream_t)stream);
}

namespace {
__global__ __launch_bounds__(256) void keep_mask_kernel(const int64_t *__restrict__ seed, uint32_t stream, uint32_t thr, int64_t n,
                                                        uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (row, lane half)
  if (i >= 2 * n) return;
  const int64_t row = i >> 1;
  const int h = (int)(i & 1);
  uint32_t w[8];
  drop_bytes((uint64_t)seed[0], stream, row, h, w);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t o = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) o |= (((w[c] >> (8 * e)) & 0xffu) >= thr ? 1u : 0u) << (8 * e);
    *reinterpret_cast<uint32_t *>(mask + row * DG_D + 32 * (c >> 2) + 8 * (c & 3) + 4 * h) = o;
  }
}
}  // namespace

// -- the random inputs of one training_losses2 call: per-row timesteps and Gaussian noise ----------------
// (t.randint(0, steps, (batch,)) and t.randn_like(x_start), DiffGraph-Rec/Model.py:352-353.)  The draws come from
// Philox-4x32-10 keyed by state[0] (the seed); the counter holds the row, the float4 column and state[1], the
// number of draws made so far, which draw_advance_kernel increments afterwards -- all on the device, so a recorded
// step draws fresh inputs on every replay, and the same state always gives the same inputs.
namespace {
constexpr uint32_t kDrawNoise = 0x100u, kDrawTs = 0x200u, kDrawDropSeed = 0x300u;

__global__ __launch_bounds__(256) void draw_kernel(const int64_t *__restrict__ state, int64_t n, int32_t n_steps,
                                                   int64_t *__restrict__ ts, f32x4 *__restrict__ noise) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (row, float4 column)
  if (i >= n * (DG_D / 4)) return;
  const uint64_t seed = (uint64_t)state[0], draw = (uint64_t)state[1];
  const uint64_t row = (uint64_t)(i >> 4);
  const uint32_t j = (uint32_t)(i & 15), hi = (uint32_t)(draw >> 32) << 16;
  if (noise != nullptr) {
    // two Box-Muller pairs from four 24-bit uniforms: u1 in (0,1], u2 in [0,1)
    const dg_u32x4 r = dg_philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), kDrawNoise | j | hi, (uint32_t)draw, (uint32_t)seed,
                                        (uint32_t)(seed >> 32));
    const float k = 1.0f / 16777216.0f;
    const float ra = sqrtf(-2.0f * logf(((float)(r.x >> 8) + 1.0f) * k)), rb = sqrtf(-2.0f * logf(((float)(r.z >> 8) + 1.0f) * k));
    float sa, ca, sb, cb;
    sincospif(2.0f * (float)(r.y >> 8) * k, &sa, &ca);
    sincospif(2.0f * (float)(r.w >> 8) * k, &sb, &cb);
    noise[i] = f32x4{ra * ca, ra * sa, rb * cb, rb * sb};
  }
  if (j == 0 && ts != nullptr) {
    const dg_u32x4 r = dg_philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), kDrawTs | hi, (uint32_t)draw, (uint32_t)seed,
                                        (uint32_t)(seed >> 32));
    ts[row] = (int64_t)(((uint64_t)r.x * (uint64_t)n_steps) >> 32);
  }
}

// state[1] += 1, and state[2] = the key of this draw's dropout masks (what dg_loss_task.drop_seed may point at)
__global__ void draw_advance_kernel(int64_t *state) {
  const uint64_t seed = (uint64_t)state[0], draw = (uint64_t)state[1];
  const dg_u32x4 r = dg_philox4x32_10(0u, 0u, kDrawDropSeed | ((uint32_t)(draw >> 32) << 16), (uint32_t)draw, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
  state[2] = (int64_t)((((uint64_t)r.x << 32) | r.y) >> 2);
  state[1] = (int64_t)(draw + 1);
}
}  // namespace

extern "C" int dg_diffusion_draw(int64_t *state, int64_t n, int32_t n_steps, int32_t d, int64_t *ts, float *noise, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_diffusion_draw: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!state || n < 0 || n_steps <= 0) return dg_fail(DG_ERR_ARG, "dg_diffusion_draw: bad argument");
  if (((uintptr_t)noise & 15) || ((uintptr_t)state & 7)) return dg_fail(DG_ERR_ARG, "dg_diffusion_draw: misaligned pointer");
  const int64_t blocks = (n * (DG_D / 4) + 255) / 256;
  if (blocks > 0x7fffffff) return dg_fail(DG_ERR_SIZE, "dg_diffusion_draw: too many rows");
  if (blocks > 0 && (ts || noise))
    hipLaunchKernelGGL(draw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, state, n, n_steps, ts,
                       reinterpret_cast<f32x4 *>(noise));
  hipLaunchKernelGGL(draw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_dropout_keep_mask(const int64_t *drop_seed, int32_t task, float drop_p, int64_t n, int32_t d, uint8_t *mask,
                                    void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_dropout_keep_mask: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!drop_seed || !mask || n < 0 || task < 0) return dg_fail(DG_ERR_ARG, "dg_dropout_keep_mask: bad argument");
  if (!(drop_p > 0.f && drop_p < 1.f)) return dg_fail(DG_ERR_ARG, "dg_dropout_keep_mask: drop_p must lie in (0,1)");
  if ((uintptr_t)mask & 3) return dg_fail(DG_ERR_ARG, "dg_dropout_keep_mask: mask must be 4-byte aligned");
  const uint32_t thr = (uint32_t)lrint((double)drop_p * 256.0);
  if (thr == 0 || thr >= 256) return dg_fail(DG_ERR_ARG, "dg_dropout_keep_mask: a drawn mask takes rates of k/256, 0 < k < 256 (got %g)", (double)drop_p);
  if (n == 0) return 0;
  const int64_t blocks = (2 * n + 255) / 256;
  hipLaunchKernelGGL(keep_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, drop_seed, (uint32_t)task,
                     thr, n, mask);
  DG_HIP(hipGetLastError());
  return 0;
}
