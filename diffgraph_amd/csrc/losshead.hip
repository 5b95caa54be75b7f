// The loss head of the recommendation model for gfx950: BPR + embedding regulariser + the
// batch means of the two diffusion losses, forward and backward, in four small launches.
//
// Replaces the chain of gathers, products, reductions and (in the backward pass) sort-based
// index_put accumulations that autograd records for HGDM.cal_loss (DiffGraph-Rec/Model.py:101-111)
// and pairPredict (DiffGraph-Rec/Utils/Utils.py:4-8): ~100 launches of a few microseconds each.
//
// A batch entry is owned by a 16-lane group (one float4 of each 64-float row per lane).  The
// backward pass first writes the per-entry gradients of the three gathered rows and counts, with
// integer atomics, how many entries name each table row and which comes first; a second launch
// writes the rows named once directly and, for rows named several times, lets the first entry add
// the later ones IN BATCH ORDER (their indices come from a few per-row slots filled beside the
// counts, or, for rows named more often than that, from a scan of the batch): no float atomics, the
// same bits on every run.
#include "dg_internal.h"

#include <math.h>

#include <climits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kSlots = 6;       // entries of one table row remembered per row; rows named more often go to head_heavy_kernel
constexpr int kMaxBatch = 8192;  // the three index lists of a batch sit in LDS as int32: 96 KiB at most

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum16(float v) {  // over the 16 lanes of a group, left in every lane
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Index lists come from the caller (a sampler may hand over -1 for a user with nothing left to draw).
__device__ __forceinline__ bool in_range(int64_t a, int64_t p, int64_t n, int n_user, int n_item) {
  return (uint64_t)a < (uint64_t)n_user && (uint64_t)p < (uint64_t)n_item && (uint64_t)n < (uint64_t)n_item;
}

struct head_rows {
  f32x4 A, P, N;
};

__device__ __forceinline__ head_rows load_rows(const float *ui, const float *out_u, const float *out_i, int64_t a,
                                               int64_t p, int64_t n, int n_user, int l) {
  const f32x4 *U = reinterpret_cast<const f32x4 *>(ui), *OU = reinterpret_cast<const f32x4 *>(out_u),
              *OI = reinterpret_cast<const f32x4 *>(out_i);
  head_rows r;
  r.A = U[a * 16 + l] + OU[a * 16 + l];                  // usrEmbeds + diff_usrEmbeds (Model.py:103)
  r.P = U[(n_user + p) * 16 + l] + OI[p * 16 + l];       // itmEmbeds + diff_itemEmbeds (Model.py:104)
  r.N = U[(n_user + n) * 16 + l] + OI[n * 16 + l];
  return r;
}

// entries_f[b] = {log sigmoid(s), sigmoid(s), |A|^2, |P|^2, |N|^2, 0, 0, 0}; entries_d[b] = {row_loss_u, row_loss_i}
__global__ __launch_bounds__(256) void head_fwd_kernel(const float *ui, const float *out_u, const float *out_i,
                                                       const double *rl_u, const double *rl_i, const int64_t *ancs,
                                                       const int64_t *poss, const int64_t *negs, int n_batch, int n_user,
                                                       int n_item, float *entries_f, double *entries_d) {
  const int b = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  if (b >= n_batch) return;
  const int64_t a = ancs[b], p = poss[b], n = negs[b];
  if (!in_range(a, p, n, n_user, n_item)) {  // nothing is gathered through a bad index: the entry, and with it every loss, is NaN
    if (l == 0) {
      float *e = entries_f + (size_t)b * 8;
      for (int k = 0; k < 8; ++k) e[k] = __builtin_nanf("");
      entries_d[(size_t)b * 2] = entries_d[(size_t)b * 2 + 1] = __builtin_nan("");
    }
    return;
  }
  const head_rows r = load_rows(ui, out_u, out_i, a, p, n, n_user, l);
  const float s = sum16(dot4(r.A, r.P)) - sum16(dot4(r.A, r.N));  // pairPredict
  const float sig = 1.0f / (1.0f + expf(-s));
  const float ra = sum16(dot4(r.A, r.A)), rp = sum16(dot4(r.P, r.P)), rn = sum16(dot4(r.N, r.N));
  if (l == 0) {
    float *e = entries_f + (size_t)b * 8;
    e[0] = logf(sig); e[1] = sig; e[2] = ra; e[3] = rp; e[4] = rn; e[5] = 0.f; e[6] = 0.f; e[7] = 0.f;
    entries_d[(size_t)b * 2] = rl_u[a];
    entries_d[(size_t)b * 2 + 1] = rl_i[p];
  }
}

// One workgroup: strided partial sums per thread, then a fixed tree.  results = {loss, bpr, reg, diff}.
__global__ __launch_bounds__(256) void head_reduce_kernel(const float *entries_f, const double *entries_d, int n_batch,
                                                          double batch_div, double reg_coef, double *results) {
  __shared__ double red[4][256];
  double s_log = 0, s_reg = 0, s_du = 0, s_di = 0;
  for (int b = threadIdx.x; b < n_batch; b += 256) {
    const float *e = entries_f + (size_t)b * 8;
    s_log += e[0];
    s_reg += (double)e[2] + (double)e[3] + (double)e[4];
    s_du += entries_d[(size_t)b * 2];
    s_di += entries_d[(size_t)b * 2 + 1];
  }
  red[0][threadIdx.x] = s_log; red[1][threadIdx.x] = s_reg; red[2][threadIdx.x] = s_du; red[3][threadIdx.x] = s_di;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double bpr = -red[0][0] / batch_div;
    const double reg = red[1][0] * reg_coef / batch_div;
    const double diff = red[2][0] / n_batch + red[3][0] / n_batch;
    results[0] = bpr + reg + diff; results[1] = bpr; results[2] = reg; results[3] = diff;
  }
}

// Per-entry gradients of the three gathered rows -> scratch [3][n_batch][64]; and, per table row,
// how many entries of its list name it (cnt) and which of them comes first (tag = list length -
// entry index, kept with an integer max).  The user list is ancs, the item list poss then negs.
__global__ __launch_bounds__(256) void head_bwd_entries_kernel(const float *ui, const float *out_u, const float *out_i,
                                                               const int64_t *ancs, const int64_t *poss,
                                                               const int64_t *negs, int n_batch, int n_user, int n_rows,
                                                               double batch_div, double reg_coef, const float *entries_f,
                                                               const double *g, float *scratch, int *tag, int *cnt,
                                                               int *slots, int *heavy) {
  const int b = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  if (blockIdx.x == 0 && threadIdx.x == 0) heavy[0] = 0;  // the list head_scatter_kernel fills starts empty
  if (b >= n_batch) return;
  const int64_t a = ancs[b], p = poss[b], n = negs[b];
  f32x4 *G = reinterpret_cast<f32x4 *>(scratch);
  int *idx32 = reinterpret_cast<int *>(scratch + (size_t)3 * n_batch * DG_D);  // [3][n_batch]: ancs | poss | negs
  if (!in_range(a, p, n, n_user, n_rows - n_user)) {  // (the forward made the losses NaN) no row is named, nothing is written
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) G[((size_t)k * n_batch + b) * 16 + l] = zero;
    if (l < 3) idx32[l * n_batch + b] = -1;
    return;
  }
  const head_rows r = load_rows(ui, out_u, out_i, a, p, n, n_user, l);
  const float c_bpr = (float)((g[0] + g[1]) / batch_div);
  const float c_reg = (float)((g[0] + g[2]) * 2.0 * reg_coef / batch_div);
  const float ds = -c_bpr * (1.0f - entries_f[(size_t)b * 8 + 1]);  // d(-log sigmoid(s))/ds = -(1 - sigmoid(s))
  G[((size_t)0 * n_batch + b) * 16 + l] = (r.P - r.N) * ds + r.A * c_reg;
  G[((size_t)1 * n_batch + b) * 16 + l] = r.A * ds + r.P * c_reg;
  G[((size_t)2 * n_batch + b) * 16 + l] = r.A * (-ds) + r.N * c_reg;
  // lanes 0..2 register the entry with its table row: first entry (max), count, and -- in arrival
  // order, which the scatter kernel undoes by sorting -- the entry index in one of the row's kSlots slots
  if (l < 3) {
    const int trow = l == 0 ? (int)a : n_user + (int)(l == 1 ? p : n);
    const int e = l == 2 ? n_batch + b : b;                 // index in the row's list (ancs; poss then negs)
    atomicMax(tag + trow, (l == 0 ? n_batch : 2 * n_batch) - e);
    const int k = atomicAdd(cnt + trow, 1);
    if (k < kSlots) slots[(size_t)trow * kSlots + k] = e;
    idx32[l * n_batch + b] = l == 0 ? (int)a : (int)(l == 1 ? p : n);
  }
}

// Entry e of the user list or of the item list writes its row of the gradient tables when it is
// the only entry naming that row; when several do, the first of them adds the later ones in list
// order (from the row's slots) and the others leave.  Rows named more often than there are slots
// -- popular items turn up a hundred times in a batch of real interactions -- are handed to
// head_heavy_kernel through a list (its order does not matter: a row is summed by one workgroup).
__global__ __launch_bounds__(256) void head_scatter_kernel(int n_batch, int n_user, const double *g, const float *scratch,
                                                           const int *tag, const int *cnt, const int *slots, int *heavy,
                                                           float *d_ui, float *d_out_u, float *d_out_i, double *d_rl_u,
                                                           double *d_rl_i) {
  const int e_all = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  if (e_all >= 3 * n_batch) return;
  const bool user = e_all < n_batch;
  const int e = user ? e_all : e_all - n_batch, m = user ? n_batch : 2 * n_batch;
  const int *idx32 = reinterpret_cast<const int *>(scratch + (size_t)3 * n_batch * DG_D);  // ancs | poss | negs as int32
  const int row = idx32[e_all];
  if (row < 0) return;  // entry with an index out of range
  const int trow = user ? row : n_user + row, c = cnt[trow];
  if (c != 1 && m - tag[trow] != e) return;  // not the only entry naming this row, nor the first of several
  if (c > kSlots) {
    if (l == 0) heavy[1 + atomicAdd(heavy, 1)] = e_all;
    return;
  }
  const f32x4 *G = reinterpret_cast<const f32x4 *>(scratch) + (user ? (size_t)0 : (size_t)n_batch * 16);
  f32x4 acc = G[(size_t)e * 16 + l];
  int n_loss = user || e < n_batch ? 1 : 0;  // entries that carry a diffusion-loss row: ancs, poss
  if (c > 1) {
    // the row's entries from its slots, put in ascending order (they were stored in arrival order)
    int s[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) s[k] = k < c ? slots[(size_t)trow * kSlots + k] : INT_MAX;
#pragma unroll
    for (int i = 0; i < kSlots; ++i)
#pragma unroll
      for (int k = 0; k + 1 < kSlots - i; ++k) {
        const int lo = min(s[k], s[k + 1]), hi = max(s[k], s[k + 1]);
        s[k] = lo; s[k + 1] = hi;
      }
#pragma unroll
    for (int k = 1; k < kSlots; ++k)   // s[0] == e, the first entry
      if (k < c) {
        acc += G[(size_t)s[k] * 16 + l];
        n_loss += (user || s[k] < n_batch) ? 1 : 0;
      }
  }
  reinterpret_cast<f32x4 *>(d_ui)[(size_t)trow * 16 + l] = acc;
  reinterpret_cast<f32x4 *>(user ? d_out_u : d_out_i)[(size_t)row * 16 + l] = acc;
  if (l == 0) (user ? d_rl_u : d_rl_i)[row] = (g[0] + g[3]) * n_loss / n_batch;
}

// Rows with more entries than slots, one 1024-thread workgroup per row at a time.  The row's list (the users' or
// the items', as int32 in LDS) is cut into 64 equal runs from the row's first entry on; each 16-lane group looks
// through one run, 16 entries per step, and adds the gradients of the entries naming the row in list order; the 64
// sums are then added in run order.  Fixed runs, fixed order: the result does not depend on timing.  (The adds of a
// run wait for their loads one after the other, so a row named 200 times wants many short runs.)
constexpr int kHeavyBlocks = 128, kHeavyRuns = 64;
__global__ __launch_bounds__(16 * kHeavyRuns) void head_heavy_kernel(int n_batch, int n_user, const double *g,
                                                                     const float *scratch, const int *heavy, float *d_ui,
                                                                     float *d_out_u, float *d_out_i, double *d_rl_u,
                                                                     double *d_rl_i) {
  extern __shared__ int idx[];  // [3 * n_batch]: ancs | poss | negs
  __shared__ f32x4 red[kHeavyRuns][16];
  __shared__ int red_n[kHeavyRuns];
  const int n_heavy = min(heavy[0], n_batch);  // (at most 3 n_batch / 7 rows can be named seven times)
  if ((int)blockIdx.x >= n_heavy) return;      // (the whole workgroup)
  const int *idx32 = reinterpret_cast<const int *>(scratch + (size_t)3 * n_batch * DG_D);
  for (int i = threadIdx.x; i < 3 * n_batch; i += 16 * kHeavyRuns) idx[i] = idx32[i];
  __syncthreads();
  const int k = threadIdx.x >> 4, l = threadIdx.x & 15, q = k & 3;
  for (int h = blockIdx.x; h < n_heavy; h += kHeavyBlocks) {
    const int e_all = min(max(heavy[1 + h], 0), 3 * n_batch - 1);
    const bool user = e_all < n_batch;
    const int e = user ? e_all : e_all - n_batch, m = user ? n_batch : 2 * n_batch;
    const int *list = user ? idx : idx + n_batch;
    const int row = list[e];
    const f32x4 *G = reinterpret_cast<const f32x4 *>(scratch) + (user ? (size_t)0 : (size_t)n_batch * 16);
    const int e0 = e & ~15, steps = (m - e0 + 15) >> 4;
    const int run = ((steps + kHeavyRuns - 1) / kHeavyRuns) << 4;  // entries per run: whole 16-entry steps
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int n_loss = 0;
    for (int j0 = e0 + k * run; j0 < min(m, e0 + (k + 1) * run); j0 += 16) {
      const int j = j0 + l;
      const bool hit = j >= e && j < m && list[j] == row;
      unsigned mask = (unsigned)(__ballot(hit) >> (16 * q)) & 0xffffu;
      while (mask) {
        const int jj = j0 + __builtin_ctz(mask);
        mask &= mask - 1;
        acc += G[(size_t)jj * 16 + l];
        n_loss += (user || jj < n_batch) ? 1 : 0;
      }
    }
    red[k][l] = acc;
    if (l == 0) red_n[k] = n_loss;
    __syncthreads();
    if (k == 0) {
      f32x4 sum = red[0][l];
      int nl = red_n[0];
#pragma unroll 8
      for (int r = 1; r < kHeavyRuns; ++r) { sum += red[r][l]; nl += red_n[r]; }
      const int trow = user ? row : n_user + row;
      reinterpret_cast<f32x4 *>(d_ui)[(size_t)trow * 16 + l] = sum;
      reinterpret_cast<f32x4 *>(user ? d_out_u : d_out_i)[(size_t)row * 16 + l] = sum;
      if (l == 0) (user ? d_rl_u : d_rl_i)[row] = (g[0] + g[3]) * nl / n_batch;
    }
    __syncthreads();
  }
}

// flags[row] = 1 for every table row a batch entry names (users: ancs; items, behind the users: poss and negs)
__global__ __launch_bounds__(256) void mark_rows_kernel(const int64_t *__restrict__ ancs, const int64_t *__restrict__ poss,
                                                        const int64_t *__restrict__ negs, int n_batch, int n_user, int n_item,
                                                        int *__restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * n_batch) return;
  const int which = i / n_batch, b = i - which * n_batch;
  const int64_t idx = which == 0 ? ancs[b] : which == 1 ? poss[b] : negs[b];
  if ((uint64_t)idx >= (uint64_t)(which == 0 ? n_user : n_item)) return;
  flags[(which == 0 ? 0 : n_user) + (int)idx] = 1;   // (several entries may name a row: the same value)
}

// The flagged rows of the user table and of the item table, each as an ascending list of row numbers inside its table
// and a count.  A workgroup takes kCompact consecutive flags of one table (one per thread): it first counts the flags
// set before its run -- every workgroup on its own, a few coalesced reads per thread instead of a scan across
// workgroups -- then numbers its own by ballots and writes them behind.  The table's last workgroup also writes the
// count.  (Small workgroups: the launch runs beside a propagation launch that fills the chip, and finds room sooner.)
constexpr int kCompact = 256;
__global__ __launch_bounds__(kCompact) void compact_rows_kernel(const int *__restrict__ flags, int n_user, int n_item, int blocks_u,
                                                                int *__restrict__ rows_u, int cap_u, int *__restrict__ rows_i,
                                                                int cap_i, int *__restrict__ counts) {
  constexpr int kWaves = kCompact / 64;
  __shared__ int wave_sum[kWaves], before_sum[kWaves];
  const bool items = (int)blockIdx.x >= blocks_u;
  const int *f = flags + (items ? n_user : 0);
  const int len = items ? n_item : n_user, cap = items ? cap_i : cap_u;
  int *rows = items ? rows_i : rows_u;
  const int chunk0 = ((int)blockIdx.x - (items ? blocks_u : 0)) * kCompact;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int before = 0;
  for (int r = threadIdx.x; r < chunk0; r += kCompact) before += f[r] != 0;
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  const int r = chunk0 + (int)threadIdx.x;
  const bool mine = r < len && f[r] != 0;
  const unsigned long long bal = __ballot(mine);
  if (lane == 0) { before_sum[wave] = before; wave_sum[wave] = __popcll(bal); }
  __syncthreads();
  int at = 0, total = 0;
  for (int w = 0; w < kWaves; ++w) {
    at += before_sum[w];
    if (w < wave) at += wave_sum[w];
    total += before_sum[w] + wave_sum[w];
  }
  at += __popcll(bal & ((1ull << lane) - 1ull));
  if (mine && at < cap) rows[at] = r;
  if (threadIdx.x == 0 && chunk0 + kCompact >= len) counts[items ? 1 : 0] = min(total, cap);
}

}  // namespace

extern "C" int dg_active_rows(const int64_t *ancs, const int64_t *poss, const int64_t *negs, int32_t n_batch, int32_t n_user,
                              int32_t n_item, int32_t *flags, int32_t *rows_u, int32_t *rows_i, int32_t *counts, void *stream) {
  if (!ancs || !poss || !negs || !flags || !rows_u || !rows_i || !counts) return dg_fail(DG_ERR_ARG, "dg_active_rows: NULL argument");
  if (n_batch <= 0 || n_batch > kMaxBatch) return dg_fail(DG_ERR_SIZE, "dg_active_rows: n_batch must be in [1,%d] (got %d)", kMaxBatch, n_batch);
  if (n_user <= 0 || n_item <= 0) return dg_fail(DG_ERR_ARG, "dg_active_rows: bad sizes");
  DG_HIP(hipMemsetAsync(flags, 0, ((size_t)n_user + (size_t)n_item) * sizeof(int32_t), (hipStream_t)stream));
  hipLaunchKernelGGL(mark_rows_kernel, dim3((3 * n_batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, ancs, poss, negs, n_batch,
                     n_user, n_item, flags);
  const int blocks_u = (n_user + kCompact - 1) / kCompact, blocks_i = (n_item + kCompact - 1) / kCompact;
  hipLaunchKernelGGL(compact_rows_kernel, dim3(blocks_u + blocks_i), dim3(kCompact), 0, (hipStream_t)stream, flags, n_user, n_item, blocks_u,
                     rows_u, n_batch, rows_i, 2 * n_batch, counts);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_rec_head_fwd(const float *ui, const float *out_u, const float *out_i, const double *row_loss_u,
                               const double *row_loss_i, const int64_t *ancs, const int64_t *poss, const int64_t *negs,
                               int32_t n_batch, int32_t n_user, int32_t n_item, double batch_div, double reg_coef,
                               int32_t d, float *entries_f, double *entries_d, double *results, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_rec_head_fwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!ui || !out_u || !out_i || !row_loss_u || !row_loss_i || !ancs || !poss || !negs || !entries_f || !entries_d)
    return dg_fail(DG_ERR_ARG, "dg_rec_head_fwd: NULL argument");
  if (n_batch <= 0 || n_batch > kMaxBatch) return dg_fail(DG_ERR_SIZE, "dg_rec_head_fwd: n_batch must be in [1,%d] (got %d)", kMaxBatch, n_batch);
  if (n_user <= 0 || n_item <= 0 || batch_div == 0.0) return dg_fail(DG_ERR_ARG, "dg_rec_head_fwd: bad sizes");
  hipLaunchKernelGGL(head_fwd_kernel, dim3((n_batch + 15) / 16), dim3(256), 0, (hipStream_t)stream, ui, out_u, out_i,
                     row_loss_u, row_loss_i, ancs, poss, negs, n_batch, n_user, n_item, entries_f, entries_d);
  if (results)
    hipLaunchKernelGGL(head_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, entries_f, entries_d, n_batch, batch_div,
                       reg_coef, results);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_rec_head_reduce(const float *entries_f, const double *entries_d, int32_t n_batch, double batch_div,
                                  double reg_coef, double *results, void *stream) {
  if (!entries_f || !entries_d || !results) return dg_fail(DG_ERR_ARG, "dg_rec_head_reduce: NULL argument");
  if (n_batch <= 0 || n_batch > kMaxBatch || batch_div == 0.0) return dg_fail(DG_ERR_SIZE, "dg_rec_head_reduce: bad sizes");
  hipLaunchKernelGGL(head_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, entries_f, entries_d, n_batch, batch_div,
                     reg_coef, results);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_rec_head_bwd(const float *ui, const float *out_u, const float *out_i, const int64_t *ancs,
                               const int64_t *poss, const int64_t *negs, int32_t n_batch, int32_t n_user, int32_t n_item,
                               double batch_div, double reg_coef, int32_t d, const float *entries_f, const double *g,
                               float *scratch, int32_t *row_tags, float *d_ui, float *d_out_u, float *d_out_i,
                               double *d_row_loss_u, double *d_row_loss_i, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_rec_head_bwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!ui || !out_u || !out_i || !ancs || !poss || !negs || !entries_f || !g || !scratch || !row_tags || !d_ui || !d_out_u ||
      !d_out_i || !d_row_loss_u || !d_row_loss_i)
    return dg_fail(DG_ERR_ARG, "dg_rec_head_bwd: NULL argument");
  if (n_batch <= 0 || n_batch > kMaxBatch) return dg_fail(DG_ERR_SIZE, "dg_rec_head_bwd: n_batch must be in [1,%d] (got %d)", kMaxBatch, n_batch);
  if (n_user <= 0 || n_item <= 0 || batch_div == 0.0) return dg_fail(DG_ERR_ARG, "dg_rec_head_bwd: bad sizes");
  if (((uintptr_t)d_ui | (uintptr_t)d_out_u | (uintptr_t)d_out_i | (uintptr_t)scratch) & 15)
    return dg_fail(DG_ERR_ARG, "dg_rec_head_bwd: gradient tables and scratch must be 16-byte aligned");
  const int n_rows = n_user + n_item;
  int *tag = row_tags, *cnt = row_tags + n_rows, *slots = row_tags + 2 * (size_t)n_rows;
  // rows named more than kSlots times: a counter and a list behind the int32 copy of the index lists
  int *heavy = reinterpret_cast<int *>(scratch + (size_t)3 * n_batch * (DG_D + 1));
  hipLaunchKernelGGL(head_bwd_entries_kernel, dim3((n_batch + 15) / 16), dim3(256), 0, (hipStream_t)stream, ui, out_u, out_i,
                     ancs, poss, negs, n_batch, n_user, n_rows, batch_div, reg_coef, entries_f, g, scratch, tag, cnt, slots, heavy);
  hipLaunchKernelGGL(head_scatter_kernel, dim3((3 * n_batch + 15) / 16), dim3(256), 0, (hipStream_t)stream, n_batch, n_user, g,
                     scratch, tag, cnt, slots, heavy, d_ui, d_out_u, d_out_i, d_row_loss_u, d_row_loss_i);
  static std::atomic<uint64_t> opted{0};
  DG_HIP(dg_lds_opt_in(reinterpret_cast<const void *>(head_heavy_kernel), 3 * kMaxBatch * (int)sizeof(int), opted));
  hipLaunchKernelGGL(head_heavy_kernel, dim3(kHeavyBlocks), dim3(16 * kHeavyRuns), 3 * (size_t)n_batch * sizeof(int), (hipStream_t)stream,
                     n_batch, n_user, g, scratch, heavy, d_ui, d_out_u, d_out_i, d_row_loss_u, d_row_loss_i);
  DG_HIP(hipGetLastError());
  return 0;
}

// The zeroed buffer dg_rec_head_bwd starts from (autograd's zero-initialised gradient tables of the three batch gathers,
// DiffGraph-Rec/Model.py:101-104), cleared by the runtime's memset on the caller's stream -- recorded into a HIP graph
// it is a memset node, not a kernel.
extern "C" int dg_clear(void *dst, size_t bytes, void *stream) {
  if (!dst && bytes) return dg_fail(DG_ERR_ARG, "dg_clear: NULL buffer");
  if (bytes == 0) return 0;
  DG_HIP(hipMemsetAsync(dst, 0, bytes, (hipStream_t)stream));
  return 0;
}
