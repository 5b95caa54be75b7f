// The node-classification model's two dense layers for gfx950.
//
// Replaces, in DiffGraph_NC/Model.py:
//   * transform_layer = nn.Linear(f_dim, latdim) applied to the feature matrix (:38-40, :47) -- forward and the
//     weight / bias gradients autograd forms for it (the features themselves take no gradient);
//   * dense = nn.Linear(latdim, 4), F.log_softmax(.., dim=1), the batch rows and F.nll_loss (:84-89), forward and
//     backward (gradients of the embeddings, the weight and the bias), and the plain logits of get_allembeds (:107).
// Both are small (4 057 x 334 x 64 and 4 057 x 64 x 4 on DBLP): plain fp32 FMA tiles staged through LDS, every sum in
// a fixed order (k ascending; table rows in strips whose partial sums a second launch adds in strip order), so the
// results are the same bits on every run.
#include "dg_internal.h"

#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTile = 64;   // columns of an output tile; its rows: 64 (4 x 4 outputs per thread) or 16 (1 x 4)
constexpr int kStep = 16;   // reduction steps staged per round

// C[i][j] (+)= sum_r A(r, i) * B(r, j) over r in [r0, r1): A(r, i) = a[i * lda_i + r * lda_r], B likewise.
// The workgroup computes the 64 x 64 tile (i0.., j0..); out-of-range i / j / r read as zero.
struct tile_args {
  const float *a, *b;
  int64_t lda_i, lda_r, ldb_j, ldb_r;
  int ni, nj;
};

// TI = rows of the tile (64 or 16); thread (ty = tid >> 4, tx = tid & 15) owns rows (TI / 16) ty .., columns 4 tx ..
// Consecutive threads read consecutive addresses of an operand whichever of its two strides is 1; the next round's
// elements are requested before the current round is multiplied.  colsum (threads < TI, when asked for): the sum
// over r of A(r, i0 + tid), rounds and rows in order.
template <int TI, bool COLSUM>
__device__ __forceinline__ void tile_product(const tile_args &t, int i0, int j0, int64_t r0, int64_t r1, float (&acc)[TI / 16][4],
                                             float &colsum) {
  constexpr int U = TI / 16;
  __shared__ float as[kStep][TI + 4], bs[kStep][kTile + 4];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;
  colsum = 0.f;
  const bool a_along_r = t.lda_r == 1, b_along_r = t.ldb_r == 1;
  int a_i[U], a_r[U], b_j[4], b_r[4];
#pragma unroll
  for (int e = 0; e < U; ++e) {
    const int idx = tid + 256 * e;
    a_i[e] = a_along_r ? idx >> 4 : idx % TI;
    a_r[e] = a_along_r ? idx & 15 : idx / TI;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e;
    b_j[e] = b_along_r ? idx >> 4 : idx & 63;
    b_r[e] = b_along_r ? idx & 15 : idx >> 6;
  }
  float av[U], bv[4];
  auto fetch = [&](int64_t r) {
#pragma unroll
    for (int e = 0; e < U; ++e) {
      const int64_t row = r + a_r[e];
      av[e] = 0.f;
      if (row < r1 && i0 + a_i[e] < t.ni) av[e] = t.a[(int64_t)(i0 + a_i[e]) * t.lda_i + row * t.lda_r];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = r + b_r[e];
      bv[e] = 0.f;
      if (row < r1 && j0 + b_j[e] < t.nj) bv[e] = t.b[(int64_t)(j0 + b_j[e]) * t.ldb_j + row * t.ldb_r];
    }
  };
  fetch(r0);
  for (int64_t r = r0; r < r1; r += kStep) {
#pragma unroll
    for (int e = 0; e < U; ++e) as[a_r[e]][a_i[e]] = av[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) bs[b_r[e]][b_j[e]] = bv[e];
    __syncthreads();
    if (r + kStep < r1) fetch(r + kStep);
#pragma unroll
    for (int rr = 0; rr < kStep; ++rr) {
      float ar[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ar[u] = as[rr][U * ty + u];
      const f32x4 br = *reinterpret_cast<const f32x4 *>(&bs[rr][4 * tx]);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(ar[u], br[v], acc[u][v]);
    }
    if (COLSUM && tid < TI) {
#pragma unroll
      for (int rr = 0; rr < kStep; ++rr) colsum += as[rr][tid];
    }
    __syncthreads();
  }
}

// y[n][m] = x[n][k] . w[m][k]^T + bias[m]; 16-row tiles: a few thousand rows then make a few hundred workgroups
constexpr int kFwdRows = 16;
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, int n, int k, int m,
                                                         float *__restrict__ y) {
  tile_args t = {x, w, (int64_t)k, 1, (int64_t)k, 1, n, m};
  const int i0 = blockIdx.x * kFwdRows, j0 = blockIdx.y * kTile;
  float acc[1][4], unused;
  tile_product<kFwdRows, false>(t, i0, j0, 0, k, acc, unused);
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int i = i0 + ty;
  if (i >= n) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int j = j0 + 4 * tx + v;
    if (j < m) y[(int64_t)i * m + j] = acc[0][v] + (bias ? bias[j] : 0.f);
  }
}

// partial[s][m][k] = sum over the strip's rows of dy[row][o] * x[row][c]; partial[s][m * k + o] = column sums of dy
__global__ __launch_bounds__(256) void linear_wgrad_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                   int n, int k, int m, int rows_per_strip,
                                                                   float *__restrict__ partial) {
  tile_args t = {dy, x, 1, (int64_t)m, 1, (int64_t)k, m, k};
  const int i0 = blockIdx.x * kTile, j0 = blockIdx.y * kTile, s = blockIdx.z;
  const int64_t r0 = (int64_t)s * rows_per_strip, r1 = min(r0 + rows_per_strip, (int64_t)n);
  float acc[4][4], colsum;
  tile_product<kTile, true>(t, i0, j0, r0, r1, acc, colsum);
  float *P = partial + (size_t)s * ((size_t)m * k + m);
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = i0 + 4 * ty + u;
    if (o >= m) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c = j0 + 4 * tx + v;
      if (c < k) P[(size_t)o * k + c] = acc[u][v];
    }
  }
  if (blockIdx.y == 0 && (int)threadIdx.x < kTile) {   // the bias gradient of this strip: rows in order
    const int o = i0 + threadIdx.x;
    if (o < m) P[(size_t)m * k + o] = colsum;
  }
}

__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ partial, int n_strips, int count,
                                                                  int mk, float *__restrict__ d_w, float *__restrict__ d_b) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= count) return;
  float s = 0.f;
  for (int g = 0; g < n_strips; ++g) s += partial[(size_t)g * count + f];
  if (f < mk) d_w[f] = s;
  else d_b[f - mk] = s;
}

// ---- dense + log_softmax + nll over the batch rows ----------------------------------------------------
constexpr int kMaxClasses = 16;

// logits[row][c] = <e[row], w[c]> + b[c]: a 16-lane group per row
__global__ __launch_bounds__(256) void nc_logits_kernel(const f32x4 *__restrict__ e, const f32x4 *__restrict__ w,
                                                        const float *__restrict__ b, int64_t n, int n_cls,
                                                        float *__restrict__ logits) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  if (row >= n) return;
  const f32x4 v = e[row * 16 + l];
  for (int c = 0; c < n_cls; ++c) {
    const f32x4 wc = w[c * 16 + l];
    float d = v.x * wc.x + v.y * wc.y + v.z * wc.z + v.w * wc.w;
    d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4); d += __shfl_xor(d, 8);
    if (l == 0) logits[row * n_cls + c] = d + b[c];
  }
}

// One workgroup.  Forward: nll = -mean_b log_softmax(logits[anc_b])[label_b] (fp32, as F.nll_loss); per entry the
// softmax is kept for the backward call.  entries: [n_batch][n_cls] probabilities.
__global__ __launch_bounds__(256) void nc_nll_fwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ ancs,
                                                         const int64_t *__restrict__ labels, int n_batch, int64_t n, int n_cls,
                                                         float *__restrict__ entries, float *__restrict__ loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int b = threadIdx.x; b < n_batch; b += 256) {
    const int64_t a = ancs[b], y = labels[b];
    float *pe = entries + (size_t)b * n_cls;
    if ((uint64_t)a >= (uint64_t)n || (uint64_t)y >= (uint64_t)n_cls) {   // not gathered through: the loss is NaN
      for (int c = 0; c < n_cls; ++c) pe[c] = __builtin_nanf("");
      s += __builtin_nanf("");
      continue;
    }
    const float *z = logits + a * n_cls;
    float mx = z[0];
    for (int c = 1; c < n_cls; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.f;
    for (int c = 0; c < n_cls; ++c) se += expf(z[c] - mx);
    const float lse = mx + logf(se);
    for (int c = 0; c < n_cls; ++c) pe[c] = expf(z[c] - lse);
    s += -(z[y] - lse);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / (float)n_batch;
}

// One workgroup of 64 threads (thread = embedding column), batch entries in order: d_e[anc_b] += dl_b . W,
// d_w[c] += dl_b[c] e[anc_b], d_b += dl_b with dl_b = g (softmax_b - onehot(label_b)) / n_batch.  d_e arrives zeroed.
__global__ __launch_bounds__(64) void nc_nll_bwd_kernel(const float *__restrict__ e, const float *__restrict__ w,
                                                        const int64_t *__restrict__ ancs, const int64_t *__restrict__ labels,
                                                        const float *__restrict__ entries, const float *__restrict__ g,
                                                        int n_batch, int64_t n, int n_cls, float *__restrict__ d_e,
                                                        float *__restrict__ d_w, float *__restrict__ d_b) {
  const int t = threadIdx.x;
  float wc[kMaxClasses], dw[kMaxClasses], db = 0.f;
  for (int c = 0; c < kMaxClasses; ++c) {
    wc[c] = c < n_cls ? w[c * DG_D + t] : 0.f;
    dw[c] = 0.f;
  }
  const float scale = g[0] / (float)n_batch;
  for (int b = 0; b < n_batch; ++b) {
    const int64_t a = ancs[b], y = labels[b];
    if ((uint64_t)a >= (uint64_t)n || (uint64_t)y >= (uint64_t)n_cls) continue;
    const float ev = e[a * DG_D + t];
    float de = 0.f;
    for (int c = 0; c < n_cls; ++c) {
      const float dl = scale * (entries[(size_t)b * n_cls + c] - (c == (int)y ? 1.f : 0.f));
      de = fmaf(dl, wc[c], de);
      dw[c] = fmaf(dl, ev, dw[c]);
      if (t == c) db += dl;
    }
    d_e[a * DG_D + t] += de;   // entries in order, one thread per column: a row named twice is added twice, in order
  }
  for (int c = 0; c < n_cls; ++c) d_w[c * DG_D + t] = dw[c];
  if (t < n_cls) d_b[t] = db;
}

// The whole loss head of DiffGraph_NC's cal_loss over the batch rows only (Model.py:79-89), one workgroup; a 16-lane
// group per batch entry b (entries b, b + 16, ... per group):
//   e = target[a] + diff[a];  z = e . w^T + bias;  label = argmax(onehot[a]) (first maximum) or labels[b];
//   nll = -mean_b log_softmax(z)[label] (fp32)   and   diff_loss = mean_b row_loss[a] (fp64).
// Kept for the backward call: the entries' softmax and labels.
__global__ __launch_bounds__(256) void nc_head_fwd_kernel(const f32x4 *__restrict__ target, const f32x4 *__restrict__ diff,
                                                          const double *__restrict__ row_loss, const f32x4 *__restrict__ w,
                                                          const float *__restrict__ bias, const int64_t *__restrict__ ancs,
                                                          const float *__restrict__ onehot, const int64_t *__restrict__ labels,
                                                          int n_batch, int64_t n, int n_cls, float *__restrict__ entries,
                                                          int32_t *__restrict__ label_out, float *__restrict__ nll,
                                                          double *__restrict__ diff_loss) {
  __shared__ float red_f[16];
  __shared__ double red_d[16];
  const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
  float s = 0.f;
  double sd = 0.0;
  for (int b = grp; b < n_batch; b += 16) {
    const int64_t a = ancs[b];
    float *pe = entries + (size_t)b * n_cls;
    int y = -1;
    if ((uint64_t)a < (uint64_t)n) {
      if (onehot) {
        y = 0;
        float best = onehot[a * n_cls];
        for (int c = 1; c < n_cls; ++c) {
          const float v = onehot[a * n_cls + c];
          if (v > best) { best = v; y = c; }
        }
      } else {
        const int64_t yy = labels[b];
        y = (uint64_t)yy < (uint64_t)n_cls ? (int)yy : -1;
      }
    }
    if (l == 0) label_out[b] = y;
    if (y < 0) {   // not gathered through: both losses are NaN
      if (l < n_cls) pe[l] = __builtin_nanf("");
      s += __builtin_nanf("");
      sd += (double)__builtin_nanf("");
      continue;
    }
    const f32x4 v = target[a * 16 + l] + diff[a * 16 + l];
    float z[kMaxClasses];
    for (int c = 0; c < n_cls; ++c) {
      const f32x4 wc = w[c * 16 + l];
      float d = v.x * wc.x + v.y * wc.y + v.z * wc.z + v.w * wc.w;
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4); d += __shfl_xor(d, 8);
      z[c] = d + bias[c];
    }
    float mx = z[0];
    for (int c = 1; c < n_cls; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.f;
    for (int c = 0; c < n_cls; ++c) se += expf(z[c] - mx);
    const float lse = mx + logf(se);
    float zy = 0.f;
    for (int c = 0; c < n_cls; ++c) {
      if (l == 0) pe[c] = expf(z[c] - lse);
      if (c == y) zy = z[c];
    }
    s += -(zy - lse);
    sd += row_loss[a];
  }
  if (l == 0) { red_f[grp] = s; red_d[grp] = sd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    double td = 0.0;
    for (int g = 0; g < 16; ++g) { t += red_f[g]; td += red_d[g]; }
    nll[0] = t / (float)n_batch;
    diff_loss[0] = td / (double)n_batch;
  }
}

// Backward of nc_head_fwd_kernel, one workgroup of 64 threads (thread = embedding column), batch entries in order:
// d_e[a] += dl_b . W (the gradient of target AND of diff: they enter as a sum), d_w[c] += dl_b[c] e[a], d_b += dl_b
// with dl_b = g_nll (softmax_b - onehot(label_b)) / n_batch, and d_row_loss[a] += g_diff / n_batch.  d_e and
// d_row_loss arrive zeroed.
__global__ __launch_bounds__(64) void nc_head_bwd_kernel(const float *__restrict__ target, const float *__restrict__ diff,
                                                         const float *__restrict__ w, const int64_t *__restrict__ ancs,
                                                         const int32_t *__restrict__ label_in, const float *__restrict__ entries,
                                                         const float *__restrict__ g_nll, const double *__restrict__ g_diff,
                                                         int n_batch, int n_cls, float *__restrict__ d_e,
                                                         double *__restrict__ d_row_loss, float *__restrict__ d_w,
                                                         float *__restrict__ d_b) {
  const int t = threadIdx.x;
  float wc[kMaxClasses], dw[kMaxClasses], db = 0.f;
  for (int c = 0; c < kMaxClasses; ++c) {
    wc[c] = c < n_cls ? w[c * DG_D + t] : 0.f;
    dw[c] = 0.f;
  }
  const float scale = g_nll[0] / (float)n_batch;
  const double dscale = g_diff[0] / (double)n_batch;
  for (int b = 0; b < n_batch; ++b) {
    const int y = label_in[b];
    if (y < 0) continue;
    const int64_t a = ancs[b];
    const float ev = target[a * DG_D + t] + diff[a * DG_D + t];
    float de = 0.f;
    for (int c = 0; c < n_cls; ++c) {
      const float dl = scale * (entries[(size_t)b * n_cls + c] - (c == y ? 1.f : 0.f));
      de = fmaf(dl, wc[c], de);
      dw[c] = fmaf(dl, ev, dw[c]);
      if (t == c) db += dl;
    }
    d_e[a * DG_D + t] += de;   // entries in order, one thread per column: a row named twice is added twice, in order
    if (t == 0) d_row_loss[a] += dscale;
  }
  for (int c = 0; c < n_cls; ++c) d_w[c * DG_D + t] = dw[c];
  if (t < n_cls) d_b[t] = db;
}

}  // namespace

extern "C" int dg_nc_head_fwd(const float *target, const float *diff, const double *row_loss, const float *w, const float *b,
                              const int64_t *ancs, const float *onehot, const int64_t *labels, int32_t n_batch, int64_t n,
                              int32_t d, int32_t n_cls, float *entries, int32_t *label_out, float *nll, double *diff_loss,
                              void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_nc_head_fwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!target || !diff || !row_loss || !w || !b || !ancs || !entries || !label_out || !nll || !diff_loss)
    return dg_fail(DG_ERR_ARG, "dg_nc_head_fwd: NULL argument");
  if (!onehot && !labels) return dg_fail(DG_ERR_ARG, "dg_nc_head_fwd: neither a one-hot label table nor labels given");
  if (n_batch <= 0 || n <= 0 || n_cls <= 0 || n_cls > kMaxClasses) return dg_fail(DG_ERR_ARG, "dg_nc_head_fwd: bad sizes");
  if (((uintptr_t)target | (uintptr_t)diff | (uintptr_t)w) & 15) return dg_fail(DG_ERR_ARG, "dg_nc_head_fwd: tables must be 16-byte aligned");
  hipLaunchKernelGGL(nc_head_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4 *>(target),
                     reinterpret_cast<const f32x4 *>(diff), row_loss, reinterpret_cast<const f32x4 *>(w), b, ancs, onehot, labels,
                     n_batch, n, n_cls, entries, label_out, nll, diff_loss);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_nc_head_bwd(const float *target, const float *diff, const float *w, const int64_t *ancs,
                              const int32_t *label_in, const float *entries, const float *g_nll, const double *g_diff,
                              int32_t n_batch, int64_t n, int32_t d, int32_t n_cls, float *d_e, double *d_row_loss, float *d_w,
                              float *d_b, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_nc_head_bwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!target || !diff || !w || !ancs || !label_in || !entries || !g_nll || !g_diff || !d_e || !d_row_loss || !d_w || !d_b)
    return dg_fail(DG_ERR_ARG, "dg_nc_head_bwd: NULL argument");
  if (n_batch <= 0 || n <= 0 || n_cls <= 0 || n_cls > kMaxClasses) return dg_fail(DG_ERR_ARG, "dg_nc_head_bwd: bad sizes");
  hipLaunchKernelGGL(nc_head_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, target, diff, w, ancs, label_in, entries, g_nll,
                     g_diff, n_batch, n_cls, d_e, d_row_loss, d_w, d_b);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_linear_fwd(const float *x, const float *w, const float *bias, int32_t n, int32_t k, int32_t m, float *y,
                             void *stream) {
  if (!x || !w || !y) return dg_fail(DG_ERR_ARG, "dg_linear_fwd: NULL argument");
  if (n < 0 || k <= 0 || m <= 0) return dg_fail(DG_ERR_ARG, "dg_linear_fwd: bad sizes");
  if (n == 0) return 0;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((n + kFwdRows - 1) / kFwdRows, (m + kTile - 1) / kTile), dim3(256), 0, (hipStream_t)stream,
                     x, w, bias, n, k, m, y);
  DG_HIP(hipGetLastError());
  return 0;
}

constexpr int kWgradStrips = 32;

extern "C" size_t dg_linear_wgrad_workspace(int32_t k, int32_t m) {
  if (k <= 0 || m <= 0) return 0;
  return (size_t)kWgradStrips * ((size_t)m * k + m) * sizeof(float);
}

extern "C" int dg_linear_wgrad(const float *dy, const float *x, int32_t n, int32_t k, int32_t m, void *workspace,
                               size_t workspace_bytes, float *d_w, float *d_b, void *stream) {
  if (!dy || !x || !workspace || !d_w || !d_b) return dg_fail(DG_ERR_ARG, "dg_linear_wgrad: NULL argument");
  if (n < 0 || k <= 0 || m <= 0) return dg_fail(DG_ERR_ARG, "dg_linear_wgrad: bad sizes");
  if (workspace_bytes < dg_linear_wgrad_workspace(k, m))
    return dg_fail(DG_ERR_SIZE, "dg_linear_wgrad: workspace of %zu bytes, %zu needed", workspace_bytes, dg_linear_wgrad_workspace(k, m));
  int rows = (n + kWgradStrips - 1) / kWgradStrips;
  rows = (rows + kStep - 1) / kStep * kStep;
  if (rows == 0) rows = kStep;
  hipLaunchKernelGGL(linear_wgrad_partial_kernel, dim3((m + kTile - 1) / kTile, (k + kTile - 1) / kTile, kWgradStrips), dim3(256), 0,
                     (hipStream_t)stream, dy, x, n, k, m, rows, (float *)workspace);
  const int count = m * k + m;
  hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float *)workspace, kWgradStrips, count, m * k, d_w, d_b);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_nc_logits(const float *e, const float *w, const float *b, int64_t n, int32_t d, int32_t n_cls, float *logits,
                            void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_nc_logits: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!e || !w || !b || !logits) return dg_fail(DG_ERR_ARG, "dg_nc_logits: NULL argument");
  if (n < 0 || n_cls <= 0 || n_cls > kMaxClasses) return dg_fail(DG_ERR_ARG, "dg_nc_logits: n_cls must be in [1,%d]", kMaxClasses);
  if (((uintptr_t)e | (uintptr_t)w) & 15) return dg_fail(DG_ERR_ARG, "dg_nc_logits: tables must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(nc_logits_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4 *>(e), reinterpret_cast<const f32x4 *>(w), b, n, n_cls, logits);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_nc_nll_fwd(const float *logits, const int64_t *ancs, const int64_t *labels, int32_t n_batch, int64_t n,
                             int32_t n_cls, float *entries, float *loss, void *stream) {
  if (!logits || !ancs || !labels || !entries || !loss) return dg_fail(DG_ERR_ARG, "dg_nc_nll_fwd: NULL argument");
  if (n_batch <= 0 || n <= 0 || n_cls <= 0 || n_cls > kMaxClasses) return dg_fail(DG_ERR_ARG, "dg_nc_nll_fwd: bad sizes");
  hipLaunchKernelGGL(nc_nll_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ancs, labels, n_batch, n, n_cls, entries,
                     loss);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_nc_nll_bwd(const float *e, const float *w, const int64_t *ancs, const int64_t *labels, const float *entries,
                             const float *g, int32_t n_batch, int64_t n, int32_t d, int32_t n_cls, float *d_e, float *d_w,
                             float *d_b, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_nc_nll_bwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!e || !w || !ancs || !labels || !entries || !g || !d_e || !d_w || !d_b) return dg_fail(DG_ERR_ARG, "dg_nc_nll_bwd: NULL argument");
  if (n_batch <= 0 || n <= 0 || n_cls <= 0 || n_cls > kMaxClasses) return dg_fail(DG_ERR_ARG, "dg_nc_nll_bwd: bad sizes");
  hipLaunchKernelGGL(nc_nll_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e, w, ancs, labels, entries, g, n_batch, n, n_cls,
                     d_e, d_w, d_b);
  DG_HIP(hipGetLastError());
  return 0;
}
