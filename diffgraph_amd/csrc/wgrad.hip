// Weight gradients of the denoiser MLP for gfx950, from the four [n][64] tables the loss
// backward kernel leaves behind (dout_tot, hid, dpre, xn) and the per-row timesteps.
//
// Replaces what autograd records for the three nn.Linear layers of Denoise
// (DiffGraph-Rec/Model.py:184-191, used at :218-232) when training_losses2 is differentiated
// (loss.backward(), DiffGraph-Rec/main.py:127):
//     dW_out          = dout^T . hid            db_out = colsum(dout)
//     dW_in[:, :64]   = dpre^T . xn             db_in  = colsum(dpre)
//     dW_in[:, 64:72] = dpre^T . emb[ts]        (emb[t] = emb_layer([cos(t f), sin(t f)]), 8 wide)
//     dW_emb          = W_in[:,64:72]^T . (dpre^T . te[ts])      db_emb = W_in[:,64:72]^T . db_in
// All of them contract over the n table rows, so the rows are the K dimension of exact-fp32
// v_mfma_f32_16x16x4_f32 tiles: a lane (j = lane&15, k = lane>>4) loads one float4 of table row
// base+k -- 16 lanes cover a whole 256-B row, a wavefront load four rows -- and component e of
// that float4 (feature 4j+e) is the operand of the tile with index e.  Sixteen 16x16 tiles make
// one 64x64 product.  Every workgroup sums a contiguous strip of rows; a second launch adds the
// per-workgroup partial results in workgroup order, so the gradients are the same bits on every
// run (no float atomics) and nothing depends on a BLAS workspace or semaphore.
#include "dg_internal.h"
#include "dg_wgrad.h"

#include <math.h>
#include <mutex>

using dg_wg::f32x4;
using dg_wg::kPartBin;
using dg_wg::kPartBout;
using dg_wg::kPartFloats;
using dg_wg::kPartTab;
using dg_wg::kPartWin;
using dg_wg::kPartWout;
using dg_wg::kTabCols;
using dg_wg::mma64;
using dg_wg::store64;

namespace {

constexpr int kGroups = 256;          // workgroups of the partial launch = partial results to add

struct wg_args {
  const float *dout, *hid, *dpre, *xn;
  const int64_t *ts;
  const float *tab;   // [n_steps][16]
  float *partials;    // [kGroups][kPartFloats]
  int64_t n;
  int32_t n_steps;
  int64_t rows_per_group;  // multiple of 16
};

struct tab_args {
  const float *w_emb, *b_emb;
  float freqs[4];
  int32_t n_steps;
  float *tab;
};

// Up to kMaxTasks denoisers per launch (the user and the item denoiser of a training step): blockIdx.y picks one.
constexpr int kMaxTasks = 2;
template <typename T>
struct per_task {
  T t[kMaxTasks];
};

// tab[t] = [ emb_layer(te_t) | te_t ],  te_t = [cos(t f_0..3), sin(t f_0..3)]  (Model.py:215-218)
__global__ __launch_bounds__(256) void wgrad_tab_kernel(const per_task<tab_args> all) {
  const tab_args &a = all.t[blockIdx.y];
  float *__restrict__ tab = a.tab;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.n_steps) return;
  float te[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float arg = (float)t * a.freqs[k];
    te[k] = cosf(arg);
    te[k + 4] = sinf(arg);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float emb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) emb += te[k] * a.w_emb[e * 8 + k];
    tab[t * kTabCols + e] = emb + a.b_emb[e];
    tab[t * kTabCols + 8 + e] = te[e];
  }
}

struct wg_operands {
  f32x4 a, b;
  float t;
};

__device__ __forceinline__ wg_operands load_operands(const wg_args &a, const float *A, const float *B, bool with_tab,
                                                     int64_t row, int64_t r1, int j) {
  wg_operands o = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, 0.f};
  if (row < r1) {
    o.a = *reinterpret_cast<const f32x4 *>(A + row * DG_D + 4 * j);
    o.b = *reinterpret_cast<const f32x4 *>(B + row * DG_D + 4 * j);
    if (with_tab) {
      const int t = min(max((int)a.ts[row], 0), a.n_steps - 1);
      o.t = a.tab[t * kTabCols + j];
    }
  }
  return o;
}

// Waves 0..3 of a workgroup: dout^T.hid + colsum(dout); waves 4..7: dpre^T.xn, dpre^T.tab[ts],
// colsum(dpre).  The four waves of a half take every fourth 4-row quad of the workgroup's strip
// (the next quad's rows are requested before the current one is multiplied) and hand their sums
// to the half's first wave through LDS, one after the other.
__global__ __launch_bounds__(512) void wgrad_partial_kernel(const per_task<wg_args> all) {
  const wg_args &a = all.t[blockIdx.y];
  __shared__ f32x4 xch[2][21][64];  // per half: 16 + 4 + 1 accumulator float4s per lane
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = wave >> 2, sub = wave & 3;
  const int j = lane & 15, k = lane >> 4;
  const bool with_tab = half == 1;
  const float *A = half == 0 ? a.dout : a.dpre;
  const float *B = half == 0 ? a.hid : a.xn;
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_group;
  const int64_t r1 = min(r0 + a.rows_per_group, a.n);

  f32x4 acc[4][4], acc_t[4], csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    acc_t[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int64_t q = r0 + 4 * sub;
  wg_operands cur = load_operands(a, A, B, with_tab, q + k, r1, j);
  for (; q < r1; q += 16) {
    const wg_operands nxt = load_operands(a, A, B, with_tab, q + 16 + k, r1, j);
    mma64(cur.a, cur.b, acc);
    csum += cur.a;
    if (with_tab) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        acc_t[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[mb], cur.t, acc_t[mb], 0, 0, 0);
    }
    cur = nxt;
  }
  // column sums: add the four k-lanes that share j (fixed order)
  for (int c = 0; c < 4; ++c) {
    float v = csum[c];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    csum[c] = v;
  }
  for (int w = 1; w < 4; ++w) {
    if (sub == w) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) xch[half][mb * 4 + nb][lane] = acc[mb][nb];
        xch[half][16 + mb][lane] = acc_t[mb];
      }
      xch[half][20][lane] = csum;
    }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] += xch[half][mb * 4 + nb][lane];
        acc_t[mb] += xch[half][16 + mb][lane];
      }
      csum += xch[half][20][lane];
    }
    __syncthreads();
  }
  if (sub != 0) return;
  float *P = a.partials + (size_t)blockIdx.x * kPartFloats;
  store64(P + (half == 0 ? kPartWout : kPartWin), lane, acc);
  if (k == 0) *reinterpret_cast<f32x4 *>(P + (half == 0 ? kPartBout : kPartBin) + 4 * j) = csum;
  if (half == 1) {
    // (dpre^T . tab)[o][c]: acc_t[mb][r] of lane l is o = 16 (l>>4) + 4 r + mb, c = l & 15
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[kPartTab + (16 * k + 4 * r + mb) * kTabCols + j] = acc_t[mb][r];
  }
}

struct fin_args {
  const float *partials;
  int32_t n_partials;
  float *d_w_in, *d_b_in, *d_w_out, *d_b_out;
  float *mk;  // [64][8] dpre^T . te[ts], for wgrad_emb_kernel
};

// A workgroup finishes 32 floats of the result: thread (slice, f) adds 1/8 of the partials of float f in
// order, then the eight slice sums are added in slice order -- a fixed order overall.
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const per_task<fin_args> all) {
  const fin_args &a = all.t[blockIdx.y];
  __shared__ float part[8][32];
  const int slice = threadIdx.x >> 5, f = blockIdx.x * 32 + (threadIdx.x & 31);
  const int per = (a.n_partials + 7) / 8;
  float s = 0.f;
  if (f < kPartFloats)
#pragma unroll 8
    for (int g = slice * per; g < min((slice + 1) * per, a.n_partials); ++g) s += a.partials[(size_t)g * kPartFloats + f];
  part[slice][threadIdx.x & 31] = s;
  __syncthreads();
  if (slice != 0 || f >= kPartFloats) return;
  for (int k = 1; k < 8; ++k) s += part[k][threadIdx.x];
  if (f < kPartWin) a.d_w_out[f] = s;
  else if (f < kPartTab) {
    const int o = (f - kPartWin) >> 6, i = (f - kPartWin) & 63;
    a.d_w_in[o * 72 + i] = s;
  } else if (f < kPartBout) {
    const int o = (f - kPartTab) >> 4, c = (f - kPartTab) & 15;
    if (c < 8) a.d_w_in[o * 72 + 64 + c] = s;
    else a.mk[o * 8 + (c - 8)] = s;
  } else if (f < kPartBin) a.d_b_out[f - kPartBout] = s;
  else a.d_b_in[f - kPartBin] = s;
}

// dW_emb[e][c] = sum_o W_in[o][64+e] mk[o][c];  db_emb[e] = sum_o W_in[o][64+e] db_in[o]
struct emb_args {
  const float *w_in, *mk, *d_b_in;
  float *d_w_emb, *d_b_emb;
};

__global__ __launch_bounds__(128) void wgrad_emb_kernel(const per_task<emb_args> all) {
  const emb_args &a = all.t[blockIdx.y];
  const float *__restrict__ w_in = a.w_in, *__restrict__ mk = a.mk, *__restrict__ d_b_in = a.d_b_in;
  float *__restrict__ d_w_emb = a.d_w_emb, *__restrict__ d_b_emb = a.d_b_emb;
  const int t = threadIdx.x;
  if (t < 64) {
    const int e = t >> 3, c = t & 7;
    float s = 0.f;
    for (int o = 0; o < 64; ++o) s += w_in[o * 72 + 64 + e] * mk[o * 8 + c];
    d_w_emb[t] = s;
  } else if (t < 72) {
    const int e = t - 64;
    float s = 0.f;
    for (int o = 0; o < 64; ++o) s += w_in[o * 72 + 64 + e] * d_b_in[o];
    d_b_emb[e] = s;
  }
}

}  // namespace

extern "C" size_t dg_denoiser_wgrad_workspace(int32_t n_steps) {
  if (n_steps <= 0) return 0;
  return ((size_t)n_steps * kTabCols + 64 * 8 + (size_t)kGroups * kPartFloats) * sizeof(float);
}

extern "C" int dg_denoiser_wgrad_batched(const dg_wgrad_task *tasks, int32_t n_tasks, int32_t n_steps, int32_t d,
                                         void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_denoiser_wgrad: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!tasks || n_tasks <= 0 || n_tasks > kMaxTasks)
    return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad: n_tasks must be in [1,%d] (got %d)", kMaxTasks, n_tasks);
  if (n_steps <= 0) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad: bad sizes");
  per_task<tab_args> ta;
  per_task<wg_args> wa;
  per_task<fin_args> fa;
  per_task<emb_args> ea;
  for (int k = 0; k < n_tasks; ++k) {
    const dg_wgrad_task &t = tasks[k];
    const dg_denoiser *dn = t.dn;
    if (!dn || !dn->w_emb || !dn->b_emb || !dn->w_in || !dn->freqs_host || !t.ts || !t.dout_tot || !t.hid || !t.dpre || !t.xn ||
        !t.workspace || !t.d_w_emb || !t.d_b_emb || !t.d_w_in || !t.d_b_in || !t.d_w_out || !t.d_b_out)
      return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad: NULL argument (task %d)", k);
    if (t.n < 0) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad: bad sizes (task %d)", k);
    if (t.workspace_bytes < dg_denoiser_wgrad_workspace(n_steps))
      return dg_fail(DG_ERR_SIZE, "dg_denoiser_wgrad: workspace of %zu bytes, %zu needed", t.workspace_bytes,
                     dg_denoiser_wgrad_workspace(n_steps));
    if (((uintptr_t)t.dout_tot | (uintptr_t)t.hid | (uintptr_t)t.dpre | (uintptr_t)t.xn | (uintptr_t)t.workspace) & 15)
      return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad: tables and workspace must be 16-byte aligned");
    for (int j = 0; j < k; ++j)
      if (tasks[j].workspace == t.workspace) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad: tasks %d and %d share a workspace", j, k);
    float *tab = (float *)t.workspace;
    float *mk = tab + (size_t)n_steps * kTabCols;
    float *partials = mk + 64 * 8;
    tab_args &a = ta.t[k];
    a.w_emb = dn->w_emb; a.b_emb = dn->b_emb; a.n_steps = n_steps; a.tab = tab;
    for (int f = 0; f < 4; ++f) a.freqs[f] = dn->freqs_host[f];
    wg_args &w = wa.t[k];
    w.dout = t.dout_tot; w.hid = t.hid; w.dpre = t.dpre; w.xn = t.xn; w.ts = t.ts; w.tab = tab; w.partials = partials;
    w.n = t.n; w.n_steps = n_steps;
    w.rows_per_group = ((t.n + kGroups - 1) / kGroups + 15) / 16 * 16;
    if (w.rows_per_group == 0) w.rows_per_group = 16;
    fin_args &f = fa.t[k];
    f.partials = partials; f.n_partials = kGroups; f.d_w_in = t.d_w_in; f.d_b_in = t.d_b_in; f.d_w_out = t.d_w_out; f.d_b_out = t.d_b_out; f.mk = mk;
    emb_args &e = ea.t[k];
    e.w_in = dn->w_in; e.mk = mk; e.d_b_in = t.d_b_in; e.d_w_emb = t.d_w_emb; e.d_b_emb = t.d_b_emb;
  }
  for (int k = n_tasks; k < kMaxTasks; ++k) { ta.t[k] = ta.t[0]; wa.t[k] = wa.t[0]; fa.t[k] = fa.t[0]; ea.t[k] = ea.t[0]; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad_tab_kernel, dim3((n_steps + 255) / 256, n_tasks), dim3(256), 0, s, ta);
  hipLaunchKernelGGL(wgrad_partial_kernel, dim3(kGroups, n_tasks), dim3(512), 0, s, wa);
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3((kPartFloats + 31) / 32, n_tasks), dim3(256), 0, s, fa);
  hipLaunchKernelGGL(wgrad_emb_kernel, dim3(1, n_tasks), dim3(128), 0, s, ea);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_denoiser_wgrad(const dg_denoiser *dn, int32_t n_steps, const int64_t *ts, const float *dout_tot,
                                 const float *hid, const float *dpre, const float *xn, int64_t n, int32_t d,
                                 void *workspace, size_t workspace_bytes, float *d_w_emb, float *d_b_emb, float *d_w_in,
                                 float *d_b_in, float *d_w_out, float *d_b_out, void *stream) {
  dg_wgrad_task t;
  t.dn = dn; t.ts = ts; t.dout_tot = dout_tot; t.hid = hid; t.dpre = dpre; t.xn = xn; t.n = n;
  t.workspace = workspace; t.workspace_bytes = workspace_bytes;
  t.d_w_emb = d_w_emb; t.d_b_emb = d_b_emb; t.d_w_in = d_w_in; t.d_b_in = d_b_in; t.d_w_out = d_w_out; t.d_b_out = d_b_out;
  return dg_denoiser_wgrad_batched(&t, 1, n_steps, d, stream);
}

// -- the weight gradients from partial results the loss backward kernel formed itself -------------------
// (dg_loss_task.wg_partials: one partial per 128 table rows, nothing of the four [n][64] operand tables is written)
extern "C" int64_t dg_wgrad_partial_floats(void) { return kPartFloats; }

extern "C" int dg_denoiser_wgrad_tab(const dg_denoiser *const *dns, float *const *tabs, int32_t n_tabs, int32_t n_steps,
                                     void *stream) {
  if (!dns || !tabs || n_tabs <= 0 || n_tabs > kMaxTasks)
    return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_tab: n_tabs must be in [1,%d] (got %d)", kMaxTasks, n_tabs);
  if (n_steps <= 0) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_tab: n_steps must be positive");
  per_task<tab_args> ta;
  for (int k = 0; k < n_tabs; ++k) {
    const dg_denoiser *dn = dns[k];
    if (!dn || !dn->w_emb || !dn->b_emb || !dn->freqs_host || !tabs[k]) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_tab: NULL argument");
    tab_args &a = ta.t[k];
    a.w_emb = dn->w_emb; a.b_emb = dn->b_emb; a.n_steps = n_steps; a.tab = tabs[k];
    for (int f = 0; f < 4; ++f) a.freqs[f] = dn->freqs_host[f];
  }
  for (int k = n_tabs; k < kMaxTasks; ++k) ta.t[k] = ta.t[0];
  hipLaunchKernelGGL(wgrad_tab_kernel, dim3((n_steps + 255) / 256, n_tabs), dim3(256), 0, (hipStream_t)stream, ta);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_denoiser_wgrad_reduce(const dg_wgrad_reduce_task *tasks, int32_t n_tasks, void *stream) {
  if (!tasks || n_tasks <= 0 || n_tasks > kMaxTasks)
    return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_reduce: n_tasks must be in [1,%d] (got %d)", kMaxTasks, n_tasks);
  per_task<fin_args> fa;
  per_task<emb_args> ea;
  for (int k = 0; k < n_tasks; ++k) {
    const dg_wgrad_reduce_task &t = tasks[k];
    if (!t.dn || !t.dn->w_in || !t.partials || !t.mk || !t.d_w_emb || !t.d_b_emb || !t.d_w_in || !t.d_b_in || !t.d_w_out || !t.d_b_out)
      return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_reduce: NULL argument (task %d)", k);
    if (t.n_partials < 0) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_reduce: negative partial count (task %d)", k);
    fin_args &f = fa.t[k];
    f.partials = t.partials; f.n_partials = t.n_partials; f.d_w_in = t.d_w_in; f.d_b_in = t.d_b_in; f.d_w_out = t.d_w_out;
    f.d_b_out = t.d_b_out; f.mk = t.mk;
    emb_args &e = ea.t[k];
    e.w_in = t.dn->w_in; e.mk = t.mk; e.d_b_in = t.d_b_in; e.d_w_emb = t.d_w_emb; e.d_b_emb = t.d_b_emb;
  }
  for (int k = n_tasks; k < kMaxTasks; ++k) { fa.t[k] = fa.t[0]; ea.t[k] = ea.t[0]; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3((kPartFloats + 31) / 32, n_tasks), dim3(256), 0, s, fa);
  hipLaunchKernelGGL(wgrad_emb_kernel, dim3(1, n_tasks), dim3(128), 0, s, ea);
  DG_HIP(hipGetLastError());
  return 0;
}

// -- the same without a workspace argument, for callers that cannot allocate --------------------------
// The workspace of dg_denoiser_wgrad (timestep operand table, one partial result per workgroup of the
// partial launch) lives in the code object's own device memory, sized for kStaticMaxSteps timesteps.
// Every GPU the library runs on holds its own copy of the array, at its own address, until the process ends.
namespace {
constexpr int kStaticMaxSteps = 1024;
constexpr size_t kStaticFloats = (size_t)kStaticMaxSteps * kTabCols + 64 * 8 + (size_t)kGroups * kPartFloats;
static __device__ __align__(256) float scratch[kStaticFloats];   // 9.19 MiB

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
float *g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each device, looked up on first use

// The array's name is only a host-side handle: its device address on the current device comes from the runtime.
int scratch_on_current_device(float **out) {
  int dev = -1;
  DG_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_static: device %d not supported (max %d)", dev, kMaxDevices);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    float *p = nullptr;
    DG_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
    if (!p) return dg_fail(DG_ERR_ARG, "dg_denoiser_wgrad_static: no device address for the scratch array");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace

extern "C" int32_t dg_denoiser_wgrad_static_max_steps(void) { return kStaticMaxSteps; }

extern "C" int dg_denoiser_wgrad_static(const dg_denoiser *dn, int32_t n_steps, const int64_t *ts, const float *dout_tot,
                                        const float *hid, const float *dpre, const float *xn, int64_t n, int32_t d,
                                        float *d_w_emb, float *d_b_emb, float *d_w_in, float *d_b_in, float *d_w_out,
                                        float *d_b_out, void *stream) {
  if (n_steps > kStaticMaxSteps)
    return dg_fail(DG_ERR_SIZE, "dg_denoiser_wgrad_static: %d timesteps exceed the built-in scratch (at most %d); use "
                   "dg_denoiser_wgrad with a workspace", n_steps, kStaticMaxSteps);
  float *ws = nullptr;
  int rc = scratch_on_current_device(&ws);
  if (rc) return rc;
  // Nothing in the scratch has to start at zero: the table kernel writes all n_steps operand rows, every workgroup
  // of the partial launch writes its whole partial result (zeros when it has no rows), the finalize launch all of mk.
  return dg_denoiser_wgrad(dn, n_steps, ts, dout_tot, hid, dpre, xn, n, d, ws, kStaticFloats * sizeof(float), d_w_emb,
                           d_b_emb, d_w_in, d_b_in, d_w_out, d_b_out, stream);
}
