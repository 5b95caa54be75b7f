// Adam step for an embedding table on gfx950, with the gradient handed over as the list of tables
// whose sum it is.
//
// Replaces, for the [n_user + n_item, 64] embedding table, torch.optim.Adam.step() as the reference
// runs it (DiffGraph-Rec/main.py:117-119,127-128: Adam(lr, weight_decay=0), default betas / eps) and
// the last addition of the backward pass (the gradient of the table is the sum over the behaviour
// graphs of two tables each, DiffGraph-Rec/Model.py:57,74 feed the same table to every graph):
// one pass reads the gradient parts, the parameter and the two moments and writes the parameter and
// the moments back -- the summed gradient never goes to memory.  The step count lives on the
// device so that a captured HIP graph advances it on every replay.
#include "dg_internal.h"

#include <math.h>

#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kMaxParts = 16;

struct adam_args {
  const float *parts[kMaxParts];
  int32_t n_parts;
  float *param, *exp_avg, *exp_avg_sq;
  float *state;  // [4]: step count, 1 - beta1^t, sqrt(1 - beta2^t), spare
  float lr, beta1, beta2, eps, weight_decay;
  int64_t n4;
};

// t <- t + 1 and the two bias corrections (one thread; the table kernel reads them after it)
__global__ void adam_prepare_kernel(float *state, float beta1, float beta2) {
  const float t = state[0] + 1.0f;
  state[0] = t;
  state[1] = 1.0f - powf(beta1, t);
  state[2] = sqrtf(1.0f - powf(beta2, t));
}

__global__ __launch_bounds__(256) void adam_table_kernel(const adam_args a) {
  const float bc1 = a.state[1], bc2_sqrt = a.state[2];
  const float step_size = a.lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (int64_t)gridDim.x * 256) {
    f32x4 g = reinterpret_cast<const f32x4 *>(a.parts[0])[i];
    for (int k = 1; k < a.n_parts; ++k) g += reinterpret_cast<const f32x4 *>(a.parts[k])[i];  // left to right
    f32x4 p = reinterpret_cast<f32x4 *>(a.param)[i];
    f32x4 m = reinterpret_cast<f32x4 *>(a.exp_avg)[i];
    f32x4 v = reinterpret_cast<f32x4 *>(a.exp_avg_sq)[i];
    if (a.weight_decay != 0.f) g += p * a.weight_decay;
    m = m + (g - m) * (1.0f - a.beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + g * g * (1.0f - a.beta2);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    f32x4 d;
    d.x = sqrtf(v.x) / bc2_sqrt + a.eps; d.y = sqrtf(v.y) / bc2_sqrt + a.eps;
    d.z = sqrtf(v.z) / bc2_sqrt + a.eps; d.w = sqrtf(v.w) / bc2_sqrt + a.eps;
    p.x -= step_size * (m.x / d.x); p.y -= step_size * (m.y / d.y);
    p.z -= step_size * (m.z / d.z); p.w -= step_size * (m.w / d.w);
    reinterpret_cast<f32x4 *>(a.param)[i] = p;
    reinterpret_cast<f32x4 *>(a.exp_avg)[i] = m;
    reinterpret_cast<f32x4 *>(a.exp_avg_sq)[i] = v;
  }
}

// Up to kMaxTensors small parameter tensors in ONE launch (the denoiser / dense-layer weights: a few thousand
// elements each).  A workgroup takes kAdamPerBlock consecutive elements of one tensor (blk_begin: the tensors' first
// workgroups).  Every workgroup derives the step count t = state[0] + 1 and the bias corrections itself; the
// workgroup that finishes last (a counter in state[3]) writes t back and clears the counter, so that a recorded
// launch advances the count on every replay without a launch of its own.
constexpr int kMaxTensors = 32, kAdamPerBlock = 1024;

struct adam_multi_args {
  float *param[kMaxTensors];
  const float *grad[kMaxTensors];
  float *exp_avg[kMaxTensors], *exp_avg_sq[kMaxTensors];
  int32_t count[kMaxTensors], blk_begin[kMaxTensors];
  int32_t n_tensors;
  float *state;
  float lr, beta1, beta2, eps, weight_decay;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(const adam_multi_args a) {
  const float t = a.state[0] + 1.0f;
  const float bc1 = 1.0f - powf(a.beta1, t), bc2_sqrt = sqrtf(1.0f - powf(a.beta2, t));
  const float step_size = a.lr / bc1;
  int k = 0;
  for (int j = 1; j < a.n_tensors; ++j)
    if ((int)blockIdx.x >= a.blk_begin[j]) k = j;
  float *p = a.param[k], *m = a.exp_avg[k], *v = a.exp_avg_sq[k];
  const float *g = a.grad[k];
  const int base = ((int)blockIdx.x - a.blk_begin[k]) * kAdamPerBlock, end = min(a.count[k], base + kAdamPerBlock);
  for (int i = base + threadIdx.x; i < end; i += 256) {
    float gi = g[i];
    const float pi = p[i];
    if (a.weight_decay != 0.f) gi += pi * a.weight_decay;
    const float mi = m[i] + (gi - m[i]) * (1.0f - a.beta1);
    const float vi = v[i] * a.beta2 + gi * gi * (1.0f - a.beta2);
    p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + a.eps));
    m[i] = mi;
    v[i] = vi;
  }
  // all reads of state[0] are done once every workgroup has passed this point
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned *done = reinterpret_cast<unsigned *>(a.state + 3);
    __threadfence();
    if (atomicAdd(done, 1u) == gridDim.x - 1) {
      __atomic_store_n(done, 0u, __ATOMIC_RELAXED);
      a.state[0] = t;
    }
  }
}

}  // namespace

extern "C" int dg_adam_multi(float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                             float *const *exp_avg_sq_host, const int64_t *counts_host, int32_t n_tensors, float *state,
                             float lr, float beta1, float beta2, float eps, float weight_decay, void *stream) {
  if (!params_host || !grads_host || !exp_avg_host || !exp_avg_sq_host || !counts_host || !state)
    return dg_fail(DG_ERR_ARG, "dg_adam_multi: NULL argument");
  if (n_tensors < 1 || n_tensors > kMaxTensors) return dg_fail(DG_ERR_ARG, "dg_adam_multi: n_tensors must be in [1,%d]", kMaxTensors);
  adam_multi_args a;
  int64_t blocks = 0;
  for (int k = 0; k < kMaxTensors; ++k) {
    const int j = k < n_tensors ? k : 0;
    if (!params_host[j] || !grads_host[j] || !exp_avg_host[j] || !exp_avg_sq_host[j])
      return dg_fail(DG_ERR_ARG, "dg_adam_multi: tensor %d: NULL pointer", j);
    if (counts_host[j] < 0 || counts_host[j] > (int64_t)1 << 28) return dg_fail(DG_ERR_SIZE, "dg_adam_multi: tensor %d: bad element count", j);
    a.param[k] = params_host[j]; a.grad[k] = grads_host[j]; a.exp_avg[k] = exp_avg_host[j]; a.exp_avg_sq[k] = exp_avg_sq_host[j];
    a.count[k] = (int32_t)counts_host[j];
    a.blk_begin[k] = (int32_t)blocks;
    if (k < n_tensors) blocks += (counts_host[j] + kAdamPerBlock - 1) / kAdamPerBlock;
  }
  if (blocks == 0) blocks = 1;   // (nothing but the step count to advance)
  if (blocks > (1 << 20)) return dg_fail(DG_ERR_SIZE, "dg_adam_multi: these tensors are not small (use dg_adam_step for tables)");
  a.n_tensors = n_tensors;
  a.state = state; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_adam_prepare(float *state, float beta1, float beta2, void *stream) {
  if (!state) return dg_fail(DG_ERR_ARG, "dg_adam_prepare: NULL argument");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
  DG_HIP(hipGetLastError());
  return 0;
}

static int adam_step(float *param, const float *const *grad_parts_host, int32_t n_parts, float *exp_avg, float *exp_avg_sq,
                     float *state, int64_t n_elems, float lr, float beta1, float beta2, float eps, float weight_decay,
                     bool prepare, void *stream) {
  if (!param || !grad_parts_host || !exp_avg || !exp_avg_sq || !state) return dg_fail(DG_ERR_ARG, "dg_adam_step: NULL argument");
  if (n_parts < 1 || n_parts > kMaxParts) return dg_fail(DG_ERR_ARG, "dg_adam_step: n_parts must be in [1,%d]", kMaxParts);
  if (n_elems < 0 || n_elems % 4 != 0) return dg_fail(DG_ERR_ARG, "dg_adam_step: element count must be a multiple of 4");
  if (n_elems == 0) return 0;
  adam_args a;
  for (int k = 0; k < kMaxParts; ++k) {
    a.parts[k] = grad_parts_host[k < n_parts ? k : 0];
    if (!a.parts[k] || ((uintptr_t)a.parts[k] & 15)) return dg_fail(DG_ERR_ARG, "dg_adam_step: gradient part %d NULL or misaligned", k);
  }
  if (((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return dg_fail(DG_ERR_ARG, "dg_adam_step: tables must be 16-byte aligned");
  a.n_parts = n_parts; a.param = param; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.state = state;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.n4 = n_elems / 4;
  if (prepare) hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
  int64_t blocks = (a.n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_table_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_adam_step(float *param, const float *const *grad_parts_host, int32_t n_parts, float *exp_avg,
                            float *exp_avg_sq, float *state, int64_t n_elems, float lr, float beta1, float beta2, float eps,
                            float weight_decay, void *stream) {
  return adam_step(param, grad_parts_host, n_parts, exp_avg, exp_avg_sq, state, n_elems, lr, beta1, beta2, eps, weight_decay,
                   true, stream);
}

extern "C" int dg_adam_step_prepared(float *param, const float *const *grad_parts_host, int32_t n_parts, float *exp_avg,
                                     float *exp_avg_sq, float *state, int64_t n_elems, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, void *stream) {
  return adam_step(param, grad_parts_host, n_parts, exp_avg, exp_avg_sq, state, n_elems, lr, beta1, beta2, eps, weight_decay,
                   false, stream);
}
