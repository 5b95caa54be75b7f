// All-rank evaluation for gfx950: scores of every test user against every item, training
// interactions masked out, top-k per user, Recall@k / NDCG@k.
//
// Replaces the evaluation loop of Coach.testEpoch (DiffGraph-Rec/main.py:153-160): per batch of
// 1024 users a dense [1024, n_item] train-mask built on the host row by row
// (AllRankTestData.__getitem__, DiffGraph-Rec/DataHandler.py:268-272), t.mm(usr, item^T) *
// (1 - mask) - mask * 1e8, t.topk, a device->host copy and the Python loops of Coach.calcRes
// (main.py:168-187).  Here the score matrix never exists in memory: a wavefront owns 32 users
// (their embeddings stay in registers as the B operand of v_mfma_f32_32x32x2_f32), item tiles of
// 32 rows are staged through LDS as the A operand, and each lane keeps the k best scores of its
// user over the items it sees (a sorted list in registers, updated by a compare-and-shift network).
// The user's sorted training row is walked in step with the item tiles, so masking costs a register
// compare per tile; a score is compared with the worst entry of the lane's list, and of the list
// the other half-wavefront keeps for the same user, before anything else happens to it.
//
// The scores themselves come off the bf16 matrix instruction at 6/16 of the exact-fp32 instruction's cycles: an fp32
// value is the sum of three bf16 numbers exactly (its 24 significand bits in three runs of eight), products of bf16
// numbers are exact in the fp32 accumulator, and of the nine cross products the six that matter leave out less than
// one fp32 rounding of the score (as in the reverse loop, diffusion.hip).  DG_EVAL_FP32=1 keeps the fp32 instruction.
#include "dg_internal.h"

#include <stdlib.h>

#include <math.h>

#include <algorithm>
#include <climits>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kMaxK = 32;
constexpr int kUsersPerBlock = 128;      // 4 wavefronts x 32 users
constexpr int kTile = 32;                // items per tile
constexpr int kPitch = 68;               // floats per staged item row (64 + 4: conflict-free ds_read_b128)
constexpr float kMasked = -1e8f;         // main.py:157
constexpr int kBuf = 16;                 // parked candidates per lane between two list merges (two half tiles)
constexpr int kPlanePitch = 9;           // u32x4 per staged item row of one bf16 plane (8 + 1: conflict-free ds_read_b128)
constexpr int kPlaneTile = kTile * kPlanePitch;
constexpr int kMaxLocalTiles = 65536 / kTile;   // the bf16 form parks an item as a 16-bit offset into its workgroup's range

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// eight fp32 -> three bf16 planes with hi + mid + lo == x up to the bits below the 24th (the remainders are exact)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 &hi, u32x4 &mid, u32x4 &lo) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = x[2 * k], b = x[2 * k + 1];
    const unsigned h = pack_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = pack_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    hi[k] = h; mid[k] = m; lo[k] = pack_bf16(sa, sb);
  }
}

struct topk_args {
  const float *usr_emb, *item_emb;
  const int64_t *test_users;
  const int32_t *trn_rowptr, *trn_col;
  float *cand_score;   // [n_test][2 * n_split][k]
  int32_t *cand_idx;
  int32_t n_test, n_item, k, n_split, tiles_per_split;
};

// First position in the ascending list col[b, e) whose entry is >= item.
__device__ __forceinline__ int lower_bound(const int32_t *__restrict__ col, int b, int e, int item) {
  while (b < e) {
    const int m = (b + e) >> 1;
    if (col[m] < item) b = m + 1; else e = m;
  }
  return b;
}

// KPAD: list slots per lane (k rounded up to a multiple of 4).  The list lives in registers, sorted by
// falling score; slots [0, KPAD - k) hold +inf and never move, so the lane's worst entry is always
// the last slot.  A new score enters through a compare-and-shift over all slots (no indexed access).
// B3: the scores on the bf16 instruction.  The staging threads cut the item rows into three bf16 planes once per
// tile (a thread's eight floats are exactly one lane's operand of one 16-feature chunk: features 32 h + 8 c .. + 7 on
// lane half h, for the items as for the users -- the matrix instruction only needs both operands to agree on which
// k is which), and parked items are 16-bit offsets into the workgroup's item range, so that three workgroups still
// share a CU's LDS.
template <int KPAD, bool B3>
__global__ __launch_bounds__(256, 3) void score_topk_kernel(const topk_args a) {
  __shared__ u32x4 tile_raw[B3 ? 2 * 3 * kPlaneTile : 2 * kTile * kPitch / 4];
  __shared__ float bs[kBuf * 256];  // parked scores [slot][thread] ...
  typedef typename std::conditional<B3, uint16_t, int>::type park_t;
  __shared__ park_t bi[kBuf * 256];  // ... and their item ids (B3: offsets from the first item of the workgroup's range)
  float *tile = reinterpret_cast<float *>(tile_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const int k = a.k;
  const int t_idx = blockIdx.x * kUsersPerBlock + wave * 32 + j;
  const bool valid = t_idx < a.n_test;
  const int64_t user = valid ? a.test_users[t_idx] : 0;

  // B operand: features 32h .. 32h+31 of this lane's user
  float ub[32];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid) v = *reinterpret_cast<const f32x4 *>(a.usr_emb + user * DG_D + 32 * h + 4 * c);
    ub[4 * c] = v.x; ub[4 * c + 1] = v.y; ub[4 * c + 2] = v.z; ub[4 * c + 3] = v.w;
  }
  u32x4 uh[4], um[4], ul[4];   // B3: the same features as bf16 planes, chunk c = features 32 h + 8 c .. + 7
  if (B3) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x[8] = {ub[8 * c], ub[8 * c + 1], ub[8 * c + 2], ub[8 * c + 3], ub[8 * c + 4], ub[8 * c + 5], ub[8 * c + 6], ub[8 * c + 7]};
      split8(x, uh[c], um[c], ul[c]);
    }
  }
  float ls[KPAD];
  int li[KPAD];
#pragma unroll
  for (int p = 0; p < KPAD; ++p) { ls[p] = p < KPAD - k ? INFINITY : -INFINITY; li[p] = -1; }

  // Scores that beat the list's last entry are parked in a small per-lane buffer (LDS, in the order seen)
  // and merged into the lists by ALL lanes together whenever some lane's buffer is full: the
  // compare-and-shift network then runs for 64 lanes at once instead of once per lucky lane.
  // A score can only be among the user's k best if it beats the lane's worst entry, is not below the worst entry of
  // the list the other half-wavefront keeps for the same user over the other items of each tile, and is not below
  // the smaller of the two lists' ceil(k/2)-th entries (the two lists together hold k entries at least that good).
  // The lists change only in compact(), for all lanes at once, so the largest of the three bounds is worked out
  // there and a tile costs ONE compare per score.  (VALU and MFMA instructions do not overlap on this chip -- the
  // co-execution counter stays at zero -- so every VALU instruction per tile adds to the kernel's time.)
  float bound = valid ? -INFINITY : INFINITY;
  const int mid_slot = KPAD - k + (k + 1) / 2 - 1;
  int nb = 0;
  const int tile0 = blockIdx.y * a.tiles_per_split;
  const int id0 = B3 ? tile0 * kTile : 0;
  auto compact = [&]() {
#pragma nounroll
    for (int b = 0; b < kBuf; ++b) {
      const bool have = b < nb;
      const float s = have ? bs[b * 256 + tid] : -INFINITY;
      const int id = have ? id0 + (int)bi[b * 256 + tid] : -1;
      if (!__any(have)) break;
      if (s > ls[KPAD - 1]) {
        // slot p takes its upper neighbour when s beats that one too, s when it beats only slot p's entry
        bool above = s > ls[KPAD - 2];
#pragma unroll
        for (int p = KPAD - 1; p >= 1; --p) {
          const bool here = p == KPAD - 1 ? true : s > ls[p];
          ls[p] = above ? ls[p - 1] : (here ? s : ls[p]);
          li[p] = above ? li[p - 1] : (here ? id : li[p]);
          if (p >= 2) above = s > ls[p - 2];
        }
        if (s > ls[0]) { ls[0] = s; li[0] = id; }
      }
    }
    nb = 0;
    float mid = ls[KPAD - 1];
#pragma unroll
    for (int p = 0; p < KPAD - 1; ++p) mid = p == mid_slot ? ls[p] : mid;
    const float tau = ls[KPAD - 1];
    const float other = fmaxf(__shfl_xor(tau, 32), fminf(mid, __shfl_xor(mid, 32)));
    bound = valid ? fmaxf(tau, other) : INFINITY;
  };

  const int n_tiles_all = (a.n_item + kTile - 1) / kTile;
  const int tile1 = min(tile0 + a.tiles_per_split, n_tiles_all);
  // the user's training items, walked in step with the item tiles (both ascend)
  int cur = 0, row_e = 0, nxt = INT_MAX;
  if (valid) {
    row_e = a.trn_rowptr[user + 1];
    cur = lower_bound(a.trn_col, a.trn_rowptr[user], row_e, tile0 * kTile);
    if (cur < row_e) nxt = a.trn_col[cur];
  }
  // staging: thread -> (row = tid >> 3, two float4 at columns 8 (tid & 7) and + 4)
  auto fetch = [&](int t, f32x4 &v0, f32x4 &v1) {
    const int item = t * kTile + (tid >> 3);
    v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (item < a.n_item) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(a.item_emb + (size_t)item * DG_D + 8 * (tid & 7));
      v0 = src[0]; v1 = src[1];
    }
  };
  auto stage = [&](int buf, f32x4 v0, f32x4 v1) {
    if (B3) {
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      u32x4 hi, mid, lo;
      split8(x, hi, mid, lo);
      u32x4 *dst = tile_raw + buf * 3 * kPlaneTile + (tid >> 3) * kPlanePitch + (tid & 7);
      dst[0] = hi; dst[kPlaneTile] = mid; dst[2 * kPlaneTile] = lo;
      return;
    }
    float *dst = tile + (buf * kTile + (tid >> 3)) * kPitch + 8 * (tid & 7);
    *reinterpret_cast<f32x4 *>(dst) = v0;
    *reinterpret_cast<f32x4 *>(dst + 4) = v1;
  };
  f32x4 p0, p1;
  if (tile0 < tile1) { fetch(tile0, p0, p1); stage(0, p0, p1); }
  __syncthreads();
  for (int t = tile0; t < tile1; ++t) {
    const int buf = (t - tile0) & 1;
    if (t + 1 < tile1) fetch(t + 1, p0, p1);
    // A operand: features 32h .. 32h+31 of item (t*32 + j)
    const float *arow = tile + (buf * kTile + j) * kPitch + 32 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (B3) {
      const u32x4 *ap = tile_raw + buf * 3 * kPlaneTile + j * kPlanePitch + 4 * h;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, ap[c]), am = __builtin_bit_cast(bf16x8, ap[kPlaneTile + c]),
                     al = __builtin_bit_cast(bf16x8, ap[2 * kPlaneTile + c]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, uh[c]), bm = __builtin_bit_cast(bf16x8, um[c]),
                     bl = __builtin_bit_cast(bf16x8, ul[c]);
        // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(arow + 4 * c);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, ub[4 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, ub[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, ub[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, ub[4 * c + 3], acc, 0, 0, 0);
      }
    }
    // acc[r]: user j, item t*32 + 8 (r >> 2) + 4 h + (r & 3)
    const int item0 = t * kTile;
    while (nxt < item0 + kTile) {  // training items inside this tile: s * (1 - 1) - 1e8 (rare)
      const int local = nxt - item0;
      if (((local >> 2) & 1) == h) {
        const int rr = ((local >> 3) << 2) | (local & 3);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = r == rr ? kMasked : acc[r];
      }
      ++cur;
      nxt = cur < row_e ? a.trn_col[cur] : INT_MAX;
    }
    if (item0 + kTile > a.n_item) {  // the last tile's rows past the table
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (item0 + 8 * (r >> 2) + 4 * h + (r & 3) >= a.n_item) acc[r] = -INFINITY;
    }
    // Park what may be among the k best (>= : a score equal to the other list's bound may still win there on the
    // item id; one equal to the lane's own worst entry is dropped again by compact()).  Two halves of eight scores,
    // so that room for a whole half is made at most once per half.
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (__any(nb > kBuf - 8)) compact();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int rr = 8 * half + r;
        const bool cand = acc[rr] >= bound;
        if (__any(cand)) {
          if (cand) {
            bs[nb * 256 + tid] = acc[rr];
            bi[nb * 256 + tid] = (park_t)(item0 - id0 + 8 * (rr >> 2) + 4 * h + (rr & 3));
            ++nb;
          }
        }
      }
    }
    if (t + 1 < tile1) stage(buf ^ 1, p0, p1);
    __syncthreads();
  }
  compact();
  if (!valid) return;
  const size_t o = ((size_t)t_idx * 2 * a.n_split + 2 * blockIdx.y + h) * k;
#pragma unroll
  for (int p = 0; p < KPAD; ++p)
    if (p >= KPAD - k) { a.cand_score[o + p - (KPAD - k)] = ls[p]; a.cand_idx[o + p - (KPAD - k)] = li[p]; }
}

// One wavefront per test user: the k best of its n_cand candidates, by falling score, equal
// scores by rising item id (k rounds of a wavefront-wide arg-max).
template <int kPer>  // candidates per lane: n_cand <= 64 * kPer
__global__ __launch_bounds__(256) void merge_topk_kernel(const float *__restrict__ cand_score,
                                                         const int32_t *__restrict__ cand_idx, int n_test, int n_cand,
                                                         int k, int32_t *__restrict__ top_idx,
                                                         float *__restrict__ top_score) {
  const int lane = threadIdx.x & 63, u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= n_test) return;
  float sc[kPer];
  int ix[kPer];
#pragma unroll
  for (int c = 0; c < kPer; ++c) {
    const int q = c * 64 + lane;
    sc[c] = -INFINITY; ix[c] = INT_MAX;
    if (q < n_cand) {
      const int id = cand_idx[(size_t)u * n_cand + q];
      if (id >= 0) { sc[c] = cand_score[(size_t)u * n_cand + q]; ix[c] = id; }
    }
  }
  for (int round = 0; round < k; ++round) {
    float bs = sc[0];
    int bi = ix[0];
#pragma unroll
    for (int c = 1; c < kPer; ++c)
      if (sc[c] > bs || (sc[c] == bs && ix[c] < bi)) { bs = sc[c]; bi = ix[c]; }
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(bs, off);
      const int oi = __shfl_xor(bi, off);
      if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
    }
    if (lane == 0) {
      top_idx[(size_t)u * k + round] = bi == INT_MAX ? -1 : bi;
      top_score[(size_t)u * k + round] = bs;
    }
#pragma unroll
    for (int c = 0; c < kPer; ++c)
      if (ix[c] == bi) { sc[c] = -INFINITY; ix[c] = INT_MAX; }  // an item id is a candidate of one lane only
  }
}

// Coach.calcRes for one user per thread (main.py:172-186): walks the user's test items in stored
// order; inv_log2[p] = 1 / log2(p + 2) and max_dcg[t] = sum_{p < t} inv_log2[p] come from the host
// in fp64, evaluated there exactly as the reference evaluates them.
__global__ __launch_bounds__(256) void eval_metrics_kernel(const int32_t *__restrict__ top_idx, int n_test, int k,
                                                           const int64_t *__restrict__ test_users,
                                                           const int32_t *__restrict__ tst_rowptr,
                                                           const int32_t *__restrict__ tst_col,
                                                           const double *__restrict__ inv_log2,
                                                           const double *__restrict__ max_dcg, double *__restrict__ recall,
                                                           double *__restrict__ ndcg) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= n_test) return;
  const int64_t user = test_users[u];
  const int b = tst_rowptr[user], e = tst_rowptr[user + 1], tst_num = e - b;
  int hits = 0;
  double dcg = 0.0;
  for (int p = b; p < e; ++p) {
    const int val = tst_col[p];
    for (int q = 0; q < k; ++q)
      if (top_idx[(size_t)u * k + q] == val) {  // list.index(val): the first position
        ++hits;
        dcg += inv_log2[q];
        break;
      }
  }
  recall[u] = tst_num > 0 ? (double)hits / (double)tst_num : 0.0;
  ndcg[u] = tst_num > 0 ? dcg / max_dcg[min(tst_num, k)] : 0.0;
}

// Item-range splits per user tile: as many as keep all workgroups resident at once -- three per CU at the scoring
// kernel's register count, 768 on the chip (one more split and a second, mostly empty round follows: measured 2.01 ms
// with 750 workgroups, 2.32 ms with 1000) -- and at most 256 / k (the merge kernel takes 2 x splits x k <= 512 candidates per user).
int pick_split(int32_t n_test, int32_t n_item, int32_t k) {
  const int tiles = (n_test + kUsersPerBlock - 1) / kUsersPerBlock;
  const int n_tiles = (n_item + kTile - 1) / kTile;
  return std::max(1, std::min(std::min(768 / tiles, 256 / std::max(k, 1)), n_tiles));
}

}  // namespace

extern "C" size_t dg_eval_topk_workspace(int32_t n_test, int32_t n_item, int32_t k) {
  if (n_test <= 0 || n_item <= 0 || k <= 0) return 0;
  return (size_t)n_test * 2 * pick_split(n_test, n_item, k) * k * (sizeof(float) + sizeof(int32_t));
}

extern "C" int dg_eval_topk(const float *usr_emb, const float *item_emb, int32_t n_user, int32_t n_item,
                            const int64_t *test_users, int32_t n_test, const int32_t *trn_rowptr, const int32_t *trn_col,
                            int32_t k, int32_t d, void *workspace, size_t workspace_bytes, int32_t *top_idx,
                            float *top_score, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_eval_topk: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!usr_emb || !item_emb || !test_users || !trn_rowptr || !trn_col || !workspace || !top_idx || !top_score)
    return dg_fail(DG_ERR_ARG, "dg_eval_topk: NULL argument");
  if (n_user <= 0 || n_item <= 0 || n_test < 0) return dg_fail(DG_ERR_ARG, "dg_eval_topk: bad sizes");
  if (k <= 0 || k > kMaxK) return dg_fail(DG_ERR_SIZE, "dg_eval_topk: k must be in [1,%d] (got %d)", kMaxK, k);
  if (n_test == 0) return 0;
  if (((uintptr_t)usr_emb | (uintptr_t)item_emb) & 15) return dg_fail(DG_ERR_ARG, "dg_eval_topk: tables must be 16-byte aligned");
  const size_t need = dg_eval_topk_workspace(n_test, n_item, k);
  if (workspace_bytes < need) return dg_fail(DG_ERR_SIZE, "dg_eval_topk: workspace of %zu bytes, %zu needed", workspace_bytes, need);
  const int tiles = (n_test + kUsersPerBlock - 1) / kUsersPerBlock;
  const int n_split = pick_split(n_test, n_item, k);
  const int n_tiles = (n_item + kTile - 1) / kTile;
  topk_args a;
  a.usr_emb = usr_emb; a.item_emb = item_emb; a.test_users = test_users; a.trn_rowptr = trn_rowptr; a.trn_col = trn_col;
  a.cand_score = (float *)workspace;
  a.cand_idx = (int32_t *)(a.cand_score + (size_t)n_test * 2 * n_split * k);
  a.n_test = n_test; a.n_item = n_item; a.k = k; a.n_split = n_split;
  a.tiles_per_split = (n_tiles + n_split - 1) / n_split;
  static void (*const kerns[2][8])(const topk_args) = {
      {score_topk_kernel<4, false>, score_topk_kernel<8, false>, score_topk_kernel<12, false>, score_topk_kernel<16, false>,
       score_topk_kernel<20, false>, score_topk_kernel<24, false>, score_topk_kernel<28, false>, score_topk_kernel<32, false>},
      {score_topk_kernel<4, true>, score_topk_kernel<8, true>, score_topk_kernel<12, true>, score_topk_kernel<16, true>,
       score_topk_kernel<20, true>, nullptr, nullptr, nullptr}};
  const char *fp32_env = getenv("DG_EVAL_FP32");   // (read at every call: a test switches it)
  // (lists of more than 20 entries leave the bf16 form short of registers at three workgroups per CU)
  const bool b3 = k <= 20 && a.tiles_per_split <= kMaxLocalTiles && !(fp32_env && atoi(fp32_env) > 0);
  void (*kern)(const topk_args) = kerns[b3 ? 1 : 0][(k + 3) / 4 - 1];  // list slots: k rounded up to a multiple of 4
  hipLaunchKernelGGL(kern, dim3(tiles, n_split), dim3(256), 0, (hipStream_t)stream, a);
  const int n_cand = 2 * n_split * k;
  auto merge = n_cand <= 128 ? merge_topk_kernel<2> : n_cand <= 256 ? merge_topk_kernel<4> : merge_topk_kernel<8>;  // <= 512
  hipLaunchKernelGGL(merge, dim3((n_test + 3) / 4), dim3(256), 0, (hipStream_t)stream, a.cand_score, a.cand_idx, n_test, n_cand, k,
                     top_idx, top_score);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_eval_metrics(const int32_t *top_idx, int32_t n_test, int32_t k, const int64_t *test_users,
                               const int32_t *tst_rowptr, const int32_t *tst_col, const double *inv_log2,
                               const double *max_dcg, double *recall, double *ndcg, void *stream) {
  if (!top_idx || !test_users || !tst_rowptr || !tst_col || !inv_log2 || !max_dcg || !recall || !ndcg)
    return dg_fail(DG_ERR_ARG, "dg_eval_metrics: NULL argument");
  if (k <= 0 || n_test < 0) return dg_fail(DG_ERR_ARG, "dg_eval_metrics: bad sizes");
  if (n_test == 0) return 0;
  hipLaunchKernelGGL(eval_metrics_kernel, dim3((n_test + 255) / 256), dim3(256), 0, (hipStream_t)stream, top_idx, n_test, k,
                     test_users, tst_rowptr, tst_col, inv_log2, max_dcg, recall, ndcg);
  DG_HIP(hipGetLastError());
  return 0;
}
