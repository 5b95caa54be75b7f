// Normalised-adjacency propagation for gfx950: CSR row gather with a fused
// LeakyReLU / row-L2-normalise / layer-sum epilogue.
//
// Replaces, per call, the chain  cat -> out_degrees/pow -> scale -> update_all(copy_u,sum)
// -> in_degrees/pow -> scale -> LeakyReLU  of DGLLayer.forward (DiffGraph-Rec/Model.py:142-174)
// and UUGCNLayer.forward (DiffGraph_NC/Model.py:184-214), plus F.normalize and the running
// layer sum of HGDM.forward (DiffGraph-Rec/Model.py:67-70,81-84).
//
// Mapping.  An embedding row is 64 fp32 = 256 B.  A row is owned by a GROUP of 16 lanes,
// each lane holding one float4 of it, so one wavefront load instruction gathers four whole
// neighbour rows (1 KiB, four fully used 256-B segments).  A group reads 16 neighbour ids
// with one coalesced load, then hands them round with DPP row broadcasts (no LDS).  Rows
// are binned by in-degree on the host:
//   group rows (deg <= group_max): one group per row, four rows per wavefront, in six degree
//                                  classes (like lengths share a wavefront), node order inside;
//   cut rows   (deg >  group_max): the neighbour list is cut by source-node range (8 ranges, one
//                                  per XCD, so that each XCD's L2 serves one eighth of the source
//                                  table) and into pieces of <= seg_len neighbours; a piece is
//                                  summed by a wavefront (four groups take every fourth chunk of
//                                  16 neighbours) or, when short, by one group, into a slab row;
//                                  a second, small launch adds each cut row's slab rows in order.
// A group row's first 13 neighbour ids, its row id and degree sit in one 64-B head record
// (graph.hip), and every edge carries the gathered node's degree norm next to its id, so
// the dependent-load chain of a short row is: head -> neighbour rows -> store.
// Every sum has a fixed order, so results are bitwise reproducible run to run.
// Neighbour gathers go through a buffer descriptor: the id -1 of a padding slot turns
// into an out-of-range offset, which the hardware answers with zeros and no memory access.
#include "dg_gather.h"

#include <algorithm>
#include <climits>
#include <cmath>

namespace {

// What the epilogue needs from memory; asked for before the neighbour rows so that it
// arrives while they load.
struct row_ctx {
  float sc;    // lane 0 of the group: norm of the row node; lane 1: its norm as a source; lane 2: saved ||y||
  f32x4 a_in;  // forward: running layer sum (or e0 under DG_ACC_REBUILD); backward: saved activation
  f32x4 g_in;  // backward: gradient wrt the layer sum; forward under DG_ACC_REBUILD: the row's own x
};

// Loads only -- no arithmetic on the loaded values here, so that the requests stay in flight
// while the caller gathers the neighbour rows (or works on the previous row).
__device__ __forceinline__ row_ctx load_row_ctx(const dg_task_dev &T, int row, int l) {
  // which tables feed a_in / g_in / rn is a per-task (wavefront-uniform) choice: pick the pointers
  // with scalar selects so that each value is ONE load from a loop-invariant base
  const bool bwd = (T.flags & DG_EPI_BWD) != 0;
  const float *a_src = bwd ? T.e_saved : (T.acc_out ? T.acc_in : nullptr);
  const float *g_src = bwd ? (T.e_saved ? T.d_sum : nullptr) : ((T.flags & DG_ACC_REBUILD) ? T.x : nullptr);
  const float *n_src = (bwd && (T.flags & DG_BWD_NORM_FEED)) ? T.row_norm : nullptr;
  row_ctx c;
  const size_t o = (size_t)row * (DG_D / 4) + l;
  const float *s_src = l == 0 ? T.dinv_row : (l == 1 ? T.dinv_out : (l == 2 ? n_src : nullptr));
  c.sc = 0.f;
  if (s_src) c.sc = s_src[row];  // one load instruction for the three per-row scalars
  c.a_in = f32x4{0.f, 0.f, 0.f, 0.f};
  c.g_in = c.a_in;
  if (a_src) c.a_in = load_stream(a_src, o);
  if (g_src) c.g_in = load_stream(g_src, o);
  return c;
}

// Backward of LeakyReLU + F.normalize + layer sum for one row (see dg_gcn_layer_bwd).
// (`next_in`, when given, receives what was stored for the next gather to read.)
__device__ __forceinline__ void epilogue_bwd(const dg_task_dev &T, int row, const row_ctx &rc, f32x4 acc, int l,
                                             f32x4 *next_in = nullptr) {
  const size_t o = (size_t)row * (DG_D / 4) + l;
  const float di = bcast16f<0>(rc.sc), dsrc = bcast16f<1>(rc.sc), rn = bcast16f<2>(rc.sc);
  f32x4 G = acc * di;
  if (!T.e_saved) {
    store_stream(T.e_out, o, G);
    if (next_in) *next_in = G;
    return;
  }
  const f32x4 e = rc.a_in, g = rc.g_in;
  const float c = (T.flags & DG_BWD_SAVED_PRESCALED) ? di : 1.0f;  // e = c * E_l
  const float re = sqrtf(sum16(e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w));
  const bool feed = (T.flags & DG_BWD_NORM_FEED) != 0;  // e is the normalised activation; its length was saved
  const float r = feed ? rn : re / c;                // ||y_l||
  const f32x4 gt = feed ? g + G : g;                    // what reaches the normalisation from above
  f32x4 dE;
  if (r > 1e-12f) {  // y/||y||: d = (g - n (n.g)) / ||y||
    const float inv_re = 1.0f / re, inv_r = 1.0f / r;
    const f32x4 n = e * inv_re;
    const float ng = sum16(n.x * gt.x + n.y * gt.y + n.z * gt.z + n.w * gt.w);
    dE = (gt - n * ng) * inv_r;
  } else {           // clamp_min(norm, eps) is active: y / eps
    dE = gt * 1e12f;
  }
  if (!feed) dE += G;
  const float slope = T.slope;
  f32x4 dz;
  dz.x = e.x > 0.f ? dE.x : dE.x * slope;
  dz.y = e.y > 0.f ? dE.y : dE.y * slope;
  dz.z = e.z > 0.f ? dE.z : dE.z * slope;
  dz.w = e.w > 0.f ? dE.w : dE.w * slope;
  dz *= dsrc;
  store_stream(T.e_out, o, dz);
  if (next_in) *next_in = dz;
}

// rst*norm, LeakyReLU, F.normalize, layer sum, next layer's input (Model.py:165-172,67-70).
__device__ __forceinline__ void epilogue(const dg_task_dev &T, int row, const row_ctx &rc, f32x4 acc, int l,
                                         f32x4 *next_in = nullptr) {
  if (T.flags & DG_EPI_BWD) return epilogue_bwd(T, row, rc, acc, l, next_in);
  if (T.partial_out != nullptr && row >= T.partial_from) {  // another rank holds the rest of this row
    reinterpret_cast<f32x4 *>(T.partial_out)[(size_t)(row - T.partial_from) * (DG_D / 4) + l] = acc;
    return;
  }
  const float slope = T.slope;
  const float di = bcast16f<0>(rc.sc), dsrc = bcast16f<1>(rc.sc);
  f32x4 y = acc * di;
  y.x = y.x > 0.f ? y.x : y.x * slope;
  y.y = y.y > 0.f ? y.y : y.y * slope;
  y.z = y.z > 0.f ? y.z : y.z * slope;
  y.w = y.w > 0.f ? y.w : y.w * slope;
  const size_t o = (size_t)row * (DG_D / 4) + l;
  const bool want_norm = (T.acc_out != nullptr) || (T.flags & DG_OUT_NORM);
  f32x4 yn = y;
  if (want_norm) {
    float ss = sum16(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w);
    yn = y * inv_norm(ss);
    if (T.row_norm && l == 0) T.row_norm[row] = sqrtf(ss);
  }
  if (T.acc_out) {
    f32x4 a = rc.a_in;
    if (T.flags & DG_ACC_REBUILD) {  // a = e0[i] + normalize(E_1[i]) from the stored p = dinv_src[i] * E_1[i]:
      const f32x4 p = rc.g_in;       // ||E_1|| = ||p|| / dsrc, so normalize(E_1) = p / max(||p||, 1e-12 * dsrc)
      const float ps = sum16(p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w);
      a += p * fminf(__builtin_amdgcn_rsqf(ps), 1e12f * __builtin_amdgcn_rcpf(dsrc));
    }
    store_stream(T.acc_out, o, a + yn);
  }
  if (T.e_out) {
    f32x4 e = (T.flags & DG_OUT_NORM) ? yn : y;
    if (T.flags & DG_OUT_PRESCALE) e *= dsrc;
    store_stream(T.e_out, o, e);
    if (next_in) *next_in = e;
  }
}

// Blocks of one task, longest work first: long pieces of cut rows (one wavefront each, partial
// sums into the slab), short pieces (one group each), group rows.
template <bool SCALE>
__device__ __forceinline__ void layer_block(const dg_task_dev &T, int lb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, l = lane & 15;
  const int nb_seg = T.n_seg_slots >> 2, nb_gseg = T.n_gseg_slots >> 4;
  const __amdgpu_buffer_rsrc_t xr = table_rsrc(T.x, T.n);
  if (lb < nb_seg) {
    const int4 d = reinterpret_cast<const int4 *>(T.seg_desc)[lb * 4 + wave];  // begin, end, slab row
    if (d.z >= 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      gather_rows<true, SCALE>(xr, T.colidx, T.colscale, d.x + 16 * q, d.y, 64, l, acc);
      acc = add_groups(acc);
      if (q == 0) reinterpret_cast<f32x4 *>(T.slab)[(size_t)d.z * (DG_D / 4) + l] = acc;
    }
  } else if (lb < nb_seg + nb_gseg) {
    const int4 d = reinterpret_cast<const int4 *>(T.gseg_desc)[(lb - nb_seg) * 16 + wave * 4 + q];
    if (d.z >= 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      gather_rows<true, SCALE>(xr, T.colidx, T.colscale, d.x, d.y, 16, l, acc);
      reinterpret_cast<f32x4 *>(T.slab)[(size_t)d.z * (DG_D / 4) + l] = acc;
    }
  } else {
    // group rows: four records per wavefront; the record names its row
    const int k = (lb - nb_seg - nb_gseg) * 16 + wave * 4 + q;
    if (k < T.n_group) {
      const int hv = T.head[(size_t)k * 16 + l];
      const float hs = SCALE ? T.head_scale[(size_t)k * 16 + l] : 0.f;
      const int deg = bcast16<14>(hv), tail = bcast16<15>(hv);
      const int row = bcast16<13>(hv);
      const row_ctx rc = load_row_ctx(T, row, l);
      if (deg >= 0) {
        const int c = l < 13 ? hv : -1;
        const unsigned lane_off = (unsigned)l * 16u;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        gather4<0, SCALE>(xr, c, hs, lane_off, acc);
        if (deg > 4) gather4<4, SCALE>(xr, c, hs, lane_off, acc);
        if (deg > 8) gather4<8, SCALE>(xr, c, hs, lane_off, acc);
        if (deg > 12) gather4<12, SCALE>(xr, c, hs, lane_off, acc);
        if (deg > 13) gather_rows<true, SCALE>(xr, T.colidx, T.colscale, tail, tail + deg - 13, 16, l, acc);
        epilogue(T, row, rc, acc, l);
      }
    }
  }
}

__device__ __forceinline__ const dg_task_dev &find_task(const dg_task_batch &batch, int b, bool hub) {
  int ti = 0;
  for (int k = 1; k < batch.n_tasks; ++k)
    if (b >= (hub ? batch.t[k].hub_blk_begin : batch.t[k].blk_begin)) ti = k;
  return batch.t[__builtin_amdgcn_readfirstlane(ti)];
}

__global__ __launch_bounds__(256) void gcn_layer_kernel(const dg_task_batch batch) {
  const dg_task_dev &T = find_task(batch, blockIdx.x, false);
  if (T.flags & DG_IN_PRESCALED) layer_block<false>(T, blockIdx.x - T.blk_begin);
  else layer_block<true>(T, blockIdx.x - T.blk_begin);
}

// Hub rows: one wavefront per row adds the row's slab rows in order, then the epilogue.  Rows of
// LDS-split tasks have thousands of slab rows (parts x workgroups of window_rows_kernel): there a
// whole workgroup takes one row, each wavefront a quarter of the slab rows, and wavefront 0 adds
// the four sums in order.  (Sixteen wavefronts per row instead of four: no faster -- the rows with
// the most parts are bound by what one CU can read.)
__global__ __launch_bounds__(256) void hub_reduce_kernel(const dg_task_batch batch) {
  __shared__ f32x4 red[4][16];
  const dg_task_dev &T = find_task(batch, blockIdx.x, true);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, l = lane & 15;
  if (T.lds_split > 0) {
    const int h = blockIdx.x - T.hub_blk_begin;  // < n_hub by construction of the grid
    const int4 d = reinterpret_cast<const int4 *>(T.hub_desc)[h];
    const int first = d.y * T.win_blk_count, total = (d.z - d.y) * T.win_blk_count;
    const int per = ((total + 255) >> 8) << 6;  // a quarter, rounded up to whole 64-row steps
    const int begin = first + wave * per, end = min(first + total, begin + per);
    const row_ctx rc = load_row_ctx(T, d.x, l);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    gather_rows<false, false>(table_rsrc(T.slab, T.n_parts * T.win_blk_count), nullptr, nullptr, begin + 16 * q, end, 64,
                              l, acc);
    acc = add_groups(acc);
    if (q == 0) red[wave][l] = acc;
    __syncthreads();
    if (wave == 0 && q == 0) epilogue(T, d.x, rc, ((red[0][l] + red[1][l]) + red[2][l]) + red[3][l], l);
    return;
  }
  const int h = (blockIdx.x - T.hub_blk_begin) * 4 + wave;
  if (h >= T.n_hub) return;
  const int4 d = reinterpret_cast<const int4 *>(T.hub_desc)[h];
  const row_ctx rc = load_row_ctx(T, d.x, l);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  gather_rows<false, false>(table_rsrc(T.slab, T.n_seg), nullptr, nullptr, d.y + 16 * q, d.z, 64, l, acc);
  acc = add_groups(acc);
  if (q == 0) epilogue(T, d.x, rc, acc, l);
}

// Row-sharded layer, second half: sum the ranks' partial neighbour sums of a row in rank
// order, then the usual epilogue.  16 lanes per row; all graphs of the layer in one launch.
__global__ __launch_bounds__(256) void partial_finalize_kernel(const dg_task_batch batch) {
  const dg_task_dev &T = find_task(batch, blockIdx.x, false);
  const int l = threadIdx.x & 15;
  const int r = (blockIdx.x - T.blk_begin) * 16 + (threadIdx.x >> 4);
  if (r >= T.fin_rows) return;
  const int row = T.fin_row_begin + r;
  const row_ctx rc = load_row_ctx(T, row, l);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < T.fin_parts; ++p)
    acc += reinterpret_cast<const f32x4 *>(T.fin_partials)[((size_t)p * T.fin_stride + r) * (DG_D / 4) + l];
  epilogue(T, row, rc, acc, l);
}

// Sum of the LDS table rows on one large-side row's list: `w` is this lane's dword of the row's 64-B record (slot 0 =
// the degree, then 31 neighbours as 16-bit indices into the table, padding = the zero row n_small); longer lists go on
// in the CSR.  `t` = table + lane's float4.  The additions run in list order.
// PAIR: the first eight slots are read together whatever the degree (padding slots name the zero row, and a sum that
// began at +0 never is -0, so adding it changes no bit): one wait for two blocks where few wavefronts share the CU.
template <bool PAIR = false>
__device__ __forceinline__ f32x4 lds_row_sum(const dg_task_dev &T, const f32x4 *__restrict__ t, unsigned w, int row, int n_small,
                                             int split, int l) {
  const int deg = bcast16<0>((int)w) & 0xffff;
  // slot 0 of the record is the degree, not a neighbour: read the zero row for it
  const unsigned ww = l == 0 ? ((w & 0xffff0000u) | (unsigned)n_small) : w;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (PAIR) {
    const unsigned a0 = (unsigned)bcast16<0>((int)ww), b0 = (unsigned)bcast16<1>((int)ww);
    const unsigned a1 = (unsigned)bcast16<2>((int)ww), b1 = (unsigned)bcast16<3>((int)ww);
    const f32x4 v0 = t[(a0 & 0xffffu) * 16], v1 = t[(a0 >> 16) * 16], v2 = t[(b0 & 0xffffu) * 16], v3 = t[(b0 >> 16) * 16];
    const f32x4 v4 = t[(a1 & 0xffffu) * 16], v5 = t[(a1 >> 16) * 16], v6 = t[(b1 & 0xffffu) * 16], v7 = t[(b1 >> 16) * 16];
    acc += v0; acc += v1; acc += v2; acc += v3;
    acc += v4; acc += v5; acc += v6; acc += v7;
  }
#define DG_LDS4(J)                                                                                    \
  {                                                                                                   \
    const unsigned a = (unsigned)bcast16<2 * (J)>((int)ww), b = (unsigned)bcast16<2 * (J) + 1>((int)ww); \
    const f32x4 v0 = t[(a & 0xffffu) * 16], v1 = t[(a >> 16) * 16];                                   \
    const f32x4 v2 = t[(b & 0xffffu) * 16], v3 = t[(b >> 16) * 16];                                   \
    acc += v0; acc += v1; acc += v2; acc += v3;                                                       \
  }
  if (!PAIR) {
    DG_LDS4(0)
    if (deg >= 4) DG_LDS4(1)
  }
  if (deg >= 8) DG_LDS4(2)
  if (deg >= 12) DG_LDS4(3)
  if (deg >= 16) {
    DG_LDS4(4)
    if (deg >= 20) DG_LDS4(5)
    if (deg >= 24) DG_LDS4(6)
    if (deg >= 28) DG_LDS4(7)
  }
#undef DG_LDS4
  if (deg > 31) {  // rare: the rest of the list from the CSR
    const int e = T.rowptr[row + 1];
    for (int p = T.rowptr[row] + 31; p < e; p += 16) {
      int c = n_small;
      if (p + l < e) c = T.colidx[p + l] - split;
      const int cnt = e - p;
#define DG_LDS4T(K0)                                                                                  \
      {                                                                                               \
        const f32x4 v0 = t[bcast16<K0>(c) * 16], v1 = t[bcast16<K0 + 1>(c) * 16];                     \
        const f32x4 v2 = t[bcast16<K0 + 2>(c) * 16], v3 = t[bcast16<K0 + 3>(c) * 16];                 \
        acc += v0; acc += v1; acc += v2; acc += v3;                                                   \
      }
      DG_LDS4T(0)
      if (cnt > 4) DG_LDS4T(4)
      if (cnt > 8) DG_LDS4T(8)
      if (cnt > 12) DG_LDS4T(12)
#undef DG_LDS4T
    }
  }
  return acc;
}

// Rows [0, lds_split) of a bipartite graph whose other block fits in LDS (see dg_csr::lds_split).
// A 1024-thread workgroup copies that block's table rows (times their source norm unless the
// table is already scaled) into LDS once, then walks its share of the rows: 64 rows at a time,
// one 16-lane group per row.  A row's degree and first 31 neighbours come in one 64-B record of
// 16-bit ids; neighbour rows are read with ds_read_b128 (four rows per wavefront instruction;
// 256-B aligned rows make every 16-lane read group conflict-free).  The next row's record and
// epilogue inputs are requested before the current row is summed, so the loop has no exposed
// global-memory latency.  Rows are taken in natural order: outputs stream out in full lines.
__global__ __launch_bounds__(1024) void lds_rows_kernel(const dg_task_batch batch) {
  extern __shared__ f32x4 tab[];  // [(n - split) + 1][16]; the last row is zeros (padding neighbour)
  int ti = 0;
  for (int k = 0; k < batch.n_tasks; ++k)
    if (batch.t[k].lds_blk_count > 0 && (int)blockIdx.x >= batch.t[k].lds_blk_begin) ti = k;
  const dg_task_dev &T = batch.t[__builtin_amdgcn_readfirstlane(ti)];  // scalar loads for every field
  const int split = T.lds_split, n_small = T.n - split;
  const bool scale = !(T.flags & DG_IN_PRESCALED);
  for (int i = threadIdx.x; i < (n_small + 1) * 16; i += blockDim.x) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int r = i >> 4;
    if (r < n_small) {
      v = reinterpret_cast<const f32x4 *>(T.x)[(size_t)(split + r) * 16 + (i & 15)];
      if (scale) v *= T.dinv_gather[split + r];
    }
    tab[i] = v;
  }
  __syncthreads();
  const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
  // Each workgroup takes tiles of 64*R consecutive rows; a group owns R of them, 64 apart, and
  // asks for the next tile's records and epilogue inputs before summing the current tile, so
  // R rows' worth of requests per group are in flight (the workgroup is alone on its CU).
  constexpr int R = 4;
  const int stride = T.lds_blk_count * 64 * R;
  const f32x4 *t = tab + l;
  int base = (blockIdx.x - T.lds_blk_begin) * 64 * R + g;
  unsigned w_next[R];
  row_ctx rc_next[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    w_next[j] = 0;
    rc_next[j] = row_ctx{};
    const int row = base + 64 * j;
    if (row < split) {
      w_next[j] = T.lhead[(size_t)row * 16 + l];
      rc_next[j] = load_row_ctx(T, row, l);
    }
  }
  for (; base < split; base += stride) {
    unsigned w_cur[R];
    row_ctx rc_cur[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      w_cur[j] = w_next[j];
      rc_cur[j] = rc_next[j];
      const int row = base + stride + 64 * j;
      if (row < split) {
        w_next[j] = T.lhead[(size_t)row * 16 + l];
        rc_next[j] = load_row_ctx(T, row, l);
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int row = base + 64 * j;
      if (row >= split) continue;
      const f32x4 acc = lds_row_sum(T, t, w_cur[j], row, n_small, split, l);
      epilogue(T, row, rc_cur[j], acc, l);
    }
  }
}

// The last layer's rows [0, lds_split) of ALL graphs of a layer call in one walk (DG_ACC_SHARED): a 512-thread
// workgroup takes tiles of 32*R consecutive rows and, tile by tile, goes through the graphs one after the other -- the
// LDS table is refilled with the next graph's small-side rows between two graphs -- while each 16-lane group keeps
// its R rows of e0 and of the running sum over tasks 1.. in registers.  Per row the launch reads e0 once and writes
// two tables (task 0's layer sum; the sum over the other tasks' layer sums, added in task order) instead of reading
// e0 once per graph, writing a layer sum per graph and adding them up in a further pass.  With up to three graphs
// the tiles alternate between task order and its reverse, so that a tile starts on the table the one before ended
// on (two refills per tile, not three; a + b = b + a keeps the sum of two what it was).  The next phase's table and
// row inputs are asked for before the current phase is summed and wait in registers.  Same arithmetic per (row, graph) as lds_rows_kernel under DG_IN_PRESCALED | DG_ACC_REBUILD, in the same
// order: the same bits.
template <int R>
__global__ __launch_bounds__(512) void lds_sum_rows_kernel(const dg_task_batch batch) {
  extern __shared__ f32x4 tab[];  // [(n - split) + 1][16]; the last row is zeros (padding neighbour)
  constexpr int kGroups = 32, kTile = kGroups * R, kFill = (DG_LDS_MAX_ROWS + 1) * 16 / 512;
  const int G = batch.n_tasks;
  const int split = batch.t[0].lds_split, n_small = batch.t[0].n - split;
  const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int n_tiles = (split + kTile - 1) / kTile, fill_n = (n_small + 1) * 16;
  const f32x4 *t = tab + l;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;  // (the whole workgroup)

  static_assert(R <= kFill, "the next tile's e0 rows wait in the table's registers (a tile never begins with a refill)");
  f32x4 fill[kFill], p_next[R], e0[R], het[R];
  unsigned w_next[R];
  float di_next[R], ds_next[R];   // the row's norm, and its norm as a source
  // Loads only: what the phase after the current one needs -- its table (or, when it begins a tile on the table in
  // place, the tile's e0 rows, in the same registers), each row's record, norms and stored activation.  They stay in
  // flight while the current phase is summed.
  // (Every load goes through a buffer descriptor and is issued whatever the row or the table entry -- beyond the
  // table it returns zeros -- so that the compiler can count the loads in flight: loads skipped lane by lane would
  // make it wait for ALL of them before the first use of any.)
  const int n = batch.t[0].n;
  auto vec_rsrc = [](const void *base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
  };
  auto load4 = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  };
  const __amdgpu_buffer_rsrc_t r_e0 = table_rsrc(batch.t[0].acc_in, n);
  auto fetch = [&](int g_next, int tile_next, bool table, bool first) {
    const dg_task_dev &N = batch.t[__builtin_amdgcn_readfirstlane(g_next)];
    const __amdgpu_buffer_rsrc_t r_x = table_rsrc(N.x, n), r_head = vec_rsrc(N.lhead, split * 64),
                                 r_row = vec_rsrc(N.dinv_row, n * 4), r_out = vec_rsrc(N.dinv_out, n * 4);
    if (table) {
#pragma unroll
      for (int k = 0; k < kFill; ++k)   // table row n_small (the zero row) and beyond: past the end of x, zeros
        fill[k] = load4(r_x, (unsigned)split * 256u + (unsigned)(threadIdx.x + 512 * k) * 16u);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const unsigned row = (unsigned)(tile_next * kTile + kGroups * j + grp);
      w_next[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r_head, row * 64u + (unsigned)l * 4u, 0, 0);   // past the last row: 0
      di_next[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_row, row * 4u, 0, 0));
      ds_next[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_out, row * 4u, 0, 0));
      p_next[j] = load4(r_x, row * 256u + (unsigned)l * 16u);
      if (first) fill[j] = load4(r_e0, row * 256u + (unsigned)l * 16u);   // (the phase has no refill: see below)
    }
  };
  const bool snake = G <= 3;
  // One loop body issues every prefetch (the first pass through it only fetches): a separate first fetch would land
  // in other registers, and the waits the compiler then needs for that path would hold up every pass.
  bool rev = false, table = false, first = false, have = false;
  int gi = 0, seen = 0;  // seen: tasks 1.. of this tile already summed
  for (;;) {
    const dg_task_dev &T = batch.t[__builtin_amdgcn_readfirstlane(gi)];
    if (table) {
      __syncthreads();  // every group is done with the previous table
#pragma unroll
      for (int k = 0; k < kFill; ++k) {
        const int i = threadIdx.x + 512 * k;
        if (i < fill_n) tab[i] = fill[k];
      }
      __syncthreads();
    } else if (first) {
#pragma unroll
      for (int j = 0; j < R; ++j) e0[j] = fill[j];
    }
    unsigned w[R];
    float di_[R], ds_[R];
    f32x4 p[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      w[j] = w_next[j];
      di_[j] = di_next[j];
      ds_[j] = ds_next[j];
      p[j] = p_next[j];
    }
    if (first) seen = 0;
    // what comes after this phase
    int g_next = gi + (rev ? -1 : 1), tile_next = tile;
    bool first_next = false, more = true, table_next = true, rev_next = rev;
    if (!have) {                      // the first phase itself
      g_next = 0;
      first_next = true;
    } else if (g_next < 0 || g_next >= G) {  // the tile ends here
      tile_next = tile + (int)gridDim.x;
      more = tile_next < n_tiles;
      first_next = true;
      if (snake) {
        rev_next = !rev;
        g_next = gi;
        table_next = false;
      } else {
        g_next = 0;
      }
    }
    // (a tile's last phase still reads e0: its successor's e0 rows go to the table's registers; a tile that begins
    // with a refill -- the first one, and every one without the alternating order -- loads them itself)
    if (more) fetch(g_next, tile_next, table_next, first_next && !table_next);
    if (have) {
      if (first && table) {
#pragma unroll
        for (int j = 0; j < R; ++j) e0[j] = load4(r_e0, (unsigned)(tile * kTile + kGroups * j + grp) * 256u + (unsigned)l * 16u);
      }
      if (gi > 0) ++seen;
      const float slope = T.slope;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int row = tile * kTile + kGroups * j + grp;
        if (row >= split) continue;
        const f32x4 acc = lds_row_sum<true>(T, t, w[j], row, n_small, split, l);
        // (epilogue() under DG_IN_PRESCALED | DG_ACC_REBUILD with acc_out and no e_out)
        const float di = di_[j], dsrc = ds_[j];
        f32x4 y = acc * di;
        y.x = y.x > 0.f ? y.x : y.x * slope;
        y.y = y.y > 0.f ? y.y : y.y * slope;
        y.z = y.z > 0.f ? y.z : y.z * slope;
        y.w = y.w > 0.f ? y.w : y.w * slope;
        const float ss = sum16(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w);
        const f32x4 yn = y * inv_norm(ss);
        f32x4 a = e0[j];
        const f32x4 pv = p[j];
        const float ps = sum16(pv.x * pv.x + pv.y * pv.y + pv.z * pv.z + pv.w * pv.w);
        a += pv * fminf(__builtin_amdgcn_rsqf(ps), 1e12f * __builtin_amdgcn_rcpf(dsrc));
        const f32x4 res = a + yn;
        const size_t o = (size_t)row * 16 + l;
        if (gi == 0) {
          store_stream(T.acc_out, o, res);
        } else {
          het[j] = seen == 1 ? res : (rev ? res + het[j] : het[j] + res);
          if (seen == G - 1) store_stream(T.acc_out, o, het[j]);
        }
      }
    }
    if (!more) break;
    have = true;
    first = first_next;
    gi = g_next;
    tile = tile_next;
    rev = rev_next;
    table = table_next;
  }
}

// Rows [lds_split, n) of the same graphs (the small side; each has neighbours all over the large
// side).  A 1024-thread workgroup walks a contiguous run of windows of DG_WIN_ROWS large-side rows:
// it copies the window's table rows (times their source norm unless already scaled) and the
// window's entry blocks into LDS and every 16-lane group adds, for each part it owns
// (dg_csr::wblocks), the window rows on that part's list: one ds_read_b32 gives four list entries,
// four ds_read_b128 the rows.  The windows alternate between two LDS buffers (all 160 KB of the
// CU): while window w is summed out of one, window w + 1 -- requested from memory one window
// earlier -- is written into the other and window w + 2 is requested; one barrier per window.
// The running sums of a group's (up to 8) parts stay in registers for the whole run and go to
// the slab at the end, one slab row per (part, workgroup); hub_reduce_kernel adds a row's slab
// rows in order.  All orders are fixed, so results are reproducible run to run.
constexpr int kWinTabF4 = (DG_WIN_ROWS + 1) * 16;   // float4 of one staged window (the last row is zeros)
constexpr size_t kWinLds = 2 * ((size_t)DG_WIN_ENT_DWORDS * 4 + (size_t)kWinTabF4 * 16);
static_assert(kWinLds <= 160 * 1024, "two staged windows and their entry blocks fill the CU's LDS exactly");
__global__ __launch_bounds__(1024) void window_rows_kernel(const dg_task_batch batch) {
  // [2][DG_WIN_ENT_DWORDS] entry blocks, then [2][DG_WIN_ROWS + 1][16] rows: the dword read past the end of an
  // entry buffer (see below) then lies inside the allocation
  extern __shared__ f32x4 win_lds[];
  uint32_t *ebuf = reinterpret_cast<uint32_t *>(win_lds);
  f32x4 *tab = win_lds + 2 * DG_WIN_ENT_DWORDS / 4;
  int ti = 0;
  for (int k = 0; k < batch.n_tasks; ++k)
    if (batch.t[k].win_blk_count > 0 && (int)blockIdx.x >= batch.t[k].win_blk_begin) ti = k;
  const dg_task_dev &T = batch.t[__builtin_amdgcn_readfirstlane(ti)];
  const int b = blockIdx.x - T.win_blk_begin, B = T.win_blk_count;
  const int per = (T.n_windows + B - 1) / B;
  const int w0 = b * per, w1 = min(T.n_windows, w0 + per);
  const int split = T.lds_split;
  const bool scale = !(T.flags & DG_IN_PRESCALED);
  const int g = threadIdx.x >> 4, l = threadIdx.x & 15;

  f32x4 acc[DG_WIN_SLOTS];
#pragma unroll
  for (int k = 0; k < DG_WIN_SLOTS; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a window on its way from memory: its rows (4 float4 per thread) and their norms, 4 dwords of its entry
  // blocks, and where this group's block starts among them
  f32x4 pre[4];
  float ps[4];
  u32x4 pent;
  int goff_next = 0;
  auto fetch = [&](int w) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = threadIdx.x + 1024 * p, r = i >> 4, u = w * DG_WIN_ROWS + r;
      pre[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      ps[p] = 1.0f;
      if (r < DG_WIN_ROWS && u < split) {
        pre[p] = reinterpret_cast<const f32x4 *>(T.x)[(size_t)u * 16 + (i & 15)];
        if (scale) ps[p] = T.dinv_gather[u];
      }
    }
    const unsigned o0 = T.woff[w * DG_WIN_GROUPS], o1 = T.woff[(w + 1) * DG_WIN_GROUPS];
    pent = u32x4{~0u, ~0u, ~0u, ~0u};
    if (4 * threadIdx.x < o1 - o0) pent = reinterpret_cast<const u32x4 *>(T.wblocks + o0)[threadIdx.x];
    goff_next = (int)(T.woff[w * DG_WIN_GROUPS + g] - o0);
  };
  auto stage = [&](int buf) {   // the fetched window -> LDS buffer `buf`
#pragma unroll
    for (int p = 0; p < 4; ++p) tab[buf * kWinTabF4 + threadIdx.x + 1024 * p] = pre[p] * ps[p];  // row 255 gets zeros
    reinterpret_cast<u32x4 *>(ebuf + buf * DG_WIN_ENT_DWORDS)[threadIdx.x] = pent;
  };
  int goff = 0;
  if (w0 < w1) {   // (a workgroup whose run is empty still writes its -- zero -- slab rows below)
    fetch(w0);
    stage(0);
    goff = goff_next;
    if (w0 + 1 < w1) fetch(w0 + 1);
    __syncthreads();
  }
  for (int w = w0; w < w1; ++w) {
    const int cur = (w - w0) & 1;
    const f32x4 *t = tab + cur * kWinTabF4 + l;
    const uint32_t *e = ebuf + cur * DG_WIN_ENT_DWORDS + goff;
    if (w + 1 < w1) {   // the other buffer was last read a barrier ago
      stage(cur ^ 1);
      goff = goff_next;
      if (w + 2 < w1) fetch(w + 2);
    }
    const unsigned h0 = e[0], h1 = e[1];
    unsigned dn = e[DG_WIN_HDR];  // the entry dword after the one being summed is always on its way
    e += DG_WIN_HDR + 1;
#pragma unroll
    for (int k = 0; k < DG_WIN_SLOTS; ++k) {
      const int c = ((k < 4 ? h0 : h1) >> (8 * (k & 3))) & 0xff;
      for (int j = 0; j < c; ++j) {
        const unsigned d = dn;
        dn = *e++;  // at the end of the last block this reads the dword behind the buffer
        const f32x4 v0 = t[(d & 0xffu) * 16], v1 = t[(d >> 8 & 0xffu) * 16];
        const f32x4 v2 = t[(d >> 16 & 0xffu) * 16], v3 = t[(d >> 24) * 16];
        acc[k] += v0; acc[k] += v1; acc[k] += v2; acc[k] += v3;
      }
    }
    __syncthreads();  // this window has been read by every group, the next one is in place
  }
#pragma unroll
  for (int k = 0; k < DG_WIN_SLOTS; ++k) {
    const int part = T.witems[g * DG_WIN_SLOTS + k];
    if (part >= 0) reinterpret_cast<f32x4 *>(T.slab)[((size_t)part * B + b) * 16 + l] = acc[k];
  }
}

constexpr int kDenseWaves = DG_DENSE_TILES_PER_BLOCK;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kDensePitch = 144;                        // bytes of one staged column: 64 bf16 + 16 B (conflict-free 16-B reads)
constexpr int kDensePlane = 64 * kDensePitch;           // one plane of one 64-row stage
constexpr int kDenseStage = 3 * kDensePlane;

// Dense graphs (dg_csr::dense_bits): Y = A . X on the matrix cores with the adjacency bitmap expanded to 0.0 / 1.0
// in registers.  The table arrives as three bf16 planes whose sum is the fp32 value exactly (dense_split_kernel:
// hi = the top 8 significand bits, mid = the next 8, lo = the last 8), stored transposed, plane[p][feature][row], so
// that the 8 consecutive table rows a lane supplies to v_mfma_f32_32x32x16_bf16 are 16 contiguous bytes.  Every
// product 1.0 * plane value is exact and the accumulator is fp32: the result is the fp32 sum of the neighbour rows,
// its rounding errors those of fp32 additions (in the instruction's order of adding, not ascending neighbour order),
// at a sixteenth of the cycles the fp32 matrix instruction takes even with three planes to add.
// A workgroup owns kDenseWaves consecutive 32-row tiles of A (one per wavefront) and one range of its columns
// (= table rows); the range passes through LDS 64 table rows at a time (two buffers: the next 64 are on their way
// while the current ones are multiplied), so every wavefront of the workgroup reads them from there.  Lane
// (i = lane & 31, kk = lane >> 5) holds the 64-bit word of row i of its tile for the current 64 columns; in step t
// byte 2 t + kk of it becomes the A operand (columns 16 t + 8 kk ..+8) and plane[p][i or 32 + i][16 t + 8 kk ..+8]
// the B operands of the two 32-feature halves.  The tile's 32 x 64 partial sums go to slab rows
// row * dense_splits + range; hub_reduce_kernel adds a row's ranges in order and applies the epilogue.
__global__ __launch_bounds__(64 * kDenseWaves) void dense_rows_kernel(const dg_task_batch batch) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * kDenseStage];
  if ((int)blockIdx.x < batch.gather_blocks) {
    // the batch's sparse graphs: their few gather workgroups ride along instead of taking a launch of their own
    const dg_task_dev &G = find_task(batch, blockIdx.x, false);
    if (G.flags & DG_IN_PRESCALED) layer_block<false>(G, blockIdx.x - G.blk_begin);
    else layer_block<true>(G, blockIdx.x - G.blk_begin);
    return;
  }
  const int blk = (int)blockIdx.x - batch.gather_blocks;
  int ti = 0;
  for (int k = 0; k < batch.n_tasks; ++k)
    if (batch.t[k].dense_splits > 0 && blk >= batch.t[k].dense_blk_begin) ti = k;
  const dg_task_dev &T = batch.t[__builtin_amdgcn_readfirstlane(ti)];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5;
  const int lb = blk - T.dense_blk_begin;
  const int S = T.dense_splits, group = lb / S, split = lb - group * S;
  const int tile = group * kDenseWaves + wave;   // (the bitmap is allocated for whole tile groups: a missing tile reads zeros)
  const unsigned row_bytes = (unsigned)T.dense_chunks * 128u;      // one feature of one plane: 64 bf16 per chunk
  const unsigned char *planes = reinterpret_cast<const unsigned char *>(T.dense_x);
  const int c0 = split * T.dense_cps, c1 = min(c0 + T.dense_cps, T.dense_chunks);
  const uint64_t *bits = T.dense_bits + ((size_t)tile * T.dense_chunks) * 32 + i;
  constexpr int kPer = 3 * 512 / (64 * kDenseWaves);   // 16-B pieces per thread of one stage (3 planes x 64 features x 8)
  u32x4 pre[kPer];
  auto fetch = [&](int c) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const unsigned idx = threadIdx.x + 64 * kDenseWaves * j, p = idx >> 9, f = (idx >> 3) & 63, q = idx & 7;
      pre[j] = *reinterpret_cast<const u32x4 *>(planes + (size_t)(p * 64 + f) * row_bytes + (unsigned)c * 128u + q * 16u);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const unsigned idx = threadIdx.x + 64 * kDenseWaves * j, p = idx >> 9, f = (idx >> 3) & 63, q = idx & 7;
      *reinterpret_cast<u32x4 *>(xs + buf * kDenseStage + p * kDensePlane + f * kDensePitch + q * 16) = pre[j];
    }
  };
  f32x16 acc0 = {0.f}, acc1 = {0.f};
  fetch(c0);
  uint64_t wn = bits[(size_t)c0 * 32];
  stash(0);
  __syncthreads();
  int cur = 0;
  for (int c = c0; c < c1; ++c) {
    const uint64_t wb = wn;
    if (c + 1 < c1) {
      fetch(c + 1);
      wn = bits[(size_t)(c + 1) * 32];
    }
    const unsigned char *xb = xs + cur * kDenseStage + i * kDensePitch + kk * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned ab = (unsigned)(wb >> (16 * t + 8 * kk)) & 0xffu;
      u32x4 av;   // two bf16 per dword: 1.0 = 0x3F80
#pragma unroll
      for (int d = 0; d < 4; ++d) av[d] = ((ab >> (2 * d)) & 1u) * 0x3F80u + ((ab >> (2 * d + 1)) & 1u) * 0x3F800000u;
      const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(xb + p * kDensePlane + t * 32);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(xb + p * kDensePlane + 32 * kDensePitch + t * 32);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1, 0, 0, 0);
      }
    }
    if (c + 1 < c1) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (tile >= T.dense_tiles) return;
  // register r of lane (i, kk) is tile row (r & 3) + 8 (r >> 2) + 4 kk, feature i (acc0) / 32 + i (acc1)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
    if (row < T.n) {
      float *dst = T.slab + ((size_t)row * S + split) * DG_D + i;
      dst[0] = acc0[r];
      dst[32] = acc1[r];
    }
  }
}

// hi / mid / lo of one fp32 value: three bf16 numbers (upper 16 bits each) whose sum is the value exactly.
__device__ __forceinline__ void split3(float x, unsigned short &hi, unsigned short &mid, unsigned short &lo) {
  const unsigned hb = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, hb);
  const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
  const float r2 = r1 - __builtin_bit_cast(float, mb);
  hi = (unsigned short)(hb >> 16);
  mid = (unsigned short)(mb >> 16);
  lo = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
}

// Does this task leave (DG_OUT_PLANES) / find (DG_IN_PLANES) its graph's table as bf16 planes?  Only graphs on the
// bitmap path whose rows have at most 16 column ranges (one 16-lane group then adds a row's slab rows in the order
// hub_reduce_kernel does); both sides of the hand-over decide by this same rule.
__host__ __device__ __forceinline__ bool planes_path(const dg_task_dev &t, int flag) {
  return t.dense_splits > 0 && t.dense_splits <= 16 && (t.flags & flag) != 0;
}

// hub_reduce_kernel for the rows of a bitmap graph whose output the same graph gathers next (DG_OUT_PLANES): a
// workgroup takes one chunk of 64 rows, a 16-lane group per row -- the row's slab rows added in range order (none:
// the row-wise first step of a backward pass), then the usual epilogue -- and also leaves the stored values as the
// three bf16 planes dense_rows_kernel reads (what dense_split_kernel would make of them: the next layer over this
// graph, flagged DG_IN_PLANES, needs no split launch).
__global__ __launch_bounds__(1024) void dense_reduce_kernel(const dg_task_batch batch) {
  __shared__ unsigned short tr[3][64][72];   // [plane][feature][row of the chunk]
  int ti = 0;
  for (int k = 0; k < batch.n_tasks; ++k)
    if (planes_path(batch.t[k], DG_OUT_PLANES) && (int)blockIdx.x >= batch.t[k].red_blk_begin) ti = k;
  const dg_task_dev &T = batch.t[__builtin_amdgcn_readfirstlane(ti)];
  const int c = blockIdx.x - T.red_blk_begin, r = threadIdx.x >> 4, l = threadIdx.x & 15, row = 64 * c + r;
  f32x4 e = {0.f, 0.f, 0.f, 0.f};   // rows beyond the table: zeros, as dense_split_kernel pads
  if (row < T.n) {
    const row_ctx rc = load_row_ctx(T, row, l);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (T.red_parts > 0)
      gather_rows<false, false>(table_rsrc(T.slab, T.n_seg), nullptr, nullptr, row * T.red_parts, (row + 1) * T.red_parts, 16, l, acc);
    epilogue(T, row, rc, acc, l, &e);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) split3(e[j], tr[0][4 * l + j][r], tr[1][4 * l + j][r], tr[2][4 * l + j][r]);
  __syncthreads();
  unsigned char *planes = reinterpret_cast<unsigned char *>(T.dense_x);
  const unsigned row_bytes = (unsigned)T.dense_chunks * 128u;
  for (unsigned idx = threadIdx.x; idx < 3 * 512; idx += 1024) {
    const unsigned p = idx >> 9, f = (idx >> 3) & 63, q = idx & 7;
    *reinterpret_cast<u32x4 *>(planes + (size_t)(p * 64 + f) * row_bytes + (unsigned)c * 128u + q * 16u) =
        *reinterpret_cast<const u32x4 *>(&tr[p][f][8 * q]);
  }
}

// The table of a dense graph as dense_rows_kernel reads it: times the norm of the gathered node when the caller's is
// not scaled yet (the first layer), cut into three bf16 planes that add up to the fp32 value exactly -- truncation
// keeps every remainder representable: hi = the value's upper 16 bits, mid = those of value - hi, lo = the rest,
// at most 8 significant bits -- and transposed to plane[p][feature][row], rows padded with zeros to whole 64-row
// chunks.  One workgroup per chunk of 64 table rows.
__global__ __launch_bounds__(256) void dense_split_kernel(const dg_task_batch batch) {
  __shared__ unsigned short tr[3][64][72];   // [plane][feature][row of the chunk]
  const dg_task_dev &T = batch.t[blockIdx.y];
  if (T.dense_splits <= 0 || (int)blockIdx.x >= T.dense_chunks || planes_path(T, DG_IN_PLANES)) return;   // (the whole workgroup)
  const int c = blockIdx.x, r = threadIdx.x >> 2, f0 = (threadIdx.x & 3) * 16, row = 64 * c + r;
  const bool scale = !(T.flags & DG_IN_PRESCALED);
  float s = 0.f;
  if (row < T.n) s = scale ? T.dinv_gather[row] : 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < T.n) {
      v = reinterpret_cast<const f32x4 *>(T.x)[(size_t)row * (DG_D / 4) + (f0 >> 2) + j];
      if (scale) v *= s;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = f0 + 4 * j + e;
      split3(v[e], tr[0][f][r], tr[1][f][r], tr[2][f][r]);
    }
  }
  __syncthreads();
  unsigned char *planes = reinterpret_cast<unsigned char *>(T.dense_x);
  const unsigned row_bytes = (unsigned)T.dense_chunks * 128u;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const unsigned idx = threadIdx.x + 256 * j, p = idx >> 9, f = (idx >> 3) & 63, q = idx & 7;
    *reinterpret_cast<u32x4 *>(planes + (size_t)(p * 64 + f) * row_bytes + (unsigned)c * 128u + q * 16u) =
        *reinterpret_cast<const u32x4 *>(&tr[p][f][8 * q]);
  }
}

// Rows with nothing to gather (last layer of the backward pass): epilogue only, 16 lanes per row.
__global__ __launch_bounds__(256) void rowwise_kernel(const dg_task_batch batch) {
  const dg_task_dev &T = find_task(batch, blockIdx.x, false);
  const int l = threadIdx.x & 15;
  const int row = (blockIdx.x - T.blk_begin) * 16 + (threadIdx.x >> 4);
  if (row >= T.n) return;
  const row_ctx rc = load_row_ctx(T, row, l);
  epilogue(T, row, rc, f32x4{0.f, 0.f, 0.f, 0.f}, l);
}

struct sum_args {
  const float *in[16];
  int32_t n_in;
};

__global__ __launch_bounds__(256) void sum_tables_kernel(sum_args a, f32x4 *__restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4 *>(a.in[0])[i];
    for (int k = 1; k < a.n_in; ++k) v += reinterpret_cast<const f32x4 *>(a.in[k])[i];
    out[i] = v;
  }
}

int fill_task(const dg_layer_task &in, int transpose, dg_task_dev &o) {
  const dg_graph *g = in.graph;
  if (!g) return dg_fail(DG_ERR_ARG, "layer task: graph is NULL");
  if (!in.x) return dg_fail(DG_ERR_ARG, "layer task: x is NULL");
  if ((in.acc_out != nullptr) != (in.acc_in != nullptr))
    return dg_fail(DG_ERR_ARG, "layer task: acc_in and acc_out must both be set or both be NULL");
  if (!in.e_out && !in.acc_out) return dg_fail(DG_ERR_ARG, "layer task: no output requested");
  if (in.e_out == in.x) return dg_fail(DG_ERR_ARG, "layer task: e_out must not alias x");
  if ((in.flags & DG_ACC_REBUILD) && (!(in.flags & DG_IN_PRESCALED) || !in.acc_out))
    return dg_fail(DG_ERR_ARG, "layer task: DG_ACC_REBUILD needs DG_IN_PRESCALED and acc_in/acc_out");
  if ((in.flags & DG_OUT_PLANES) && !in.e_out)
    return dg_fail(DG_ERR_ARG, "layer task: DG_OUT_PLANES needs e_out (the planes are a second form of it)");
  if ((in.flags & DG_IN_PLANES) && !(in.flags & DG_IN_PRESCALED))
    return dg_fail(DG_ERR_ARG, "layer task: DG_IN_PLANES needs DG_IN_PRESCALED (the planes hold the table as it was stored)");
  if (g->weighted && (transpose || (in.flags & (DG_IN_PRESCALED | DG_EPI_BWD))))
    return dg_fail(DG_ERR_ARG, "layer task: a weighted operator runs forward only, over a table that is not scaled yet");
  const dg_csr &c = transpose ? g->bwd : g->fwd;
  o.rowptr = c.rowptr; o.lhead = c.lhead; o.lds_split = c.lds_split; o.lds_blk_begin = 0; o.lds_blk_count = 0;
  o.dinv_gather = transpose ? g->dinv_dst : g->dinv_src;
  o.wblocks = c.wblocks; o.woff = c.woff; o.witems = c.witems; o.n_windows = c.n_windows; o.n_parts = c.n_parts;
  o.win_blk_begin = 0; o.win_blk_count = 0;
  o.dense_bits = c.dense_bits; o.dense_x = g->dense_x; o.dense_splits = c.dense_splits; o.dense_chunks = c.dense_chunks;
  o.dense_cps = c.dense_cps; o.dense_tiles = c.dense_tiles; o.dense_blk_begin = 0;
  o.red_blk_begin = 0; o.red_parts = c.dense_splits;
  o.colidx = c.colidx; o.head = c.head; o.hub_desc = c.hub_desc; o.seg_desc = c.seg_desc; o.gseg_desc = c.gseg_desc;
  o.colscale = c.colscale; o.head_scale = c.head_scale;
  o.dinv_row = transpose ? g->dinv_src : g->dinv_dst;
  o.dinv_out = transpose ? g->dinv_dst : g->dinv_src;
  o.x = in.x; o.slab = g->slab; o.e_out = in.e_out; o.acc_in = in.acc_in; o.acc_out = in.acc_out;
  o.n = g->n; o.n_hub = c.n_hub; o.n_wave = c.n_wave; o.n_group = c.n_slots; o.n_seg = c.n_seg; o.n_seg_slots = c.n_seg_slots; o.n_gseg_slots = c.n_gseg_slots;
  o.flags = in.flags; o.slope = in.slope;
  o.e_saved = nullptr; o.d_sum = nullptr;
  o.fin_partials = nullptr; o.fin_stride = 0; o.fin_parts = o.fin_row_begin = o.fin_rows = 0;
  o.partial_out = in.partial_out; o.partial_from = in.partial_from;
  o.row_norm = in.row_norm_out;
  return 0;
}

constexpr int kExchangeCus = 8;  // CUs kept free of lds_rows_kernel workgroups under DG_ROWS_FULL_ONLY

// `total` workgroups shared out in proportion to the weights (largest remainders; every weighted task gets one at least).
void apportion(int total, const double *w, int n, int32_t *cnt) {
  double wsum = 0, frac[DG_MAX_TASKS];
  for (int k = 0; k < n; ++k) wsum += w[k];
  int used = 0;
  for (int k = 0; k < n; ++k) {
    cnt[k] = 0;
    frac[k] = -1.0;
    if (w[k] > 0) {
      const double x = total * w[k] / wsum;
      cnt[k] = std::max(1, (int)x);
      frac[k] = x - (int)x;
      used += cnt[k];
    }
  }
  while (used < total) {
    int best = -1;
    for (int k = 0; k < n; ++k)
      if (frac[k] >= 0 && (best < 0 || frac[k] > frac[best])) best = k;
    if (best < 0) break;
    ++cnt[best];
    frac[best] = -1.0;
    ++used;
  }
}

int launch_batch(dg_task_batch &batch, const dg_graph *const *graphs, hipStream_t stream) {
  const int n_tasks = batch.n_tasks;
  int32_t blocks = 0, hub_blocks = 0, red_blocks = 0;
  // which rows this call computes (row-sharded runs may ask for the partial-sum rows and the rest separately)
  constexpr int32_t kRowsMask = DG_ROWS_PARTIAL_ONLY | DG_ROWS_FULL_ONLY;
  const int32_t rows_sel = batch.t[0].flags & kRowsMask;
  if (rows_sel == kRowsMask) return dg_fail(DG_ERR_ARG, "layer task: DG_ROWS_PARTIAL_ONLY and DG_ROWS_FULL_ONLY exclude each other");
  int device = 0;
  DG_HIP(hipGetDevice(&device));
  for (int k = 0; k < n_tasks; ++k) {
    const dg_task_dev &t = batch.t[k];
    if (graphs[k]->device != device)
      return dg_fail(DG_ERR_ARG, "layer task %d: its graph was created on device %d, the current device is %d", k,
                     graphs[k]->device, device);
    if ((t.flags & kRowsMask) != rows_sel)
      return dg_fail(DG_ERR_ARG, "layer tasks 0 and %d differ in DG_ROWS_PARTIAL_ONLY / DG_ROWS_FULL_ONLY", k);
    if (rows_sel && (t.lds_split <= 0 || (t.partial_out && t.partial_from != t.lds_split)))
      return dg_fail(DG_ERR_ARG, "layer task %d: DG_ROWS_*_ONLY needs a graph with one side held in LDS and, when partial sums are "
                     "asked for, partial_from at that side's first row (%d), got partial_from %d", k, t.lds_split,
                     t.partial_out ? t.partial_from : -1);
  }
  const bool full_rows = rows_sel != DG_ROWS_PARTIAL_ONLY, partial_rows = rows_sel != DG_ROWS_FULL_ONLY;
  for (int k = 0; k < n_tasks; ++k) {
    for (int j = 0; j < k; ++j)
      if (graphs[j] == graphs[k] && batch.t[k].n_seg > 0)
        return dg_fail(DG_ERR_ARG, "layer tasks %d and %d share a graph with hub rows (one slab per graph)", j, k);
    dg_task_dev &t = batch.t[k];
    blocks = (blocks + 7) & ~7;  // a task's block b and blockIdx.x agree mod 8 (the segment order counts on it)
    t.blk_begin = blocks;
    t.hub_blk_begin = hub_blocks;
    blocks += (t.n_seg_slots >> 2) + (t.n_gseg_slots >> 4) + ((t.n_group + 15) >> 4);
    t.red_blk_begin = red_blocks;
    if (planes_path(t, DG_OUT_PLANES)) red_blocks += t.dense_chunks;   // dense_reduce_kernel instead of hub_reduce_kernel
    else hub_blocks += t.lds_split > 0 ? t.n_hub : (t.n_hub + 3) >> 2;
  }
  // workgroups of the two LDS kernels: one per CU in total each, shared out by the tasks' work
  int32_t lds_blocks = 0, lds_rows_max = 0, win_blocks = 0;
  {
    double w[DG_MAX_TASKS], wsum = 0;
    for (int k = 0; k < n_tasks; ++k) {
      // a row costs its streamed record/inputs/outputs (~0.8 KB from HBM) plus one LDS row read per edge,
      // which is worth about 1/50 of that
      w[k] = batch.t[k].lds_split > 0 ? 0.5 * (double)graphs[k]->nnz + 48.0 * batch.t[k].lds_split : 0.0;
      wsum += w[k];
    }
    int dev = 0, cus = 256;
    if (wsum > 0 && hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // A call for the non-partial rows alone runs beside the exchange of the partial sums: lds_rows_kernel holds a whole
    // CU per workgroup until it ends, so a few CUs are left to the collective's kernel.
    const int lds_cus = rows_sel == DG_ROWS_FULL_ONLY ? std::max(1, cus - kExchangeCus) : cus;
    int32_t share[DG_MAX_TASKS];
    apportion(lds_cus, w, n_tasks, share);
    for (int k = 0; k < n_tasks; ++k) {
      dg_task_dev &t = batch.t[k];
      t.lds_blk_begin = lds_blocks;
      if (w[k] > 0) {
        const int32_t cnt = std::min(share[k], (t.lds_split + 63) / 64);
        t.lds_blk_count = cnt;
        lds_blocks += cnt;
        lds_rows_max = std::max(lds_rows_max, t.n - t.lds_split);
      } else {
        t.lds_blk_count = 0;
      }
    }
    // small-side rows: a window costs its list entries plus staging 64 KB (worth ~1500 entries)
    wsum = 0;
    for (int k = 0; k < n_tasks; ++k) {
      w[k] = batch.t[k].lds_split > 0 ? 0.5 * (double)graphs[k]->nnz + 1500.0 * batch.t[k].n_windows : 0.0;
      wsum += w[k];
    }
    apportion(cus, w, n_tasks, share);
    for (int k = 0; k < n_tasks; ++k) {
      dg_task_dev &t = batch.t[k];
      t.win_blk_begin = win_blocks;
      t.win_blk_count = 0;
      if (w[k] > 0) {
        t.win_blk_count = std::min(std::min(share[k], t.n_windows), DG_WIN_MAX_BLOCKS);
        win_blocks += t.win_blk_count;
      }
    }
  }
  int32_t dense_blocks = 0, dense_chunks_max = 0;   // (dense_chunks_max: of the tasks whose table still has to be split)
  for (int k = 0; k < n_tasks; ++k) {
    dg_task_dev &t = batch.t[k];
    t.dense_blk_begin = dense_blocks;
    if (t.dense_splits > 0) {
      for (int j = 0; j < k; ++j)
        if (graphs[j] == graphs[k]) return dg_fail(DG_ERR_ARG, "layer tasks %d and %d share a dense graph (one slab per graph)", j, k);
      dense_blocks += ((t.dense_tiles + kDenseWaves - 1) / kDenseWaves) * t.dense_splits;
      if (!planes_path(t, DG_IN_PLANES)) dense_chunks_max = max(dense_chunks_max, t.dense_chunks);
    }
  }
  for (int k = n_tasks; k < DG_MAX_TASKS; ++k) batch.t[k] = batch.t[0];
  // A batch of dense graphs with a little gather-path work beside them (DBLP: the 11 113-edge co-author graph next to
  // two bitmap graphs): those workgroups go first in the dense launch rather than into a launch of their own.
  constexpr int32_t kRideAlongBlocks = 1024;
  batch.gather_blocks = (dense_blocks > 0 && blocks <= kRideAlongBlocks) ? blocks : 0;
  if (dense_blocks > 0) {
    if (dense_chunks_max > 0) hipLaunchKernelGGL(dense_split_kernel, dim3(dense_chunks_max, n_tasks), dim3(256), 0, stream, batch);
    hipLaunchKernelGGL(dense_rows_kernel, dim3(batch.gather_blocks + dense_blocks), dim3(64 * kDenseWaves), 0, stream, batch);
    if (batch.gather_blocks > 0) blocks = 0;
  }
  if (batch.t[0].flags & DG_ACC_SHARED) {
    // the rows below lds_split of all tasks in one walk, two output tables (lds_sum_rows_kernel)
    if (rows_sel != DG_ROWS_FULL_ONLY)
      return dg_fail(DG_ERR_ARG, "layer task: DG_ACC_SHARED comes with DG_ROWS_FULL_ONLY (the other rows are summed per task)");
    constexpr int32_t need = DG_IN_PRESCALED | DG_ACC_REBUILD | DG_ACC_SHARED;
    for (int k = 0; k < n_tasks; ++k) {
      const dg_task_dev &t = batch.t[k];
      if ((t.flags & need) != need || t.e_out || !t.acc_out || t.row_norm || t.n != batch.t[0].n ||
          t.lds_split != batch.t[0].lds_split || t.acc_in != batch.t[0].acc_in)
        return dg_fail(DG_ERR_ARG, "layer task %d: DG_ACC_SHARED needs DG_IN_PRESCALED | DG_ACC_REBUILD on every task, no e_out, no "
                       "row_norm_out, one acc_in, and graphs of one size with the same side held in LDS", k);
      if (k >= 1 && (t.acc_out != batch.t[1].acc_out || t.acc_out == batch.t[0].acc_out))
        return dg_fail(DG_ERR_ARG, "layer task %d: under DG_ACC_SHARED tasks 1.. name one acc_out table, task 0 another", k);
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    constexpr int rows_per_group = 6;   // (8 rows per group spill registers)
    const int32_t split = batch.t[0].lds_split, tile = 32 * rows_per_group;
    const int32_t grid = std::min(std::max(1, cus - kExchangeCus), (split + tile - 1) / tile);
    const size_t lds_bytes = (size_t)(batch.t[0].n - split + 1) * DG_D * sizeof(float);
    for (int k = n_tasks; k < DG_MAX_TASKS; ++k) batch.t[k] = batch.t[0];
    static std::atomic<uint64_t> opted{0};
    DG_HIP(dg_lds_opt_in(reinterpret_cast<const void *>(lds_sum_rows_kernel<rows_per_group>),
                         (DG_LDS_MAX_ROWS + 1) * DG_D * (int)sizeof(float), opted));
    hipLaunchKernelGGL(lds_sum_rows_kernel<rows_per_group>, dim3(grid), dim3(512), lds_bytes, stream, batch);
    DG_HIP(hipGetLastError());
    return 0;
  }
  if (lds_blocks > 0 && full_rows) {
    const size_t lds_bytes = (size_t)(lds_rows_max + 1) * DG_D * sizeof(float);
    static std::atomic<uint64_t> lds_opted{0};
    DG_HIP(dg_lds_opt_in(reinterpret_cast<const void *>(lds_rows_kernel), (DG_LDS_MAX_ROWS + 1) * DG_D * (int)sizeof(float),
                         lds_opted));
    hipLaunchKernelGGL(lds_rows_kernel, dim3(lds_blocks), dim3(1024), lds_bytes, stream, batch);
  }
  if (win_blocks > 0 && partial_rows) {
    static std::atomic<uint64_t> win_opted{0};
    DG_HIP(dg_lds_opt_in(reinterpret_cast<const void *>(window_rows_kernel), (int)kWinLds, win_opted));
    hipLaunchKernelGGL(window_rows_kernel, dim3(win_blocks), dim3(1024), kWinLds, stream, batch);
  }
  // (under DG_ROWS_*_ONLY no task has gcn_layer_kernel rows, and every hub row is a partial-sum row)
  if (blocks > 0) hipLaunchKernelGGL(gcn_layer_kernel, dim3(blocks), dim3(256), 0, stream, batch);
  if (hub_blocks > 0 && partial_rows) hipLaunchKernelGGL(hub_reduce_kernel, dim3(hub_blocks), dim3(256), 0, stream, batch);
  if (red_blocks > 0) hipLaunchKernelGGL(dense_reduce_kernel, dim3(red_blocks), dim3(1024), 0, stream, batch);
  DG_HIP(hipGetLastError());
  return 0;
}

int launch_layer(const dg_layer_task *tasks, int32_t n_tasks, int transpose, hipStream_t stream) {
  dg_task_batch batch;
  const dg_graph *graphs[DG_MAX_TASKS];
  batch.n_tasks = n_tasks;
  for (int k = 0; k < n_tasks; ++k) {
    int rc = fill_task(tasks[k], transpose, batch.t[k]);
    if (rc) return rc;
    graphs[k] = tasks[k].graph;
  }
  return launch_batch(batch, graphs, stream);
}

}  // namespace

extern "C" int dg_gcn_layer_fwd(const dg_layer_task *tasks, int32_t n_tasks, int32_t d, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_gcn_layer_fwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!tasks || n_tasks <= 0 || n_tasks > DG_MAX_TASKS)
    return dg_fail(DG_ERR_ARG, "dg_gcn_layer_fwd: n_tasks must be in [1,%d] (got %d)", DG_MAX_TASKS, n_tasks);
  return launch_layer(tasks, n_tasks, 0, (hipStream_t)stream);
}

extern "C" int dg_spmm_norm_act_fwd(dg_graph *g, const float *x, float *y, int32_t d, float slope, int transpose,
                                    void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_spmm_norm_act_fwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!g || !x || !y) return dg_fail(DG_ERR_ARG, "dg_spmm_norm_act_fwd: NULL argument");
  dg_layer_task t;
  t.graph = g; t.x = x; t.e_out = y; t.acc_in = nullptr; t.acc_out = nullptr; t.flags = 0; t.slope = slope;
  t.partial_out = nullptr; t.partial_from = 0; t.row_norm_out = nullptr;
  return launch_layer(&t, 1, transpose, (hipStream_t)stream);
}

extern "C" int dg_gcn_partial_finalize_batched(const dg_finalize_task *tasks, int32_t n_tasks, int32_t d, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_gcn_partial_finalize: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!tasks || n_tasks <= 0 || n_tasks > DG_MAX_TASKS)
    return dg_fail(DG_ERR_ARG, "dg_gcn_partial_finalize: n_tasks must be in [1,%d] (got %d)", DG_MAX_TASKS, n_tasks);
  dg_task_batch batch;
  batch.n_tasks = 0;
  int32_t blocks = 0;
  for (int k = 0; k < n_tasks; ++k) {
    const dg_finalize_task &f = tasks[k];
    if (!f.graph || !f.partials || f.n_parts <= 0) return dg_fail(DG_ERR_ARG, "dg_gcn_partial_finalize: task %d: NULL argument", k);
    if (f.part_stride_rows < f.n_rows) return dg_fail(DG_ERR_ARG, "dg_gcn_partial_finalize: task %d: part stride smaller than a part", k);
    if (f.row_begin < 0 || f.n_rows < 0 || (int64_t)f.row_begin + f.n_rows > f.graph->n)
      return dg_fail(DG_ERR_ARG, "dg_gcn_partial_finalize: task %d: rows [%d,%d) outside the graph's %d nodes", k, f.row_begin,
                     f.row_begin + f.n_rows, f.graph->n);
    if ((f.flags & DG_ACC_REBUILD) && !f.x_prev)
      return dg_fail(DG_ERR_ARG, "dg_gcn_partial_finalize: task %d: DG_ACC_REBUILD needs x_prev", k);
    if (f.n_rows == 0) continue;
    dg_layer_task lt;
    lt.graph = f.graph; lt.x = f.x_prev ? f.x_prev : f.partials; lt.e_out = f.e_out; lt.acc_in = f.acc_in; lt.acc_out = f.acc_out;
    lt.flags = f.flags; lt.slope = f.slope; lt.partial_out = nullptr; lt.partial_from = 0; lt.row_norm_out = nullptr;
    dg_task_dev &t = batch.t[batch.n_tasks];
    int rc = fill_task(lt, 0, t);
    if (rc) return rc;
    t.fin_partials = f.partials; t.fin_parts = f.n_parts; t.fin_stride = f.part_stride_rows;
    t.fin_row_begin = f.row_begin; t.fin_rows = f.n_rows;
    t.blk_begin = blocks;
    blocks += (f.n_rows + 15) >> 4;
    batch.n_tasks++;
  }
  if (blocks == 0) return 0;
  for (int k = batch.n_tasks; k < DG_MAX_TASKS; ++k) batch.t[k] = batch.t[0];
  hipLaunchKernelGGL(partial_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, batch);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_gcn_partial_finalize(dg_graph *graph, const float *partials, int32_t n_parts,
                                       int64_t part_stride_rows, int32_t row_begin, int32_t n_rows, const float *x_prev,
                                       float *e_out, const float *acc_in, float *acc_out, int32_t flags, float slope,
                                       int32_t d, void *stream) {
  dg_finalize_task f;
  f.graph = graph; f.partials = partials; f.n_parts = n_parts; f.part_stride_rows = part_stride_rows;
  f.row_begin = row_begin; f.n_rows = n_rows; f.x_prev = x_prev; f.e_out = e_out; f.acc_in = acc_in; f.acc_out = acc_out;
  f.flags = flags; f.slope = slope;
  return dg_gcn_partial_finalize_batched(&f, 1, d, stream);
}

extern "C" int dg_gcn_layer_bwd(const dg_bwd_task *tasks, int32_t n_tasks, int32_t d, void *stream) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "dg_gcn_layer_bwd: embedding width %d unsupported (built for %d)", d, DG_D);
  if (!tasks || n_tasks <= 0 || n_tasks > DG_MAX_TASKS)
    return dg_fail(DG_ERR_ARG, "dg_gcn_layer_bwd: n_tasks must be in [1,%d] (got %d)", DG_MAX_TASKS, n_tasks);
  // tasks with a gather (w_in) go through the transposed propagation launch; the others are row-wise
  dg_task_batch gather, rowwise;
  const dg_graph *ggraphs[DG_MAX_TASKS];
  gather.n_tasks = rowwise.n_tasks = 0;
  int32_t row_blocks = 0, row_red_blocks = 0;
  for (int k = 0; k < n_tasks; ++k) {
    const dg_bwd_task &b = tasks[k];
    if (!b.graph || !b.w_out) return dg_fail(DG_ERR_ARG, "backward task %d: NULL graph or output", k);
    if (!b.w_in && !b.e_saved) return dg_fail(DG_ERR_ARG, "backward task %d: neither w_in nor e_saved given", k);
    if (b.e_saved && !b.d_sum) return dg_fail(DG_ERR_ARG, "backward task %d: e_saved needs d_sum", k);
    if (b.w_out == b.w_in) return dg_fail(DG_ERR_ARG, "backward task %d: w_out must not alias w_in", k);
    if ((b.flags & DG_BWD_NORM_FEED) && b.e_saved && !b.row_norm)
      return dg_fail(DG_ERR_ARG, "backward task %d: DG_BWD_NORM_FEED needs row_norm", k);
    dg_layer_task lt;
    lt.graph = b.graph; lt.x = b.w_in ? b.w_in : b.w_out; lt.e_out = b.w_out; lt.acc_in = nullptr; lt.acc_out = nullptr;
    lt.flags = DG_IN_PRESCALED | DG_EPI_BWD | (b.flags & (DG_BWD_SAVED_PRESCALED | DG_OUT_PLANES)); lt.slope = b.slope;
    if (b.w_in) lt.flags |= b.flags & DG_IN_PLANES;
    if (b.e_saved) lt.flags |= b.flags & DG_BWD_NORM_FEED;
    lt.partial_out = nullptr; lt.partial_from = 0; lt.row_norm_out = nullptr;
    dg_task_batch &dst = b.w_in ? gather : rowwise;
    dg_task_dev &t = dst.t[dst.n_tasks];
    lt.x = b.w_in ? b.w_in : b.d_sum;  // fill_task only checks x for NULL / aliasing
    int rc = fill_task(lt, 1, t);
    if (rc) return rc;
    t.e_saved = b.e_saved; t.d_sum = b.d_sum;
    t.row_norm = const_cast<float *>(b.row_norm);
    if (b.w_in) {
      ggraphs[dst.n_tasks] = b.graph;
    } else {
      t.blk_begin = row_blocks;
      t.red_blk_begin = row_red_blocks;
      t.red_parts = 0;   // nothing to add up: the epilogue alone
      if (planes_path(t, DG_OUT_PLANES)) row_red_blocks += t.dense_chunks;
      else row_blocks += (t.n + 15) >> 4;
    }
    dst.n_tasks++;
  }
  if (rowwise.n_tasks > 0) {
    for (int k = rowwise.n_tasks; k < DG_MAX_TASKS; ++k) rowwise.t[k] = rowwise.t[0];
    if (row_blocks > 0) hipLaunchKernelGGL(rowwise_kernel, dim3(row_blocks), dim3(256), 0, (hipStream_t)stream, rowwise);
    if (row_red_blocks > 0) hipLaunchKernelGGL(dense_reduce_kernel, dim3(row_red_blocks), dim3(1024), 0, (hipStream_t)stream, rowwise);
    DG_HIP(hipGetLastError());
  }
  if (gather.n_tasks > 0) return launch_batch(gather, ggraphs, (hipStream_t)stream);
  return 0;
}

extern "C" int dg_sum_tables(const float *const *in, int32_t n_in, float *out, int64_t n_elems, void *stream) {
  if (!in || !out || n_in < 1 || n_in > 16) return dg_fail(DG_ERR_ARG, "dg_sum_tables: n_in must be in [1,16]");
  if (n_elems % 4 != 0) return dg_fail(DG_ERR_ARG, "dg_sum_tables: element count must be a multiple of 4");
  sum_args a;
  a.n_in = n_in;
  for (int k = 0; k < 16; ++k) a.in[k] = in[k < n_in ? k : 0];
  const int64_t n4 = n_elems / 4;
  if (n4 == 0) return 0;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sum_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a,
                     reinterpret_cast<f32x4 *>(out), n4);
  DG_HIP(hipGetLastError());
  return 0;
}
