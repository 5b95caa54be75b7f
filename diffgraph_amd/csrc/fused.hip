// Row-fused propagation over a SET of graphs on the same nodes (gfx950).
//
// HGDM.forward (DiffGraph-Rec/Model.py:53-92) runs the same two-layer propagation over the target graph and over
// every auxiliary behaviour graph, all starting from the one embedding table, and only ever uses the target graph's
// layer sum and the SUM of the behaviour graphs' layer sums (Model.py:87).  Run graph by graph, every graph's
// epilogue re-reads the layer-0 row, writes its own layer sum, and a further pass adds the behaviour sums up.  Here
// one 16-lane group takes row r of ALL the graphs back to back: their neighbour lists lie one after the other in
// one padded list, the group adds each list into its own register sum, and then applies the graphs' epilogues
// together -- the layer-0 row is read once, the target sum and the behaviour sum are written once each, and the
// per-behaviour sums never exist in memory.  The graphs' activation tables are stacked, [graphs][n][64], so that
// the second layer's neighbour ids are pre-offset by graph and the whole list is gathered through one descriptor.
//
// Row classes as in spmm.hip: group rows (the padded lists of all graphs together hold at most quad_max quads of
// four slots) are summed whole by one group, four rows per wavefront; longer rows are cut into pieces at the graph
// boundaries, at the limits of 8 ranges of the stacked ids (one per XCD) and into at most seg_len neighbours, summed
// into slab rows by a wavefront or a group each, and a second small launch adds a row's slab rows graph by graph
// and applies the epilogues.  Every sum has a fixed order.
#include "dg_gather.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <vector>

struct dg_graphset {
  int32_t n = 0, n_graphs = 0;
  int64_t nnz = 0;
  int device = 0;
  int32_t quad_max = 0, seg_len = 0;
  // One padded neighbour list per row: graph 0's neighbours, padded with -1 to a multiple of four slots, then graph
  // 1's, ...; col_plain holds node ids (first layer: every graph gathers from the one table), col_stacked holds
  // graph * n + id (later layers: stacked tables); colscale the gathered node's norm in its graph (0: padding).
  int32_t *col_plain = nullptr, *col_stacked = nullptr;
  float *colscale = nullptr;
  // One 16-word record per node, by class of list length and node id: words 0..11 the first 12 slots, 12 the number
  // of quads (-1: a cut row, not summed here), 13 the row, 14 the quad counts after graphs 0, 1, 2 (one byte each,
  // top byte 0xff), 15 the position of slot 12 in the padded arrays.
  int32_t *head_plain = nullptr, *head_stacked = nullptr;
  float *head_scale = nullptr;
  float *norms = nullptr;      // [n][8]: the row's in-degree norm in graphs 0..3, then its out-degree norm
  int32_t *hub_desc = nullptr; // [n_hub][8]: cut row, first slab row of graphs 0..3, last slab row + 1, 0, 0
  int32_t *seg_desc = nullptr, *gseg_desc = nullptr;  // as dg_csr's
  int32_t n_hub = 0, n_seg = 0, n_seg_slots = 0, n_gseg_slots = 0;
  int32_t n_rec = 0;           // head records (group rows, plus skipped slots that pad the block order)
  int32_t side_split = 0;      // > 0: bipartite at this row, and the two sides' work is dealt to separate XCDs
  float *slab = nullptr;
};

namespace {

constexpr int kMaxSetGraphs = 4;
constexpr int kHeadSlots = 12;
constexpr int kGroupPiece = 32;
constexpr int kRuns = 8;

enum { M_FIRST = 0, M_LAST = 1, M_BWD_MID = 2, M_BWD_TOP = 3, M_BWD_LAST = 4, M_BWD_LAST_ADAM = 5 };

struct set_dev {
  const int32_t *col, *head, *hub_desc, *seg_desc, *gseg_desc;
  const float *colscale, *head_scale, *norms;
  const float *x;        // the table the neighbour rows come from ...
  int32_t x_rows;        // ... and its rows (n, or graphs * n when stacked)
  float *slab;
  int32_t n_seg;
  float *e_out;          // [graphs][n][64], or NULL
  const float *e0;       // M_LAST: the layer-0 table
  float *sum_first, *sum_rest;
  const float *e_saved;  // backward: the layer's forward activations, stacked
  const float *d_first, *d_rest;  // backward: gradients wrt the two sums
  int32_t n, n_graphs, n_hub, n_seg_slots, n_gseg_slots, n_rec;
  const int32_t *row_flags;   // set_rowwise_kernel: rows whose flag is 0 have zero gradients coming in (or NULL)
  float slope;
  // M_BWD_LAST_ADAM: the table the gradient belongs to, its Adam moments and state {steps taken, ..}
  float *adam_p, *adam_m, *adam_v;
  float *adam_state;
  float lr, beta1, beta2, eps, weight_decay;
};

struct set_ctx {
  float adam_t;    // M_BWD_LAST_ADAM: the step count of this update
  float sc;        // lanes 0..7 of the group: the row's norms
  f32x4 a, b;      // M_LAST: a = e0[row]; backward: gradients wrt the first / the other graphs' sums
  f32x4 own[kMaxSetGraphs];  // M_LAST: the row's own first-layer activations; backward: its saved activations
                             // (M_BWD_LAST_ADAM: the row of the parameter table and of its two moments)
};

// Loads only: the requests stay in flight while the neighbour rows are gathered.
template <int MODE>
__device__ __forceinline__ void load_set_ctx(const set_dev &T, int row, int l, set_ctx &c) {
  c.sc = 0.f;
  if (l < 8) c.sc = T.norms[(size_t)row * 8 + l];
  const size_t o = (size_t)row * (DG_D / 4) + l, gs = (size_t)T.n * (DG_D / 4);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  c.a = c.b = z;
#pragma unroll
  for (int g = 0; g < kMaxSetGraphs; ++g) c.own[g] = z;
  if (MODE == M_FIRST || MODE == M_BWD_LAST) return;
  if (MODE == M_BWD_LAST_ADAM) {
    c.a = load_stream(T.d_first, o);
    c.b = load_stream(T.d_rest, o);
    c.own[0] = load_stream(T.adam_p, o);
    c.own[1] = load_stream(T.adam_m, o);
    c.own[2] = load_stream(T.adam_v, o);
    return;
  }
  const float *own = MODE == M_LAST ? T.x : T.e_saved;
  if (MODE == M_LAST) {
    c.a = load_stream(T.e0, o);
  } else {
    c.a = load_stream(T.d_first, o);
    c.b = load_stream(T.d_rest, o);
  }
  c.own[0] = load_stream(own, o);
  c.own[1] = load_stream(own, gs + o);
  if (T.n_graphs > 2) c.own[2] = load_stream(own, 2 * gs + o);
  if (T.n_graphs > 3) c.own[3] = load_stream(own, 3 * gs + o);
}

__device__ __forceinline__ f32x4 leaky(f32x4 y, float slope) {
  y.x = y.x > 0.f ? y.x : y.x * slope;
  y.y = y.y > 0.f ? y.y : y.y * slope;
  y.z = y.z > 0.f ? y.z : y.z * slope;
  y.w = y.w > 0.f ? y.w : y.w * slope;
  return y;
}

// One graph's share of a row's epilogue; the arithmetic is that of epilogue() / epilogue_bwd() in spmm.hip.
// S receives the graph's layer sum (M_LAST).
template <int MODE, int GI>
__device__ __forceinline__ void set_epilogue_one(const set_dev &T, size_t o, const set_ctx &c, f32x4 acc, f32x4 &S) {
  const size_t og = (size_t)GI * (size_t)T.n * (DG_D / 4) + o;
  const float din = bcast16f<GI>(c.sc), dout = bcast16f<4 + GI>(c.sc);
  if (MODE == M_FIRST) {
    // rst * norm, LeakyReLU (Model.py:165-172), stored times the row's norm as a source (node_f * norm, Model.py:157)
    store_stream(T.e_out, og, leaky(acc * din, T.slope) * dout);
  } else if (MODE == M_LAST) {
    const f32x4 y = leaky(acc * din, T.slope);
    const float ss = sum16(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w);
    const f32x4 yn = y * inv_norm(ss);
    // e0 + normalize(E_1) from the stored p = dout * E_1: normalize(E_1) = p / max(||p||, 1e-12 * dout)
    const f32x4 p = c.own[GI];
    const float ps = sum16(p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w);
    const f32x4 a = c.a + p * fminf(__builtin_amdgcn_rsqf(ps), 1e12f * __builtin_amdgcn_rcpf(dout));
    S = a + yn;
    if (T.e_out) store_stream(T.e_out, og, y);
  } else {
    // backward roles: the row's norm as a SOURCE scales the gathered sum, its in-degree norm the stored result
    const float di = dout, dnext = din;
    const f32x4 G = acc * di;
    const f32x4 e = c.own[GI], g = GI == 0 ? c.a : c.b;
    const float cs = MODE == M_BWD_MID ? di : 1.0f;  // e = cs * E_l
    const float re = sqrtf(sum16(e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w));
    const float r = re / cs;
    f32x4 dE;
    if (r > 1e-12f) {
      const float inv_re = 1.0f / re, inv_r = 1.0f / r;
      const f32x4 nv = e * inv_re;
      const float ng = sum16(nv.x * g.x + nv.y * g.y + nv.z * g.z + nv.w * g.w);
      dE = (g - nv * ng) * inv_r;
    } else {
      dE = g * 1e12f;
    }
    dE += G;
    const float slope = T.slope;
    f32x4 dz;
    dz.x = e.x > 0.f ? dE.x : dE.x * slope;
    dz.y = e.y > 0.f ? dE.y : dE.y * slope;
    dz.z = e.z > 0.f ? dE.z : dE.z * slope;
    dz.w = e.w > 0.f ? dE.w : dE.w * slope;
    store_stream(T.e_out, og, dz * dnext);
  }
}

template <int MODE>
__device__ __forceinline__ void set_epilogue(const set_dev &T, int row, int l, const set_ctx &c, const f32x4 (&sums)[4]) {
  const size_t o = (size_t)row * (DG_D / 4) + l;
  if (MODE == M_BWD_LAST || MODE == M_BWD_LAST_ADAM) {
    // the last backward step is linear: d e0 gets sum_g dinv_g[row] * (sum over N_g(row) of dZ_1 of graph g), one table
    f32x4 g0 = sums[0] * bcast16f<4>(c.sc);
    g0 += sums[1] * bcast16f<5>(c.sc);
    if (T.n_graphs > 2) g0 += sums[2] * bcast16f<6>(c.sc);
    if (T.n_graphs > 3) g0 += sums[3] * bcast16f<7>(c.sc);
    if (MODE == M_BWD_LAST) {
      store_stream(T.sum_first, o, g0);
      return;
    }
    // ... and the table's Adam step on the spot.  Every graph's layer sum starts from e0 (Model.py:57,74), so the
    // gradient is d_first + d_rest (once per behaviour graph) + g0, added left to right; then torch.optim.Adam's
    // update with the step count t = steps taken + 1 -- the arithmetic, term for term, of adam_prepare_kernel and
    // adam_table_kernel (optim.hip), which form the same sum from the same tables: the same bits.
    f32x4 g = c.a + c.b;
    if (T.n_graphs > 2) g += c.b;
    if (T.n_graphs > 3) g += c.b;
    g += g0;
    const float t = c.adam_t;
    const float bc1 = 1.0f - powf(T.beta1, t), bc2_sqrt = sqrtf(1.0f - powf(T.beta2, t));
    const float step_size = T.lr / bc1;
    f32x4 p = c.own[0], m = c.own[1], v = c.own[2];
    if (T.weight_decay != 0.f) g += p * T.weight_decay;
    m = m + (g - m) * (1.0f - T.beta1);
    v = v * T.beta2 + g * g * (1.0f - T.beta2);
    f32x4 dn;
    dn.x = sqrtf(v.x) / bc2_sqrt + T.eps; dn.y = sqrtf(v.y) / bc2_sqrt + T.eps;
    dn.z = sqrtf(v.z) / bc2_sqrt + T.eps; dn.w = sqrtf(v.w) / bc2_sqrt + T.eps;
    p.x -= step_size * (m.x / dn.x); p.y -= step_size * (m.y / dn.y);
    p.z -= step_size * (m.z / dn.z); p.w -= step_size * (m.w / dn.w);
    store_stream(T.adam_p, o, p);
    store_stream(T.adam_m, o, m);
    store_stream(T.adam_v, o, v);
    return;
  }
  f32x4 S0, S1, S2, S3;
  S0 = S1 = S2 = S3 = f32x4{0.f, 0.f, 0.f, 0.f};
  set_epilogue_one<MODE, 0>(T, o, c, sums[0], S0);
  set_epilogue_one<MODE, 1>(T, o, c, sums[1], S1);
  if (T.n_graphs > 2) set_epilogue_one<MODE, 2>(T, o, c, sums[2], S2);
  if (T.n_graphs > 3) set_epilogue_one<MODE, 3>(T, o, c, sums[3], S3);
  if (MODE == M_LAST) {
    store_stream(T.sum_first, o, S0);
    // sum(list) over the behaviour graphs, left to right (Model.py:87)
    f32x4 rest = S1;
    if (T.n_graphs > 2) rest += S2;
    if (T.n_graphs > 3) rest += S3;
    store_stream(T.sum_rest, o, rest);
  }
}

// The padded list of a group row, a quad at a time; when the quad count reaches the end of a graph's share the
// running sum moves down a three-deep stack and starts again from zero -- exactly three times per row (empty
// shares move it at once) -- so that at the end the stack and the running sum hold graphs 0..3 in order.  All
// groups of a wavefront are at the same slot of their own lists, so the neighbour ids come out of the group's
// 16-id chunk with compile-time DPP selectors; only the moves diverge between groups.
template <bool SCALE>
__device__ __forceinline__ void walk_group_row(__amdgpu_buffer_rsrc_t xr, const int32_t *__restrict__ col,
                                               const float *__restrict__ colscale, int hv, float hs, int total, unsigned bnd,
                                               int tail, int l, f32x4 (&sums)[4]) {
  const unsigned lane_off = (unsigned)l * 16u;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc = z, s1 = z, s2 = z, s3 = z;
  int nb = (int)(bnd & 255u);
#define DG_SET_MOVE(SQ)                                                  \
  while ((SQ) == nb) {                                                   \
    s3 = s2; s2 = s1; s1 = acc; acc = z;                                 \
    bnd >>= 8; nb = (int)(bnd & 255u);                                   \
  }
  DG_SET_MOVE(0)
  const int c = l < kHeadSlots ? hv : -1;
  if (total > 0) { gather4<0, SCALE>(xr, c, hs, lane_off, acc); DG_SET_MOVE(1) }
  if (total > 1) { gather4<4, SCALE>(xr, c, hs, lane_off, acc); DG_SET_MOVE(2) }
  if (total > 2) { gather4<8, SCALE>(xr, c, hs, lane_off, acc); DG_SET_MOVE(3) }
  if (total > 3) {
    int p = tail, sq = 3;
    const int end = tail + 4 * (total - 3);
    int cc = -1;
    float sc = 0.f;
    if (p + l < end) {
      cc = col[p + l];
      if (SCALE) sc = colscale[p + l];
    }
    while (p < end) {
      const int pn = p + 16;
      // the next chunk's ids are on their way while this chunk's rows load (an unconditional load: see gather_rows)
      const int at = min(pn + l, end - 1);
      int cn = col[at];
      float sn = SCALE ? colscale[at] : 0.f;
      if (pn + l >= end) {
        cn = -1;
        sn = 0.f;
      }
      const int cnt = end - p;
      gather4<0, SCALE>(xr, cc, sc, lane_off, acc); DG_SET_MOVE(sq + 1)
      if (cnt > 4) { gather4<4, SCALE>(xr, cc, sc, lane_off, acc); DG_SET_MOVE(sq + 2) }
      if (cnt > 8) { gather4<8, SCALE>(xr, cc, sc, lane_off, acc); DG_SET_MOVE(sq + 3) }
      if (cnt > 12) { gather4<12, SCALE>(xr, cc, sc, lane_off, acc); DG_SET_MOVE(sq + 4) }
      sq += 4; cc = cn; sc = sn; p = pn;
    }
  }
#undef DG_SET_MOVE
  sums[0] = s3; sums[1] = s2; sums[2] = s1; sums[3] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void set_layer_kernel(const set_dev T) {
  constexpr bool SCALE = MODE == M_FIRST;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, l = lane & 15;
  const int nb_seg = T.n_seg_slots >> 2, nb_gseg = T.n_gseg_slots >> 4;
  const int lb = blockIdx.x;
  const __amdgpu_buffer_rsrc_t xr = table_rsrc(T.x, T.x_rows);
  // The step count moves on inside the two launches, with no launch of its own and no race: this launch reads
  // state[0] (steps taken) and leaves the new count in the spare state[3]; the cut rows' launch, which follows it,
  // reads state[3] and writes it to state[0].  Nobody reads, within a launch, the word that launch writes.
  if (MODE == M_BWD_LAST_ADAM && lb == 0 && threadIdx.x == 0) T.adam_state[3] = T.adam_state[0] + 1.0f;
  if (lb < nb_seg) {
    const int4 d = reinterpret_cast<const int4 *>(T.seg_desc)[lb * 4 + wave];  // begin, end, slab row
    if (d.z >= 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      gather_rows<true, SCALE>(xr, T.col, T.colscale, d.x + 16 * q, d.y, 64, l, acc);
      acc = add_groups(acc);
      if (q == 0) reinterpret_cast<f32x4 *>(T.slab)[(size_t)d.z * (DG_D / 4) + l] = acc;
    }
  } else if (lb < nb_seg + nb_gseg) {
    const int4 d = reinterpret_cast<const int4 *>(T.gseg_desc)[(lb - nb_seg) * 16 + wave * 4 + q];
    if (d.z >= 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      gather_rows<true, SCALE>(xr, T.col, T.colscale, d.x, d.y, 16, l, acc);
      reinterpret_cast<f32x4 *>(T.slab)[(size_t)d.z * (DG_D / 4) + l] = acc;
    }
  } else {
    const int k = (lb - nb_seg - nb_gseg) * 16 + wave * 4 + q;
    if (k < T.n_rec) {
      const int hv = T.head[(size_t)k * 16 + l];
      const float hs = SCALE ? T.head_scale[(size_t)k * 16 + l] : 0.f;
      const int total = bcast16<12>(hv), row = bcast16<13>(hv), tail = bcast16<15>(hv);
      const unsigned bnd = (unsigned)bcast16<14>(hv);
      if (total >= 0) {
        set_ctx c;
        load_set_ctx<MODE>(T, row, l, c);
        if (MODE == M_BWD_LAST_ADAM) c.adam_t = T.adam_state[0] + 1.0f;
        f32x4 sums[4];
        walk_group_row<SCALE>(xr, T.col, T.colscale, hv, hs, total, bnd, tail, l, sums);
        set_epilogue<MODE>(T, row, l, c, sums);
      }
    }
  }
}

// Cut rows: one wavefront per row adds, graph by graph, the row's slab rows in order (four groups take every
// fourth chunk of 16), then one group applies the epilogues.
template <int MODE>
__global__ __launch_bounds__(256) void set_hub_kernel(const set_dev T) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, l = lane & 15;
  const int h = blockIdx.x * 4 + wave;
  if (MODE == M_BWD_LAST_ADAM && blockIdx.x == 0 && threadIdx.x == 0) T.adam_state[0] = T.adam_state[3];
  if (h >= T.n_hub) return;
  const int4 d0 = reinterpret_cast<const int4 *>(T.hub_desc)[2 * h], d1 = reinterpret_cast<const int4 *>(T.hub_desc)[2 * h + 1];
  const int row = d0.x;
  const int sb[5] = {d0.y, d0.z, d0.w, d1.x, d1.y};
  set_ctx c;
  load_set_ctx<MODE>(T, row, l, c);
  if (MODE == M_BWD_LAST_ADAM) c.adam_t = T.adam_state[3];
  const __amdgpu_buffer_rsrc_t sr = table_rsrc(T.slab, T.n_seg);
  f32x4 sums[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    gather_rows<false, false>(sr, nullptr, nullptr, sb[g] + 16 * q, sb[g + 1], 64, l, acc);
    sums[g] = add_groups(acc);
  }
  if (q == 0) set_epilogue<MODE>(T, row, l, c, sums);
}

// The top step of the backward pass has nothing to gather: the epilogues alone, one group per row.
__global__ __launch_bounds__(256) void set_rowwise_kernel(const set_dev T) {
  const int l = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= T.n) return;
  if (T.row_flags != nullptr && T.row_flags[row] == 0) {
    // no gradient reaches this row's layer sums: dZ is zero whatever the row holds -- written without reading it
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const size_t o = (size_t)row * (DG_D / 4) + l, gs = (size_t)T.n * (DG_D / 4);
    for (int g = 0; g < T.n_graphs; ++g) store_stream(T.e_out, g * gs + o, z);
    return;
  }
  set_ctx c;
  load_set_ctx<M_BWD_TOP>(T, row, l, c);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sums[4] = {z, z, z, z};
  set_epilogue<M_BWD_TOP>(T, row, l, c, sums);
}

template <class V>
hipError_t upload(const std::vector<V> &v, V **dst) {
  *dst = nullptr;
  hipError_t e = hipMalloc((void **)dst, std::max<size_t>(v.size(), 1) * sizeof(V));
  if (e != hipSuccess) return e;
  if (!v.empty()) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice);
  return e;
}

template <class V>
hipError_t download(const V *src, size_t count, std::vector<V> &v) {
  v.resize(count);
  if (count == 0) return hipSuccess;
  return hipMemcpy(v.data(), src, count * sizeof(V), hipMemcpyDeviceToHost);
}

int fill(const dg_graphset *s, int mode, set_dev &t) {
  t.col = mode == M_FIRST ? s->col_plain : s->col_stacked;
  t.head = mode == M_FIRST ? s->head_plain : s->head_stacked;
  t.hub_desc = s->hub_desc; t.seg_desc = s->seg_desc; t.gseg_desc = s->gseg_desc;
  t.colscale = s->colscale; t.head_scale = s->head_scale; t.norms = s->norms;
  t.x = nullptr; t.x_rows = mode == M_FIRST ? s->n : s->n * s->n_graphs;
  t.slab = s->slab; t.n_seg = s->n_seg;
  t.e_out = nullptr; t.e0 = nullptr; t.sum_first = t.sum_rest = nullptr;
  t.e_saved = nullptr; t.d_first = t.d_rest = nullptr;
  t.n = s->n; t.n_graphs = s->n_graphs; t.n_hub = s->n_hub; t.n_seg_slots = s->n_seg_slots; t.n_gseg_slots = s->n_gseg_slots;
  t.n_rec = s->n_rec;
  t.slope = 0.f;
  t.row_flags = nullptr;
  t.adam_p = t.adam_m = t.adam_v = nullptr; t.adam_state = nullptr;
  t.lr = t.beta1 = t.beta2 = t.eps = t.weight_decay = 0.f;
  int device = 0;
  DG_HIP(hipGetDevice(&device));
  if (device != s->device)
    return dg_fail(DG_ERR_ARG, "graph set: created on device %d, the current device is %d", s->device, device);
  return 0;
}

template <int MODE>
int launch(const dg_graphset *s, const set_dev &t, hipStream_t stream) {
  const int blocks = (s->n_seg_slots >> 2) + (s->n_gseg_slots >> 4) + ((s->n_rec + 15) >> 4);
  hipLaunchKernelGGL(set_layer_kernel<MODE>, dim3(blocks), dim3(256), 0, stream, t);
  // (with the optimizer step inside, the second launch also moves the step count on: it runs even without cut rows)
  if (s->n_hub > 0 || MODE == M_BWD_LAST_ADAM)
    hipLaunchKernelGGL(set_hub_kernel<MODE>, dim3(std::max((s->n_hub + 3) >> 2, 1)), dim3(256), 0, stream, t);
  DG_HIP(hipGetLastError());
  return 0;
}

int check_d(const char *what, const dg_graphset *s, int32_t d) {
  if (d != DG_D) return dg_fail(DG_ERR_DIM, "%s: embedding width %d unsupported (built for %d)", what, d, DG_D);
  if (!s) return dg_fail(DG_ERR_ARG, "%s: graph set is NULL", what);
  return 0;
}

}  // namespace

extern "C" int dg_graphset_create(dg_graph *const *graphs, int32_t n_graphs, int32_t group_max, int32_t seg_len,
                                  dg_graphset **out) {
  if (!out) return dg_fail(DG_ERR_ARG, "dg_graphset_create: out is NULL");
  *out = nullptr;
  if (!graphs || n_graphs < 2 || n_graphs > kMaxSetGraphs)
    return dg_fail(DG_ERR_ARG, "dg_graphset_create: a set holds 2 to %d graphs (got %d)", kMaxSetGraphs, n_graphs);
  if (group_max <= 0) group_max = 128;
  if (seg_len <= 0) seg_len = 256;
  if (seg_len < 64 || seg_len % 64 != 0)
    return dg_fail(DG_ERR_ARG, "dg_graphset_create: seg_len must be a positive multiple of 64 (got %d)", seg_len);
  const int32_t quad_max = std::min(std::max(group_max / 4, 3), 254);
  int device = 0;
  DG_HIP(hipGetDevice(&device));
  for (int g = 0; g < n_graphs; ++g) {
    if (!graphs[g]) return dg_fail(DG_ERR_ARG, "dg_graphset_create: graph %d is NULL", g);
    if (graphs[g]->n != graphs[0]->n)
      return dg_fail(DG_ERR_ARG, "dg_graphset_create: graph %d has %d nodes, graph 0 has %d", g, graphs[g]->n, graphs[0]->n);
    if (!graphs[g]->symmetric || graphs[g]->weighted)
      return dg_fail(DG_ERR_ARG, "dg_graphset_create: graph %d is not a symmetric unweighted graph", g);
    if (graphs[g]->device != device)
      return dg_fail(DG_ERR_ARG, "dg_graphset_create: graph %d lives on device %d, the current device is %d", g,
                     graphs[g]->device, device);
  }
  const int32_t n = graphs[0]->n;
  if ((int64_t)n * n_graphs * DG_D * 4 >= ((int64_t)1 << 32))
    return dg_fail(DG_ERR_SIZE, "dg_graphset_create: %d graphs of %d nodes exceed the 4 GiB limit of a stacked table", n_graphs, n);

  std::vector<int32_t> rowptr[kMaxSetGraphs], colidx[kMaxSetGraphs];
  std::vector<float> dsrc[kMaxSetGraphs], ddst[kMaxSetGraphs];
  int64_t nnz = 0;
  for (int g = 0; g < n_graphs; ++g) {
    const dg_graph *gr = graphs[g];
    DG_HIP(download(gr->fwd.rowptr, (size_t)n + 1, rowptr[g]));
    DG_HIP(download(gr->fwd.colidx, (size_t)gr->nnz, colidx[g]));
    DG_HIP(download(gr->dinv_src, (size_t)n, dsrc[g]));
    DG_HIP(download(gr->dinv_dst, (size_t)n, ddst[g]));
    nnz += gr->nnz;
  }
  // the padded lists
  std::vector<int64_t> base((size_t)n + 1, 0);
  std::vector<int32_t> quads((size_t)n, 0);
  for (int32_t r = 0; r < n; ++r) {
    int32_t qn = 0;
    for (int g = 0; g < n_graphs; ++g) qn += (rowptr[g][r + 1] - rowptr[g][r] + 3) / 4;
    quads[r] = qn;
    base[r + 1] = base[r] + 4 * (int64_t)qn;
  }
  if (base[n] >= (int64_t)INT32_MAX) return dg_fail(DG_ERR_SIZE, "dg_graphset_create: the padded lists exceed the int32 limit");
  std::vector<int32_t> col_plain((size_t)base[n], -1), col_stacked((size_t)base[n], -1);
  std::vector<float> colscale((size_t)base[n], 0.f);
  std::vector<float> norms((size_t)n * 8, 1.0f);
  for (int32_t r = 0; r < n; ++r) {
    int64_t p = base[r];
    for (int g = 0; g < n_graphs; ++g) {
      const int32_t b = rowptr[g][r], d = rowptr[g][r + 1] - b;
      for (int32_t j = 0; j < d; ++j) {
        const int32_t c = colidx[g][b + j];
        col_plain[(size_t)(p + j)] = c;
        col_stacked[(size_t)(p + j)] = g * n + c;
        colscale[(size_t)(p + j)] = dsrc[g][c];
      }
      p += 4 * (int64_t)((d + 3) / 4);
      norms[(size_t)r * 8 + g] = ddst[g][r];
      norms[(size_t)r * 8 + 4 + g] = dsrc[g][r];
    }
  }
  // Bipartite in every graph at the same row (makeBiAdj's block form, users then items)?  Then the two sides' rows
  // gather from disjoint halves of the tables, and (DG_SET_SPLIT=1) their work is dealt to separate XCDs -- workgroup
  // b runs on XCD b mod 8 --: rows below the split to XCDs 0..3, the others to XCDs 4..7, so that an XCD's 4 MB of L2
  // serves one half of the tables only.  Both sides hold the same number of edge ends, so the halves weigh the same.
  int32_t split = n;
  for (int g = 0; g < n_graphs; ++g)
    for (int32_t r = 0; r < n; ++r)
      if (rowptr[g][r + 1] > rowptr[g][r] && colidx[g][rowptr[g][r]] > r) split = std::min(split, colidx[g][rowptr[g][r]]);
  if (split >= n) split = 0;
  for (int g = 0; g < n_graphs && split > 0; ++g)
    for (int32_t r = 0; r < n; ++r) {
      if (rowptr[g][r + 1] == rowptr[g][r]) continue;
      const int32_t lo = colidx[g][rowptr[g][r]], hi = colidx[g][rowptr[g][r + 1] - 1];
      if (r < split ? lo < split : hi >= split) { split = 0; break; }
    }
  const char *split_env = getenv("DG_SET_SPLIT");
  const bool side_split = split > 0 && split_env && atoi(split_env) > 0;
  // cut rows: pieces by graph, by range of the stacked ids, by length
  std::vector<int32_t> hub_rows;
  for (int32_t r = 0; r < n; ++r)
    if (quads[r] > quad_max) hub_rows.push_back(r);
  std::stable_sort(hub_rows.begin(), hub_rows.end(), [&](int32_t a, int32_t b) { return quads[a] > quads[b]; });
  // range limits of the runs: all 8 runs over all cut rows -- or, split by side, runs 0..3 over the cut rows below
  // the split and runs 4..7 over the others
  int32_t limit[2][kRuns + 1];
  for (int side = 0; side < 2; ++side) {
    std::vector<int32_t> ids;
    for (int32_t r : hub_rows) {
      if (side_split ? (r < split) != (side == 0) : side == 1) continue;
      for (int64_t p = base[r]; p < base[r + 1]; ++p)
        if (col_stacked[(size_t)p] >= 0) ids.push_back(col_stacked[(size_t)p]);
    }
    const int runs = side_split ? kRuns / 2 : kRuns;
    limit[side][0] = 0;
    for (int x = 1; x <= kRuns; ++x) limit[side][x] = INT32_MAX;
    for (int x = 1; x < runs; ++x) {
      if (ids.empty()) continue;
      const size_t at = ids.size() * (size_t)x / runs;
      std::nth_element(ids.begin(), ids.begin() + at, ids.end());
      limit[side][x] = ids[at];
    }
  }
  struct Piece { int32_t begin, end, slab; };
  std::vector<Piece> wave_pieces[kRuns], group_pieces[kRuns];
  std::vector<int32_t> hub_desc;
  int32_t n_seg = 0;
  for (int32_t r : hub_rows) {
    int32_t first[kMaxSetGraphs + 1];
    int64_t p0 = base[r];
    const int side = side_split && r >= split ? 1 : 0;
    const int runs = side_split ? kRuns / 2 : kRuns, run0 = side * (kRuns / 2);
    for (int g = 0; g < kMaxSetGraphs; ++g) {
      first[g] = n_seg;
      if (g >= n_graphs) continue;
      const int32_t d = rowptr[g][r + 1] - rowptr[g][r];
      const int32_t *cs = col_stacked.data();
      int32_t p = (int32_t)p0;
      const int32_t e = p + d;
      for (int x = 0; x < runs && p < e; ++x) {
        const int32_t pe = x + 1 == runs ? e : (int32_t)(std::lower_bound(cs + p, cs + e, limit[side][x + 1]) - cs);
        const int32_t len = pe - p;
        if (len <= 0) continue;
        const int32_t parts = (len + seg_len - 1) / seg_len;
        for (int32_t k = 0; k < parts; ++k) {
          const int32_t b0 = p + (int32_t)((int64_t)len * k / parts), b1 = p + (int32_t)((int64_t)len * (k + 1) / parts);
          (b1 - b0 > kGroupPiece ? wave_pieces : group_pieces)[run0 + x].push_back({b0, b1, n_seg++});
        }
        p = pe;
      }
      p0 += 4 * (int64_t)((d + 3) / 4);
    }
    first[kMaxSetGraphs] = n_seg;
    hub_desc.insert(hub_desc.end(), {r, first[0], first[1], first[2], first[3], first[4], 0, 0});
  }
  std::vector<int32_t> seg_desc, gseg_desc;
  int32_t n_seg_slots = 0, n_gseg_slots = 0;
  auto lay_out = [&](std::vector<Piece> (&runs)[kRuns], int per_block, std::vector<int32_t> &desc, int32_t &n_slots) {
    size_t longest = 0;
    for (int x = 0; x < kRuns; ++x) longest = std::max(longest, runs[x].size());
    const int32_t blocks_per_run = (int32_t)((longest + per_block - 1) / per_block);
    n_slots = blocks_per_run * kRuns * per_block;
    desc.assign((size_t)n_slots * 4, 0);
    for (int32_t slot = 0; slot < n_slots; ++slot) desc[(size_t)slot * 4 + 2] = -1;
    for (int x = 0; x < kRuns; ++x)
      for (size_t i = 0; i < runs[x].size(); ++i) {
        // run x's pieces go to blocks x, x + 8, ... of the class: those share an XCD and its L2
        const size_t slot = ((i / per_block) * kRuns + x) * per_block + i % per_block;
        desc[slot * 4 + 0] = runs[x][i].begin;
        desc[slot * 4 + 1] = runs[x][i].end;
        desc[slot * 4 + 2] = runs[x][i].slab;
      }
  };
  lay_out(wave_pieces, 4, seg_desc, n_seg_slots);
  lay_out(group_pieces, 16, gseg_desc, n_gseg_slots);
  // head records, by class of list length (rows that share a wavefront then walk lists of like length), node order inside
  std::vector<int32_t> order((size_t)n);
  for (int32_t r = 0; r < n; ++r) order[r] = r;
  auto list_class = [&](int32_t r) {
    const int32_t qn = quads[r];
    if (qn > quad_max) return 7;   // cut rows: skipped, last
    return qn > 24 ? 0 : qn > 16 ? 1 : qn > 11 ? 2 : qn > 7 ? 3 : qn > 3 ? 4 : qn > 0 ? 5 : 6;
  };
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return list_class(a) < list_class(b); });
  if (side_split) {
    // blocks of 16 records in rounds of 8: the first four of a round take rows below the split, the other four rows
    // above it (slots left over when a side runs out stay empty: -1 = skip)
    std::vector<int32_t> sides[2];
    for (int32_t r : order)
      if (quads[r] <= quad_max) sides[r < split ? 0 : 1].push_back(r);
    const size_t rounds = std::max((sides[0].size() + 63) / 64, (sides[1].size() + 63) / 64);
    std::vector<int32_t> dealt(rounds * 128, -1);
    for (int side = 0; side < 2; ++side)
      for (size_t i = 0; i < sides[side].size(); ++i)
        dealt[(i / 64) * 128 + side * 64 + i % 64] = sides[side][i];
    order.swap(dealt);
  }
  const int32_t n_rec = (int32_t)order.size();
  std::vector<int32_t> head_plain((size_t)n_rec * 16, -1), head_stacked((size_t)n_rec * 16, -1);
  std::vector<float> head_scale((size_t)n_rec * 16, 0.f);
  for (int32_t k = 0; k < n_rec; ++k) {
    const int32_t r = order[k];
    int32_t *hp = &head_plain[(size_t)k * 16], *hk = &head_stacked[(size_t)k * 16];
    if (r < 0) continue;                // an empty slot: word 12 stays -1
    hp[13] = hk[13] = r;
    if (quads[r] > quad_max) continue;  // word 12 stays -1
    float *hs = &head_scale[(size_t)k * 16];
    const int32_t slots = std::min(4 * quads[r], kHeadSlots);
    for (int32_t j = 0; j < slots; ++j) {
      hp[j] = col_plain[(size_t)(base[r] + j)];
      hk[j] = col_stacked[(size_t)(base[r] + j)];
      hs[j] = colscale[(size_t)(base[r] + j)];
    }
    uint32_t bnd = 0xff000000u, cum = 0;
    for (int g = 0; g < 3; ++g) {
      if (g < n_graphs) cum += (uint32_t)((rowptr[g][r + 1] - rowptr[g][r] + 3) / 4);
      bnd |= cum << (8 * g);
    }
    hp[12] = hk[12] = quads[r];
    hp[14] = hk[14] = (int32_t)bnd;
    hp[15] = hk[15] = (int32_t)(base[r] + kHeadSlots);
  }

  dg_graphset *s = new dg_graphset();
  s->n = n; s->n_graphs = n_graphs; s->nnz = nnz; s->device = device; s->quad_max = quad_max; s->seg_len = seg_len;
  s->n_hub = (int32_t)hub_rows.size(); s->n_seg = n_seg; s->n_seg_slots = n_seg_slots; s->n_gseg_slots = n_gseg_slots;
  s->n_rec = n_rec; s->side_split = side_split ? split : 0;
  hipError_t e = upload(col_plain, &s->col_plain);
  if (e == hipSuccess) e = upload(col_stacked, &s->col_stacked);
  if (e == hipSuccess) e = upload(colscale, &s->colscale);
  if (e == hipSuccess) e = upload(head_plain, &s->head_plain);
  if (e == hipSuccess) e = upload(head_stacked, &s->head_stacked);
  if (e == hipSuccess) e = upload(head_scale, &s->head_scale);
  if (e == hipSuccess) e = upload(norms, &s->norms);
  if (e == hipSuccess) e = upload(hub_desc, &s->hub_desc);
  if (e == hipSuccess) e = upload(seg_desc, &s->seg_desc);
  if (e == hipSuccess) e = upload(gseg_desc, &s->gseg_desc);
  if (e == hipSuccess) e = hipMalloc((void **)&s->slab, std::max<size_t>((size_t)n_seg, 1) * DG_D * sizeof(float));
  if (e != hipSuccess) {
    dg_graphset_destroy(s);
    return dg_fail((int)e, "dg_graphset_create: device allocation/copy failed: %s", hipGetErrorString(e));
  }
  *out = s;
  return 0;
}

extern "C" void dg_graphset_destroy(dg_graphset *s) {
  if (!s) return;
  (void)hipFree(s->col_plain); (void)hipFree(s->col_stacked); (void)hipFree(s->colscale);
  (void)hipFree(s->head_plain); (void)hipFree(s->head_stacked); (void)hipFree(s->head_scale);
  (void)hipFree(s->norms); (void)hipFree(s->hub_desc); (void)hipFree(s->seg_desc); (void)hipFree(s->gseg_desc);
  (void)hipFree(s->slab);
  delete s;
}

extern "C" int dg_graphset_counts(const dg_graphset *s, int32_t *counts_host) {
  if (!s || !counts_host) return dg_fail(DG_ERR_ARG, "dg_graphset_counts: NULL argument");
  counts_host[0] = s->n; counts_host[1] = s->n_graphs; counts_host[2] = s->n_hub; counts_host[3] = s->n_seg;
  counts_host[4] = s->quad_max; counts_host[5] = s->seg_len; counts_host[6] = s->side_split;
  return 0;
}

extern "C" int dg_set_layer_first(dg_graphset *set, const float *x, float *e_out, float slope, int32_t d, void *stream) {
  int rc = check_d("dg_set_layer_first", set, d);
  if (rc) return rc;
  if (!x || !e_out) return dg_fail(DG_ERR_ARG, "dg_set_layer_first: NULL argument");
  set_dev t;
  if ((rc = fill(set, M_FIRST, t))) return rc;
  t.x = x; t.e_out = e_out; t.slope = slope;
  return launch<M_FIRST>(set, t, (hipStream_t)stream);
}

extern "C" int dg_set_layer_last(dg_graphset *set, const float *x, const float *e0, float *sum_first, float *sum_rest,
                                 float *e_out, float slope, int32_t d, void *stream) {
  int rc = check_d("dg_set_layer_last", set, d);
  if (rc) return rc;
  if (!x || !e0 || !sum_first || !sum_rest) return dg_fail(DG_ERR_ARG, "dg_set_layer_last: NULL argument");
  if (e_out == x) return dg_fail(DG_ERR_ARG, "dg_set_layer_last: e_out must not alias x");
  set_dev t;
  if ((rc = fill(set, M_LAST, t))) return rc;
  t.x = x; t.e0 = e0; t.sum_first = sum_first; t.sum_rest = sum_rest; t.e_out = e_out; t.slope = slope;
  return launch<M_LAST>(set, t, (hipStream_t)stream);
}

extern "C" int dg_set_layer_bwd_top(dg_graphset *set, const float *e_saved, const float *d_first, const float *d_rest,
                                    float *w_out, const int32_t *row_flags, float slope, int32_t d, void *stream) {
  int rc = check_d("dg_set_layer_bwd_top", set, d);
  if (rc) return rc;
  if (!e_saved || !d_first || !d_rest || !w_out) return dg_fail(DG_ERR_ARG, "dg_set_layer_bwd_top: NULL argument");
  set_dev t;
  if ((rc = fill(set, M_BWD_TOP, t))) return rc;
  t.e_saved = e_saved; t.d_first = d_first; t.d_rest = d_rest; t.e_out = w_out; t.slope = slope; t.row_flags = row_flags;
  hipLaunchKernelGGL(set_rowwise_kernel, dim3((set->n + 15) >> 4), dim3(256), 0, (hipStream_t)stream, t);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_set_layer_bwd_mid(dg_graphset *set, const float *w_in, const float *e_saved, const float *d_first,
                                    const float *d_rest, float *w_out, float slope, int32_t d, void *stream) {
  int rc = check_d("dg_set_layer_bwd_mid", set, d);
  if (rc) return rc;
  if (!w_in || !e_saved || !d_first || !d_rest || !w_out) return dg_fail(DG_ERR_ARG, "dg_set_layer_bwd_mid: NULL argument");
  if (w_out == w_in) return dg_fail(DG_ERR_ARG, "dg_set_layer_bwd_mid: w_out must not alias w_in (e_saved it may: a row is read, then written, by the same lanes)");
  set_dev t;
  if ((rc = fill(set, M_BWD_MID, t))) return rc;
  t.x = w_in; t.e_saved = e_saved; t.d_first = d_first; t.d_rest = d_rest; t.e_out = w_out; t.slope = slope;
  return launch<M_BWD_MID>(set, t, (hipStream_t)stream);
}

extern "C" int dg_set_layer_bwd_last(dg_graphset *set, const float *w_in, float *g_out, int32_t d, void *stream) {
  int rc = check_d("dg_set_layer_bwd_last", set, d);
  if (rc) return rc;
  if (!w_in || !g_out) return dg_fail(DG_ERR_ARG, "dg_set_layer_bwd_last: NULL argument");
  set_dev t;
  if ((rc = fill(set, M_BWD_LAST, t))) return rc;
  t.x = w_in; t.sum_first = g_out; t.slope = 1.0f;
  return launch<M_BWD_LAST>(set, t, (hipStream_t)stream);
}

extern "C" int dg_set_layer_bwd_last_adam(dg_graphset *set, const float *w_in, const float *d_first, const float *d_rest,
                                          float *param, float *exp_avg, float *exp_avg_sq, float *state, float lr,
                                          float beta1, float beta2, float eps, float weight_decay, int32_t d, void *stream) {
  int rc = check_d("dg_set_layer_bwd_last_adam", set, d);
  if (rc) return rc;
  if (!w_in || !d_first || !d_rest || !param || !exp_avg || !exp_avg_sq || !state)
    return dg_fail(DG_ERR_ARG, "dg_set_layer_bwd_last_adam: NULL argument");
  if (param == w_in) return dg_fail(DG_ERR_ARG, "dg_set_layer_bwd_last_adam: the parameter table must not be the gathered table");
  set_dev t;
  if ((rc = fill(set, M_BWD_LAST, t))) return rc;
  t.x = w_in; t.d_first = d_first; t.d_rest = d_rest; t.slope = 1.0f;
  t.adam_p = param; t.adam_m = exp_avg; t.adam_v = exp_avg_sq; t.adam_state = state;
  t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.weight_decay = weight_decay;
  return launch<M_BWD_LAST_ADAM>(set, t, (hipStream_t)stream);
}
