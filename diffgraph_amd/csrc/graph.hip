// Graph handle: host-side CSR construction, degree norms and the degree-binned row schedule.
//
// Stands in for the dgl.DGLGraph the reference builds with dgl.graph((src,dst), num_nodes)
// (DiffGraph-Rec/DataHandler.py:120-132) and for graph.out_degrees()/in_degrees() +
// clamp(min=1) + pow(-0.5), which the reference recomputes on every layer call
// (DiffGraph-Rec/Model.py:151-152,165-166) and this handle computes once.
#include "dg_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

static thread_local std::string g_last_error;

int dg_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

extern "C" const char *dg_last_error(void) { return g_last_error.c_str(); }
extern "C" int dg_version(void) { return 1; }

#ifndef DG_SOURCE_DIGEST
#define DG_SOURCE_DIGEST "unknown"
#endif
// The digest of the kernel sources this library was built from (diffgraph_amd/build.py); the marker lets a
// build script read it out of the file without loading it.
static const char g_source_digest[] = "DG_SOURCE_DIGEST=" DG_SOURCE_DIGEST;
extern "C" const char *dg_source_digest(void) { return g_source_digest + sizeof("DG_SOURCE_DIGEST=") - 1; }

namespace {

constexpr int kHead = 13;  // neighbour ids carried in a group row's 16-word head
constexpr int kGroupPiece = 32;  // pieces of cut rows up to this long are summed by one 16-lane group, longer ones by a wavefront

struct HostCsr {
  std::vector<uint64_t> dense_bits;
  int32_t dense_splits = 0, dense_chunks = 0, dense_cps = 0, dense_tiles = 0;
  std::vector<int32_t> rowptr, colidx, row_order, head, wave_desc, hub_desc, seg_desc, gseg_desc;
  std::vector<float> colscale, head_scale;
  std::vector<uint32_t> lhead, wblocks, woff;
  std::vector<int32_t> witems;
  int32_t n_hub = 0, n_wave = 0, n_group = 0, n_seg = 0, lds_split = 0, n_windows = 0, n_parts = 0, n_slots = 0, slot_begin = 0, n_seg_slots = 0, n_gseg_slots = 0;
};

// CSR over `row` with ascending `col` inside each row: two stable counting sorts
// (by col, then by row), O(n + nnz).
// `perm` (optional) receives, per CSR position, the index of the edge it came from.
void build_csr(int32_t n, int64_t nnz, const int32_t *row, const int32_t *col, HostCsr &out,
               std::vector<int32_t> *perm = nullptr) {
  std::vector<int64_t> cnt((size_t)n + 1, 0);
  for (int64_t e = 0; e < nnz; ++e) cnt[(size_t)col[e] + 1]++;
  for (int32_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
  std::vector<int32_t> by_col((size_t)nnz);
  {
    std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int64_t e = 0; e < nnz; ++e) by_col[(size_t)pos[col[e]]++] = (int32_t)e;
  }
  out.rowptr.assign((size_t)n + 1, 0);
  std::vector<int64_t> rp((size_t)n + 1, 0);
  for (int64_t e = 0; e < nnz; ++e) rp[(size_t)row[e] + 1]++;
  for (int32_t i = 0; i < n; ++i) rp[i + 1] += rp[i];
  out.colidx.resize((size_t)nnz);
  {
    std::vector<int64_t> pos(rp.begin(), rp.end() - 1);
    if (perm) perm->resize((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
      int32_t e = by_col[(size_t)k];
      const int64_t at = pos[row[e]]++;
      out.colidx[(size_t)at] = col[e];
      if (perm) (*perm)[(size_t)at] = e;
    }
  }
  for (int32_t i = 0; i <= n; ++i) out.rowptr[i] = (int32_t)rp[i];
}

// Finds s such that every edge joins a row < s with a row >= s (the block form
// [[0,R],[R^T,0]] the reference's makeBiAdj builds); 0 if the graph is not of that form.
int32_t bipartite_split(int32_t n, const HostCsr &c) {
  int32_t s = n;
  for (int32_t r = 0; r < n; ++r) {
    if (c.rowptr[r + 1] == c.rowptr[r]) continue;
    int32_t lo = c.colidx[c.rowptr[r]];  // columns ascend inside a row
    if (lo > r) s = std::min(s, lo);
  }
  if (s <= 0 || s >= n) return 0;
  for (int32_t r = 0; r < n; ++r) {
    if (c.rowptr[r + 1] == c.rowptr[r]) continue;
    int32_t lo = c.colidx[c.rowptr[r]], hi = c.colidx[c.rowptr[r + 1] - 1];
    if (r < s ? lo < s : hi >= s) return 0;
  }
  return s;
}

// The small-side rows [split, n) of an LDS-split graph, regrouped for window_rows_kernel (layout:
// dg_csr::wblocks).  A row expected to meet many rows of every window is cut into parts by slice
// of the window, so that no part is much longer than a group's fair share; parts are dealt to
// the 64 groups longest first, each to the group with the least work so far.  Returns false when
// the graph is too dense for the kernel's LDS entry buffer (the caller then drops the LDS split).
bool build_windows(int32_t n, const std::vector<int32_t> &deg, HostCsr &c) {
  const int32_t s = c.lds_split, m = n - s;
  const int32_t nw = (s + DG_WIN_ROWS - 1) / DG_WIN_ROWS;
  constexpr int G = DG_WIN_GROUPS, S = DG_WIN_SLOTS;
  int64_t nnz_small = 0;
  for (int32_t i = 0; i < m; ++i) nnz_small += deg[s + i];
  const double per_window = (double)nnz_small / nw;  // entries of one window
  if (per_window * 1.5 + G * (DG_WIN_HDR + S) * 4 > DG_WIN_ENT_DWORDS * 4) return false;
  // parts per row
  std::vector<int32_t> n_cut(m), first_part(m + 1, 0);
  double share = std::max(per_window / G * 0.75, 8.0);
  for (;;) {
    int32_t total = 0;
    for (int32_t i = 0; i < m; ++i) {
      const double e = (double)deg[s + i] / nw;
      n_cut[i] = std::min(16, std::max(1, (int32_t)std::ceil(e / share)));
      total += n_cut[i];
    }
    if (total <= G * S) break;
    share *= 1.25;
  }
  for (int32_t i = 0; i < m; ++i) first_part[i + 1] = first_part[i] + n_cut[i];
  const int32_t n_parts = first_part[m];
  c.n_parts = n_parts;
  c.n_windows = nw;
  std::vector<int32_t> part_row(n_parts), order(n_parts), group_of(n_parts), slot_of(n_parts);
  std::vector<double> part_load(n_parts);
  for (int32_t i = 0; i < m; ++i)
    for (int32_t k = 0; k < n_cut[i]; ++k) {
      part_row[first_part[i] + k] = i;
      part_load[first_part[i] + k] = (double)deg[s + i] / n_cut[i];
    }
  for (int32_t v = 0; v < n_parts; ++v) order[v] = v;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return part_load[a] > part_load[b]; });
  c.witems.assign(G * S, -1);
  double load[G] = {0};
  int32_t used[G] = {0};
  for (int32_t v : order) {
    int best = -1;
    for (int g = 0; g < G; ++g)
      if (used[g] < S && (best < 0 || load[g] < load[best])) best = g;
    group_of[v] = best; slot_of[v] = used[best];
    c.witems[best * S + used[best]] = v;
    used[best]++;
    load[best] += part_load[v] + 4.0 * nw;  // a slot costs a little even when empty
  }
  auto part_of = [&](int32_t i, int32_t local) { return first_part[i] + local * n_cut[i] / DG_WIN_ROWS; };
  // entries of (window, group, slot)
  std::vector<uint16_t> cnt((size_t)nw * G * S, 0);
  for (int32_t i = 0; i < m; ++i)
    for (int32_t p = c.rowptr[s + i]; p < c.rowptr[s + i + 1]; ++p) {
      const int32_t u = c.colidx[p], w = u / DG_WIN_ROWS, v = part_of(i, u - w * DG_WIN_ROWS);
      cnt[((size_t)w * G + group_of[v]) * S + slot_of[v]]++;
    }
  c.woff.assign((size_t)nw * G + 1, 0);
  uint64_t total = 0;
  for (int32_t w = 0; w < nw; ++w) {
    total = (total + 3) & ~(uint64_t)3;
    const uint64_t begin = total;
    for (int g = 0; g < G; ++g) {
      const size_t bv = (size_t)w * G + g;
      c.woff[bv] = (uint32_t)total;
      total += DG_WIN_HDR;
      for (int k = 0; k < S; ++k) total += (cnt[bv * S + k] + 3) / 4;
    }
    if (total - begin > DG_WIN_ENT_DWORDS) return false;
  }
  c.woff[(size_t)nw * G] = (uint32_t)total;
  c.wblocks.assign((size_t)total + 4, 0xffffffffu);  // every entry byte = 255, the zero row
  std::vector<uint32_t> slot_byte((size_t)nw * G * S);  // first entry byte of each slot, within its block
  for (size_t bv = 0; bv < (size_t)nw * G; ++bv) {
    uint8_t *blk = reinterpret_cast<uint8_t *>(&c.wblocks[c.woff[bv]]);
    uint32_t byte = DG_WIN_HDR * 4;
    for (int k = 0; k < S; ++k) {
      const uint32_t quads = (cnt[bv * S + k] + 3) / 4;  // <= 64
      blk[k] = (uint8_t)quads;
      slot_byte[bv * S + k] = byte;
      byte += quads * 4;
    }
  }
  std::fill(cnt.begin(), cnt.end(), 0);  // reused as the fill cursor
  for (int32_t i = 0; i < m; ++i)
    for (int32_t p = c.rowptr[s + i]; p < c.rowptr[s + i + 1]; ++p) {  // ascending ids within a slot
      const int32_t u = c.colidx[p], w = u / DG_WIN_ROWS, local = u - w * DG_WIN_ROWS, v = part_of(i, local);
      const size_t bv = (size_t)w * G + group_of[v], sv = bv * S + slot_of[v];
      uint8_t *blk = reinterpret_cast<uint8_t *>(&c.wblocks[c.woff[bv]]);
      blk[slot_byte[sv] + cnt[sv]++] = (uint8_t)local;
    }
  // hub_reduce_kernel adds the workgroups' partial sums of a row's parts: {row, first part, last part + 1, 0}
  c.hub_desc.clear();
  for (int32_t i = 0; i < m; ++i) c.hub_desc.insert(c.hub_desc.end(), {s + i, first_part[i], first_part[i + 1], 0});
  return true;
}

// Dense graphs: the bitmap schedule of dg_csr::dense_bits.  Returns false (and leaves c alone) when the graph does
// not qualify: too sparse, too small or large, or an edge given more than once (a bit cannot count).
bool build_dense(int32_t n, HostCsr &c) {
  const int64_t nnz = (int64_t)c.colidx.size();
  if (n < DG_DENSE_MIN_NODES || n > DG_DENSE_MAX_NODES || (double)nnz < DG_DENSE_MIN_DENSITY * (double)n * (double)n) return false;
  for (int32_t r = 0; r < n; ++r)
    for (int32_t p = c.rowptr[r] + 1; p < c.rowptr[r + 1]; ++p)
      if (c.colidx[p] == c.colidx[p - 1]) return false;
  const int32_t tiles = (n + 31) / 32, chunks = (n + 63) / 64;
  // (tile group, range) workgroups: DG_DENSE_BLOCKS_PER_GRAPH per graph
  const int32_t groups = (tiles + DG_DENSE_TILES_PER_BLOCK - 1) / DG_DENSE_TILES_PER_BLOCK;
  int32_t splits = std::max(1, std::min(chunks, (DG_DENSE_BLOCKS_PER_GRAPH + groups - 1) / groups));
  const int32_t cps = (chunks + splits - 1) / splits;
  splits = (chunks + cps - 1) / cps;
  c.dense_tiles = tiles; c.dense_chunks = chunks; c.dense_cps = cps; c.dense_splits = splits;
  c.dense_bits.assign((size_t)groups * DG_DENSE_TILES_PER_BLOCK * chunks * 32, 0);   // whole tile groups
  for (int32_t r = 0; r < n; ++r)
    for (int32_t p = c.rowptr[r]; p < c.rowptr[r + 1]; ++p) {
      const int32_t col = c.colidx[p];
      c.dense_bits[((size_t)(r >> 5) * chunks + (col >> 6)) * 32 + (r & 31)] |= (uint64_t)1 << (col & 63);
    }
  c.hub_desc.clear();
  for (int32_t r = 0; r < n; ++r) c.hub_desc.insert(c.hub_desc.end(), {r, r * splits, (r + 1) * splits, 0});
  c.n_hub = n; c.n_seg = n * splits;
  c.n_wave = c.n_group = c.n_slots = c.slot_begin = c.n_seg_slots = c.n_gseg_slots = 0;
  c.lds_split = 0; c.n_windows = c.n_parts = 0;
  c.row_order.clear(); c.seg_desc.clear(); c.gseg_desc.clear(); c.wave_desc.clear(); c.head.clear(); c.head_scale.clear();
  c.lhead.clear(); c.wblocks.clear(); c.woff.clear(); c.witems.clear();
  return true;
}

// Two row classes: group rows (in-degree <= group_max: summed whole by one 16-lane group, in node
// order) and cut rows (longer: summed in pieces through the slab).  row_order lists cut rows, then
// group rows, each by falling degree.  Rows below c.lds_split (if set) are left to lds_rows_kernel
// and appear in no class.
void build_schedule(int32_t n, int32_t group_max, int32_t seg_len, HostCsr &c, const std::vector<char> *owned = nullptr) {
  c.dense_bits.clear();
  c.dense_splits = c.dense_chunks = c.dense_cps = c.dense_tiles = 0;
  if (!owned && build_dense(n, c)) return;
  std::vector<int32_t> deg(n);
  for (int32_t i = 0; i < n; ++i) deg[i] = c.rowptr[i + 1] - c.rowptr[i];
  c.lds_split = 0;
  c.lhead.clear(); c.wblocks.clear(); c.woff.clear(); c.witems.clear();
  c.n_windows = c.n_parts = 0;
  {
    const int32_t s = owned ? 0 : bipartite_split(n, c);  // row-block shards take the general path
    // worth it only when the large side is much larger than the LDS-resident side
    if (s > 0 && n - s <= DG_LDS_MAX_ROWS && s >= 64 * (n - s)) {
      c.lds_split = s;
      if (!build_windows(n, deg, c)) {
        c.lds_split = 0;
        c.wblocks.clear(); c.woff.clear(); c.witems.clear(); c.hub_desc.clear();
        c.n_windows = c.n_parts = 0;
      }
    }
  }
  c.row_order.clear();
  for (int32_t i = c.lds_split; i < n; ++i)
    if (!owned || (*owned)[i]) c.row_order.push_back(i);
  if (c.lds_split > 0) {
    // rows below the split: lds_rows_kernel; the others: window_rows_kernel, then hub_reduce_kernel
    const uint32_t pad = (uint32_t)(n - c.lds_split);  // the zero row of the LDS table
    c.lhead.assign((size_t)c.lds_split * 16, pad | (pad << 16));
    for (int32_t r = 0; r < c.lds_split; ++r) {
      uint16_t hw[32];
      for (int k = 0; k < 32; ++k) hw[k] = (uint16_t)pad;
      hw[0] = (uint16_t)std::min(deg[r], 65535);
      for (int32_t k = 0; k < std::min(deg[r], 31); ++k) hw[k + 1] = (uint16_t)(c.colidx[c.rowptr[r] + k] - c.lds_split);
      for (int k = 0; k < 16; ++k) c.lhead[(size_t)r * 16 + k] = (uint32_t)hw[2 * k] | ((uint32_t)hw[2 * k + 1] << 16);
    }
    c.n_hub = n - c.lds_split; c.n_wave = c.n_group = c.n_seg = c.n_slots = c.slot_begin = c.n_seg_slots = c.n_gseg_slots = 0;
    c.seg_desc.clear(); c.gseg_desc.clear(); c.wave_desc.clear(); c.head.clear(); c.head_scale.clear();
    return;
  }
  std::stable_sort(c.row_order.begin(), c.row_order.end(),
                   [&](int32_t a, int32_t b) { return deg[a] > deg[b]; });
  n = (int32_t)c.row_order.size();  // rows scheduled in the two classes
  c.n_hub = c.n_wave = c.n_group = 0;
  for (int32_t k = 0; k < n; ++k) {
    if (deg[c.row_order[k]] > group_max) c.n_hub++;
    else c.n_group++;
  }
  // Cut rows.  The node ids are split into 8 ranges holding equal shares of the cut rows' neighbours;
  // a cut row's (ascending) neighbour list is cut at the range limits, and pieces longer than seg_len
  // again into equal parts.  Range x is summed by the workgroups whose index is x (mod 8) -- blocks
  // b, b + 8, ... share an XCD -- so each XCD keeps gathering from the same eighth of the source table
  // and finds it in its L2.  Every piece writes one slab row; a row's slab rows are consecutive, in
  // ascending neighbour order.
  c.hub_desc.clear(); c.seg_desc.clear(); c.gseg_desc.clear(); c.wave_desc.clear();
  c.n_seg = 0;
  constexpr int kRuns = 8;
  int32_t limit[kRuns + 1];
  {
    std::vector<int32_t> ids;
    for (int32_t h = 0; h < c.n_hub; ++h) {
      const int32_t r = c.row_order[h];
      ids.insert(ids.end(), c.colidx.begin() + c.rowptr[r], c.colidx.begin() + c.rowptr[r + 1]);
    }
    limit[0] = 0;
    limit[kRuns] = INT32_MAX;
    for (int x = 1; x < kRuns; ++x) {
      if (ids.empty()) { limit[x] = INT32_MAX; continue; }
      const size_t q = ids.size() * (size_t)x / kRuns;
      std::nth_element(ids.begin(), ids.begin() + q, ids.end());
      limit[x] = ids[q];
    }
  }
  struct Piece { int32_t begin, end, slab; };
  std::vector<Piece> wave_pieces[kRuns], group_pieces[kRuns];
  for (int32_t h = 0; h < c.n_hub; ++h) {
    const int32_t r = c.row_order[h];
    const int32_t *col = c.colidx.data();
    const int32_t s0 = c.n_seg;
    int32_t p = c.rowptr[r];
    const int32_t e = c.rowptr[r + 1];
    for (int x = 0; x < kRuns && p < e; ++x) {
      const int32_t pe = x + 1 == kRuns ? e : (int32_t)(std::lower_bound(col + p, col + e, limit[x + 1]) - col);
      const int32_t len = pe - p;
      if (len <= 0) continue;
      const int32_t parts = (len + seg_len - 1) / seg_len;
      for (int32_t k = 0; k < parts; ++k) {
        const int32_t b0 = p + (int32_t)((int64_t)len * k / parts), b1 = p + (int32_t)((int64_t)len * (k + 1) / parts);
        (b1 - b0 > kGroupPiece ? wave_pieces : group_pieces)[x].push_back({b0, b1, c.n_seg++});
      }
      p = pe;
    }
    c.hub_desc.insert(c.hub_desc.end(), {r, s0, c.n_seg, 0});
  }
  // slot tables: run x's pieces go to blocks x, x + 8, x + 16, ... of the class, `per_block` pieces each
  auto lay_out = [&](std::vector<Piece> (&runs)[kRuns], int per_block, std::vector<int32_t> &desc, int32_t &n_slots) {
    size_t longest = 0;
    for (int x = 0; x < kRuns; ++x) longest = std::max(longest, runs[x].size());
    const int32_t blocks_per_run = (int32_t)((longest + per_block - 1) / per_block);
    n_slots = blocks_per_run * kRuns * per_block;
    desc.assign((size_t)n_slots * 4, 0);
    for (int32_t slot = 0; slot < n_slots; ++slot) desc[(size_t)slot * 4 + 2] = -1;
    for (int x = 0; x < kRuns; ++x)
      for (size_t i = 0; i < runs[x].size(); ++i) {
        const size_t slot = ((i / per_block) * kRuns + x) * per_block + i % per_block;
        desc[slot * 4 + 0] = runs[x][i].begin;
        desc[slot * 4 + 1] = runs[x][i].end;
        desc[slot * 4 + 2] = runs[x][i].slab;
      }
  };
  lay_out(wave_pieces, 4, c.seg_desc, c.n_seg_slots);
  lay_out(group_pieces, 16, c.gseg_desc, c.n_gseg_slots);
  // One head record per node (rows of the other class, and rows this rank does not own, carry degree -1
  // and are skipped).
  // The records span the owned rows only (a rank of a row-block partition owns one block).
  const int32_t n_nodes = (int32_t)deg.size();
  int32_t lo = 0, hi = n_nodes;
  if (owned) {
    while (lo < hi && !(*owned)[lo]) ++lo;
    while (hi > lo && !(*owned)[hi - 1]) --hi;
  }
  c.slot_begin = lo;
  c.n_slots = hi - lo;
  c.head.assign((size_t)c.n_slots * 16, -1);
  c.head_scale.assign((size_t)c.n_slots * 16, 0.f);
  // Records go in six coarse degree classes, longest rows first, node order inside a class: the four rows
  // that share a wavefront then issue about the same number of gathers (a wavefront load instruction costs
  // the same whether one or four of its row groups still have neighbours), while the epilogue's tables
  // are still walked in long ascending runs.
  std::vector<int32_t> slot_row;
  for (int32_t r = lo; r < hi; ++r) slot_row.push_back(r);
  auto degree_class = [&](int32_t r) {
    if ((owned && !(*owned)[r]) || deg[r] > group_max) return 6;   // skipped rows last
    const int32_t d = deg[r];
    return d > 29 ? 0 : d > 13 ? 1 : d > 8 ? 2 : d > 4 ? 3 : d > 0 ? 4 : 5;
  };
  std::stable_sort(slot_row.begin(), slot_row.end(), [&](int32_t a, int32_t b) { return degree_class(a) < degree_class(b); });
  for (int32_t k = 0; k < c.n_slots; ++k) {
    const int32_t r = slot_row[k];
    int32_t *hd = &c.head[(size_t)k * 16];
    hd[13] = r;
    if ((owned && !(*owned)[r]) || deg[r] > group_max) continue;  // hd[14] stays -1
    float *hs = &c.head_scale[(size_t)k * 16];
    const int32_t b = c.rowptr[r], d = deg[r];
    for (int32_t j = 0; j < std::min(d, kHead); ++j) {
      hd[j] = c.colidx[b + j];
      hs[j] = c.colscale[b + j];
    }
    hd[14] = d; hd[15] = b + kHead;
  }
}

template <class T>
hipError_t upload(const std::vector<T> &v, T **dst) {
  *dst = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  hipError_t e = hipMalloc((void **)dst, bytes);
  if (e != hipSuccess) return e;
  if (!v.empty()) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

hipError_t upload_csr(const HostCsr &h, dg_csr &d) {
  hipError_t e;
  if ((e = upload(h.rowptr, &d.rowptr)) != hipSuccess) return e;
  if ((e = upload(h.colidx, &d.colidx)) != hipSuccess) return e;
  if ((e = upload(h.colscale, &d.colscale)) != hipSuccess) return e;
  if ((e = upload(h.row_order, &d.row_order)) != hipSuccess) return e;
  if ((e = upload(h.head, &d.head)) != hipSuccess) return e;
  if ((e = upload(h.head_scale, &d.head_scale)) != hipSuccess) return e;
  if ((e = upload(h.wave_desc, &d.wave_desc)) != hipSuccess) return e;
  if ((e = upload(h.hub_desc, &d.hub_desc)) != hipSuccess) return e;
  if ((e = upload(h.seg_desc, &d.seg_desc)) != hipSuccess) return e;
  if ((e = upload(h.gseg_desc, &d.gseg_desc)) != hipSuccess) return e;
  if ((e = upload(h.lhead, &d.lhead)) != hipSuccess) return e;
  if ((e = upload(h.wblocks, &d.wblocks)) != hipSuccess) return e;
  if ((e = upload(h.woff, &d.woff)) != hipSuccess) return e;
  if ((e = upload(h.witems, &d.witems)) != hipSuccess) return e;
  if ((e = upload(h.dense_bits, &d.dense_bits)) != hipSuccess) return e;
  d.dense_splits = h.dense_splits; d.dense_chunks = h.dense_chunks; d.dense_cps = h.dense_cps; d.dense_tiles = h.dense_tiles;
  d.n_windows = h.n_windows; d.n_parts = h.n_parts;
  d.n_hub = h.n_hub; d.n_wave = h.n_wave; d.n_group = h.n_group; d.n_seg = h.n_seg;
  d.lds_split = h.lds_split; d.n_slots = h.n_slots; d.slot_begin = h.slot_begin; d.n_seg_slots = h.n_seg_slots; d.n_gseg_slots = h.n_gseg_slots;
  return hipSuccess;
}

void free_csr(dg_csr &d) {
  (void)hipFree(d.rowptr); (void)hipFree(d.colidx); (void)hipFree(d.colscale); (void)hipFree(d.row_order);
  (void)hipFree(d.head); (void)hipFree(d.head_scale); (void)hipFree(d.wave_desc); (void)hipFree(d.hub_desc);
  (void)hipFree(d.seg_desc); (void)hipFree(d.gseg_desc); (void)hipFree(d.lhead);
  (void)hipFree(d.wblocks); (void)hipFree(d.woff); (void)hipFree(d.witems); (void)hipFree(d.dense_bits);
  d = dg_csr();
}

}  // namespace

extern "C" int dg_graph_create(int32_t n, int64_t nnz, const int32_t *src, const int32_t *dst,
                               int32_t group_max, int32_t seg_len, dg_graph **out) {
  return dg_graph_create_ex(n, nnz, src, dst, nullptr, nullptr, group_max, seg_len, out);
}

extern "C" int dg_graph_create_ex(int32_t n, int64_t nnz, const int32_t *src, const int32_t *dst,
                                  const int32_t *out_deg_host, const int32_t *in_deg_host, int32_t group_max,
                                  int32_t seg_len, dg_graph **out) {
  if (!out) return dg_fail(DG_ERR_ARG, "dg_graph_create: out is NULL");
  *out = nullptr;
  if (n <= 0) return dg_fail(DG_ERR_ARG, "dg_graph_create: n_nodes must be positive (got %d)", n);
  if (nnz < 0 || (nnz > 0 && (!src || !dst)))
    return dg_fail(DG_ERR_ARG, "dg_graph_create: bad edge arrays");
  if (nnz >= (int64_t)INT32_MAX)
    return dg_fail(DG_ERR_SIZE, "dg_graph_create: %lld edges exceed the int32 CSR limit", (long long)nnz);
  if ((int64_t)n * DG_D * 4 >= ((int64_t)1 << 32))
    return dg_fail(DG_ERR_SIZE, "dg_graph_create: %d nodes exceed the 4 GiB table limit of the gather path", n);
  if (group_max <= 0) group_max = 64;
  if (seg_len <= 0) seg_len = 256;
  if (seg_len < 64 || seg_len % 64 != 0)
    return dg_fail(DG_ERR_ARG, "dg_graph_create: seg_len must be a positive multiple of 64 (got %d)", seg_len);
  if (group_max > seg_len) group_max = seg_len;
  for (int64_t e = 0; e < nnz; ++e)
    if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n)
      return dg_fail(DG_ERR_ARG, "dg_graph_create: edge %lld (%d -> %d) out of range [0,%d)", (long long)e, src[e], dst[e], n);

  HostCsr f, b;
  build_csr(n, nnz, dst, src, f);  // rows = destinations
  build_csr(n, nnz, src, dst, b);  // rows = sources
  const bool symmetric = (f.rowptr == b.rowptr) && (f.colidx == b.colidx);

  // clamp(min=1), pow(-0.5) in fp32: torch's CPU path evaluates pow(x,-0.5) as 1/sqrt(x)
  // with correctly rounded sqrt and divide, which is what this does.
  std::vector<float> dsrc(n), ddst(n);
  for (int32_t i = 0; i < n; ++i) {
    int32_t in_deg = in_deg_host ? in_deg_host[i] : f.rowptr[i + 1] - f.rowptr[i];
    int32_t out_deg = out_deg_host ? out_deg_host[i] : b.rowptr[i + 1] - b.rowptr[i];
    ddst[i] = 1.0f / sqrtf((float)std::max(in_deg, 1));
    dsrc[i] = 1.0f / sqrtf((float)std::max(out_deg, 1));
  }
  f.colscale.resize((size_t)nnz);
  for (int64_t p = 0; p < nnz; ++p) f.colscale[(size_t)p] = dsrc[f.colidx[(size_t)p]];  // gathered node = a source
  std::vector<char> owned;
  bool all_owned = true;
  if (in_deg_host) {
    owned.resize(n);
    for (int32_t i = 0; i < n; ++i) {
      owned[i] = in_deg_host[i] >= 0;
      all_owned = all_owned && owned[i];
    }
  }
  build_schedule(n, group_max, seg_len, f, all_owned ? nullptr : &owned);
  if (!symmetric) {
    b.colscale.resize((size_t)nnz);
    for (int64_t p = 0; p < nnz; ++p) b.colscale[(size_t)p] = ddst[b.colidx[(size_t)p]];
    build_schedule(n, group_max, seg_len, b);
  }

  dg_graph *g = new dg_graph();
  g->n = n; g->nnz = nnz; g->symmetric = symmetric ? 1 : 0; g->n_rows = n;
  g->group_max = group_max; g->seg_len = seg_len;
  hipError_t e = hipGetDevice(&g->device);
  if (e == hipSuccess) e = upload(dsrc, &g->dinv_src);
  if (e == hipSuccess) e = upload(ddst, &g->dinv_dst);
  if (e == hipSuccess) e = upload_csr(f, g->fwd);
  if (e == hipSuccess) {
    if (symmetric) g->bwd = g->fwd;
    else e = upload_csr(b, g->bwd);
  }
  if (e == hipSuccess) {
    size_t segs = (size_t)std::max(g->fwd.n_seg, g->bwd.n_seg);
    // LDS-split graphs: one slab row per (part of a small-side row, workgroup of window_rows_kernel)
    segs = std::max(segs, (size_t)std::max(g->fwd.n_parts, g->bwd.n_parts) * DG_WIN_MAX_BLOCKS);
    e = hipMalloc((void **)&g->slab, std::max<size_t>(segs, 1) * DG_D * sizeof(float));
  }
  if (e == hipSuccess && (g->fwd.dense_splits > 0 || g->bwd.dense_splits > 0))
    e = hipMalloc((void **)&g->dense_x, (size_t)3 * DG_D * std::max(g->fwd.dense_chunks, g->bwd.dense_chunks) * 64 * sizeof(uint16_t));
  if (e != hipSuccess) {
    dg_graph_destroy(g);
    return dg_fail((int)e, "dg_graph_create: device allocation/copy failed: %s", hipGetErrorString(e));
  }
  *out = g;
  return 0;
}

// A rectangular operator with a weight per entry, run by the same row schedule and kernels as a graph:
// out[r] = sum over the entries (r, c) of weight * x[c].  The handle has max(n_rows, n_cols) nodes of which the first
// n_rows are computed; both degree norms are 1, the weights sit where a graph keeps its sources' norms.
extern "C" int dg_graph_create_weighted(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *row, const int32_t *col,
                                        const float *weight, int32_t group_max, int32_t seg_len, dg_graph **out) {
  if (!out) return dg_fail(DG_ERR_ARG, "dg_graph_create_weighted: out is NULL");
  *out = nullptr;
  if (n_rows <= 0 || n_cols <= 0) return dg_fail(DG_ERR_ARG, "dg_graph_create_weighted: empty shape %d x %d", n_rows, n_cols);
  if (nnz < 0 || (nnz > 0 && (!row || !col || !weight))) return dg_fail(DG_ERR_ARG, "dg_graph_create_weighted: bad entry arrays");
  if (nnz >= (int64_t)INT32_MAX)
    return dg_fail(DG_ERR_SIZE, "dg_graph_create_weighted: %lld entries exceed the int32 CSR limit", (long long)nnz);
  const int32_t n = std::max(n_rows, n_cols);
  if ((int64_t)n * DG_D * 4 >= ((int64_t)1 << 32))
    return dg_fail(DG_ERR_SIZE, "dg_graph_create_weighted: %d rows exceed the 4 GiB table limit of the gather path", n);
  if (group_max <= 0) group_max = 64;
  if (seg_len <= 0) seg_len = 256;
  if (seg_len < 64 || seg_len % 64 != 0)
    return dg_fail(DG_ERR_ARG, "dg_graph_create_weighted: seg_len must be a positive multiple of 64 (got %d)", seg_len);
  if (group_max > seg_len) group_max = seg_len;
  for (int64_t e = 0; e < nnz; ++e)
    if (row[e] < 0 || row[e] >= n_rows || col[e] < 0 || col[e] >= n_cols)
      return dg_fail(DG_ERR_ARG, "dg_graph_create_weighted: entry %lld (%d, %d) outside %d x %d", (long long)e, row[e], col[e],
                     n_rows, n_cols);
  HostCsr f;
  std::vector<int32_t> perm;
  build_csr(n, nnz, row, col, f, &perm);
  f.colscale.resize((size_t)nnz);
  for (int64_t p = 0; p < nnz; ++p) f.colscale[(size_t)p] = weight[perm[(size_t)p]];
  std::vector<char> owned((size_t)n, 0);
  std::fill(owned.begin(), owned.begin() + n_rows, 1);
  build_schedule(n, group_max, seg_len, f, &owned);   // (an ownership list keeps it on the gather path)
  const std::vector<float> ones((size_t)n, 1.0f);
  dg_graph *g = new dg_graph();
  g->n = n; g->nnz = nnz; g->symmetric = 1; g->weighted = 1; g->n_rows = n_rows;
  g->group_max = group_max; g->seg_len = seg_len;
  hipError_t e = hipGetDevice(&g->device);
  if (e == hipSuccess) e = upload(ones, &g->dinv_src);
  if (e == hipSuccess) e = upload(ones, &g->dinv_dst);
  if (e == hipSuccess) e = upload_csr(f, g->fwd);
  if (e == hipSuccess) {
    g->bwd = g->fwd;
    e = hipMalloc((void **)&g->slab, std::max<size_t>((size_t)g->fwd.n_seg, 1) * DG_D * sizeof(float));
  }
  if (e != hipSuccess) {
    dg_graph_destroy(g);
    return dg_fail((int)e, "dg_graph_create_weighted: device allocation/copy failed: %s", hipGetErrorString(e));
  }
  *out = g;
  return 0;
}

extern "C" void dg_graph_destroy(dg_graph *g) {
  if (!g) return;
  if (!g->symmetric) free_csr(g->bwd);
  free_csr(g->fwd);
  (void)hipFree(g->dinv_src); (void)hipFree(g->dinv_dst); (void)hipFree(g->slab); (void)hipFree(g->dense_x);
  delete g;
}

extern "C" int32_t dg_graph_num_nodes(const dg_graph *g) { return g ? g->n : 0; }
extern "C" int32_t dg_graph_num_rows(const dg_graph *g) { return g ? g->n_rows : 0; }
extern "C" int64_t dg_graph_num_edges(const dg_graph *g) { return g ? g->nnz : 0; }
extern "C" int dg_graph_is_symmetric(const dg_graph *g) { return g ? g->symmetric : 0; }

extern "C" int dg_graph_export(const dg_graph *g, int which, void *dst, size_t bytes) {
  if (!g || !dst) return dg_fail(DG_ERR_ARG, "dg_graph_export: NULL argument");
  const void *src = nullptr;
  size_t need = 0;
  int32_t counts[6] = {g->fwd.n_hub, g->fwd.n_wave, g->fwd.n_group, g->fwd.n_seg, g->fwd.lds_split, g->fwd.dense_splits};
  switch (which) {
    case 0: src = g->fwd.rowptr; need = ((size_t)g->n + 1) * 4; break;
    case 1: src = g->fwd.colidx; need = (size_t)g->nnz * 4; break;
    case 2: src = g->dinv_src; need = (size_t)g->n * 4; break;
    case 3: src = g->dinv_dst; need = (size_t)g->n * 4; break;
    case 4:
      src = g->fwd.row_order;
      need = g->fwd.dense_splits > 0 ? 0 : (size_t)(g->fwd.n_hub + g->fwd.n_wave + g->fwd.n_group) * 4;
      break;
    case 5:
      if (bytes < sizeof(counts)) return dg_fail(DG_ERR_SIZE, "dg_graph_export: buffer too small");
      memcpy(dst, counts, sizeof(counts));
      return 0;
    default: return dg_fail(DG_ERR_ARG, "dg_graph_export: unknown selector %d", which);
  }
  if (bytes < need) return dg_fail(DG_ERR_SIZE, "dg_graph_export: buffer too small (%zu < %zu)", bytes, need);
  if (need) DG_HIP(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
  return 0;
}
