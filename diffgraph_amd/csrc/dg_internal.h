// Internal declarations shared by the translation units of libdiffgraph_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <atomic>

#include "../../include/diffgraph_hip.h"

#define DG_ERR_ARG (-1)
#define DG_ERR_DIM (-2)
#define DG_ERR_SIZE (-3)

#define DG_EPI_BWD 0x100  // internal flag: backward epilogue
#define DG_LDS_MAX_ROWS 383  // small-side rows kept in LDS (+1 zero row) = 96 KiB
// window_rows_kernel: the small-side rows of such a graph are summed from windows of DG_WIN_ROWS
// consecutive large-side rows staged in LDS (+1 zero row = 64 KiB).  Each of the 64 16-lane groups
// of a workgroup owns up to DG_WIN_SLOTS "parts"; a part is a small-side row, or -- for rows that
// meet many rows of every window -- that row restricted to a slice of the window.
#define DG_WIN_ROWS 255
#define DG_WIN_SLOTS 8
#define DG_WIN_GROUPS 64
#define DG_WIN_HDR 2           // dwords of per-slot entry-dword counts at the head of a (window, group) block
#define DG_WIN_ENT_DWORDS 4096 // LDS room for one window's entry blocks (16 KiB)
#define DG_WIN_MAX_BLOCKS 256
#define DG_DENSE_MIN_DENSITY 0.10  // above it the bitmap contraction beats gathering 256 B per edge through L2
#define DG_DENSE_MIN_NODES 256
// dense_rows_kernel: 32-row tiles (one per wavefront) that share the table rows a workgroup stages, and the number of
// workgroups a graph is cut into (about one per CU, so that one dense graph alone fills the chip)
#define DG_DENSE_TILES_PER_BLOCK 4
#ifndef DG_DENSE_BLOCKS_PER_GRAPH
#define DG_DENSE_BLOCKS_PER_GRAPH 256
#endif
#define DG_DENSE_MAX_NODES 32768  // bitmap of n^2 / 8 bytes: 128 MiB at the limit

// Records a printf-style message retrievable through dg_last_error(); returns `code`.
int dg_fail(int code, const char *fmt, ...);

#define DG_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return dg_fail((int)e_, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// A kernel that takes more than 64 KiB of dynamic LDS has to be opted in once on every device it runs on
// (`done`: one bit per device id, owned by the call site).
inline hipError_t dg_lds_opt_in(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// One direction of a graph: CSR over "rows" (the nodes that receive the sum) with the
// degree-binned schedule the propagation kernel walks.
struct dg_csr {
  int32_t *rowptr = nullptr;     // [n+1]
  int32_t *colidx = nullptr;     // [nnz], ascending inside a row
  float *colscale = nullptr;     // [nnz] degree norm of the gathered node colidx[p]
  int32_t *row_order = nullptr;  // [n]: hub rows, then wave rows, then group rows (each by falling degree)
  // One 16-word record per node of [slot_begin, slot_begin + n_slots), by degree class and node id: words 0..12 the first 13 neighbour ids (-1 = none),
  // word 13 the row id, word 14 its degree -- or -1 when the row is not a group row (summed as a
  // wave or hub row, or not owned by this rank) -- word 15 the colidx position of neighbour 13.
  // One coalesced 64-B read gives a short row everything it needs.
  int32_t *head = nullptr;       // [n_slots][16]
  float *head_scale = nullptr;   // [n_slots][16] norms of the 13 head neighbours
  int32_t n_slots = 0, slot_begin = 0;  // all nodes, or the span of the rows this rank owns (none for LDS-split graphs)
  int32_t *wave_desc = nullptr;  // unused (kept for layout compatibility of the export selectors)
  int32_t *hub_desc = nullptr;   // [n_hub][4] cut row, first slab row, last slab row + 1, 0
  // Pieces of cut rows, in the order the workgroups take them (see build_schedule for the XCD-aware
  // order); a slot is {colidx begin, end, slab row (-1: empty slot), 0}.  seg_desc: pieces summed by a
  // wavefront, four per block; gseg_desc: short pieces summed by a 16-lane group, sixteen per block.
  int32_t *seg_desc = nullptr;   // [n_seg_slots][4]
  int32_t *gseg_desc = nullptr;  // [n_gseg_slots][4]
  int32_t n_seg_slots = 0, n_gseg_slots = 0;
  int32_t n_hub = 0, n_wave = 0, n_group = 0, n_seg = 0;
  // Bipartite graphs whose second block is small enough for LDS (Industry shape: 1M users x 361
  // items): rows [0, lds_split) gather only from rows [lds_split, n), whose table (<= 96 KB)
  // lds_rows_kernel keeps in LDS; those rows are then left out of the three classes above.
  int32_t lds_split = 0;
  // [lds_split][16] words = 32 halfwords per row below the split: halfword 0 the degree, 1..31 the
  // first 31 neighbours as ids relative to the split (padding = n - lds_split, the zero row).
  uint32_t *lhead = nullptr;
  // Rows [lds_split, n) (the small side) in the same graphs.  Their neighbour lists are regrouped by
  // window of DG_WIN_ROWS large-side rows and by owning group: block (w, g) starts at dword
  // woff[w * 64 + g] of wblocks and holds DG_WIN_HDR dwords = 8 byte counts (slot k: number of
  // entry dwords), then the entry dwords slot after slot.  An entry dword is four one-byte row ids
  // relative to the window (255 = the zero row, padding).  woff[w * 64] is a multiple of 4 and
  // a window's blocks take at most DG_WIN_ENT_DWORDS dwords.
  // witems[g * 8 + k] = part owned by group g in slot k, or -1; the parts of small-side row i are
  // hub_desc[i].y .. hub_desc[i].z - 1 (n_parts in all).
  uint32_t *wblocks = nullptr, *woff = nullptr;
  int32_t *witems = nullptr;
  int32_t n_windows = 0, n_parts = 0;
  // Dense graphs (at least DG_DENSE_MIN_DENSITY of all node pairs are edges, no repeated edge): the adjacency as
  // a bitmap, contracted with the table on the matrix cores by dense_rows_kernel instead of gathered row by
  // row.  dense_bits[(tile * dense_chunks + c) * 32 + i] holds columns [64 c, 64 c + 64) of row 32 tile + i
  // (bit b = column 64 c + b).  The columns are cut into dense_splits ranges of dense_cps chunks; a wavefront sums
  // one (32-row tile, range) into slab rows row * dense_splits + range, and hub_reduce_kernel adds a row's ranges
  // in order (hub_desc lists every node).  No row of such a graph is in the other classes.
  uint64_t *dense_bits = nullptr;
  int32_t dense_splits = 0, dense_chunks = 0, dense_cps = 0, dense_tiles = 0;
};

struct dg_graph {
  int32_t n = 0;
  int64_t nnz = 0;
  int symmetric = 0;
  int weighted = 0;           // dg_graph_create_weighted: colscale holds caller-given weights, rows [0, n_rows) are computed
  int32_t n_rows = 0;
  int32_t group_max = 0, seg_len = 0;
  float *dinv_src = nullptr;  // [n] max(out_deg,1)^-1/2
  float *dinv_dst = nullptr;  // [n] max(in_deg,1)^-1/2
  dg_csr fwd;                 // rows = destinations, columns = sources
  dg_csr bwd;                 // rows = sources, columns = destinations (aliases fwd when symmetric)
  float *slab = nullptr;      // [max(fwd.n_seg,bwd.n_seg)][64] partial sums of hub-row segments
  // dense graphs only: the layer's input table as three bf16 planes [3][64 features][64 dense_chunks rows] that
  // add up to the fp32 values (times the gathered node's norm when the caller's table is not scaled yet)
  float *dense_x = nullptr;
  int device = 0;
};

// Device-visible description of one graph's share of a batched layer launch.
struct dg_task_dev {
  const int32_t *rowptr;
  const uint32_t *lhead;
  const int32_t *colidx, *head, *hub_desc, *seg_desc, *gseg_desc;
  const float *colscale, *head_scale;
  const float *dinv_row;     // norm of the receiving (row) node
  const float *dinv_out;     // norm applied by DG_OUT_PRESCALE (the row node's norm as a future source)
  const float *x;
  float *slab;
  float *e_out;
  const float *acc_in;
  float *acc_out;
  float *partial_out;    // sharded: raw neighbour sums of rows >= partial_from (NULL: off)
  int32_t partial_from;
  const float *e_saved;  // backward: saved forward activation of this layer (NULL: layer 0)
  const float *d_sum;    // backward: gradient wrt the layer sum
  float *row_norm;       // forward: ||y|| out (NULL: off); backward with DG_BWD_NORM_FEED: ||y|| in
  int32_t n, n_hub, n_wave, n_group, n_seg;  // n_group: head records (dg_csr::n_slots)
  int32_t n_seg_slots, n_gseg_slots;
  int32_t flags;
  float slope;
  int32_t lds_split;      // > 0: rows below it are summed by lds_rows_kernel
  int32_t lds_blk_begin, lds_blk_count;  // this task's workgroups in the lds_rows_kernel launch
  const float *dinv_gather;  // norm of a gathered node (by node id), for staging the LDS table
  const uint32_t *wblocks, *woff;  // see dg_csr
  const int32_t *witems;
  int32_t n_windows, n_parts;
  int32_t win_blk_begin, win_blk_count;  // this task's workgroups in the window_rows_kernel launch
  const float *fin_partials;  // dg_gcn_partial_finalize: partial sums [n_parts][stride rows][64] ...
  int64_t fin_stride;         // ... rows between parts
  int32_t fin_parts, fin_row_begin, fin_rows;  // ... for rows [fin_row_begin, fin_row_begin + fin_rows)
  const uint64_t *dense_bits;   // see dg_csr
  float *dense_x;
  int32_t dense_splits, dense_chunks, dense_cps, dense_tiles;
  int32_t dense_blk_begin;      // first block of this task in the dense_rows_kernel launch
  int32_t red_blk_begin;        // first block of this task in the dense_reduce_kernel launch
  int32_t red_parts;            // slab rows per row that kernel adds up (0: none, the row-wise backward step)
  int32_t blk_begin;      // first block of this task in the main launch
  int32_t hub_blk_begin;  // first block of this task in the hub-row reduce launch
};

struct dg_task_batch {
  dg_task_dev t[DG_MAX_TASKS];
  int32_t n_tasks;
  int32_t gather_blocks = 0;  // dense_rows_kernel: its first gather_blocks workgroups do gcn_layer_kernel's work (0: none)
};
