// Boolean sparse-sparse product for gfx950: the pattern of A.B for CSR patterns A [n_rows x n_mid]
// and B [n_mid x n_cols].
//
// Replaces the meta-path composition of DiffGraph_NC/DataHandler.py:147-165 -- Q1.dot(Q2) on
// scipy matrices followed by (Hc != 0): with the softmax weights of the relation mixture all
// positive, only the PATTERN of the product survives the binarisation, so the arithmetic is a
// boolean row merge.  One wavefront owns one output row: it ORs the B rows named by the A row
// into a bitmap of n_cols bits held in LDS (ds_or_b32), writes the bitmap out and counts its
// bits; after a prefix sum over the row counts a second kernel expands the bitmaps into
// ascending column ids.  Dense-ish results (the author-paper-term-paper-author graph of DBLP
// fills 43 % of its cells) cost no more than sparse ones.
#include "dg_internal.h"

namespace {

constexpr int kMaxWords = 8192;  // bitmap words per row kept in LDS: n_cols <= 262144

__global__ __launch_bounds__(256) void spgemm_bitmap_kernel(const int32_t *__restrict__ a_rowptr,
                                                            const int32_t *__restrict__ a_col, int n_rows,
                                                            const int32_t *__restrict__ b_rowptr,
                                                            const int32_t *__restrict__ b_col, int words,
                                                            uint32_t *__restrict__ bitmap, int32_t *__restrict__ row_nnz) {
  extern __shared__ uint32_t bm_all[];  // [4][words]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
  const int row = blockIdx.x * 4 + wave;
  uint32_t *bm = bm_all + (size_t)wave * words;
  for (int w = lane; w < words; w += 64) bm[w] = 0u;
  __syncthreads();
  if (row < n_rows) {
    const int pa0 = a_rowptr[row], pa1 = a_rowptr[row + 1];
    // the four 16-lane groups take every fourth entry of the A row; a group's lanes stride over that B row
    for (int p = pa0 + g; p < pa1; p += 4) {
      const int k = a_col[p];
      const int q1 = b_rowptr[k + 1];
      for (int q = b_rowptr[k] + l; q < q1; q += 16) {
        const int c = b_col[q];
        atomicOr(&bm[c >> 5], 1u << (c & 31));
      }
    }
  }
  __syncthreads();
  if (row >= n_rows) return;
  int cnt = 0;
  for (int w = lane; w < words; w += 64) {
    const uint32_t v = bm[w];
    bitmap[(size_t)row * words + w] = v;
    cnt += __popc(v);
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane == 0) row_nnz[row] = cnt;
}

// Row bitmaps -> ascending column ids at c_col[c_rowptr[row] ...]; one wavefront per row.
__global__ __launch_bounds__(256) void spgemm_expand_kernel(const uint32_t *__restrict__ bitmap, int n_rows, int words,
                                                            const int32_t *__restrict__ c_rowptr,
                                                            int32_t *__restrict__ c_col) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n_rows) return;
  int base = c_rowptr[row];
  for (int w0 = 0; w0 < words; w0 += 64) {
    const int w = w0 + lane;
    uint32_t v = w < words ? bitmap[(size_t)row * words + w] : 0u;
    const int n = __popc(v);
    int incl = n;  // inclusive scan over the wavefront
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    int o = base + incl - n;
    while (v) {
      c_col[o++] = w * 32 + __builtin_ctz(v);
      v &= v - 1;
    }
    base += __shfl(incl, 63);
  }
}

}  // namespace

extern "C" int32_t dg_spgemm_bitmap_words(int32_t n_cols) { return n_cols <= 0 ? 0 : (n_cols + 31) / 32; }

extern "C" int dg_spgemm_pattern_rows(const int32_t *a_rowptr, const int32_t *a_col, int32_t n_rows, const int32_t *b_rowptr,
                                      const int32_t *b_col, int32_t n_mid, int32_t n_cols, uint32_t *bitmap, int32_t *row_nnz,
                                      void *stream) {
  if (!a_rowptr || !a_col || !b_rowptr || !b_col || !bitmap || !row_nnz)
    return dg_fail(DG_ERR_ARG, "dg_spgemm_pattern_rows: NULL argument");
  if (n_rows < 0 || n_mid <= 0 || n_cols <= 0) return dg_fail(DG_ERR_ARG, "dg_spgemm_pattern_rows: bad sizes");
  const int words = dg_spgemm_bitmap_words(n_cols);
  if (words > kMaxWords)
    return dg_fail(DG_ERR_SIZE, "dg_spgemm_pattern_rows: %d columns exceed the %d the row bitmap holds", n_cols, kMaxWords * 32);
  if (n_rows == 0) return 0;
  const size_t lds = (size_t)4 * words * sizeof(uint32_t);
  static std::atomic<uint64_t> opted{0};
  if (lds > 64 * 1024)
    DG_HIP(dg_lds_opt_in(reinterpret_cast<const void *>(spgemm_bitmap_kernel), 4 * kMaxWords * (int)sizeof(uint32_t), opted));
  hipLaunchKernelGGL(spgemm_bitmap_kernel, dim3((n_rows + 3) / 4), dim3(256), lds, (hipStream_t)stream, a_rowptr, a_col, n_rows,
                     b_rowptr, b_col, words, bitmap, row_nnz);
  DG_HIP(hipGetLastError());
  return 0;
}

extern "C" int dg_spgemm_pattern_fill(const uint32_t *bitmap, int32_t n_rows, int32_t n_cols, const int32_t *c_rowptr,
                                      int32_t *c_col, void *stream) {
  if (!bitmap || !c_rowptr || !c_col) return dg_fail(DG_ERR_ARG, "dg_spgemm_pattern_fill: NULL argument");
  if (n_rows < 0 || n_cols <= 0) return dg_fail(DG_ERR_ARG, "dg_spgemm_pattern_fill: bad sizes");
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(spgemm_expand_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, bitmap, n_rows,
                     dg_spgemm_bitmap_words(n_cols), c_rowptr, c_col);
  DG_HIP(hipGetLastError());
  return 0;
}
