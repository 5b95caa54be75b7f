/*
 * diffgraph_hip.h -- C ABI of libdiffgraph_hip.so (MI355X / gfx950).
 *
 * Hand-written HIP kernels for DiffGraph's per-epoch compute hot path: normalised-adjacency
 * propagation with fused LeakyReLU / row-L2-normalise / layer-sum epilogue, and the latent
 * diffusion q_sample / denoiser / reverse p_sample loop.
 *
 * The reference (preetam1407/DiffGraph) is pure Python over DGL + PyTorch and defines no
 * FFI of its own; each entry point below names the Python method (file:line, relative to
 * the reference checkout) whose arithmetic it replaces.
 *
 * Conventions
 *   - Every `const float*` / `float*` / `const int64_t*` data argument is a DEVICE pointer
 *     owned by the caller (in practice a torch tensor's data_ptr()); the library never
 *     frees or retains it.  Host-pointer arguments say so in their name (`*_host`).
 *   - Embedding tables are row-major fp32 [rows][64]; D must be 64 (other values -> -2).
 *   - `stream` is a hipStream_t passed as void*; all launches are asynchronous on it.
 *   - Return value: 0 ok; <0 argument error (dg_last_error() has the text);
 *     >0 a hipError_t from the runtime.
 *   - Re-entrant from any host thread; a dg_graph handle is immutable after creation and
 *     may be shared between threads/streams, except for its hub scratch slab, which
 *     calls on the same handle share (serialise them on one stream).
 */
#ifndef DIFFGRAPH_HIP_H
#define DIFFGRAPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_D 64                 /* embedding width the kernels are built for */
#define DG_MAX_TASKS 8          /* graphs per batched layer launch */

int dg_version(void);
/* Digest (16 hex digits) of the kernel sources the library was built from, "unknown" for a build without one. */
const char *dg_source_digest(void);
const char *dg_last_error(void);

/* ------------------------------------------------------------------------------------
 * Graph handle: replaces the dgl.DGLGraph the reference builds in
 * DiffGraph-Rec/DataHandler.py:120-132 (makeBiAdj: dgl.graph((src,dst), num_nodes)) and
 * DiffGraph_NC/DataHandler.py:174 (dgl.from_scipy).  Edges are directed src -> dst;
 * duplicates count (multigraph), as in DGL.  The handle owns device copies of the CSR
 * (by destination), of out/in-degree norms  max(deg,1)^-1/2  (fp32, computed once; the
 * reference recomputes them on every layer call, Model.py:151-152,165-166) and of a
 * degree-binned row schedule.
 * ---------------------------------------------------------------------------------- */
typedef struct dg_graph dg_graph;

/* group_max / seg_len: schedule knobs, 0 = library default (rows with in-degree <=
 * group_max are summed by one 16-lane group, <= seg_len by one wavefront, longer rows are
 * split into seg_len-neighbour segments whose partial sums go through a scratch slab). */
int dg_graph_create(int32_t n_nodes, int64_t n_edges, const int32_t *src_host,
                    const int32_t *dst_host, int32_t group_max, int32_t seg_len,
                    dg_graph **out);
/* Same, with the node degrees supplied by the caller (int32[n_nodes], NULL = count the given
 * edges).  A rank that holds only a shard of the edges passes the GLOBAL degrees so that its
 * norms equal the whole graph's.  in_deg_host[i] < 0 marks node i as NOT OWNED by this rank:
 * the forward propagation then neither computes nor writes row i (a rank of a row-block
 * partition owns one block of rows and holds only the edges that end in it). */
int dg_graph_create_ex(int32_t n_nodes, int64_t n_edges, const int32_t *src_host,
                       const int32_t *dst_host, const int32_t *out_deg_host,
                       const int32_t *in_deg_host, int32_t group_max, int32_t seg_len,
                       dg_graph **out);
/* A rectangular operator with one fp32 weight per entry, run by the same schedule and kernels as a graph:
 *     out[r] = sum over the entries (r, c) of weight * x[c],   r < n_rows, c < n_cols
 * (weight * x[c] rounded before the sum, entries of a row in ascending c).  The handle counts max(n_rows, n_cols)
 * nodes -- the row count of the table x a layer task gathers from -- of which rows [0, n_rows) are computed and
 * written; its degree norms are 1.  It runs forward only (dg_gcn_layer_fwd / dg_spmm_norm_act_fwd without transpose,
 * no DG_IN_PRESCALED); with slope 1 and no normalisation asked for, the layer epilogue is the identity.
 * Used to add up, in one launch and one output table, the last step of the backward pass over all the graphs of a
 * layer: d e0 = sum_g A_g^T dZ_g (the reference's autograd adds these per graph, DiffGraph-Rec/Model.py:57,74 feed
 * the same table to every graph). */
int dg_graph_create_weighted(int32_t n_rows, int32_t n_cols, int64_t n_entries, const int32_t *row_host,
                             const int32_t *col_host, const float *weight_host, int32_t group_max,
                             int32_t seg_len, dg_graph **out);
void dg_graph_destroy(dg_graph *g);
int32_t dg_graph_num_nodes(const dg_graph *g);
/* rows a layer over this handle computes (= num_nodes except for weighted operators) */
int32_t dg_graph_num_rows(const dg_graph *g);
int64_t dg_graph_num_edges(const dg_graph *g);
/* 1 if the edge set equals its transpose (then forward and backward share one CSR). */
int dg_graph_is_symmetric(const dg_graph *g);
/* Copy schedule/CSR facts back to the host for tests: which = 0 rowptr[n+1], 1 colidx[nnz],
 * 2 dinv_src[n] (float bits), 3 dinv_dst[n], 4 row_order[n_hub + n_wave + n_group] (the owned rows;
 * with lds_split > 0 only those from lds_split on),
 * 5 counts {n_hub,n_wave,n_group,n_seg,lds_split} (lds_split > 0: rows below it gather from an
 * LDS-resident copy of the rows above it). */
int dg_graph_export(const dg_graph *g, int which, void *dst_host, size_t dst_bytes);

/* ------------------------------------------------------------------------------------
 * a1: one propagation layer.  Replaces DGLLayer.forward(graph,u_f,v_f)
 * (DiffGraph-Rec/Model.py:142-174) and UUGCNLayer.forward(graph,u_f)
 * (DiffGraph_NC/Model.py:184-214):
 *     y = LeakyReLU_slope( dinv_dst (.) sum_{j->i} dinv_src[j] * x[j] )
 * x, y: [n_nodes][64].  slope = 1.0f gives the activation-free form.
 * transpose != 0 propagates along reversed edges (used by the backward pass).
 * ---------------------------------------------------------------------------------- */
int dg_spmm_norm_act_fwd(dg_graph *g, const float *x, float *y, int32_t d, float slope,
                         int transpose, void *stream);

/* ------------------------------------------------------------------------------------
 * a1+a4 fused, batched over the graphs of one propagation layer.  Replaces one iteration
 * of the per-graph layer loops in HGDM.forward (DiffGraph-Rec/Model.py:61-70,73-87;
 * DiffGraph_NC/Model.py:52-71): layer call + F.normalize(p=2,dim=1) + running layer sum.
 * For task k, with y = LeakyReLU(...) as above and yn = y / max(||y||_2, 1e-12):
 *     if e_out:   e_out[i]   = (DG_OUT_NORM ? yn : y) * (DG_OUT_PRESCALE ? dinv_src[i] : 1)
 *     if acc_out: acc_out[i] = acc_in[i] + yn
 * DG_IN_PRESCALED says x already carries the dinv_src[j] factor (written by a previous
 * call with DG_OUT_PRESCALE), so the gather is a plain sum.
 * All tasks run in ONE launch (plus one for hub-row partial sums when any graph has them).
 * ---------------------------------------------------------------------------------- */
enum { DG_IN_PRESCALED = 1, DG_OUT_NORM = 2, DG_OUT_PRESCALE = 4, DG_ACC_REBUILD = 16 };
/* DG_ACC_REBUILD (second layer only, needs DG_IN_PRESCALED): the first layer did not write a
 * running sum; acc_in points at the layer-0 table e0 and the sum so far is rebuilt per row as
 * e0[i] + normalize(x[i])  (x[i] = dinv_src[i] * E_1[i], so normalize(x[i]) is the first
 * layer's normalised activation).  Saves one table write and one cold table read per graph. */
enum { DG_OUT_PLANES = 0x200, DG_IN_PLANES = 0x400 };
/* Graphs on the bitmap path (dense graphs) gather from a copy of the table cut into three bf16 planes, made by a
 * launch of its own before each layer.  DG_OUT_PLANES: this task's output is what the NEXT layer task over the same
 * graph gathers -- leave it as planes too (e_out is still written); DG_IN_PLANES (with DG_IN_PRESCALED) on that next
 * task: the planes are in place, skip the cut.  Both are hints: ignored for graphs on other paths (and for bitmap
 * graphs with more than 16 column ranges per row), by the same rule on both sides, so a pair of tasks always agrees.
 * Also accepted in dg_bwd_task::flags, where the output is w_out and the gather source w_in.  Same bits either way. */
enum { DG_ROWS_PARTIAL_ONLY = 64, DG_ROWS_FULL_ONLY = 128 };
/* Row-sharded runs may cut a layer call in two so that the exchange of the partial sums overlaps
 * the rest of the layer: a call with DG_ROWS_PARTIAL_ONLY computes only the rows >= partial_from
 * (their partial sums), a call with DG_ROWS_FULL_ONLY only the rows below; the two together equal
 * one call with neither flag.  All tasks of a call carry the same choice.  Accepted for graphs
 * whose schedule keeps the two row classes in separate launches (one small side held in LDS, with
 * partial_from equal to the first row of that side); an argument error otherwise.  Without
 * partial_out the two calls cut the layer at the same row (the first row of the side held in
 * LDS) and the rows of the first call get the usual epilogue. */
enum { DG_ACC_SHARED = 0x800 };
/* With DG_ROWS_FULL_ONLY on the last layer (DG_IN_PRESCALED | DG_ACC_REBUILD, no e_out): the tasks' rows are walked
 * together, row by row through all graphs, and only two tables are written -- task 0's acc_out as usual, and ONE table
 * named as acc_out by every task from 1 on, which receives the sum of those tasks' layer sums, added in task order
 * (HGDM.forward only ever adds them up, DiffGraph-Rec/Model.py:73-87).  Each value equals, bit for bit, what separate
 * tasks followed by dg_sum_tables give.  All tasks: the same acc_in, graphs of one size with the same side in LDS. */

typedef struct dg_layer_task {
  dg_graph *graph;
  const float *x;        /* gather source [n][64] */
  float *e_out;          /* next layer's input, or NULL */
  const float *acc_in;   /* running layer sum in, or NULL iff acc_out is NULL */
  float *acc_out;        /* running layer sum out (may alias acc_in), or NULL */
  int32_t flags;
  float slope;
  /* Row-sharded runs: rows with id >= partial_from only hold part of their neighbour list on
   * this rank; their plain neighbour sum (no norm, no activation) goes to
   * partial_out[row - partial_from][64] and nothing else is written for them.
   * partial_out == NULL disables this. */
  float *partial_out;
  int32_t partial_from;
  /* [n] ||y[i]||_2 before normalisation, or NULL.  A backward pass through a layer whose
   * NORMALISED activation fed the next layer (DG_OUT_NORM) needs it: e_out then no longer
   * carries the length.  Written only when the row is normalised (acc_out or DG_OUT_NORM). */
  float *row_norm_out;
} dg_layer_task;

int dg_gcn_layer_fwd(const dg_layer_task *tasks, int32_t n_tasks, int32_t d, void *stream);

/* Second half of a row-sharded layer: for rows [row_begin, row_begin + n_rows) of `graph`,
 * add the n_parts partial neighbour sums (p = 0..n_parts-1 in order, gathered from the ranks;
 * part p's row r starts at partials + (p * part_stride_rows + r) * 64) and apply the epilogue of
 * dg_gcn_layer_fwd (norm of the receiving node, LeakyReLU, normalise, layer sum, next layer's
 * input) with the same flags. */
int dg_gcn_partial_finalize(dg_graph *graph, const float *partials, int32_t n_parts,
                            int64_t part_stride_rows, int32_t row_begin, int32_t n_rows,
                            const float *x_prev, float *e_out,
                            const float *acc_in, float *acc_out, int32_t flags, float slope,
                            int32_t d, void *stream);   /* x_prev: the layer's gather source, read only
                                                           under DG_ACC_REBUILD (may be NULL otherwise) */

/* The same for several graphs in ONE launch (the graphs of a layer, as dg_gcn_layer_fwd). */
typedef struct dg_finalize_task {
  dg_graph *graph;
  const float *partials;
  int32_t n_parts;
  int64_t part_stride_rows;
  int32_t row_begin, n_rows;
  const float *x_prev;
  float *e_out;
  const float *acc_in;
  float *acc_out;
  int32_t flags;
  float slope;
} dg_finalize_task;

int dg_gcn_partial_finalize_batched(const dg_finalize_task *tasks, int32_t n_tasks, int32_t d, void *stream);

/* The exchange between the two halves of a row-sharded layer, for callers with an RCCL communicator of their own: one
 * ncclAllGather of `rows_per_rank` rows of 64 floats from every rank into `full` ([ranks][rows_per_rank][64], rank
 * order -- the layout dg_gcn_partial_finalize reads with part_stride_rows = rows_per_rank).  comm: an ncclComm_t
 * passed as void*.  RCCL is looked up when first needed (in the process, else librccl.so): the library does not link
 * against it.  No counterpart in the reference, which runs on one device (DiffGraph_NC/script.sh:1 exports three GPUs
 * and uses one); diffgraph_amd.dist makes the same exchange through torch.distributed. */
int dg_allgather_rows(void *comm, const float *local, float *full, int64_t rows_per_rank, int32_t d, void *stream);

/* ------------------------------------------------------------------------------------
 * a11: backward of one fused propagation layer, batched over graphs.  Replaces what
 * autograd records for DGLLayer.forward + F.normalize + the layer sum
 * (loss.backward(), DiffGraph-Rec/main.py:127) -- DGL's SpMM backward is the SpMM over
 * reversed edges.  With node j, layer l, saved activation E_l and d_sum = dLoss/d(layer sum):
 *     G[j]     = dinv_src[j] * sum_{j->i} w_in[i]                (0 when w_in == NULL)
 *     dE[j]    = G[j] + normalize_backward(d_sum[j]; E_l[j])
 *     w_out[j] = dinv_dst[j] * LeakyReLU'(E_l[j]) * dE[j]        (dZ_l, scaled for the next gather)
 * and, when e_saved == NULL (layer 0, the embedding table itself):  w_out[j] = G[j].
 * DG_BWD_SAVED_PRESCALED says e_saved holds dinv_src[j]*E_l[j] (what a forward call with
 * DG_OUT_PRESCALE wrote); otherwise it holds E_l[j].
 * ---------------------------------------------------------------------------------- */
enum { DG_BWD_SAVED_PRESCALED = 8, DG_BWD_NORM_FEED = 32 };
/* DG_BWD_NORM_FEED: the layer sum AND the next layer were fed the normalised activation
 * (DiffGraph_NC/Model.py:53-55: E_l = normalize(layer(E_{l-1}))).  Then e_saved holds the
 * normalised activation (times dinv_src[j] under DG_BWD_SAVED_PRESCALED), row_norm the
 * lengths ||y_l[j]|| the forward wrote through row_norm_out, and
 *     dE[j] = normalize_backward(d_sum[j] + G[j]; y_l[j]). */

typedef struct dg_bwd_task {
  dg_graph *graph;
  const float *w_in;     /* dZ_{l+1} scaled by dinv_dst, [n][64], or NULL for the last layer */
  const float *e_saved;  /* activation of layer l from the forward pass, or NULL for layer 0 */
  const float *d_sum;    /* gradient wrt this graph's layer sum, [n][64] (unused when e_saved == NULL) */
  float *w_out;
  int32_t flags;
  float slope;
  const float *row_norm; /* [n], only read under DG_BWD_NORM_FEED */
} dg_bwd_task;

int dg_gcn_layer_bwd(const dg_bwd_task *tasks, int32_t n_tasks, int32_t d, void *stream);

/* out[i] = ((in[0][i] + in[1][i]) + in[2][i]) ... left to right; replaces Python
 * sum(list) over per-behaviour sums (DiffGraph-Rec/Model.py:87).  n_in <= 16. */
int dg_sum_tables(const float *const *in_host, int32_t n_in, float *out, int64_t n_elems,
                  void *stream);

/* ------------------------------------------------------------------------------------
 * a2 as ONE row walk over all graphs: HGDM.forward (DiffGraph-Rec/Model.py:53-92) runs the same
 * two-layer propagation over the target graph (Model.py:61-70) and every behaviour graph
 * (Model.py:73-84), all from the one embedding table, and uses only the target graph's layer
 * sum and sum(list) of the behaviour graphs' layer sums (Model.py:87,92).  A dg_graphset holds
 * 2..4 symmetric graphs over the same nodes (graph 0 = the target graph) with ONE row schedule:
 * a 16-lane group sums row r's neighbours in every graph back to back and applies the graphs'
 * epilogues together, so the layer-0 row is read once per row instead of once per graph and
 * the per-behaviour sums never reach memory.  Per-graph tables are STACKED: [graphs][n][64].
 * With y_g = LeakyReLU_slope(dinv_g[i] * sum_{j in N_g(i)} .) as in dg_spmm_norm_act_fwd:
 *   dg_set_layer_first:  e_out[g][i] = dinv_g[i] * y_g        (gathering dinv_g[j] * x[j], x = [n][64])
 *   dg_set_layer_last:   S_g = (e0[i] + normalize(x[g][i])) + normalize(y_g)   (gathering x[g][j], the first call's e_out)
 *                        sum_first[i] = S_0;  sum_rest[i] = (S_1 + S_2) + S_3;  e_out[g][i] = y_g (if not NULL)
 * Backward (a11), with d_first / d_rest the gradients wrt sum_first / sum_rest (graph 0 takes d_first, the others
 * d_rest) and the formulas of dg_gcn_layer_bwd:
 *   dg_set_layer_bwd_top:  w_out[g] = dZ_2 of graph g from e_saved[g] = the last call's e_out   (no gather)
 *   dg_set_layer_bwd_mid:  w_out[g] = dZ_1 of graph g from w_in = the top step's w_out (gathered) and
 *                          e_saved[g] = the first call's e_out
 *   dg_set_layer_bwd_last: g_out[i] = sum_g dinv_g[i] * sum_{j in N_g(i)} w_in[g][j]   (w_in = the mid step's w_out): the
 *                          gradient that reaches the layer-0 table through the propagation, over all graphs, ONE table
 * In the top and mid calls w_out may be e_saved itself (the result written over the activations just consumed).
 * The graphs must outlive nothing: the set copies what it needs.  Calls on one set share its slab: serialise them
 * on one stream.
 * ---------------------------------------------------------------------------------- */
typedef struct dg_graphset dg_graphset;
/* group_max: rows whose padded lists hold at most this many slots in all graphs together are summed by one group
 * (0 = 128); seg_len as in dg_graph_create. */
int dg_graphset_create(dg_graph *const *graphs, int32_t n_graphs, int32_t group_max, int32_t seg_len,
                       dg_graphset **out);
void dg_graphset_destroy(dg_graphset *set);
/* counts_host[7] = {nodes, graphs, cut rows, slab rows, quads a group row may hold, seg_len, side split (0: none)} */
int dg_graphset_counts(const dg_graphset *set, int32_t *counts_host);
int dg_set_layer_first(dg_graphset *set, const float *x, float *e_out, float slope, int32_t d, void *stream);
int dg_set_layer_last(dg_graphset *set, const float *x, const float *e0, float *sum_first, float *sum_rest,
                      float *e_out, float slope, int32_t d, void *stream);
/* row_flags (nullable; DEVICE int32[nodes], e.g. the flags of dg_active_rows): rows whose flag is 0 have zero rows in
 * d_first and d_rest -- their rows of w_out are written as zeros without reading anything else. */
int dg_set_layer_bwd_top(dg_graphset *set, const float *e_saved, const float *d_first, const float *d_rest,
                         float *w_out, const int32_t *row_flags, float slope, int32_t d, void *stream);
int dg_set_layer_bwd_mid(dg_graphset *set, const float *w_in, const float *e_saved, const float *d_first,
                         const float *d_rest, float *w_out, float slope, int32_t d, void *stream);
int dg_set_layer_bwd_last(dg_graphset *set, const float *w_in, float *g_out, int32_t d, void *stream);
/* dg_set_layer_bwd_last with the embedding table's optimizer step in its epilogue (main.py:127-128 runs
 * loss.backward() and opt.step() back to back): a row's gradient -- d_first + d_rest once per behaviour graph (every
 * graph's layer sum starts from the table, Model.py:57,74) + g_out, added left to right -- goes straight into
 * torch.optim.Adam's update of param / exp_avg / exp_avg_sq at step count state[0] + 1 and never reaches memory.
 * Same bits as dg_set_layer_bwd_last, then dg_adam_prepare + dg_adam_step_prepared over those parts.  `state` is
 * dg_adam_step's [4]: the call also moves its step count state[0] on by one (through the spare word state[3]; the
 * bias corrections in state[1..2] are left as they were -- dg_adam_prepare rebuilds them from the count). */
int dg_set_layer_bwd_last_adam(dg_graphset *set, const float *w_in, const float *d_first, const float *d_rest, float *param,
                               float *exp_avg, float *exp_avg_sq, float *state, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int32_t d, void *stream);

/* ------------------------------------------------------------------------------------
 * a6: q_sample.  Replaces GaussianDiffusion.q_sample + _extract_into_tensor
 * (DiffGraph-Rec/Model.py:302-312):  x_t[i] = sqrt_ac[t_i]*x0[i] + sqrt_1mac[t_i]*noise[i].
 * sqrt_ac / sqrt_1mac: DEVICE fp32 tables of length n_steps (the fp64 host tables cast to
 * fp32, which equals the reference's index-then-cast).  t: int64[n], each in [0,n_steps).
 * ---------------------------------------------------------------------------------- */
int dg_q_sample(const float *x0, const float *noise, const int64_t *t, const float *sqrt_ac,
                const float *sqrt_1mac, int32_t n_steps, int64_t n, int32_t d, float *x_t,
                void *stream);

/* Denoiser parameters, torch nn.Linear layout [out][in] (DiffGraph-Rec/Model.py:184-191
 * with dims='[64]', d_emb_size=8): emb_layer 8x8, in_layers[0] 64x72, out_layers[0] 64x64.
 * freqs: the 4 fp32 frequencies exp(-ln(1e4)*k/4) (Model.py:215), HOST pointer. */
typedef struct dg_denoiser {
  const float *w_emb, *b_emb;   /* [8][8], [8]   device */
  const float *w_in, *b_in;     /* [64][72], [64] device */
  const float *w_out, *b_out;   /* [64][64], [64] device */
  const float *freqs_host;      /* [4] host */
  int32_t norm;                 /* Denoise.norm (args.norm, default True) */
} dg_denoiser;

/* Workspace the denoiser kernels need: a [n_steps][64] fp32 per-timestep bias table
 * b_in + W_in[:,64:72] . emb_layer([cos(t f), sin(t f)])  (Model.py:215-220,225-227).
 * dg_time_bias_table fills it (device pointer, n_steps*64 floats). */
int dg_time_bias_table(const dg_denoiser *dn, int32_t n_steps, float *table, void *stream);
/* The tables of up to four denoisers (a training step's usr_denoiser and item_denoiser) in one launch. */
int dg_time_bias_tables(const dg_denoiser *const *dns, float *const *tables, int32_t n_tables,
                        int32_t n_steps, void *stream);

/* a7: Denoise.forward(x, timesteps, mess_dropout=False) (DiffGraph-Rec/Model.py:214-234),
 * eval-mode arithmetic.  x,out: [n][64]; t: int64[n] in [0,n_steps); table from above. */
int dg_denoise_fwd(const dg_denoiser *dn, const float *time_bias, int32_t n_steps, const float *x,
                   const int64_t *t, int64_t n, int32_t d, float *out, void *stream);

/* The same in train() mode with mess_dropout=True (Model.py:221-224): nn.Dropout(drop_p) between F.normalize and the
 * first Linear, over the caller's keep_mask (uint8 [n][64], nonzero = kept) or a mask drawn from the device word
 * *drop_seed (task number 0 of dg_dropout_keep_mask) -- see dg_loss_task for the two forms.  drop_p = 0: dg_denoise_fwd. */
int dg_denoise_fwd_train(const dg_denoiser *dn, const float *time_bias, int32_t n_steps, const float *x,
                         const int64_t *t, int64_t n, int32_t d, const uint8_t *keep_mask,
                         const int64_t *drop_seed, float drop_p, float *out, void *stream);

/* a8: GaussianDiffusion.p_sample(model, x_start, steps=0) reverse loop
 * (DiffGraph-Rec/Model.py:287-300 with p_mean_variance :314-325): for i = n_steps-1..0:
 *     x <- coef1[i]*Denoise(x, i) + coef2[i]*x
 * run entirely in registers, one launch.  coef1/coef2: DEVICE fp32 tables (posterior mean
 * coefficients cast to fp32).  x_T is the caller's job (steps==0 -> x_start itself;
 * otherwise dg_q_sample first).  x_T and x_0 may alias.
 * The contractions run on the bf16 matrix instruction with every fp32 value cut into three
 * bf16 numbers (six products per fp32 product: what is left out is below one fp32 rounding),
 * or, with DG_PSAMPLE_FP32=1 in the environment, on the exact-fp32 matrix instruction; the
 * two forms agree to a few 1e-7 of a row's largest entry after all steps. */
int dg_p_sample_loop(const dg_denoiser *dn, const float *time_bias, const float *coef1,
                     const float *coef2, int32_t n_steps, const float *x_T, int64_t n, int32_t d,
                     float *x_0, void *stream);

/* The same loop for several (denoiser, table) pairs in ONE launch -- e.g. the user and the item
 * denoiser of HGDM.predict (DiffGraph-Rec/Model.py:114-115) -- so that a small table does not
 * leave most of the chip idle.  n_tasks <= 4. */
typedef struct dg_psample_task {
  const dg_denoiser *dn;
  const float *time_bias;
  const float *x_T;
  float *x_0;
  int64_t n;
} dg_psample_task;

int dg_p_sample_loop_batched(const dg_psample_task *tasks, int32_t n_tasks, const float *coef1,
                             const float *coef2, int32_t n_steps, int32_t d, void *stream);

/* a9: GaussianDiffusion.training_losses2 forward (DiffGraph-Rec/Model.py:348-366) with the
 * random draws supplied by the caller: x_t = q_sample(x_start, ts, noise) (skipped when
 * sqrt_ac == NULL, the noise_scale == 0 case), out = Denoise(x_t, ts),
 * row_loss[i] = loss_w[ts_i] * mean_j (target[i][j]-out[i][j])^2   (fp64, weight table
 * loss_w = SNR(t-1)-SNR(t), 1 at t=0, DEVICE fp64[n_steps]).  The caller gathers
 * row_loss[batch].  x_t_out (nullable) receives x_t for the backward pass. */
int dg_diffusion_loss_fwd(const dg_denoiser *dn, const float *time_bias, int32_t n_steps,
                          const float *target, const float *x_start, const float *noise,
                          const int64_t *ts, const float *sqrt_ac, const float *sqrt_1mac,
                          const double *loss_w, int64_t n, int32_t d, float *out, double *row_loss,
                          float *x_t_out, void *stream);

/* a11: backward of dg_diffusion_loss_fwd (what autograd records for training_losses2,
 * DiffGraph-Rec/Model.py:348-366).  g_row_loss: DEVICE fp64[n] = dL/d row_loss; g_out: DEVICE
 * [n][64] = dL/d out from downstream users of `out`, or NULL.  Writes dL/d x_start and
 * dL/d target, and the four [n][64] tables whose products give the weight gradients:
 *     dW_out = dout_tot^T . hid      db_out = colsum(dout_tot)
 *     dW_in[:, :64] = dpre^T . xn    d time_bias[t] = sum_{i: ts_i = t} dpre[i]
 * (dg_denoiser_wgrad computes them). */
int dg_diffusion_loss_bwd(const dg_denoiser *dn, const float *time_bias, int32_t n_steps,
                          const float *target, const float *x_start, const float *noise,
                          const int64_t *ts, const float *sqrt_ac, const float *sqrt_1mac,
                          const double *loss_w, const double *g_row_loss, const float *g_out,
                          int64_t n, int32_t d, float *d_x_start, float *d_target, float *dout_tot,
                          float *hid, float *dpre, float *xn, void *stream);

/* dg_diffusion_loss_fwd / _bwd for several (denoiser, table) pairs in ONE launch each -- the user
 * and the item denoiser of HGDM.cal_loss (DiffGraph-Rec/Model.py:96-97).  Fields as the arguments
 * of the single-pair entry points; the forward reads the first block, the backward also the second.
 * d_target_add (nullable): a [n][64] table added into d_target, for a gradient that reached
 * `target` by another route (Model.py:103-104 adds the denoiser output to the same embeddings the
 * loss compares it with).  n_tasks <= 4. */
typedef struct dg_loss_task {
  const dg_denoiser *dn;
  const float *time_bias;
  const float *target, *x_start, *noise;
  const int64_t *ts;
  int64_t n;
  float *out;               /* forward */
  double *row_loss;
  const double *g_row_loss; /* backward */
  const float *g_out, *d_target_add;
  float *d_x_start, *d_target, *dout_tot, *hid, *dpre, *xn;
  /* Training-mode dropout of Denoise.forward (nn.Dropout between F.normalize and the first Linear,
   * DiffGraph-Rec/Model.py:193,221-224; active while the module is in train() mode, i.e. the reference's
   * first epoch, main.py:144).  drop_p = 0: off.  With keep_mask (uint8 [n][64]): an element is multiplied by
   * 1/(1-drop_p) where keep_mask[row][col] != 0 and by 0 elsewhere (torch's arithmetic on a mask torch drew).
   * Without: the mask is drawn here, 8 random bits per element, from a counter-based generator keyed by the
   * DEVICE word *drop_seed, the task's position in the call and the row -- so the rate applied is
   * q = round(256 drop_p)/256 (argument error unless 0 < q < 1) and kept elements are multiplied by 1/(1-q),
   * which keeps the expectation (drop_p = 0.5, the reference's rate: q = drop_p).  The backward call redraws the
   * same mask from the same word; dg_dropout_keep_mask writes it out. */
  const uint8_t *keep_mask;
  const int64_t *drop_seed;
  float drop_p;
  /* Backward only.  wg_partials != NULL: the kernel forms the weight gradients' partial sums itself, one partial
   * result of dg_wgrad_partial_floats() floats per 128 rows ((n + 127) / 128 of them, every one written whole),
   * for dg_denoiser_wgrad_reduce; dout_tot / hid / dpre / xn are then not written and may be NULL.  wg_tab: the
   * denoiser's operand table from dg_denoiser_wgrad_tab ([n_steps][16]). */
  const float *wg_tab;
  float *wg_partials;
  /* rows != NULL: only the table rows rows[0 .. *n_rows) take part (DEVICE int32[rows_cap], ascending; *n_rows a DEVICE
   * word <= rows_cap <= n; see dg_active_rows).  The forward then writes out / row_loss of those rows only; the backward
   * writes d_x_start / d_target of those rows only (hand in zeroed tables; d_target may be d_target_add itself) and
   * takes g_row_loss / g_out as zero elsewhere -- which they are when the loss is only read at listed rows
   * (diff_loss[batch] and the batch gathers of HGDM.cal_loss, DiffGraph-Rec/Model.py:96-104: a row outside the
   * batch gets no gradient and its output is never read).  wg_partials: (rows_cap + 127) / 128 partial results. */
  const int32_t *rows, *n_rows;
  int32_t rows_cap;
} dg_loss_task;

/* The random inputs of one training_losses2 call (t.randint(0, steps, (batch,)) and t.randn_like(x_start),
 * DiffGraph-Rec/Model.py:352-353) for n rows: ts int64[n] uniform in [0, n_steps), noise fp32 [n][64] standard
 * normal (either may be NULL).  state: int64[4] on the device = {seed, draws made so far, key of this draw's
 * dropout masks, spare}; the draw is a function of (seed, draws made) alone, the call then increments state[1] and
 * writes state[2] (dg_loss_task.drop_seed may point at it) -- all on the device, so a recorded step draws anew on
 * every replay.  The streams are not torch's: only the distribution is the reference's. */
int dg_diffusion_draw(int64_t *state, int64_t n, int32_t n_steps, int32_t d, int64_t *ts, float *noise, void *stream);

/* The keep mask (uint8 [n][64], 1 = kept) the loss entry points draw for task number `task` from *drop_seed. */
int dg_dropout_keep_mask(const int64_t *drop_seed, int32_t task, float drop_p, int64_t n, int32_t d, uint8_t *mask,
                         void *stream);

int dg_diffusion_loss_fwd_batched(const dg_loss_task *tasks, int32_t n_tasks, int32_t n_steps,
                                  const float *sqrt_ac, const float *sqrt_1mac, const double *loss_w,
                                  int32_t d, void *stream);
int dg_diffusion_loss_bwd_batched(const dg_loss_task *tasks, int32_t n_tasks, int32_t n_steps,
                                  const float *sqrt_ac, const float *sqrt_1mac, const double *loss_w,
                                  int32_t d, void *stream);

/* a11: weight gradients of the denoiser from the four tables dg_diffusion_loss_bwd wrote and the
 * per-row timesteps -- what autograd records for the three nn.Linear layers of Denoise
 * (DiffGraph-Rec/Model.py:184-191, used at :218-232) under loss.backward() (DiffGraph-Rec/main.py:127):
 *     d_w_out = dout_tot^T . hid     d_b_out = colsum(dout_tot)
 *     d_w_in  = [ dpre^T . xn | dpre^T . emb[ts] ]  (64x72)     d_b_in = colsum(dpre)
 *     d_w_emb, d_b_emb: through emb[t] = emb_layer([cos(t f), sin(t f)])
 * Outputs are DEVICE arrays in torch's [out][in] layout, overwritten.  Sums run in a fixed order
 * (same bits on every run).  workspace: DEVICE scratch of dg_denoiser_wgrad_workspace(n_steps)
 * bytes, 16-byte aligned; ts: int64[n] in [0,n_steps). */
size_t dg_denoiser_wgrad_workspace(int32_t n_steps);
int dg_denoiser_wgrad(const dg_denoiser *dn, int32_t n_steps, const int64_t *ts, const float *dout_tot,
                      const float *hid, const float *dpre, const float *xn, int64_t n, int32_t d,
                      void *workspace, size_t workspace_bytes, float *d_w_emb, float *d_b_emb,
                      float *d_w_in, float *d_b_in, float *d_w_out, float *d_b_out, void *stream);

/* dg_denoiser_wgrad without the workspace argument.  No counterpart in the reference: it is for callers
 * that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  The scratch (timestep operand table + per-workgroup partial results, 9.2 MiB) is a
 * device array inside the library, one copy per GPU, held until the process ends; the call allocates and
 * frees nothing.  n_steps must not exceed dg_denoiser_wgrad_static_max_steps() (1024), else DG error -3.
 * Same kernels and summation order as dg_denoiser_wgrad: same bits.  All calls on one GPU share that GPU's
 * scratch: calls that overlap on one GPU (different streams or threads) race; serialise them. */
int32_t dg_denoiser_wgrad_static_max_steps(void);
int dg_denoiser_wgrad_static(const dg_denoiser *dn, int32_t n_steps, const int64_t *ts, const float *dout_tot,
                             const float *hid, const float *dpre, const float *xn, int64_t n, int32_t d,
                             float *d_w_emb, float *d_b_emb, float *d_w_in, float *d_b_in, float *d_w_out,
                             float *d_b_out, void *stream);

/* The same for up to two denoisers (a training step's usr_denoiser and item_denoiser,
 * DiffGraph-Rec/Model.py:34-35) in one set of four launches; every task brings its own workspace. */
typedef struct dg_wgrad_task {
  const dg_denoiser *dn;
  const int64_t *ts;
  const float *dout_tot, *hid, *dpre, *xn;
  int64_t n;
  void *workspace;
  size_t workspace_bytes;
  float *d_w_emb, *d_b_emb, *d_w_in, *d_b_in, *d_w_out, *d_b_out;
} dg_wgrad_task;
int dg_denoiser_wgrad_batched(const dg_wgrad_task *tasks, int32_t n_tasks, int32_t n_steps, int32_t d,
                              void *stream);

/* The same gradients without the four operand tables going through memory: dg_diffusion_loss_bwd_batched forms
 * the partial sums (dg_loss_task.wg_partials / wg_tab) and dg_denoiser_wgrad_reduce adds them up in partial order
 * (the same bits on every run) and finishes emb_layer's gradients.
 * dg_denoiser_wgrad_tab: the per-timestep operand table [n_steps][16] = emb_layer(te_t) | te_t of up to two
 * denoisers (depends on emb_layer's weights only).  mk: DEVICE scratch of 512 floats per task. */
int64_t dg_wgrad_partial_floats(void);
int dg_denoiser_wgrad_tab(const dg_denoiser *const *dns, float *const *tabs, int32_t n_tabs, int32_t n_steps, void *stream);
typedef struct dg_wgrad_reduce_task {
  const dg_denoiser *dn;
  const float *partials;
  int32_t n_partials;
  float *mk;
  float *d_w_emb, *d_b_emb, *d_w_in, *d_b_in, *d_w_out, *d_b_out;
} dg_wgrad_reduce_task;
int dg_denoiser_wgrad_reduce(const dg_wgrad_reduce_task *tasks, int32_t n_tasks, void *stream);

/* The table rows a batch of HGDM.cal_loss names (DiffGraph-Rec/Model.py:96-104: diff_loss[ancs] / [poss] and the
 * gathers usrEmbeds[ancs], itmEmbeds[poss], itmEmbeds[negs]): rows_u = the distinct ancs, ascending (DEVICE
 * int32[n_batch]), rows_i = the distinct poss and negs, ascending (DEVICE int32[2 n_batch]), counts = DEVICE int32[2]
 * = how many of each.  Indices outside their table are left out.  flags: DEVICE scratch, int32[n_user + n_item].
 * For dg_loss_task.rows: nothing else of training_losses2's tables is ever read. */
int dg_active_rows(const int64_t *ancs, const int64_t *poss, const int64_t *negs, int32_t n_batch, int32_t n_user,
                   int32_t n_item, int32_t *flags, int32_t *rows_u, int32_t *rows_i, int32_t *counts, void *stream);

/* ------------------------------------------------------------------------------------
 * a10: the loss head of HGDM.cal_loss (DiffGraph-Rec/Model.py:101-111; pairPredict,
 * DiffGraph-Rec/Utils/Utils.py:4-8), forward and backward.  With, per batch entry b,
 *     A = ui[ancs_b] + out_u[ancs_b],  P = ui[n_user+poss_b] + out_i[poss_b],  N likewise for negs_b,
 *     s_b = <A,P> - <A,N>:
 *     bpr  = -sum_b log(sigmoid(s_b)) / batch_div
 *     reg  = reg_coef * sum_b (|A|^2 + |P|^2 + |N|^2) / batch_div
 *     diff = mean_b row_loss_u[ancs_b] + mean_b row_loss_i[poss_b]            (fp64)
 *     loss = bpr + reg + diff
 * ui: [n_user+n_item][64] layer sums of the target graph; out_u/out_i: denoised embeddings and
 * row_loss_u/row_loss_i: fp64 per-row losses from dg_diffusion_loss_fwd; ancs/poss/negs: DEVICE
 * int64[n_batch] (each in range; not checked on the device).  n_batch <= 8192.
 * results: DEVICE fp64[4] = {loss, bpr, reg, diff}.  entries: DEVICE workspace, 8 floats and
 * 2 doubles per batch entry (float[8*n_batch] then, 8-byte aligned, double[2*n_batch]), kept
 * for the backward call.
 * ---------------------------------------------------------------------------------- */
int dg_rec_head_fwd(const float *ui, const float *out_u, const float *out_i, const double *row_loss_u,
                    const double *row_loss_i, const int64_t *ancs, const int64_t *poss,
                    const int64_t *negs, int32_t n_batch, int32_t n_user, int32_t n_item,
                    double batch_div, double reg_coef, int32_t d, float *entries_f,
                    double *entries_d, double *results, void *stream);
/* results == NULL above leaves out the last launch (the sums over the entries); dg_rec_head_reduce is that launch
 * alone -- the backward call needs the entries only, so the sums may run beside it on another stream. */
int dg_rec_head_reduce(const float *entries_f, const double *entries_d, int32_t n_batch, double batch_div,
                       double reg_coef, double *results, void *stream);

/* Clears `bytes` bytes at dst on `stream` (hipMemsetAsync): the zero state dg_rec_head_bwd's buffer has to be handed
 * over in -- the zero-initialised gradient tables autograd creates for the batch gathers of DiffGraph-Rec/Model.py:101-104. */
int dg_clear(void *dst, size_t bytes, void *stream);

/* Backward of dg_rec_head_fwd.  g: DEVICE fp64[4] = dL/d{loss,bpr,reg,diff} (what autograd hands
 * back; usually {1,0,0,0}).  Writes the dense gradient tables d_ui [n_user+n_item][64],
 * d_out_u [n_user][64], d_out_i [n_item][64], d_row_loss_u fp64[n_user], d_row_loss_i fp64[n_item]:
 * rows of batch entries get their gradient (entries that name the same row are added in a fixed
 * order -- batch order, or 64 runs of the batch for rows named more than six times -- so the result
 * does not depend on timing); ALL OTHER ROWS MUST BE ZERO ON ENTRY (the caller clears the tables).
 * scratch: DEVICE float[3*n_batch*65 + n_batch + 4], 16-byte aligned like the tables.
 * row_tags: DEVICE int32[8*(n_user+n_item)] whose first 2*(n_user+n_item) words are ZERO ON ENTRY
 * (per-row first-entry marks and entry counts, then six entry slots per row; left dirty). */
int dg_rec_head_bwd(const float *ui, const float *out_u, const float *out_i, const int64_t *ancs,
                    const int64_t *poss, const int64_t *negs, int32_t n_batch, int32_t n_user,
                    int32_t n_item, double batch_div, double reg_coef, int32_t d,
                    const float *entries_f, const double *g, float *scratch, int32_t *row_tags,
                    float *d_ui, float *d_out_u, float *d_out_i, double *d_row_loss_u,
                    double *d_row_loss_i, void *stream);

/* ------------------------------------------------------------------------------------
 * The node-classification model's two dense layers (DiffGraph_NC/Model.py).
 * dg_linear_fwd: y[n][m] = x[n][k] . w[m][k]^T + bias[m] (bias may be NULL) -- transform_layer = nn.Linear(f_dim,
 * latdim) on the feature matrix (:38-40,47).  dg_linear_wgrad: what autograd forms for it, d_w[m][k] = dy^T . x and
 * d_b[m] = column sums of dy, the table rows added in 32 strips whose partial sums are added in strip order (the same
 * bits on every run); workspace: DEVICE, dg_linear_wgrad_workspace(k, m) bytes.
 * dg_nc_logits: logits[n][n_cls] = e[n][64] . w[n_cls][64]^T + b (self.dense, :84,107; n_cls <= 16).
 * dg_nc_nll_fwd: loss[0] = F.nll_loss(F.log_softmax(logits, 1)[ancs], labels) (:84-89; mean over the batch, fp32);
 * entries: DEVICE float[n_batch][n_cls], the batch rows' softmax, kept for dg_nc_nll_bwd, which writes d_w / d_b of
 * the dense layer and ADDS the batch rows' gradients into d_e[n][64] (zeroed by the caller; entries in batch order,
 * so a row named twice is added twice in that order); g: DEVICE float[1], the upstream gradient of the loss.  An
 * index outside [0, n) or a label outside [0, n_cls) makes the loss NaN and is not gathered through.
 * ---------------------------------------------------------------------------------- */
int dg_linear_fwd(const float *x, const float *w, const float *bias, int32_t n, int32_t k, int32_t m, float *y,
                  void *stream);
size_t dg_linear_wgrad_workspace(int32_t k, int32_t m);
int dg_linear_wgrad(const float *dy, const float *x, int32_t n, int32_t k, int32_t m, void *workspace,
                    size_t workspace_bytes, float *d_w, float *d_b, void *stream);
int dg_nc_logits(const float *e, const float *w, const float *b, int64_t n, int32_t d, int32_t n_cls, float *logits,
                 void *stream);
int dg_nc_nll_fwd(const float *logits, const int64_t *ancs, const int64_t *labels, int32_t n_batch, int64_t n,
                  int32_t n_cls, float *entries, float *loss, void *stream);
int dg_nc_nll_bwd(const float *e, const float *w, const int64_t *ancs, const int64_t *labels, const float *entries,
                  const float *g, int32_t n_batch, int64_t n, int32_t d, int32_t n_cls, float *d_e, float *d_w,
                  float *d_b, void *stream);
/* The loss head of DiffGraph_NC's cal_loss over the batch rows only, in one launch each way (Model.py:79-89):
 *     e = target[a] + diff[a]  (all_embeddings, :83);  nll[0] = F.nll_loss(F.log_softmax(dense(e)), label)  (fp32);
 *     diff_loss[0] = mean_b row_loss[a_b]  (fp64: training_losses2's per-row loss gathered by the batch and averaged, :80-82)
 * label_b = argmax(onehot[a_b]) (DEVICE float[n][n_cls], first maximum, :86) or, with onehot NULL, labels[b].
 * entries (DEVICE float[n_batch][n_cls]) and label_out (DEVICE int32[n_batch], -1 = index out of range: both losses
 * NaN, the entry not gathered through) are kept for dg_nc_head_bwd, which ADDS the batch rows' gradients into
 * d_e[n][64] (= the gradient of target and of diff alike) and d_row_loss[n] (fp64) -- both zeroed by the caller,
 * entries in batch order -- and writes d_w / d_b of the dense layer; g_nll / g_diff: DEVICE float[1] / double[1]. */
int dg_nc_head_fwd(const float *target, const float *diff, const double *row_loss, const float *w, const float *b,
                   const int64_t *ancs, const float *onehot, const int64_t *labels, int32_t n_batch, int64_t n,
                   int32_t d, int32_t n_cls, float *entries, int32_t *label_out, float *nll, double *diff_loss,
                   void *stream);
int dg_nc_head_bwd(const float *target, const float *diff, const float *w, const int64_t *ancs,
                   const int32_t *label_in, const float *entries, const float *g_nll, const double *g_diff,
                   int32_t n_batch, int64_t n, int32_t d, int32_t n_cls, float *d_e, double *d_row_loss, float *d_w,
                   float *d_b, void *stream);

/* ------------------------------------------------------------------------------------
 * Adam step of one table, the gradient given as the n_parts (<= 16) tables whose left-to-right sum
 * it is (HOST array of DEVICE pointers).  Replaces torch.optim.Adam.step() for the embedding table
 * (DiffGraph-Rec/main.py:117-119,127-128; amsgrad off) together with the final sum of the backward
 * pass.  exp_avg / exp_avg_sq: DEVICE moment tables like param; state: DEVICE float[4] whose first
 * word is the number of steps taken so far (0 before the first call; the call advances it, so a
 * captured launch steps on every replay).  n_elems % 4 == 0.
 * ---------------------------------------------------------------------------------- */
int dg_adam_step(float *param, const float *const *grad_parts_host, int32_t n_parts, float *exp_avg,
                 float *exp_avg_sq, float *state, int64_t n_elems, float lr, float beta1, float beta2,
                 float eps, float weight_decay, void *stream);
/* The same in two calls, so that the first (one thread: the step count and the two bias corrections of the step
 * to come) can be issued early, on another stream: dg_adam_prepare, then dg_adam_step_prepared ordered after it. */
int dg_adam_prepare(float *state, float beta1, float beta2, void *stream);
int dg_adam_step_prepared(float *param, const float *const *grad_parts_host, int32_t n_parts, float *exp_avg,
                          float *exp_avg_sq, float *state, int64_t n_elems, float lr, float beta1, float beta2,
                          float eps, float weight_decay, void *stream);

/* torch.optim.Adam.step() (same reference lines; amsgrad off) for n_tensors (<= 32) small parameter tensors in one
 * launch: HOST arrays of DEVICE pointers (parameter, its gradient, its two moments) and of element counts.  state:
 * DEVICE float[4], zero before the first step: word 0 the steps taken so far, advanced by the launch itself (a
 * recorded launch steps on every replay), word 3 a counter the launch uses and leaves at zero. */
int dg_adam_multi(float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                  float *const *exp_avg_sq_host, const int64_t *counts_host, int32_t n_tensors, float *state,
                  float lr, float beta1, float beta2, float eps, float weight_decay, void *stream);

/* ------------------------------------------------------------------------------------
 * All-rank evaluation.  Replaces the per-batch body of Coach.testEpoch
 * (DiffGraph-Rec/main.py:153-160) with the dense train-mask rows of AllRankTestData
 * (DiffGraph-Rec/DataHandler.py:268-272):
 *     allPreds = usr_emb[u] . item_emb^T * (1 - trnMask) - trnMask * 1e8 ;  topk(allPreds, k)
 * for the n_test users test_users[] (DEVICE int64).  The training interactions come as a CSR over
 * users (DEVICE int32 trn_rowptr[n_user+1], trn_col[] ascending inside a row) instead of dense mask
 * rows.  top_idx / top_score: DEVICE [n_test][k], by falling score; equal scores by rising item id
 * (torch.topk leaves that order open); -1 where a user has fewer than k items.  k <= 32.
 * workspace: DEVICE scratch of dg_eval_topk_workspace() bytes.
 * The scores are fp32 dot products; for k <= 20 they are formed on the bf16 matrix instruction from the fp32
 * values cut into three bf16 numbers each (six products per term: what is left out is below one fp32 rounding of
 * the score), with DG_EVAL_FP32=1 in the environment on the exact-fp32 instruction -- the two differ the way two
 * fp32 summation orders differ.
 * ---------------------------------------------------------------------------------- */
size_t dg_eval_topk_workspace(int32_t n_test, int32_t n_item, int32_t k);
int dg_eval_topk(const float *usr_emb, const float *item_emb, int32_t n_user, int32_t n_item,
                 const int64_t *test_users, int32_t n_test, const int32_t *trn_rowptr,
                 const int32_t *trn_col, int32_t k, int32_t d, void *workspace, size_t workspace_bytes,
                 int32_t *top_idx, float *top_score, void *stream);

/* Coach.calcRes (DiffGraph-Rec/main.py:168-187) per test user: recall[u] = hits / tstNum,
 * ndcg[u] = dcg / maxDcg in fp64, the user's test items (DEVICE int32 CSR tst_rowptr[n_user+1],
 * tst_col[] in the order the test matrix stores them) walked in order.  inv_log2: DEVICE fp64[k] =
 * 1/log2(p+2); max_dcg: DEVICE fp64[k+1], max_dcg[t] = sum_{p<t} inv_log2[p], both evaluated by
 * the caller on the host the way the reference evaluates them.  The caller adds the per-user
 * values up (the reference adds them in test-user order). */
int dg_eval_metrics(const int32_t *top_idx, int32_t n_test, int32_t k, const int64_t *test_users,
                    const int32_t *tst_rowptr, const int32_t *tst_col, const double *inv_log2,
                    const double *max_dcg, double *recall, double *ndcg, void *stream);

/* ------------------------------------------------------------------------------------
 * Negative sampling.  Replaces PairwiseTrnData.negSampling (DiffGraph-Rec/DataHandler.py:282-289):
 * for pair p with user rows[p], negs[p] = an item id drawn uniformly from [0, n_item) and re-drawn
 * while the user has a training interaction with it (training CSR over users as in dg_eval_topk:
 * DEVICE int32 trn_rowptr[n_user+1], trn_col[] ascending inside a row).  rows, negs: DEVICE
 * int32[n_pairs].  The draws are a pure function of (seed, p): the reference draws from numpy's
 * global generator instead, so only the distribution is comparable.  negs[p] = -1 for a user who
 * has interacted with every item (the reference would not return).
 * ---------------------------------------------------------------------------------- */
int dg_neg_sampling(const int32_t *trn_rowptr, const int32_t *trn_col, const int32_t *rows,
                    int64_t n_pairs, int32_t n_user, int32_t n_item, uint64_t seed, int32_t *negs,
                    void *stream);

/* ------------------------------------------------------------------------------------
 * Meta-path composition.  Replaces Hc = Q1.dot(Q2); Hc = (Hc != 0) of
 * DiffGraph_NC/DataHandler.py:158-161 (scipy on the host): the PATTERN of the product of two CSR
 * patterns A [n_rows x n_mid] and B [n_mid x n_cols] (DEVICE int32 rowptr / column arrays; values do
 * not matter -- the reference's mixture weights are softmax outputs, all positive).
 * dg_spgemm_pattern_rows writes one bitmap of dg_spgemm_bitmap_words(n_cols) uint32 words per output
 * row (bit c of the row = column c present) and the row's entry count; the caller forms
 * c_rowptr = exclusive prefix sum of row_nnz, and dg_spgemm_pattern_fill expands the bitmaps into
 * ascending column ids.  n_cols <= 262144.
 * ---------------------------------------------------------------------------------- */
int32_t dg_spgemm_bitmap_words(int32_t n_cols);
int dg_spgemm_pattern_rows(const int32_t *a_rowptr, const int32_t *a_col, int32_t n_rows,
                           const int32_t *b_rowptr, const int32_t *b_col, int32_t n_mid, int32_t n_cols,
                           uint32_t *bitmap, int32_t *row_nnz, void *stream);
int dg_spgemm_pattern_fill(const uint32_t *bitmap, int32_t n_rows, int32_t n_cols,
                           const int32_t *c_rowptr, int32_t *c_col, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFGRAPH_HIP_H */
