/*
 * wgamd_ext.h — entry points that have NO counterpart in libwholegraph's C ABI.
 *
 * (1) Mini-batch aggregation.  The reference ships no SpMM/SDDMM kernel: its models call
 *     torch_geometric.nn.SAGEConv / GATConv (third party; call sites
 *     /root/reference/python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59,119-125,178-199).
 *     These entry points are what a PyTorch-ROCm autograd.Function binds to get the same maths
 *     (PyG formulas, fp32) from hand-written gfx950 kernels over the sampler's per-hop CSR
 *     (row_ptr = sample offsets, col = raw_to_unique mapping; graph_structure.py:186-195).
 * (2) A no-host-sync variant of the sampling + renumbering walk for the loader's steady state:
 *     same results as chaining the wgamd_ops.h ops, but every size stays on the device and
 *     outputs go to caller-provided capacity buffers, so a whole multi-hop mini-batch is a
 *     fixed sequence of launches (hipGraph-capturable) with no D2H round trip per op
 *     (the reference pays >= 5 stream syncs per hop).
 *
 * Plain pointers and sizes only; all pointers are device pointers unless stated; `stream` is a
 * hipStream_t passed as void*.  All calls are asynchronous on `stream`.
 */
#ifndef WGAMD_EXT_H_
#define WGAMD_EXT_H_

#include "wgamd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) aggregation -------------------------------------------------------------------- */

/* out[i, 0:F] = REDUCE_{e in [row_ptr[i], row_ptr[i+1])} x[src(e), 0:F]
 *   src(e) = col[e]                      when src_ids == NULL
 *          = src_ids[col[e]]             otherwise (fused feature fetch: x is then the global
 *                                        feature table and src_ids the batch's local->global map;
 *                                        src_ids_dtype = WHOLEMEMORY_DT_INT | WHOLEMEMORY_DT_INT64)
 *   REDUCE = sum (mean == 0) or sum / max(deg_i, 1) (mean != 0).  fp32, summed in CSR order.
 * row_ptr int32[n_rows+1], col int32[nnz]; ldx / ldo = row strides in elements. */
wholememory_error_code_t wgamd_spmm_csr_f32(const int* row_ptr,
                                            const int* col,
                                            int64_t n_rows,
                                            const float* x,
                                            int64_t ldx,
                                            int F,
                                            const void* src_ids,
                                            wholememory_dtype_t src_ids_dtype,
                                            int mean,
                                            float* out,
                                            int64_t ldo,
                                            void* stream);

/* SAGEConv input builder: the mean (or sum) aggregate AND the root term side by side, so that
 *   lin_l(mean_j x_j) + lin_r(x_i)  is ONE dense GEMM  [agg | x_self] @ [W_l | W_r]^T :
 *   out[i, 0:F]  = REDUCE_{e in row i} x[col[e], :]
 *   out[i, F:2F] = x[self_rows[i], :]
 * self_rows int64[n_rows] = row of destination i inside x (i itself for a single mini-batch, the
 * block-diagonal row for a call group); ldo >= 2F. */
wholememory_error_code_t wgamd_sage_aggregate_f32(const int* row_ptr,
                                                  const int* col,
                                                  int64_t n_rows,
                                                  const float* x,
                                                  int64_t ldx,
                                                  int F,
                                                  const int64_t* self_rows,
                                                  int mean,
                                                  float* out,
                                                  int64_t ldo,
                                                  void* stream);

/* The same with the FEATURE FETCH fused in: `table` is the global feature table and src_ids the mini-batch's
 * local -> global map (INT|INT64), so the batch feature matrix x = table[src_ids] is never materialised:
 *   out[i, 0:F]  = REDUCE_e table[src_ids[col[e]], :]      out[i, F:2F] = table[src_ids[self_rows[i]], :] */
wholememory_error_code_t wgamd_sage_aggregate_fetch_f32(const int* row_ptr,
                                                        const int* col,
                                                        int64_t n_rows,
                                                        const float* table,
                                                        int64_t ldt,
                                                        int F,
                                                        const void* src_ids,
                                                        wholememory_dtype_t src_ids_dtype,
                                                        const int64_t* self_rows,
                                                        int mean,
                                                        float* out,
                                                        int64_t ldo,
                                                        void* stream);

/* Sum aggregation over a CSR whose rows may be very long — the TRANSPOSED sampled hop of the backward pass, where a hub
 * source is a neighbour of thousands of rows.  Rows are summed in pieces of 64 entries by different lane groups and the
 * pieces of a row are then added up in order: same values from run to run (no atomics on the data), no launch that lasts as
 * long as its longest row.  out[r, :] = sum_{e in row r} x[col[e], :].  Workspace: wgamd_spmm_csr_segmented_workspace_bytes
 * (n_entries = row_ptr[n_rows], known to the caller). */
size_t wgamd_spmm_csr_segmented_workspace_bytes(int64_t n_entries, int F);
wholememory_error_code_t wgamd_spmm_csr_segmented_f32(const int* row_ptr, const int* col, int64_t n_rows, int64_t n_entries,
                                                      const float* x, int64_t ldx, int F, float* out, int64_t ldo,
                                                      void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the above w.r.t. x (src_ids == NULL form):
 *   grad_x[col[e], :] += grad_out[i, :] * (mean ? 1/max(deg_i,1) : 1)   for every edge e of row i.
 * grad_x must be zero-initialised (or hold the gradient to accumulate into) by the caller. */
wholememory_error_code_t wgamd_spmm_csr_bwd_f32(const int* row_ptr,
                                                const int* col,
                                                int64_t n_rows,
                                                const float* grad_out,
                                                int64_t ldg,
                                                int F,
                                                int mean,
                                                float* grad_x,
                                                int64_t ldx,
                                                void* stream);

/* Transpose of a sampled-hop CSR (rows = destinations, col = source rows in [0, n_src)) for the atomic-free backward
 * passes: the edges sorted by source, STABLE (edge order inside a source = edge order of the hop, so gradient sums are
 * reproducible).  One radix sort of (source, edge) pairs over ceil(log2 n_src) bits + two small kernels.
 *   row_ptr_t [n_src + 1]  : CSR offsets over the sources
 *   edge_perm [n_edges]    : edge ids in source-major order                      (may be NULL)
 *   edge_dst  [n_edges]    : destination row of every ORIGINAL edge id           (may be NULL)
 *   col_t     [n_edges]    : destination row of the k-th source-major edge = edge_dst[edge_perm[k]]   (may be NULL)
 * workspace: wgamd_csr_transpose_workspace_bytes(n_edges, n_src) bytes of device scratch. */
size_t wgamd_csr_transpose_workspace_bytes(int64_t n_edges, int64_t n_src);
wholememory_error_code_t wgamd_csr_transpose_i32(const int* row_ptr,
                                                 const int* col,
                                                 int64_t n_rows,
                                                 int64_t n_edges,
                                                 int64_t n_src,
                                                 int* row_ptr_t,
                                                 int* edge_perm,
                                                 int* edge_dst,
                                                 int* col_t,
                                                 void* workspace,
                                                 size_t workspace_bytes,
                                                 void* stream);

/* PyG COO edge_index (src, dst int64; dst ids in [0, n_dst)) -> destination-major int32 CSR for the aggregation kernels:
 * stable radix sort of (dst, edge) over ceil(log2 n_dst) bits, row_ptr [n_dst + 1] from the run boundaries,
 * col[k] = src of the k-th destination-major edge, edge_perm (may be NULL) = the edge ids in that order. */
size_t wgamd_coo_to_csr_workspace_bytes(int64_t n_edges, int64_t n_dst);
wholememory_error_code_t wgamd_coo_to_csr_i64(const int64_t* src,
                                              const int64_t* dst,
                                              int64_t n_edges,
                                              int64_t n_dst,
                                              int* row_ptr,
                                              int* col,
                                              int* edge_perm,
                                              void* workspace,
                                              size_t workspace_bytes,
                                              void* stream);

/* GATConv message passing (edge-softmax SDDMM + weighted SpMM), H heads x C channels:
 *   s_e      = leaky_relu(a_src[col[e], h] + a_dst[i, h], negative_slope)
 *   alpha_e  = softmax over the edges e of row i (per head)
 *   out[i,h,:] = sum_e alpha_e * x[col[e], h, :]
 * x [N_src, H*C] (row stride ldx), a_src [N_src, H], a_dst [n_rows, H], out [n_rows, H*C]
 * (row stride ldo), alpha_out nullable [nnz, H].  Rows without edges produce zeros. */
wholememory_error_code_t wgamd_gat_csr_f32(const int* row_ptr,
                                           const int* col,
                                           int64_t n_rows,
                                           const float* x,
                                           int64_t ldx,
                                           const float* a_src,
                                           const float* a_dst,
                                           int H,
                                           int C,
                                           float negative_slope,
                                           float* alpha_out,
                                           float* out,
                                           int64_t ldo,
                                           void* stream);

/* ---- (2) no-sync sampling walk ---------------------------------------------------------- */

/* One hop of uniform sampling + renumbering with device-resident sizes.
 *   in : targets (INT|INT64 = id_dtype)[<= target_cap], *n_targets_dev = how many are valid
 *   out: offsets int32[target_cap+1]   exclusive scan of min(deg, M); entries past n_targets
 *                                      repeat the total
 *        neighbor_lid int32[edge_cap]  raw_to_unique mapping of every sampled edge (CSR col)
 *        center_lid  int32[edge_cap]   row index of every sampled edge (nullable)
 *        edge_gid    int64[edge_cap]   CSR position of every sampled edge (nullable)
 *        unique      (id_dtype)[target_cap + edge_cap]  targets ++ new nodes (first appearance),
 *                                        capacity slack padded with -1 (a gather skips it)
 *        counts_dev  int32[2]          {n_edges, n_unique}
 * edge_cap must be >= target_cap * M (M > 0 required).  Results are identical to
 * wholegraph_csr_unweighted_sample_without_replacement followed by graph_append_unique.
 * workspace: wgamd_sample_hop_workspace_bytes(target_cap, edge_cap, id_dtype) bytes. */
size_t wgamd_sample_hop_workspace_bytes(int64_t target_cap, int64_t edge_cap, wholememory_dtype_t id_dtype);
size_t wgamd_sample_hop_weighted_workspace_bytes(int64_t target_cap, int64_t edge_cap, wholememory_dtype_t id_dtype,
                                                 int64_t max_row_len);

wholememory_error_code_t wgamd_sample_hop_nosync(const int64_t* csr_row_ptr,
                                                 const void* csr_col,
                                                 wholememory_dtype_t id_dtype,
                                                 const void* targets,
                                                 const int* n_targets_dev,
                                                 int64_t target_cap,
                                                 int max_sample_count,
                                                 unsigned long long random_seed,
                                                 int* offsets,
                                                 int* neighbor_lid,
                                                 int* center_lid,
                                                 int64_t* edge_gid,
                                                 int64_t edge_cap,
                                                 void* unique,
                                                 int* counts_dev,
                                                 void* workspace,
                                                 size_t workspace_bytes,
                                                 void* stream);

/* The same hop for a CALL GROUP of n_batches mini-batches handled by one launch sequence (the
 * idea of cugraph_pyg's local_seeds_per_call,
 * /root/reference/python/cugraph-pyg/cugraph_pyg/sampler/distributed_sampler.py:279-343): every
 * launch carries G x the work, every mini-batch gets exactly the result a single-batch call with
 * its own seed would produce (its PCG streams are numbered from the start of its own segment and
 * it is renumbered on its own).
 *   in : targets        concatenation of the batches' targets, batch b = [target_seg[b], target_seg[b+1])
 *        target_batch   int32[target_cap]  batch of every target
 *        target_seg     int32[n_batches+1] (device); target_seg[n_batches] = live target count
 *        random_seeds_dev u64[n_batches]   (device) one sampling seed per mini-batch
 *   out: offsets        int32[target_cap+1] global CSR row_ptr over all targets
 *        neighbor_row   int32[edge_cap]  GLOBAL row (into `unique`) of every sampled edge's neighbour;
 *                                        the per-batch local id is neighbor_row - unique_seg[b]
 *        center_row     int32[edge_cap]  global target row of every sampled edge (required)
 *        unique         ids[target_cap+edge_cap]  per-batch unique lists, concatenated: batch b =
 *                                        [unique_seg[b], unique_seg[b+1]) = its targets ++ its new nodes;
 *                                        slack padded with -1 (see WGAMD_HOP_NO_UNIQUE_PAD)
 *        unique_batch   int32[target_cap+edge_cap], unique_seg int32[n_batches+1]:
 *                                        feed them back as target_batch / target_seg of the next hop
 *        counts_dev     int32[2] {n_edges, n_unique}
 * Workspace as for the single-batch hop. */
wholememory_error_code_t wgamd_sample_hop_batched_nosync(const int64_t* csr_row_ptr,
                                                         const void* csr_col,
                                                         wholememory_dtype_t id_dtype,
                                                         const void* targets,
                                                         const int* target_batch,
                                                         const int* target_seg,
                                                         int n_batches,
                                                         int64_t target_cap,
                                                         int max_sample_count,
                                                         const unsigned long long* random_seeds_dev,
                                                         int* offsets,
                                                         int* neighbor_row,
                                                         int* center_row,
                                                         int64_t* edge_gid,
                                                         int64_t edge_cap,
                                                         void* unique,
                                                         int* unique_batch,
                                                         int* unique_seg,
                                                         int* counts_dev,
                                                         void* workspace,
                                                         size_t workspace_bytes,
                                                         int64_t n_vertices /* ids are < n_vertices; 0 = unknown.  A bound
                                                           lets the renumber table pack (batch, id, first position) into
                                                           one 64-bit word per slot (one atomic per key instead of two) */,
                                                         void* stream);

/* Flags of the call-group hops.  WGAMD_HOP_NO_UNIQUE_PAD: do not write the -1 padding into the capacity slack of
 * `unique` / `nodes_out` — for callers that read counts_dev / unique_seg (host or device) and never look past the live end.
 * The capacity is the worst case (every seed with fan-out^hops distinct neighbours), 3-4x the live size on real graphs, so
 * the padding is most of what the renumber step writes: walk 1.18 -> 1.09 ms per call group of 191 on the products-like graph.
 * Everything below the live end is identical with and without the flag. */
#define WGAMD_HOP_NO_UNIQUE_PAD 1u
/* WGAMD_HOP_COL_INT32: `csr_col` holds INT (32-bit) entries although id_dtype is WHOLEMEMORY_DT_INT64 — a graph with fewer
 * than 2^31 vertices kept compact behind an INT64 API (needs 0 < n_vertices < 2^31).  Targets, `unique` and the frontier
 * lists stay INT64; the sampled neighbours travel through the hop as 32-bit values (half the sector footprint of the
 * sampler's random column picks, half the bytes of every renumber pass) and are widened where `unique` is written.  Same
 * results as the INT64 column array, bit for bit.  In the _ex entry point `unique_batch` may be NULL (not produced). */
#define WGAMD_HOP_COL_INT32 2u

wholememory_error_code_t wgamd_sample_hop_batched_nosync_ex(const int64_t* csr_row_ptr,
                                                         const void* csr_col,
                                                         wholememory_dtype_t id_dtype,
                                                         const void* targets,
                                                         const int* target_batch,
                                                         const int* target_seg,
                                                         int n_batches,
                                                         int64_t target_cap,
                                                         int max_sample_count,
                                                         const unsigned long long* random_seeds_dev,
                                                         int* offsets,
                                                         int* neighbor_row,
                                                         int* center_row,
                                                         int64_t* edge_gid,
                                                         int64_t edge_cap,
                                                         void* unique,
                                                         int* unique_batch,
                                                         int* unique_seg,
                                                         int* counts_dev,
                                                         void* workspace,
                                                         size_t workspace_bytes,
                                                         int64_t n_vertices /* ids are < n_vertices; 0 = unknown.  A bound
                                                           lets the renumber table pack (batch, id, first position) into
                                                           one 64-bit word per slot (one atomic per key instead of two) */,
                                                         unsigned flags,
                                                         void* stream);

/* One hop of the PyG-style walk for a call group — what cugraph_pyg's sampling call produces
 * (pylibcugraph.*_neighbor_sample(renumber=True, prior_sources_behavior="exclude",
 * deduplicate_sources=True, retain_seeds=True); call site
 * /root/reference/python/cugraph-pyg/cugraph_pyg/sampler/distributed_sampler.py:877-908): hop k expands only
 * the vertices FIRST SEEN by hop k-1 (the frontier), and renumbers the sampled neighbours against ALL
 * vertices of the mini-batch so far.  Same kernels as the WholeGraph-style hop; per mini-batch the
 * result equals chaining wholegraph_csr_unweighted_sample_without_replacement(frontier) and
 * graph_append_unique(nodes, neighbours) with that batch's seed.
 *   nodes / node_batch / node_seg        per-batch vertex lists so far, concatenated (batch b =
 *                                        [node_seg[b], node_seg[b+1]); node_seg[G] = live count)
 *   frontier / _batch / _seg / _local0   the vertices to expand, by batch; local0[b] = local id (inside
 *                                        batch b) of its first frontier vertex
 *   offsets           int32[frontier_cap+1]  CSR row_ptr over the frontier
 *   neighbor_local / center_local  int32[edge_cap]  per-batch LOCAL ids of both ends of every sampled edge
 *                                        (PyG: row = neighbor_local, col = center_local)
 *   edge_gid          int64[edge_cap] CSR slot of every sampled edge (nullable)
 *   nodes_out / _batch / _seg            the grown per-batch lists (input lists ++ new vertices), slack = -1
 *   frontier_out / _batch / _seg / _local0   the next hop's frontier (= vertices first seen now)
 *   counts_dev        int32[2] {n_edges, n_nodes_out}
 *   neighbor_row_scratch / center_row_scratch int32[edge_cap] scratch
 * Capacities: edge_cap >= frontier_cap * M; nodes_out holds node_cap + edge_cap entries.
 * Workspace: wgamd_sample_hop_workspace_bytes(max(node_cap, frontier_cap), edge_cap, id_dtype). */
typedef struct wgamd_pyg_hop_t {
  const int64_t* csr_row_ptr;
  const void* csr_col;
  wholememory_dtype_t id_dtype;
  int n_batches;
  int max_sample_count;
  const unsigned long long* random_seeds_dev;
  const void* nodes;
  const int* node_batch;
  const int* node_seg;
  int64_t node_cap;
  const void* frontier;
  const int* frontier_batch;
  const int* frontier_seg;
  const int* frontier_local0;
  int64_t frontier_cap;
  int* offsets;
  int* neighbor_local;
  int* center_local;
  int64_t* edge_gid;
  int64_t edge_cap;
  void* nodes_out;
  int* nodes_out_batch;
  int* nodes_out_seg;
  void* frontier_out;
  int* frontier_out_batch;
  int* frontier_out_seg;
  int* frontier_out_local0;
  int* counts_dev;
  int* neighbor_row_scratch;
  int* center_row_scratch;
  void* workspace;
  size_t workspace_bytes;
  /* biased hop (NULL = uniform): weight of every CSR slot, FLOAT | DOUBLE; fan-out <= 256; max_row_len = the graph's
   * maximum degree (sizes the key slabs); workspace then = wgamd_sample_hop_weighted_workspace_bytes(...).  Rows with at
   * most max_sample_count candidates are copied whole, zero-weight edges included (the reference's kernel does the same:
   * weighted_sample_without_replacement_func.cuh:592-647).  Same results as
   * wholegraph_csr_weighted_sample_without_replacement + graph_append_unique per mini-batch. */
  const void* csr_weight;
  wholememory_dtype_t weight_dtype;
  int64_t max_row_len;
  int64_t n_vertices; /* ids are < n_vertices (0 = unknown): enables the packed renumber table, see above */
  unsigned flags;     /* WGAMD_HOP_* bits, 0 = defaults */
} wgamd_pyg_hop_t;

wholememory_error_code_t wgamd_sample_hop_pyg_nosync(const wgamd_pyg_hop_t* p, void* stream);

/* Rows of the hop's targets inside its block-diagonal `unique` list — the "x[:num_dst]" of a single mini-batch becomes
 * an index list for a call group:  rows[i] = i + unique_seg[b] - target_seg[b],  b = target_batch[i],  for i < n_targets.
 * One launch (the SAGEConv root term and the next layer's row list read it); everything is device memory. */
wholememory_error_code_t wgamd_call_group_target_rows(const int* unique_seg, const int* target_seg, const int* target_batch,
                                                      int64_t n_targets, int64_t* rows, void* stream);

/* Frontier of a node type between two hops of a heterogeneous call-group walk: batch b (of n_batches <= 4095) gained the
 * vertices [begin[b], seg[b+1] - seg[b]) of its batch-major list `nodes` (n_nodes entries, offsets seg [n_batches + 1]) since
 * the previous hop.  Writes them batch-major into ids[capacity] with their batch[capacity] and the offsets f_seg [n_batches + 1];
 * slots past f_seg[n_batches] are padding no consumer reads.  (The per-hop frontier the reference's distributed sampler keeps
 * per batch, python/cugraph-pyg/cugraph_pyg/sampler/distributed_sampler.py:808-824, for a whole call group.) */
wholememory_error_code_t wgamd_frontier_list(const int64_t* nodes, int64_t n_nodes, const int* seg, const int* begin, int n_batches,
                                             int64_t capacity, int64_t* ids, int* batch, int* f_seg, void* stream);

/* One (hop, edge type) of a heterogeneous call group (wgamd_sample_hop_pyg_nosync outputs) in the form the layers consume:
 * dst_full[j] / dst_compact[j] = row of frontier entry j in the destination type's batch-major node list — all vertices of
 * the walk (segments seg_dst [G+1]) / the vertices discovered by hops 0-1 only (compact_seg_dst [G+1], int64; NULL with
 * dst_compact NULL) — and col_full[e] / col_compact[e] = row of edge e's source in the source type's list, the same two ways
 * (row_local = the hop's `neighbor_local`).  Entry j of batch b has local id frontier_local0[b] + (j - frontier_seg[b]). */
wholememory_error_code_t wgamd_call_group_hop_rows(const int* offsets, const int* frontier_batch, const int* frontier_seg,
                                                   const int* frontier_local0, const int* row_local, int64_t n_frontier,
                                                   const int* seg_dst, const int64_t* compact_seg_dst, const int* seg_src,
                                                   const int64_t* compact_seg_src, int64_t* dst_full, int64_t* dst_compact,
                                                   int* col_full, int* col_compact, void* stream);
/* wgamd_call_group_hop_rows with the number of mini-batches given (frontier_seg has n_batches + 1 entries, the last =
 * n_frontier; the batch of an entry follows from it): one block per (batch, stretch of its frontier entries) reads the batch's
 * segment starts once and streams entries and edges with every lane — 3-4x faster at call-group sizes. */
wholememory_error_code_t wgamd_call_group_hop_rows_batched(const int* offsets, const int* frontier_seg, const int* frontier_local0,
                                                           const int* row_local, int64_t n_frontier, int n_batches,
                                                           const int* seg_dst, const int64_t* compact_seg_dst, const int* seg_src,
                                                           const int64_t* compact_seg_src, int64_t* dst_full, int64_t* dst_compact,
                                                           int* col_full, int* col_compact, void* stream);

/* ONE mini-batch of a PyG-style call group copied into fixed-size buffers (cugraph_pyg_amd.loader.PerBatchStep): the
 * reference's training loops step the optimizer once per mini-batch (pylibwholegraph/torch/gnn_model.py:119-125); with
 * every per-batch array at a fixed address and size the whole step can be captured in one hipGraph and replayed per
 * mini-batch.  For hop k (arrays of n_hops pointers): frontier entries [frontier_seg[k][batch], frontier_seg[k][batch + 1]) of
 * `offsets[k]` / `row_local[k]` (the hop's CSR over its frontier and the batch-local id of every sampled neighbour) become
 * row_ptr_out[k] (int32 [row_cap[k] + 1], from 0, row_ptr_out[k][row_cap[k]] = edge_cap[k]: the slack edges are dealt evenly
 * to the slack rows — at least one: a hop with row_cap[k] live rows overflows — with sources spread over the input rows, so
 * every entry of the fixed-size arrays is a well-formed edge of a row nobody reads), self_rows_out[k] (int64 [row_cap[k]]:
 * batch-local id of the entry = its row of x; 0 in the padding), col_out[k] (int32 [edge_cap[k]]: batch-local ids) and, when col_seg_out
 * is given, col_seg_out[k]: the same sources as rows of a trimmed layer's output, laid out as hop 0's row_cap[0] rows, then
 * hop 1's row_cap[1], ... .  n_id_out [node_cap] = the mini-batch's vertices (padding repeats the first).  sizes_out
 * (nullable, int32 [2 n_hops + 2]): live rows and edges per hop, live vertices, 1 if anything exceeded its capacity (the
 * copy is then truncated).  row_ptr_all_out (nullable, int32 [sum row_cap + 1]): the hops' CSRs back to back as ONE CSR — hop
 * k's rows start at sum(row_cap[:k]) and point at edges from sum(edge_cap[:k]); a caller that allocates the hops' col /
 * self_rows arrays back to back runs a layer over hops 0..j as one launch over a prefix of it; inv_deg_all_out (nullable, float
 * [sum row_cap]) = 1 / max(degree, 1) of every row of that CSR (what the backward of a mean aggregation scales by);
 * seed_mask_out (nullable, float [row_cap[0]]) = 1 for a live seed row, 0 for the padding (the row weights of the loss over the
 * last layer's row_cap[0] output rows: no slice, no mask arithmetic in the captured step).  One launch, no synchronisation. */
wholememory_error_code_t wgamd_call_group_stage_batch(int n_hops, const int* const* offsets, const int* const* row_local,
                                                      const int* const* frontier_seg, const int* const* frontier_local0,
                                                      const void* nodes, wholememory_dtype_t id_dtype, const int* node_seg,
                                                      int batch, const int* row_cap, const int* edge_cap, int node_cap,
                                                      int* const* row_ptr_out, int64_t* const* self_rows_out, int* const* col_out,
                                                      int* const* col_seg_out, void* n_id_out, int* sizes_out, int* row_ptr_all_out,
                                                      float* inv_deg_all_out, float* seed_mask_out, void* stream);

/* One hop of a PyG-style call group renumbered for the LAYER that consumes it (cugraph_pyg_amd.loader.CallGroup).  The
 * layer's input rows are `n_segments` segments per batch: local ids [local0[s][b], local0[s+1][b]) of batch b sit at rows
 * seg_base[s] + start[s][b] + (local id - local0[s][b]); seg_tab is int32 [2 n_segments, n_batches + 1] with row 2s =
 * local0[s], row 2s + 1 = start[s].  One segment whose start = the node-list offsets is the batch-major list of all
 * vertices (x = feat[n_id]); the output of a trimmed layer is one segment per hop it ran (that hop's frontier list).
 * Writes self_rows[j] (int64, nullable) = input row of frontier entry j itself (local id frontier_local0[b] + j -
 * frontier_seg[b]) and col[e] = input row of edge e's sampled neighbour (row_local[e] = the hop's `neighbor_local`).
 * frontier_seg has n_batches + 1 entries (the last = n_frontier); frontier_batch is not read (may be NULL). */
wholememory_error_code_t wgamd_call_group_layer_cols(const int* offsets, const int* frontier_batch, const int* frontier_seg,
                                                     const int* frontier_local0, const int* row_local, int64_t n_frontier,
                                                     int n_batches, int n_segments, const int* seg_tab, const int64_t* seg_base,
                                                     int64_t* self_rows, int* col, void* stream);

/* wgamd_gat_csr_f32 over a SUBSET of a larger destination list, optionally accumulating: row i of this launch reads
 * a_dst[dst_rows[i], :] and writes (accumulate: adds to) out[dst_rows[i], :].  This is one hop and edge type of a
 * heterogeneous call group: its rows are the frontier entries of that hop, dst_rows their places in the node list of the
 * destination type, and HeteroConv's sum over the relations ending in one node type is accumulate = 1 on stream-ordered
 * launches.  dst_rows = NULL, accumulate = 0: exactly wgamd_gat_csr_f32.  Row indirection / accumulation need H*C % 4 == 0,
 * H*C <= 256 and 16-byte aligned rows (WHOLEMEMORY_LOGIC_ERROR otherwise). */
wholememory_error_code_t wgamd_gat_csr_rows_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x, int64_t ldx,
                                                const float* a_src, const float* a_dst, int H, int C, float negative_slope,
                                                const int64_t* dst_rows, int accumulate, float* alpha_out, float* out,
                                                int64_t ldo, void* stream);

/* out[dst_rows ? dst_rows[i] : i, 0:C] = act(in[i, 0:C] + bias)  (bias nullable, relu 0/1; C % 4 == 0, 16-byte aligned rows):
 * the tail of a HeteroConv layer — bias, ReLU and the placement of a hop's rows in the destination type's list — in one pass. */
wholememory_error_code_t wgamd_bias_act_rows_f32(const float* in, int64_t ldi, int64_t n_rows, int C, const float* bias, int relu,
                                                 const int64_t* dst_rows, float* out, int64_t ldo, void* stream);

/* GAT aggregation BEFORE the dense transform (csrc/wg_aggregate.hip) — for sampled hops, where destinations are 10-20x
 * fewer than sources.  The attention-weighted sum is linear in the source rows, so
 *   agg[i, h, :] = sum_{e in row i} alpha_e^h x[col[e], :]          (x UNTRANSFORMED, F floats; out row i = [H][F])
 * followed by H small GEMMs  out[i, h, :] = agg[i, h, :] @ W[:, h*C:(h+1)*C]  over the destination rows only equals
 * GATConv's  sum_e alpha_e^h (x W)[col[e], h, :]  up to fp32 reassociation, without the lin GEMM over every source row.
 * alpha as in wgamd_gat_csr_f32 from a_src [N_src, H] (= x_src @ fold(W, att_src)) and a_dst[dst_rows ? dst_rows[i] : i, :].
 * Shapes: F % 4 == 0, F <= 256, H in {1, 2, 4, 8}, 16-byte aligned rows, ldo >= H * F. */
wholememory_error_code_t wgamd_gat_aggregate_heads_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                       int64_t ldx, int F, const float* a_src, const float* a_dst, int H,
                                                       float negative_slope, const int64_t* dst_rows, float* out, int64_t ldo,
                                                       void* stream);
/* The same aggregation reading the source rows THROUGH an id list (fetch in the layer): neighbour j's row is x[src_ids[j]] — x
 * the feature table, src_ids (int64) the call group's node list of the source type — while a_src stays indexed by j.  The
 * [n_src, F] copy of the rows (wholememory_gather's output) is never written.  src_ids NULL = wgamd_gat_aggregate_heads_f32.
 * terms_by_id: bit 0 — a_src holds the attention terms of the TABLE's rows and is read at row src_ids[j]; bit 1 — a_dst
 * likewise at row dst_ids[dst_rows ? dst_rows[i] : i] (dst_ids = the node list of the destination type).  The per-list terms
 * are then never made: a call group lists a table row once per mini-batch that sampled it. */
wholememory_error_code_t wgamd_gat_aggregate_heads_ids_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                           int64_t ldx, const int64_t* src_ids, const int64_t* dst_ids,
                                                           int terms_by_id, int F, const float* a_src,
                                                           const float* a_dst, int H, float negative_slope,
                                                           const int64_t* dst_rows, float* out, int64_t ldo, void* stream);
/* Backward of wgamd_gat_aggregate_heads[_ids]_f32 (csrc/wg_gat_bwd.hip): given grad_agg = dL/dagg [n_rows, H F],
 * ACCUMULATES (float atomics, caller-zeroed buffers) grad_a_src / grad_a_dst at the rows the forward read its terms from
 * (per listed row, or per TABLE row with terms_by_id) and, when grad_x is given, grad_x[row of x an edge read] += sum_h
 * alpha_e^h grad_agg[i, h, :].  de: scratch [edges, H] floats (holds dL/d(score) per edge and head on return).  Same
 * addressing arguments as the forward. */
wholememory_error_code_t wgamd_gat_aggregate_heads_bwd_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                           int64_t ldx, const int64_t* src_ids, const int64_t* dst_ids,
                                                           int terms_by_id, int F, const float* a_src, const float* a_dst, int H,
                                                           float negative_slope, const int64_t* dst_rows, const float* grad_agg,
                                                           int64_t ldg, float* de, float* grad_a_src, float* grad_a_dst,
                                                           float* grad_x, int64_t ldgx, float* stats, void* stream);
/* (stats, nullable: [2][n_rows][H] floats — the rows' softmax maxima and denominators, what the kernel below reads.)
 * grad_x of the same aggregation WITHOUT atomics, source-major over the transposed hop (row_ptr_t [n_src + 1], col_t = the
 * destination row of every entry: wgamd_csr_transpose_i32): every row of grad_x [n_src, F] is written exactly once.  Plain
 * addressing (a_src per source row, a_dst at dst_rows[i]): the case of a hidden-state input. */
wholememory_error_code_t wgamd_gat_aggregate_heads_bwd_gx_f32(const int* row_ptr_t, const int* col_t, int64_t n_src, int64_t n_rows,
                                                              int F, const float* a_src, const float* a_dst, int H,
                                                              float negative_slope, const int64_t* dst_rows, const float* stats,
                                                              const float* grad_agg, int64_t ldg, float* grad_x, int64_t ldgx,
                                                              void* stream);

/* The dense tail after wgamd_gat_aggregate_heads_f32 on the matrix pipe at fp32 accuracy (csrc/wg_gat_transform.hip):
 *   out[out_rows ? out_rows[i] : i, h C + c] = act( sum_k agg[i, h F + k] W[k, h C + c] (+ acc_in[i, h C + c]) (+ bias[h C + c]) )
 * — the H per-head GEMMs, HeteroConv's sum over the relations of a destination type (`acc_in`, may alias `out` when
 * out_rows is null) and the layer's bias / ReLU / row placement in one pass.  The product is the exact 3-way bf16 split of
 * wgamd_sage_layer_fused_bf16x3 (six bf16 MFMAs per fp32 product, fp32 accumulate).  `w_tiles` = W [F, H C] row-major
 * re-ordered by wgamd_gat_transform_weight_tiles into wgamd_gat_transform_weight_bytes(F, H, C) bytes.
 * Shapes (wgamd_gat_transform_heads_supported): C == 64, F in {64, 128, 256}; rows and the bias 16-byte aligned. */
int wgamd_gat_transform_heads_supported(int F, int H, int C);
size_t wgamd_gat_transform_weight_bytes(int F, int H, int C);
wholememory_error_code_t wgamd_gat_transform_weight_tiles(const float* w, int64_t ldw, int F, int H, int C, void* tiles,
                                                          void* stream);
wholememory_error_code_t wgamd_gat_transform_heads_bf16x3(const float* agg, int64_t ld_agg, int64_t n_rows, int F, int H, int C,
                                                          const void* w_tiles, const float* acc_in, int64_t ld_acc,
                                                          const float* bias, int relu, const int64_t* out_rows, float* out,
                                                          int64_t ldo, void* stream);

/* wgamd_gat_aggregate_heads_f32 + wgamd_gat_transform_heads_bf16x3 as ONE kernel (csrc/wg_gat_fused.hip): the [n_rows, H F]
 * aggregate stays in LDS.  Same arguments and semantics as the two calls (a_src [N_src, H], a_dst row dst_rows ? dst_rows[i] : i,
 * leaky-ReLU slope, then acc_in / bias / relu / out_rows / out of the transform).  Built for the deep hop of a sampled walk:
 * rows of up to 10 neighbours run from registers, longer rows are correct but slow.
 * Shapes (wgamd_gat_layer_fused_supported): F == 128, H == 4, C == 64; 16-byte aligned rows, terms and bias. */
int wgamd_gat_layer_fused_supported(int F, int H, int C);
wholememory_error_code_t wgamd_gat_layer_fused_bf16x3(const int* row_ptr, const int* col, int64_t n_rows, const float* x, int64_t ldx,
                                                      int F, const float* a_src, const float* a_dst, int H, int C,
                                                      float negative_slope, const int64_t* dst_rows, const void* w_tiles,
                                                      const float* acc_in, int64_t ld_acc, const float* bias, int relu,
                                                      const int64_t* out_rows, float* out, int64_t ldo, void* stream);
/* wgamd_gat_layer_fused_bf16x3 reading the source rows through an id list (see wgamd_gat_aggregate_heads_ids_f32). */
wholememory_error_code_t wgamd_gat_layer_fused_ids_bf16x3(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                          int64_t ldx, const int64_t* src_ids, const int64_t* dst_ids,
                                                          int terms_by_id, int F, const float* a_src,
                                                          const float* a_dst, int H, int C, float negative_slope,
                                                          const int64_t* dst_rows, const void* w_tiles, const float* acc_in,
                                                          int64_t ld_acc, const float* bias, int relu, const int64_t* out_rows,
                                                          float* out, int64_t ldo, void* stream);

/* Backward of wgamd_gat_csr_f32 (csrc/wg_gat_bwd.hip): given grad_out [n_rows, H*C] and the forward's alpha [E, H], writes
 * grad_x [n_src, H*C], grad_a_src [n_src, H], grad_a_dst [n_rows, H]; de [E, H] is scratch.  Needs the hop CSR transposed:
 * row_ptr_t [n_src+1], edge_perm [E] (edge ids sorted by source, stable), edge_dst [E] (destination row of every edge).
 * Shapes: C % 4 == 0, C/4 a power of two, H*C <= 256 — anything else returns WHOLEMEMORY_LOGIC_ERROR.
 * workspace (wgamd_gat_csr_bwd_workspace_bytes(n_entries, H, C) bytes; NULL = none): with it the source-major pass sums long
 * source rows in pieces of 64 entries and adds the pieces up in order (a power-law hop has hub sources with thousands of
 * entries).  Two entry points, one algorithm:
 *   wgamd_gat_csr_bwd_f32     — the signature of rounds 1-2 (no n_entries): the piece capacity is what the workspace holds;
 *   wgamd_gat_csr_bwd_f32_v2  — n_entries = E, the entry count of the transposed hop (row_ptr_t[n_src]): a non-NULL workspace
 *                               smaller than wgamd_gat_csr_bwd_workspace_bytes(n_entries, H, C) returns
 *                               WHOLEMEMORY_INVALID_INPUT (as wgamd_spmm_csr_segmented_f32 does).
 * Either way the slots are bounded on the device: a long row whose pieces do not fit gets none, nothing is written past the
 * workspace, and the int at byte 8 of the (256-byte aligned) workspace is set to 1 — the gradients of that row are then
 * incomplete; with the workspace size the query returns for the true E it cannot happen. */
size_t wgamd_gat_csr_bwd_workspace_bytes(int64_t n_entries, int H, int C);
wholememory_error_code_t wgamd_gat_csr_bwd_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x, int64_t ldx,
                                               const float* a_src, const float* a_dst, int H, int C, float negative_slope,
                                               const float* alpha, const float* grad_out, int64_t ldg, const int* row_ptr_t,
                                               const int* edge_perm, const int* edge_dst, int64_t n_src, float* de,
                                               float* grad_x, int64_t ldgx, float* grad_a_src, float* grad_a_dst,
                                               void* workspace, size_t workspace_bytes, void* stream);
wholememory_error_code_t wgamd_gat_csr_bwd_f32_v2(const int* row_ptr, const int* col, int64_t n_rows, const float* x, int64_t ldx,
                                                  const float* a_src, const float* a_dst, int H, int C, float negative_slope,
                                                  const float* alpha, const float* grad_out, int64_t ldg, const int* row_ptr_t,
                                                  const int* edge_perm, const int* edge_dst, int64_t n_src, float* de,
                                                  float* grad_x, int64_t ldgx, float* grad_a_src, float* grad_a_dst,
                                                  int64_t n_entries, void* workspace, size_t workspace_bytes, void* stream);

/* A whole SAGEConv layer over a sampled hop in ONE kernel (csrc/wg_sage_fused.hip):
 *   out[i,:] = act( [ mean|sum_{e in row i} X[col[e]] | X[self_rows[i]] ] @ w_t + bias ),  X[r] = x[src_ids ? src_ids[r] : r]
 * feature fetch -> aggregation (fp32, CSR order) -> fp32-MFMA transform; the [n_rows, 2F] operand lives only in LDS.
 * w_t: [2F, N] row-major (rows 0..F-1 = W_l^T, rows F..2F-1 = W_r^T), ldw >= N; bias nullable; relu != 0 applies max(.,0).
 * x_rows = number of rows of x (0 = unknown): below 4 GB the kernel uses 32-bit row offsets.
 * Shapes: F % 4 == 0, F <= 256, N in {64, 128, 256}, x rows 16-B aligned — anything else returns WHOLEMEMORY_LOGIC_ERROR
 * (callers then use wgamd_sage_aggregate_f32 + a library GEMM).  Semantics: torch_geometric.nn.SAGEConv as the reference
 * uses it (python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59). */
wholememory_error_code_t wgamd_sage_layer_fused_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                    int64_t ldx, int64_t x_rows, int F, const void* src_ids,
                                                    wholememory_dtype_t src_ids_dtype, const int64_t* self_rows, int mean,
                                                    const float* w_t, int64_t ldw, int N, const float* bias, int relu,
                                                    float* out, int64_t ldo, void* stream);

/* The same layer with the dense product evaluated on the bf16 matrix pipe at fp32 accuracy: both operands are split
 * exactly into three bf16 pieces (a = a_hi + a_mid + a_lo; 24 = 3 x 8 significand bits) and the six products of weight
 * >= 2^-16 are accumulated in fp32 (v_mfma_f32_32x32x16_bf16); the dropped terms are <= 2^-23 |a b|, the class of fp32
 * round-off.  Six bf16 MFMAs cost 6/16 of one fp32 MFMA, which takes the layer from the fp32-MFMA roof to the HBM roof.
 * `w_planes` is the weight [2F, N] re-ordered by wgamd_sage_split_weight_bf16x3 into wgamd_sage_weight_planes_bytes(2F, N)
 * bytes (do it once per weight update): since round 3 fp32 tiles [k-step][column][16 consecutive k] — the weight travels as
 * 4 bytes per element and the multiplying waves split it in registers (the pre-split planes were 6; the weight stream is what
 * the layer's time is most sensitive to).  The buffer is opaque to callers; the entry-point names are kept.  Shapes: F % 4 == 0 and small enough for two 32-row tiles of 3 bf16 planes in
 * 160 KB of LDS (F <= 208), N in {64, 128, 256}: wgamd_sage_layer_bf16x3_supported.  Inf/NaN features give NaN rows. */
size_t wgamd_sage_weight_planes_bytes(int K, int N);
int wgamd_sage_layer_bf16x3_supported(int F, int N);
/* The same buffer (and the padded bias) straight from a layer's parameters in torch.nn.Linear's layout — w_l, w_r [N, F]
 * row-major, bias [N] (nullable): the operand [W_l | W_r]^T with its columns zero-padded from N to Np, one launch.  `planes`
 * holds wgamd_sage_weight_planes_bytes(2F, Np) bytes, `bias_out` (nullable) Np floats; `full_tiles`: see WGAMD_SAGE_FULL_TILES. */
wholememory_error_code_t wgamd_sage_layer_weight_planes(const float* w_l, int64_t ldl, const float* w_r, int64_t ldr,
                                                        const float* bias, int F, int N, int Np, void* planes,
                                                        float* bias_out, int full_tiles, void* stream);
/* Tile shape of the layer kernel.  F > 148 (wgamd_sage_layer_uses_half_tiles) runs 64-row tiles whose mean half alone goes
 * through LDS, with pre-split weight planes; a launch of a few thousand rows (ONE mini-batch: 20 tiles on 256 CUs, each a serial
 * chain of row fetches) is faster on whole 32-row tiles.  Such a launch passes `relu | WGAMD_SAGE_FULL_TILES` and planes made with
 * full_tiles = 1 (fp32 tiles); both default to the throughput shape. */
#define WGAMD_SAGE_FULL_TILES 2
int wgamd_sage_layer_uses_half_tiles(int F);
wholememory_error_code_t wgamd_sage_split_weight_bf16x3(const float* w_t, int64_t ldw, int K, int N, void* planes,
                                                        void* stream);
wholememory_error_code_t wgamd_sage_layer_fused_bf16x3(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                       int64_t ldx, int64_t x_rows, int F, const void* src_ids,
                                                       wholememory_dtype_t src_ids_dtype, const int64_t* self_rows, int mean,
                                                       const void* w_planes, int N, const float* bias, int relu, float* out,
                                                       int64_t ldo, void* stream);

/* ---- training: the one-kernel layer's forward with the aggregate kept, and its weight gradient ------------------------------
 * The reference's models train (python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59,119-125; every example ends in
 * loss.backward()): these two entry points are what the autograd.Function of wholegraph_amd.nn binds.
 *
 * wgamd_sage_layer_fused_bf16x3_train: wgamd_sage_layer_fused_bf16x3 (same launch, same bits in `out`) that also writes the
 * aggregate half of its operand, agg_out[i, 0:F] = mean|sum_{e in row i} X[col[e]] (fp32, ld_agg >= F floats, rows 16-B
 * aligned; NULL = not kept): n_rows F 4 bytes written once instead of E F 4 bytes of neighbour rows fetched again in the
 * backward pass. */
wholememory_error_code_t wgamd_sage_layer_fused_bf16x3_train(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                             int64_t ldx, int64_t x_rows, int F, const void* src_ids,
                                                             wholememory_dtype_t src_ids_dtype, const int64_t* self_rows,
                                                             int mean, const void* w_planes, int N, const float* bias, int relu,
                                                             float* out, int64_t ldo, float* agg_out, int64_t ld_agg,
                                                             void* stream);

/* Weight gradient of the layer  out = act([agg | X[self_rows]] @ [W_l | W_r]^T + b)  over one hop (csrc/wg_sage_bwd.hip):
 *   dZ[i, :]      = grad_out[i, :]  where  act_out == NULL or act_out[i, :] > 0,  else 0        (ReLU mask folded in)
 *   grad_w_l[n,f] (+)= sum_i dZ[i, n] agg[i, f]          grad_w_r[n,f] (+)= sum_i dZ[i, n] X[self_rows[i], f]
 *   grad_bias[n]  (+)= sum_i dZ[i, n]                                                           (nullable)
 * X[r] = x[src_ids ? src_ids[r] : r] as in the forward; grad_w_l / grad_w_r are [N, F] row-major contiguous (the layout of
 * torch.nn.Linear.weight); accumulate != 0 adds to what they hold (a layer that ran over several hops).  A split-K product on
 * the bf16 matrix pipe at fp32 accuracy (the exact 3-way split of the forward, six products per term, fp32 accumulate): every
 * workgroup owns a contiguous range of rows, keeps its [2F, N] partial sum in registers and writes it once; the partial sums
 * are added in workgroup order by a second launch — no atomics, the same bits from run to run.
 * Shapes: F % 4 == 0, F <= 256, N <= 256; x / agg rows 16-B aligned.  workspace: wgamd_sage_wgrad_workspace_bytes(n_rows, F, N)
 * bytes of device scratch (0 = shape not supported). */
size_t wgamd_sage_wgrad_workspace_bytes(int64_t n_rows, int F, int N);
wholememory_error_code_t wgamd_sage_wgrad_bf16x3(const float* agg, int64_t ld_agg, const float* x, int64_t ldx, int F,
                                                 const void* src_ids, wholememory_dtype_t src_ids_dtype,
                                                 const int64_t* self_rows, int64_t n_rows, const float* grad_out, int64_t ldg,
                                                 const float* act_out, int64_t ld_act, int N, float* grad_w_l,
                                                 float* grad_w_r, float* grad_bias, int accumulate, void* workspace,
                                                 size_t workspace_bytes, void* stream);

/* ---- the feature fetch of the one-kernel layer over a PEER-MAPPED table ------------------------------------------------------
 * src_ids_dtype = WGAMD_IDS_BYTE_OFFSETS in wgamd_sage_layer_fused_bf16x3(_train) / wgamd_sage_wgrad_bf16x3: `src_ids` is then
 * an int64 list of BYTE offsets from `x` (row r of the layer's input starts at (char*)x + src_ids[r]) instead of row numbers —
 * what wgamd_mapped_row_offsets (include/wgamd_comm.h) makes of a call group's node ids for a CHUNKED / CONTINUOUS table whose
 * partitions live on several GPUs: the layer kernel then reads remote rows over xGMI itself, x = table[n_id] never exists. */
#define WGAMD_IDS_BYTE_OFFSETS ((wholememory_dtype_t)64)

/* Biased (A-Res) sampling, fan-out <= 32: how wholegraph_csr_weighted_sample_without_replacement and the biased call-group
 * hop find the M largest keys.  pruning = 1 (default; env WGAMD_WEIGHTED_PRUNING=0 turns it off): a cheap lower bound of
 * every |key| first, exact keys (log1pf, two divisions) only for the candidates that can still reach the M-th largest one —
 * same picks as computing every key (csrc/wg_sample.hip, "threshold pruning").  force_redo = 1 (env
 * WGAMD_WEIGHTED_FORCE_REDO=1) sends every row through the hand-back path the pruned kernels take for a row they cannot
 * decide: a test switch.  A negative argument leaves that setting as it is. */
void wgamd_set_weighted_sampling_mode(int pruning, int force_redo);
/* Uniform hops of a call group whose frontier capacity is at least `min_capacity` entries (and whose fan-out is at most 32)
 * walk the frontier grouped by vertex-id range instead of in list order: identical results (every entry keeps its own PCG
 * streams and output positions), a third of the cache-line fetches where the frontier repeats its hubs batch after batch.
 * OFF by default (the sampling kernel is bound by VALU issue on an MI355X, not by the lines it fetches: same duration, plus
 * the two launches that build the order); environment: WGAMD_SAMPLE_LOCALITY=<n>; min_capacity <= 0 switches it off. */
void wgamd_set_sample_locality_min(int64_t min_capacity);

/* Uniform neighbour sampling WITH replacement (cugraph_pyg `replace=True`; the reference forwards it to libcugraph,
 * sampler/distributed_sampler.py:775-792,864 — not in its tree, so the draw layout is this library's and is pinned by
 * the CPU restatement of the parity tests).  Same tensors, contexts and error behaviour as wholegraph_csr_unweighted_sample_without_replacement
 * (include/wgamd_ops.h); a seed with N > 0 neighbours yields exactly `sample_count` picks,
 * pick t = col[start + G(random_seed, i * sample_count + t).i31() % N] in draw order, a seed without neighbours none. */
wholememory_error_code_t wgamd_csr_uniform_sample_with_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor, int sample_count, wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context, void* output_center_localid_memory_context, void* output_edge_gid_memory_context,
  unsigned long long random_seed, wholememory_env_func_t* p_env_fns, void* stream);

/* Feature gather with a narrow product folded in, one pass over the gathered rows:
 *   out_x[i, :] = table[ids[i], :]          out_terms[i, :] = table[ids[i], :] @ v       (v [F, T] row-major, T <= 32)
 * fp32 in, exact-fp32 MFMA products, fp32 accumulation.  A negative id skips the row (out_x[i] untouched) and yields zero terms.
 * For the attention logits of GATConv (x_j . fold(W, att) per relation end — the reference's models build
 * torch_geometric.nn.GATConv, /root/reference/python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:45-59): the second pass
 * over every gathered row that a separate [n, F] x [F, T] GEMM costs disappears.
 * term_group = 0: out_terms is [n, T] rows (ldo floats apart).  term_group = 4 (T % 4 == 0): slabs [T / 4][n][4] — the four
 * terms of relation end k for all rows are contiguous (what a GAT kernel with H = 4 heads reads), ldo unused.
 * ids = NULL: the rows 0 .. n-1 of `table` as they lie; out_x = NULL: only the terms are produced (both: the narrow product
 * of a resident [n, F] matrix in one streaming pass — the layer-2 attention logits of a hidden state).
 * Shapes: F in {32, 64, 128, 256} (wgamd_gather_terms_supported), rows 16-byte aligned; others: WHOLEMEMORY_LOGIC_ERROR. */
int wgamd_gather_terms_supported(int F, int T);
wholememory_error_code_t wgamd_gather_terms_f32(const float* table, int64_t ldt, const void* ids, wholememory_dtype_t id_dtype,
                                                int64_t n, int F, const float* v, int T, float* out_x, int64_t ldx,
                                                float* out_terms, int64_t ldo, int term_group, void* stream);
/* slabs_out[k][i][0..3] = slabs_in[k][ids[i]][0..3], k < n_slabs: the attention terms of a call group's rows taken from the
 * terms of the TABLE's rows ([n_slabs][n_in][4], what wgamd_gather_terms_f32 writes with ids = NULL and term_group = 4).  A
 * call group lists a table row once per mini-batch that sampled it; when the table is shorter than the list, x @ v over the
 * table + this gather of 16-byte rows replaces x[ids] @ v over the list.  A negative or out-of-range id gives zero terms. */
wholememory_error_code_t wgamd_gather_term_slabs_f32(const float* slabs_in, int64_t n_in, int n_slabs, const void* ids,
                                                     wholememory_dtype_t id_dtype, int64_t n, float* slabs_out, void* stream);
/* dv [F, T] += x^T [F, n] . dterms [n, T] (row-major, contiguous): the weight gradient of terms = x @ v for a narrow v
 * (T <= 32) over a LONG x (a whole feature table: the reduction runs over its rows); float atomics into the caller's
 * (zeroed) dv.  F <= 256. */
wholememory_error_code_t wgamd_rows_terms_bwd_f32(const float* x, int64_t ldx, int64_t n, int F, const float* dterms, int T,
                                                  float* dv, void* stream);

/* De-duplication of an id list with a known bound (ids < id_bound, e.g. the vertex count of the table they index):
 *   distinct[0 .. *n_distinct_dev)  the distinct non-negative ids, ASCENDING (so already grouped by owner rank of a
 *                                   range-partitioned table); capacity min(n, id_bound) entries
 *   inverse[i]                      position of ids[i] in `distinct`; -1 for a negative id (a row to skip) and for an id
 *                                   >= id_bound (then *out_of_bound_dev = 1, nullable)
 * Mark -> scan over the bound -> compact -> look up: no sort, no hash table, no host synchronisation.
 * Used by the partitioned FeatureStore to send each distinct row of a call group over xGMI ONCE
 * (wholegraph_amd/tensor.py, DistributedWholeMemoryTensor.gather(dedup=...)): the reference's
 * wholememory_gather_nccl exchanges every requested id (/root/reference/cpp/src/wholememory_ops/gather_op_impl_nccl.cu:23-171).
 * Workspace: wgamd_unique_bounded_workspace_bytes(id_bound) (0 = bound not supported: must be in (0, 2^31 - 4096)), 256-byte aligned. */
size_t wgamd_unique_bounded_workspace_bytes(int64_t id_bound);
wholememory_error_code_t wgamd_unique_bounded(const void* ids, wholememory_dtype_t id_dtype, int64_t n, int64_t id_bound,
                                              int64_t* distinct, int* inverse, int* n_distinct_dev, int* out_of_bound_dev,
                                              void* workspace, size_t workspace_bytes, void* stream);
/* The same over a CAPACITY-sized list whose live length is on the device (`n_live_dev`, nullable = n): the node list of a
 * no-sync call-group walk.  Lets the de-duplication run on the walk's stream right behind the walk, its count travelling
 * with the walk's own size read-back — the call group's distinct feature rows are then fetched ONCE and the first layer
 * reads them through `inverse` (its src_ids), at N = 1 as at N > 1 (bench.py, round 6). */
wholememory_error_code_t wgamd_unique_bounded_live(const void* ids, wholememory_dtype_t id_dtype, int64_t n, const int* n_live_dev,
                                                   int64_t id_bound, int64_t* distinct, int* inverse, int* n_distinct_dev,
                                                   int* out_of_bound_dev, void* workspace, size_t workspace_bytes, void* stream);

/* Softmax cross-entropy of the seeds' logits — the loss of every training loop of the reference
 * (/root/reference/python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:119-125: F.cross_entropy inside the batch loop).
 *   loss = sum_i w_i (lse_i - x[i, t_i]) / sum_i w_i  over rows with 0 <= t_i < n_classes (a negative target = torch's
 *   ignore_index), w_i = row_weight[i] or 1;   d x[i, c] = g w_i (exp(x[i, c] - lse_i) - [c == t_i]) / sum_i w_i.
 * forward: ONE launch; writes lse [n_rows], *loss_out, and into `state` (wgamd_softmax_xent_state_bytes(n_rows) bytes, zeroed
 * by the call unless state_is_zeroed; reusable as it is left) the sum of weights the backward divides by.  Per-workgroup
 * partial sums are added in workgroup order: run-to-run deterministic.  backward: ONE launch; grad_loss (nullable = 1) is a
 * DEVICE scalar, so the pair sits inside a captured per-mini-batch step (cugraph_pyg_amd.loader.PerBatchStep). */
size_t wgamd_softmax_xent_state_bytes(int64_t n_rows);
wholememory_error_code_t wgamd_softmax_xent_forward_f32(const float* logits, int64_t ld, int64_t n_rows, int n_classes,
                                                        const int64_t* target, const float* row_weight, float* lse, void* state,
                                                        int state_is_zeroed, float* loss_out, void* stream);
wholememory_error_code_t wgamd_softmax_xent_backward_f32(const float* logits, int64_t ld, int64_t n_rows, int n_classes,
                                                         const int64_t* target, const float* row_weight, const float* lse,
                                                         const void* state, const float* grad_loss, float* grad_logits, int64_t ldg,
                                                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WGAMD_EXT_H_ */
