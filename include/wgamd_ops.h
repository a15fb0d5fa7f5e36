/*
 * wgamd_ops.h — the operator entry points of the hot path (extern "C", return
 * wholememory_error_code_t, no exception crosses the boundary).
 *
 * Each declaration names the reference interface it replaces.  Common contract
 * (/root/reference/cpp/src/wholegraph_ops/unweighted_sample_without_replacement_func.cuh:329-334,463):
 *   - `stream` is the caller's hipStream_t passed as void*; work is enqueued on it and the op
 *     synchronises that stream before returning whenever an output SIZE must be known, so
 *     every output is complete on return.
 *   - fixed-size outputs are caller-allocated tensors; variable-size outputs are allocated by
 *     the op through p_env_fns->output_fns with the per-output `void* memory_context`;
 *     a NULL optional context means "do not produce that output".
 *   - shape/dtype violations return WHOLEMEMORY_INVALID_INPUT / WHOLEMEMORY_LOGIC_ERROR and
 *     print one line to stderr.
 */
#ifndef WGAMD_OPS_H_
#define WGAMD_OPS_H_

#include "wgamd_tensor.h"
#include "wgamd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- neighbour sampling ----------------------------------------------------------------- */

/* Replaces wholegraph_csr_unweighted_sample_without_replacement
 * (/root/reference/cpp/include/wholememory/wholegraph_op.h:31-42).
 * csr_row_ptr INT64[V+1]; csr_col INT|INT64[E]; center_nodes INT|INT64[n];
 * output_sample_offset INT[n+1] (caller-allocated) = exclusive scan of min(deg, M) (M<=0: deg).
 * dest (col dtype)[cnt]; center_localid INT[cnt] (optional); edge_gid INT64[cnt] (optional).
 * Seed i with deg<=M copies its row in CSR order; otherwise position t<M is chosen by the
 * reference's Fisher-Yates with r_t = PCG(random_seed, subsequence = i*B + lane).i31 % (deg-t)
 * (B, items-per-lane from the reference's launch table) — results are BIT-IDENTICAL to the
 * reference's host oracle (cpp/tests/wholegraph_ops/graph_sampling_test_utils.cu:312-401).
 * csr_row_ptr / csr_col may be tensors over device pointers (a CSR this GPU holds whole) or tensors backed by a
 * DISTRIBUTED / CHUNKED / CONTINUOUS handle (a CSR partitioned over the GPUs of a communicator): the op then fetches the
 * row offsets of the centres and the columns at the picked positions from their owners — the reference's NCCL path,
 * cpp/src/wholegraph_ops/unweighted_sample_without_replacement_nccl_func.cuh:213-372 — and is collective over that
 * communicator; the result is the same either way. */
wholememory_error_code_t wholegraph_csr_unweighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor,
  wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor,
  int max_sample_count,
  wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context,
  void* output_center_localid_memory_context,
  void* output_edge_gid_memory_context,
  unsigned long long random_seed,
  wholememory_env_func_t* p_env_fns,
  void* stream);

/* Replaces wholegraph_csr_weighted_sample_without_replacement (wholegraph_op.h:61-73):
 * A-Res biased sampling, key_e = log2(u_e)/w_e, keep the M largest keys.  csr_weight
 * FLOAT|DOUBLE[E].  Order inside a seed (unspecified by the reference, whose tests sort per
 * segment): CSR order of the selected edges; key ties go to the lowest neighbour index. */
wholememory_error_code_t wholegraph_csr_weighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor,
  wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t wm_csr_weight_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor,
  int max_sample_count,
  wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context,
  void* output_center_localid_memory_context,
  void* output_edge_gid_memory_context,
  unsigned long long random_seed,
  wholememory_env_func_t* p_env_fns,
  void* stream);

/* Host accessors of the op RNG stream, used by the reference's Python tests
 * (wholegraph_op.h:82-94; impl cpp/src/wholegraph_ops/raft_random_gen.cu:15-97).
 * `output` is a 1-D HOST tensor: INT|INT64 for the first, FLOAT for the second. */
wholememory_error_code_t generate_random_positive_int_cpu(int64_t random_seed,
                                                          int64_t subsequence,
                                                          wholememory_tensor_t output);
wholememory_error_code_t generate_exponential_distribution_negative_float_cpu(
  int64_t random_seed, int64_t subsequence, wholememory_tensor_t output);

/* ---- renumbering / sampled-subgraph CSR ------------------------------------------------- */

/* Replaces graph_append_unique (/root/reference/cpp/include/wholememory/graph_op.h:27-33).
 * unique = targets (verbatim, ids 0..T-1) ++ neighbours not among the targets, each once, in
 * FIRST-APPEARANCE order (the order of the reference's host oracle,
 * cpp/tests/graph_ops/append_unique_test_utils.cu:52-84; the reference device op leaves it
 * unspecified).  mapping INT[E]: index in `unique` of neighbour e; pass a tensor with
 * dim==0 / NULL data pointer (or NULL) to skip it. targets/neighbours: same dtype INT|INT64. */
wholememory_error_code_t graph_append_unique(
  wholememory_tensor_t target_nodes_tensor,
  wholememory_tensor_t neighbor_nodes_tensor,
  void* output_unique_node_memory_context,
  wholememory_tensor_t output_neighbor_raw_to_unique_mapping_tensor,
  wholememory_env_func_t* p_env_fns,
  void* stream);

/* Replaces csr_add_self_loop (graph_op.h:44-48): row i -> [i] ++ row i. INT only;
 * out_row_ptr has rows+1 entries, out_col nnz+rows. */
wholememory_error_code_t csr_add_self_loop(wholememory_tensor_t csr_row_ptr_tensor,
                                           wholememory_tensor_t csr_col_ptr_tensor,
                                           wholememory_tensor_t output_csr_row_ptr_tensor,
                                           wholememory_tensor_t output_csr_col_ptr_tensor,
                                           void* stream);

/* ---- feature fetch ---------------------------------------------------------------------- */

/* Replaces wholememory_gather (/root/reference/cpp/include/wholememory/wholememory_op.h:25-30):
 * output[i,:] = convert(table[indices[i],:]); a negative index leaves row i untouched.
 * table/output 1-D or 2-D (same rank), any dtype pair the reference registers
 * (floating<->floating, or identical integer types); indices INT|INT64.
 * gather_sms is accepted for signature compatibility and ignored (grid is sized for 256 CUs). */
wholememory_error_code_t wholememory_gather(wholememory_tensor_t wholememory_tensor,
                                            wholememory_tensor_t indices_tensor,
                                            wholememory_tensor_t output_tensor,
                                            wholememory_env_func_t* p_env_fns,
                                            void* stream,
                                            int gather_sms WGAMD_DEFAULT(-1));

/* Replaces wholememory_scatter (wholememory_op.h:42-47): table[indices[i],:] = convert(input[i,:]). */
wholememory_error_code_t wholememory_scatter(wholememory_tensor_t input_tensor,
                                             wholememory_tensor_t indices_tensor,
                                             wholememory_tensor_t wholememory_tensor,
                                             wholememory_env_func_t* p_env_fns,
                                             void* stream,
                                             int scatter_sms WGAMD_DEFAULT(-1));

/* Replaces wholememory_env_test_op (wholememory_op.h:61-68; cpp/src/wholememory_ops/wholememory_test_op.cu:53-140): the
 * self-test a binding runs on its allocator callbacks.  out[i, j] = (T)(float)i + input[j] for i < entry_count is computed
 * into scratch from temporary_fns, copied into output_fixed_tensor ([entry_count, dim], caller-allocated) and into one
 * output allocated through output_fns for every non-NULL context: DEVICE, PINNED and HOST allocation types. */
wholememory_error_code_t wholememory_env_test_op(wholememory_tensor_t input_tensor,
                                                 wholememory_tensor_t output_fixed_tensor,
                                                 void* output_variable_device_tensor_handle,
                                                 void* output_variable_pinned_tensor_handle,
                                                 void* output_variable_host_tensor_handle,
                                                 int64_t output_variable_entry_count,
                                                 wholememory_env_func_t* p_env_fns,
                                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WGAMD_OPS_H_ */
