/*
 * wgamd_embedding.h — trainable embedding tables with sparse optimizers on top of DISTRIBUTED tensors.
 * Replaces /root/reference/cpp/include/wholememory/embedding.h:17-237 (same names, argument meaning, defaults and
 * error behaviour) for what exists on an MI355X node:
 *   * storage is WHOLEMEMORY_ML_DEVICE (one partition per GPU in HBM, wgamd_comm.h) or WHOLEMEMORY_ML_HOST (one partition
 *     per rank in pinned host memory, which the GPU reads and writes in place over PCIe);
 *   * a WHOLEMEMORY_AT_READWRITE policy on the table's own communicator is the reference's "device cached host
 *     embedding" (embedding.cpp:556-759): every rank keeps a set-associative WRITE-BACK cache of ITS OWN rows in private
 *     HBM — a line holds the padded embedding row and, once an optimizer is set, the row's optimizer state behind the
 *     same tag.  Ids are routed to their owner; with adjust_cache the owner first brings the rows in (writing displaced
 *     modified lines back); reads and optimizer updates go to the line when the row is resident and to the table row
 *     when it is not; writeback_cache flushes the modified lines (and keeps them), drop_all_cache flushes and empties.
 *     The table seen through wholememory_embedding_get_embedding_tensor is stale between an update and the next
 *     write-back, as in the reference;
 *   * a WHOLEMEMORY_AT_READONLY policy builds the reference's "local cached global readonly embedding"
 *     (embedding.cpp:776-894): a set-associative cache of table rows in every rank's own HBM, so that repeated reads of
 *     hot rows owned by peers stop crossing xGMI.  A gather returns the same bytes with or without the cache;
 *     adjust_cache = true lets the gather insert the rows it missed; writeback is a no-op (nothing is dirty),
 *     drop_all_cache empties it (call it after writing the table through its tensor);
 *   * round_robin_size must be 0.
 * Layout: the table rows are padded to 16 bytes (embedding.cpp:45-58, align_embedding_dim); the per-element optimizer
 * states live in ONE fp32 table [entries, n_states * padded_dim] with the same row partition, the per-row LazyAdam
 * powers in a [entries, 2] fp32 table.  Every state is reachable by name as a sub-tensor view.
 */
#ifndef WGAMD_EMBEDDING_H_
#define WGAMD_EMBEDDING_H_

#include "wgamd_comm.h"
#include "wgamd_tensor.h"
#include "wgamd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wholememory_embedding_cache_policy_* wholememory_embedding_cache_policy_t;
typedef struct wholememory_embedding_optimizer_* wholememory_embedding_optimizer_t;
typedef struct wholememory_embedding_* wholememory_embedding_t;

/* embedding.h:24-28 */
enum wholememory_access_type_t {
  WHOLEMEMORY_AT_NONE = 0,
  WHOLEMEMORY_AT_READONLY,
  WHOLEMEMORY_AT_READWRITE,
};

/* embedding.h:33-39 */
enum wholememory_optimizer_type_t {
  WHOLEMEMORY_OPT_NONE = 0,
  WHOLEMEMORY_OPT_SGD,
  WHOLEMEMORY_OPT_LAZY_ADAM,
  WHOLEMEMORY_OPT_RMSPROP,
  WHOLEMEMORY_OPT_ADAGRAD,
};

/* embedding.h:63-80.  Parameters (float*, embedding_optimizer.cpp:100-178,286-398): every optimizer "weight_decay";
 * LAZY_ADAM "epsilon" "beta1" "beta2" "adam_w" (> 0.5 = decoupled decay); ADAGRAD "epsilon"; RMSPROP "epsilon" "alpha".
 * An unknown name returns WHOLEMEMORY_INVALID_INPUT; WHOLEMEMORY_OPT_NONE returns WHOLEMEMORY_NOT_IMPLEMENTED. */
wholememory_error_code_t wholememory_create_embedding_optimizer(wholememory_embedding_optimizer_t* optimizer,
                                                                enum wholememory_optimizer_type_t optimizer_type);
wholememory_error_code_t wholememory_optimizer_set_parameter(wholememory_embedding_optimizer_t optimizer,
                                                             const char* parameter_name, void* value);
void wholememory_destroy_embedding_optimizer(wholememory_embedding_optimizer_t optimizer);

/* embedding.h:96-110 — cache_ratio outside [1/512, 1] -> WHOLEMEMORY_INVALID_VALUE (embedding.cpp:917-920); every other
 * combination is recorded and judged by wholememory_create_embedding.  destroy accepts NULL. */
wholememory_error_code_t wholememory_create_embedding_cache_policy(wholememory_embedding_cache_policy_t* cache_policy,
                                                                   wholememory_comm_t cache_level_comm,
                                                                   wholememory_memory_type_t memory_type,
                                                                   wholememory_memory_location_t memory_location,
                                                                   enum wholememory_access_type_t access_type,
                                                                   float cache_ratio);
wholememory_error_code_t wholememory_destroy_embedding_cache_policy(wholememory_embedding_cache_policy_t cache_policy);

/* embedding.h:127-144.  embedding_tensor_description: 2-D, dtype FLOAT / HALF / BF16 for trainable tables (any dtype
 * for read-only ones).  round_robin_size 0; embedding_entry_partition NULL = equal split (ignored with a cache policy,
 * embedding.cpp:1009).  cache_policy: NULL; or a WHOLEMEMORY_AT_READONLY policy — cache_ratio * entries lines (rounded up
 * to sets of 32) of private HBM per rank, whatever communicator the policy names; a cache communicator other than `comm`
 * with cache memory type DISTRIBUTED -> WHOLEMEMORY_INVALID_INPUT (embedding.cpp:986-992); or a WHOLEMEMORY_AT_READWRITE
 * policy — cache_ratio * (rows of the rank) lines of write-back cache per rank; its communicator must be `comm`
 * (embedding.cpp:1000-1004), its location WHOLEMEMORY_ML_DEVICE (:962-967) and its memory type not below the table's
 * (:968-972), else WHOLEMEMORY_INVALID_INPUT.  set_optimizer on a READONLY-cached embedding -> WHOLEMEMORY_INVALID_INPUT
 * (embedding.cpp:55-60). */
wholememory_error_code_t wholememory_create_embedding(wholememory_embedding_t* wholememory_embedding,
                                                      wholememory_tensor_description_t* embedding_tensor_description,
                                                      wholememory_comm_t comm,
                                                      wholememory_memory_type_t memory_type,
                                                      wholememory_memory_location_t memory_location,
                                                      wholememory_embedding_cache_policy_t cache_policy,
                                                      size_t* embedding_entry_partition WGAMD_DEFAULT(NULL),
                                                      int user_defined_sms WGAMD_DEFAULT(-1),
                                                      int round_robin_size WGAMD_DEFAULT(0));
wholememory_error_code_t wholememory_destroy_embedding(wholememory_embedding_t wholememory_embedding);

/* embedding.h:151-161 — the [entries, dim] view of the padded table; set_optimizer allocates and initialises the states
 * (collective over the embedding's communicator) and can be called once. */
wholememory_tensor_t wholememory_embedding_get_embedding_tensor(wholememory_embedding_t wholememory_embedding);
wholememory_error_code_t wholememory_embedding_set_optimizer(wholememory_embedding_t wholememory_embedding,
                                                             wholememory_embedding_optimizer_t optimizer);

/* embedding.h:173-178 — wholememory_gather on the embedding tensor; with a cache: hits are copied from the cache lines,
 * only the misses go to the table (and, with adjust_cache, are then inserted).  Collective when the table is. */
wholememory_error_code_t wholememory_embedding_gather(wholememory_embedding_t wholememory_embedding,
                                                      wholememory_tensor_t indices,
                                                      wholememory_tensor_t output,
                                                      bool adjust_cache,
                                                      wholememory_env_func_t* p_env_fns,
                                                      int64_t stream_int);

/* embedding.h:191-198 — collective.  indices int32|int64 [n] (global rows, duplicates allowed, negative = skipped),
 * grads fp32 [n, dim] (row stride >= dim).  Gradients of the same row — from any rank — are summed first, then the
 * optimizer updates the row once (embedding.cpp:136-313). */
wholememory_error_code_t wholememory_embedding_gather_gradient_apply(wholememory_embedding_t wholememory_embedding,
                                                                     wholememory_tensor_t indices,
                                                                     wholememory_tensor_t grads,
                                                                     bool adjust_cache,
                                                                     float lr,
                                                                     wholememory_env_func_t* p_env_fns,
                                                                     int64_t stream_int);

/* embedding.h:205-216 — NULL-terminated names ("m" "v" "beta12t" | "state_sum" | "v" | none); NULL for an unknown name. */
const char* const* wholememory_embedding_get_optimizer_state_names(wholememory_embedding_t wholememory_embedding);
wholememory_tensor_t wholememory_embedding_get_optimizer_state(wholememory_embedding_t wholememory_embedding,
                                                               const char* name);

/* embedding.h:223-233 — READONLY cache: writeback has nothing to do; drop empties every line of this rank's cache and
 * zeroes its statistics.  READWRITE cache (collective, ends with a barrier): writeback copies every modified line —
 * embedding row and optimizer state — to its table row and keeps the lines; drop does the same and empties the cache. */
wholememory_error_code_t wholememory_embedding_writeback_cache(wholememory_embedding_t wholememory_embedding,
                                                               int64_t stream_int);
wholememory_error_code_t wholememory_embedding_drop_all_cache(wholememory_embedding_t wholememory_embedding,
                                                              int64_t stream_int);

/* Not in the reference: hits / valid lookups since creation (or the last drop) and the number of cache lines of THIS
 * rank's cache; all zero without a cache.  `lines` may be NULL. */
wholememory_error_code_t wgamd_embedding_cache_stats(wholememory_embedding_t wholememory_embedding, int64_t* hits,
                                                     int64_t* lookups, int64_t* lines);

#ifdef __cplusplus
}
#endif
#endif /* WGAMD_EMBEDDING_H_ */
