/*
 * wgamd_comm.h — communicator + DISTRIBUTED memory handles: the part of libwholegraph's runtime that
 * the multi-GPU feature fetch stands on (/root/reference/cpp/include/wholememory/wholememory.h:84-420,
 * wholememory_tensor.h:27-76; implementation cpp/src/wholememory/communicator.cpp, nccl_comms.cpp,
 * memory_handle.cpp:1393-1745).
 *
 * One process per GPU.  The communicator wraps an RCCL communicator (bootstrap: 128-byte unique id moved
 * by the caller, e.g. torch.distributed.broadcast — comm.py:159-166 of the reference).  Memory types on
 * WHOLEMEMORY_ML_DEVICE:
 *   * WHOLEMEMORY_MT_DISTRIBUTED — every rank holds a contiguous range of the entries in its own HBM and remote rows
 *     are fetched by all-to-all over xGMI (wholememory_gather / wholememory_scatter accept tensors backed by such a
 *     handle; one host synchronisation per call);
 *   * WHOLEMEMORY_MT_CHUNKED / WHOLEMEMORY_MT_CONTINUOUS — the same partition, PEER-MAPPED: available when all ranks share
 *     a node (wholememory_communicator_support_type_location says so).  Every rank exports its partition through HIP IPC
 *     and opens its peers'; gather / scatter are then one kernel whose loads / stores cross xGMI directly, with no host
 *     synchronisation (the reference's mapped path, cpp/src/wholememory_ops/gather_op_impl_mapped.cu:18-67).  There is
 *     no flat global pointer: a CONTINUOUS handle is addressed like a CHUNKED one (wgamd_get_peer_pointers).
 * WHOLEMEMORY_ML_HOST (memory_handle.cpp:233-262, :432-520): the same three types with every partition in PINNED HOST
 * memory, which the GPU's loads and stores reach in place over PCIe — a hipHostMalloc block of the owning process for
 * DISTRIBUTED (and for any type on a single-rank communicator), a POSIX shared-memory segment mapped and registered by
 * every process for the peer-mapped types.  Meant for tables that outgrow HBM, behind the READWRITE device cache of
 * wgamd_embedding.h.  HIERARCHY and NVSHMEM are not reproduced and return WHOLEMEMORY_NOT_SUPPORTED.  RCCL is resolved at run time (dlopen "librccl.so"), so the library loads on a CPU-only box
 * and inside a PyTorch process shares torch's RCCL.  One collective per communicator at a time (as RCCL requires): the
 * communicator's pinned count buffer is shared by its calls, so two gathers on ONE communicator from two host threads race.
 *
 * Scratch and streams: a gather / scatter on a DISTRIBUTED handle synchronises the stream ONCE (the count read-back) and NOT
 * at its end — its scratch (RCCL send / receive buffers included) goes back to p_env_fns->temporary_fns while the last
 * kernels are still in flight.  The temporary allocator must therefore be STREAM-ORDERED ON THE `stream` ARGUMENT: a
 * caching allocator that reuses a block only for work enqueued later on that same stream (torch's, when `stream` is torch's
 * current stream — what wholegraph_amd.env hands over: get_stream() and the allocator callbacks both use the current
 * stream), or one that synchronises on free (the library default: hipFree).  An allocator keyed to another stream, or a
 * free that returns memory to other streams at once, needs an event wait in its free callback.
 * wholememory_free of a peer-mapped handle is COLLECTIVE (device sync + barrier before the mappings are closed, barrier
 * before the partition is released; memory_handle.cpp:1007-1029): call it on every rank, in the same order.
 */
#ifndef WGAMD_COMM_H_
#define WGAMD_COMM_H_

#include "wgamd_tensor.h"
#include "wgamd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* wholememory.h:91-98 — process-wide init / teardown (log_level: 0 fatal … 5 trace; only stored) */
wholememory_error_code_t wholememory_init(unsigned int flags, int log_level);
wholememory_error_code_t wholememory_finalize(void);

#define WHOLEMEMORY_UNIQUE_ID_BYTES (128)
typedef struct wholememory_unique_id_t {
  char internal[WHOLEMEMORY_UNIQUE_ID_BYTES];
} wholememory_unique_id_t;

/* wholememory.h:118-134 */
wholememory_error_code_t wholememory_create_unique_id(wholememory_unique_id_t* unique_id);
wholememory_error_code_t wholememory_create_communicator(wholememory_comm_t* comm,
                                                         wholememory_unique_id_t unique_id,
                                                         int rank,
                                                         int size);
wholememory_error_code_t wholememory_destroy_communicator(wholememory_comm_t comm);
/* wholememory.h:136-163 — COLLECTIVE over `comm`: ranks passing the same color >= 0 form one new communicator, ordered by
 * (key, rank in comm); color < 0 (WHOLEMEMORY_SPLIT_NOCOLOR) takes part and receives NULL.  Built from ncclGetUniqueId +
 * ncclCommInitRank per colour (the reference calls ncclCommSplit, communicator.cpp:753-830). */
#define WHOLEMEMORY_SPLIT_NOCOLOR (-1)
wholememory_error_code_t wholememory_split_communicator(wholememory_comm_t* new_comm,
                                                        wholememory_comm_t comm,
                                                        int color,
                                                        int key);
/* wholememory.h:160-205 */
wholememory_error_code_t wholememory_communicator_support_type_location(
  wholememory_comm_t comm, wholememory_memory_type_t memory_type, wholememory_memory_location_t memory_location);
wholememory_error_code_t wholememory_communicator_get_rank(int* rank, wholememory_comm_t comm);
wholememory_error_code_t wholememory_communicator_get_size(int* size, wholememory_comm_t comm);
wholememory_error_code_t wholememory_communicator_barrier(wholememory_comm_t comm);
/* wholememory.h:199-225.  Single node, RCCL only: local size = size (1 when the ranks span hosts), nobody is in an MNNVL
 * clique, the only backend is WHOLEMEMORY_DB_NCCL (= RCCL; NVSHMEM -> NOT_SUPPORTED). */
typedef enum wholememory_distributed_backend_t {
  WHOLEMEMORY_DB_NONE = 0,
  WHOLEMEMORY_DB_NCCL,
  WHOLEMEMORY_DB_NVSHMEM
} wholememory_distributed_backend_t;
typedef struct clique_info_t {
  int is_in_clique;
  int clique_first_rank;
  int clique_rank;
  int clique_rank_num;
  int clique_id;
  int clique_num;
} clique_info_t;
wholememory_error_code_t wholememory_communicator_get_local_size(int* local_size, wholememory_comm_t comm);
wholememory_error_code_t wholememory_communicator_get_clique_info(clique_info_t* clique_info, wholememory_comm_t comm);
bool wholememory_communicator_is_bind_to_nvshmem(wholememory_comm_t comm);
wholememory_error_code_t wholememory_communicator_set_distributed_backend(
  wholememory_comm_t comm, wholememory_distributed_backend_t distributed_backend);
wholememory_distributed_backend_t wholememory_communicator_get_distributed_backend(wholememory_comm_t comm);
/* wholememory.h:467-470 */
bool wholememory_is_intranode_communicator(wholememory_comm_t comm);
bool wholememory_is_intra_mnnvl_communicator(wholememory_comm_t comm);
bool wholememory_is_build_with_nvshmem(void);
/* wholememory.h:426 — device count asked of a forked child, so the caller can still fork its workers; -1 on error */
int fork_get_device_count(void);
/* (no reference counterpart) what RCCL itself reports for the communicator: ncclCommCount and ncclGetVersion (-1 where the
 * loaded library lacks the symbol) — evidence in bench.py's line that the exchange ran over RCCL with that many ranks */
wholememory_error_code_t wgamd_communicator_rccl_info(wholememory_comm_t comm, int* rccl_ranks, int* rccl_version);

/* wholememory.h:225-236 — total_size bytes split into entries of data_granularity bytes; rank_entry_partition
 * (entries per rank, nullable) overrides the equal split of wholememory_equal_entry_partition_plan. */
wholememory_error_code_t wholememory_malloc(wholememory_handle_t* wholememory_handle_ptr,
                                            size_t total_size,
                                            wholememory_comm_t comm,
                                            wholememory_memory_type_t memory_type,
                                            wholememory_memory_location_t memory_location,
                                            size_t data_granularity,
                                            size_t* rank_entry_partition WGAMD_DEFAULT(NULL));
wholememory_error_code_t wholememory_free(wholememory_handle_t wholememory_handle);
/* peer-mapped handles: pointer of every rank's partition as mapped into THIS process (the chunked view of
 * wholememory_get_global_reference, wholememory.h:300-330); `pointers` has room for world-size entries */
wholememory_error_code_t wgamd_get_peer_pointers(void** pointers, wholememory_handle_t wholememory_handle);

/* Row addresses of a peer-mapped (CHUNKED / CONTINUOUS, more than one rank) 2-D table, for kernels that read the rows
 * themselves (the one-kernel SAGE layer with the fetch folded in; src_ids_dtype = WGAMD_IDS_BYTE_OFFSETS, wgamd_ext.h):
 * offsets[i] = byte distance of row ids[i] (INT | INT64) from *base, the lowest partition base of this process's mapping;
 * a negative id or one past the last row gets the offset of the first row of that lowest partition (the readers dereference
 * base + offset unconditionally: the answer is always a mapped, 16-byte-aligned address).  Not collective.  WHOLEMEMORY_LOGIC_ERROR for a handle that is not
 * peer-mapped (DISTRIBUTED rows are not addressable; a single-partition handle is read through its local tensor). */
wholememory_error_code_t wgamd_mapped_row_offsets(wholememory_tensor_t table, const void* ids, wholememory_dtype_t ids_dtype,
                                                  int64_t n, int64_t* offsets, void** base, void* stream);
/* the HIP IPC steps on their own: a 64-byte handle of a hipMalloc'ed block / mapping one exported by another process */
wholememory_error_code_t wgamd_ipc_export(void* device_ptr, void* handle64);
wholememory_error_code_t wgamd_ipc_open(const void* handle64, void** device_ptr);
wholememory_error_code_t wgamd_ipc_close(void* device_ptr);
/* wholememory.h:245-330 */
wholememory_error_code_t wholememory_get_communicator(wholememory_comm_t* comm,
                                                      wholememory_handle_t wholememory_handle);
wholememory_memory_type_t wholememory_get_memory_type(wholememory_handle_t wholememory_handle);
wholememory_memory_location_t wholememory_get_memory_location(wholememory_handle_t wholememory_handle);
size_t wholememory_get_total_size(wholememory_handle_t wholememory_handle);
size_t wholememory_get_data_granularity(wholememory_handle_t wholememory_handle);
wholememory_error_code_t wholememory_get_local_memory(void** local_ptr,
                                                      size_t* local_size,
                                                      size_t* local_offset,
                                                      wholememory_handle_t wholememory_handle);
/* wholememory.h:281-309,346-392.  local/cross communicators exist for HIERARCHY handles only (-> NOT_SUPPORTED);
 * get_rank_memory answers for the caller's own rank and, on a peer-mapped handle, for every rank; the flat global pointer
 * exists when one rank holds all rows (partitions are mapped one by one here: wgamd_get_peer_pointers), else INVALID_INPUT. */
wholememory_error_code_t wholememory_get_local_communicator(wholememory_comm_t* comm,
                                                            wholememory_handle_t wholememory_handle);
wholememory_error_code_t wholememory_get_cross_communicator(wholememory_comm_t* comm,
                                                            wholememory_handle_t wholememory_handle);
wholememory_distributed_backend_t wholememory_get_distributed_backend(wholememory_handle_t wholememory_handle);
wholememory_error_code_t wholememory_get_local_size(size_t* local_size, wholememory_handle_t wholememory_handle);
wholememory_error_code_t wholememory_get_local_offset(size_t* local_offset, wholememory_handle_t wholememory_handle);
wholememory_error_code_t wholememory_get_rank_memory(void** rank_memory_ptr,
                                                     size_t* rank_memory_size,
                                                     size_t* rank_memory_offset,
                                                     int rank,
                                                     wholememory_handle_t wholememory_handle);
wholememory_error_code_t wholememory_get_global_pointer(void** global_ptr, wholememory_handle_t wholememory_handle);
/* wholememory.h:380-420 — equal split: per = ceil(total / world); rank r owns [min(r*per,total), min((r+1)*per,total)) */
wholememory_error_code_t wholememory_equal_entry_partition_plan(size_t* entry_per_rank,
                                                                size_t total_entry_count,
                                                                int world_size);
wholememory_error_code_t wholememory_get_rank_partition_sizes(size_t* rank_mem_sizes,
                                                              wholememory_handle_t wholememory_handle);
wholememory_error_code_t wholememory_get_rank_partition_offsets(size_t* rank_mem_offsets,
                                                                wholememory_handle_t wholememory_handle);

/* wholememory_tensor.h:27-76 — a tensor whose storage is a handle (dim 1 or 2, partitioned along dim 0) */
wholememory_error_code_t wholememory_create_tensor(wholememory_tensor_t* wholememory_tensor,
                                                   wholememory_tensor_description_t* tensor_description,
                                                   wholememory_comm_t comm,
                                                   wholememory_memory_type_t memory_type,
                                                   wholememory_memory_location_t memory_location,
                                                   size_t* tensor_entry_partition WGAMD_DEFAULT(NULL));
wholememory_error_code_t wholememory_make_tensor_from_handle(
  wholememory_tensor_t* wholememory_tensor,
  wholememory_handle_t wholememory_handle,
  wholememory_tensor_description_t* tensor_description);
/* wholememory_tensor.h:124-140 — partition of dim 0 over the ranks in entries: offsets has world_size + 1 values, sizes
 * world_size; a tensor over plain memory is one partition ({0, n} / {n}) */
wholememory_error_code_t wholememory_tensor_get_entry_offsets(size_t* entry_offsets,
                                                              wholememory_tensor_t wholememory_tensor);
wholememory_error_code_t wholememory_tensor_get_entry_partition_sizes(size_t* entry_partition,
                                                                      wholememory_tensor_t wholememory_tensor);
/* wholememory_tensor.h:132-160 */
wholememory_error_code_t wholememory_tensor_get_local_entry_count(size_t* local_entry_count,
                                                                  wholememory_tensor_t wholememory_tensor);
wholememory_error_code_t wholememory_tensor_get_local_entry_start(size_t* local_entry_start,
                                                                  wholememory_tensor_t wholememory_tensor);
wholememory_error_code_t wholememory_tensor_map_local_tensor(wholememory_tensor_t wholememory_tensor,
                                                             wholememory_tensor_t* local_tensor);

/* wholememory.h:422-461 — headerless binary files of `file_entry_size`-byte entries <-> the handle's entries
 * (`memory_entry_size` = row stride in bytes, payload `memory_offset` bytes into the entry).  The files of the list are
 * one concatenated array; round_robin_size == 0 keeps file order = memory order, rr > 0 deals blocks of rr entries to
 * the ranks in turn.  Collective (ends with a barrier).  store writes THIS rank's entries to its own file. */
wholememory_error_code_t wholememory_load_from_file(wholememory_handle_t wholememory_handle,
                                                    size_t memory_offset,
                                                    size_t memory_entry_size,
                                                    size_t file_entry_size,
                                                    const char** file_names,
                                                    int file_count,
                                                    int round_robin_size);
wholememory_error_code_t wholememory_store_to_file(wholememory_handle_t wholememory_handle,
                                                   size_t memory_offset,
                                                   size_t memory_entry_stride,
                                                   size_t file_entry_size,
                                                   const char* local_file_name);

#ifdef __cplusplus
}
#endif
#endif /* WGAMD_COMM_H_ */
