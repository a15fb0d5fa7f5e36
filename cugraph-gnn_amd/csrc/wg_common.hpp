// Internal helpers shared by the C-ABI translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <vector>
#include <mutex>
#include <string>
#include <unordered_map>

#include "wgamd_ops.h"
#include "wgamd_tensor.h"
#include "wgamd_types.h"

struct wholememory_tensor_ {
  void* storage_ptr;  // element 0 before storage_offset
  wholememory_tensor_description_t desc;
  wholememory_tensor_* root;      // nullptr for a root tensor
  wholememory_handle_t handle;    // nullptr unless backed by a DISTRIBUTED handle
  bool owns_handle = false;       // wholememory_create_tensor: destroy_tensor frees the handle too
  uint64_t handle_serial = 0;     // ... if it is still THAT handle: an address can be handed out again after a communicator
                                  // force-released the handles it still had (wgamd_free_if_serial)
};

// wg_comm.hip: the serial number of a live handle (0 = not live), and a free that only acts on the handle that carries it
extern "C" uint64_t wgamd_handle_serial(wholememory_handle_t h);
extern "C" wholememory_error_code_t wgamd_free_if_serial(wholememory_handle_t h, uint64_t serial);

namespace wgamd {

struct hip_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct logic_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct invalid_input : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct comm_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline std::string fmt(const char* f, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return buf;
}

#define WG_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      throw ::wgamd::hip_error(::wgamd::fmt("%s:%d HIP error %d (%s) in %s", __FILE__,       \
                                            __LINE__, (int)e__, hipGetErrorString(e__), #expr)); \
    }                                                                                        \
  } while (0)

#define WG_EXPECTS(cond, ...)                                                              \
  do {                                                                                     \
    if (!(cond)) throw ::wgamd::logic_error(::wgamd::fmt(__VA_ARGS__));                    \
  } while (0)

#define WG_REQUIRE_INPUT(cond, ...)                                                        \
  do {                                                                                     \
    if (!(cond)) throw ::wgamd::invalid_input(::wgamd::fmt(__VA_ARGS__));                  \
  } while (0)

// Runs `body`, mapping exceptions to the ABI's return codes (one stderr line each), the same
// convention as /root/reference/cpp/src/wholegraph_ops/unweighted_sample_without_replacement_impl_mapped.cu:57-66.
template <typename F>
inline wholememory_error_code_t guarded(const char* op, F&& body)
{
  try {
    body();
  } catch (const invalid_input& e) {
    fprintf(stderr, "[wholegraph_amd] %s: invalid input: %s\n", op, e.what());
    return WHOLEMEMORY_INVALID_INPUT;
  } catch (const logic_error& e) {
    fprintf(stderr, "[wholegraph_amd] %s: logic error: %s\n", op, e.what());
    return WHOLEMEMORY_LOGIC_ERROR;
  } catch (const hip_error& e) {
    fprintf(stderr, "[wholegraph_amd] %s: device error: %s\n", op, e.what());
    return WHOLEMEMORY_CUDA_ERROR;
  } catch (const comm_error& e) {
    fprintf(stderr, "[wholegraph_amd] %s: communication error: %s\n", op, e.what());
    return WHOLEMEMORY_COMMUNICATION_ERROR;
  } catch (const std::bad_alloc&) {
    fprintf(stderr, "[wholegraph_amd] %s: out of memory\n", op);
    return WHOLEMEMORY_OUT_OF_MEMORY;
  } catch (...) {
    fprintf(stderr, "[wholegraph_amd] %s: unknown error\n", op);
    return WHOLEMEMORY_UNKNOW_ERROR;
  }
  return WHOLEMEMORY_SUCCESS;
}

inline size_t dtype_size(wholememory_dtype_t d) { return wholememory_dtype_get_element_size(d); }

// pointer to element `storage_offset` of a raw-pointer tensor
inline void* tensor_data(wholememory_tensor_t t)
{
  if (t == nullptr || t->storage_ptr == nullptr) return nullptr;
  return static_cast<char*>(t->storage_ptr) + t->desc.storage_offset * (int64_t)dtype_size(t->desc.dtype);
}

// ---- scratch / output memory through the caller's callbacks --------------------------------
// (the roles of /root/reference/cpp/src/wholememory_ops/temp_memory_handle.hpp:12-62 and
//  output_memory_handle.hpp:22-35, re-expressed as small RAII objects)
class temp_buffer {
 public:
  explicit temp_buffer(wholememory_env_func_t* env) : env_(env)
  {
    env_->temporary_fns.create_memory_context_fn(&ctx_, env_->temporary_fns.global_context);
  }
  temp_buffer(const temp_buffer&)            = delete;
  temp_buffer& operator=(const temp_buffer&) = delete;
  ~temp_buffer()
  {
    if (ptr_) env_->temporary_fns.free_fn(ctx_, env_->temporary_fns.global_context);
    env_->temporary_fns.destroy_memory_context_fn(ctx_, env_->temporary_fns.global_context);
  }
  void* alloc(int64_t elt_count, wholememory_dtype_t dtype,
              wholememory_memory_allocation_type_t where = WHOLEMEMORY_MA_DEVICE)
  {
    if (ptr_) {
      env_->temporary_fns.free_fn(ctx_, env_->temporary_fns.global_context);
      ptr_ = nullptr;
    }
    wholememory_tensor_description_t d;
    wholememory_initialize_tensor_desc(&d);
    d.dim        = 1;
    d.sizes[0]   = elt_count > 0 ? elt_count : 1;  // never hand a 0-byte request to the allocator
    d.strides[0] = 1;
    d.dtype      = dtype;
    ptr_ = env_->temporary_fns.malloc_fn(&d, where, ctx_, env_->temporary_fns.global_context);
    if (!ptr_) throw std::bad_alloc();
    return ptr_;
  }
  template <typename T>
  T* device(int64_t n, wholememory_dtype_t dt)
  {
    return static_cast<T*>(alloc(n, dt, WHOLEMEMORY_MA_DEVICE));
  }
  void* bytes(int64_t n) { return alloc(n, WHOLEMEMORY_DT_INT8, WHOLEMEMORY_MA_DEVICE); }
  void* pointer() const { return ptr_; }
  void* pinned_bytes(int64_t n) { return alloc(n, WHOLEMEMORY_DT_INT8, WHOLEMEMORY_MA_PINNED); }

 private:
  wholememory_env_func_t* env_;
  void* ctx_ = nullptr;
  void* ptr_ = nullptr;
};

// Several scratch pieces behind ONE env allocation: every temp_buffer costs four callbacks into the caller's allocator
// (create context, malloc, free, destroy context) — Python callbacks in the torch binding — so an op declares its pieces,
// commits once, and addresses them by offset.
class temp_arena {
 public:
  explicit temp_arena(wholememory_env_func_t* env) : buf_(env) {}
  size_t add(size_t bytes)
  {
    const size_t off = total_;
    total_ += (bytes + 255) / 256 * 256;
    return off;
  }
  void commit() { base_ = static_cast<char*>(buf_.bytes((int64_t)(total_ ? total_ : 1))); }
  template <typename T>
  T* at(size_t off) const
  {
    return reinterpret_cast<T*>(base_ + off);
  }

 private:
  temp_buffer buf_;
  size_t total_ = 0;
  char* base_   = nullptr;
};

// Allocates an op OUTPUT of `elt_count` elements in the caller's memory_context (never freed here).
inline void* output_alloc(wholememory_env_func_t* env, void* memory_context, int64_t elt_count,
                          wholememory_dtype_t dtype)
{
  wholememory_tensor_description_t d;
  wholememory_initialize_tensor_desc(&d);
  d.dim        = 1;
  d.sizes[0]   = elt_count;
  d.strides[0] = 1;
  d.dtype      = dtype;
  void* p = env->output_fns.malloc_fn(&d, WHOLEMEMORY_MA_DEVICE, memory_context,
                                      env->output_fns.global_context);
  if (elt_count > 0 && p == nullptr) throw std::bad_alloc();
  return p;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Compute units a launch on `stream` can occupy: the popcount of the stream's CU mask (hipExtStreamCreateWithCUMask:
// a caller may give the walk and the forward pass disjoint slices of the chip), else the device's CU count.  Kernels that
// run ONE persistent workgroup per CU size their grid from this, so a masked stream never queues a second round of
// workgroups behind the first.  The device's CU count is cached per device; a non-null stream's mask is asked for on
// every call (a stream handle can be recycled by the runtime for a stream with another mask, so it is no cache key).
inline int stream_cu_count(hipStream_t stream)
{
  static std::mutex m;
  static std::unordered_map<int, int> device_cus;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> g(m);
    auto it = device_cus.find(dev);
    if (it != device_cus.end()) cus = it->second;
  }
  if (cus == 0) {
    cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    std::lock_guard<std::mutex> g(m);
    device_cus[dev] = cus;
  }
  if (stream == nullptr) return cus;
  uint32_t mask[32] = {0};
  if (hipExtStreamGetCUMask(stream, 32, mask) != hipSuccess) {
    (void)hipGetLastError();
    return cus;
  }
  int bits = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  return (bits > 0 && bits < cus) ? bits : cus;
}

// A size that lives either on the host (ABI ops: the value is known) or on the device (no-sync
// walk: `dev` points at it and `host` is only the capacity used to size the grid).
struct dev_count {
  int host;
  const int* dev;
  __device__ __forceinline__ int get() const { return dev ? *dev : host; }
};

// Where a seed's PCG streams come from.  ABI ops: one scalar `seed`, stream index = the seed's
// index in the call.  Batched no-sync walk: mini-batch b of the call group has its own 64-bit seed
// (`seeds_dev[b]`) and its streams are numbered from the START of its own segment, so every
// mini-batch reproduces exactly what a single-batch call with that seed would have drawn.
struct rng_plan {
  uint64_t seed;
  const unsigned long long* seeds_dev;  // [G] or nullptr
  const int* target_batch;              // [T] batch of every target, or nullptr (single batch)
  const int* target_seg;                // [G+1] first target of every batch
  __device__ __forceinline__ void resolve(int i, uint64_t& s, int& i_local) const
  {
    s       = seed;
    i_local = i;
    if (target_batch) {
      const int b = target_batch[i];
      i_local     = i - target_seg[b];
      s           = seeds_dev[b];
    } else if (seeds_dev) {
      s = seeds_dev[0];
    }
  }
};

// ---- device-side utilities implemented in wg_scan.hip ---------------------------------------
// out[0..n] (n+1 entries) = exclusive scan of in[0..n-1]; out[n] = total. in/out may alias.
// `tmp` needs scan_tmp_ints(n) ints.  With `n_live_dev` (device int, <= n) the inputs past the live
// count must be zero up to the end of the tile holding it; outputs are then defined for
// [0, *n_live_dev] and at [n] only, and tiles past the live count cost no memory traffic.
int64_t scan_tmp_ints(int64_t n);
constexpr int kScanTile = 2048;
void exclusive_scan_i32(const int* in, int* out, int64_t n, int* tmp, hipStream_t stream,
                        const int* n_live_dev = nullptr);
// the same with the per-tile sums (tile t = elements [t kScanTile, (t + 1) kScanTile)) already in `sums` — one launch
void exclusive_scan_i32_presummed(const int* in, int* out, int64_t n, const int* sums, hipStream_t stream,
                                  const int* n_live_dev = nullptr);
// State of one single-pass (chained) scan launch — see wg_scan_chain.hpp.  The buffers are library-owned, one set per
// (device, stream); `scan_chain_acquire` hands out the next epoch of that set (host side, no device work after the first
// call on a stream).  At most kScanChainTiles tiles per launch.
struct scan_chain {
  unsigned long long* tiles;   // [tile]: epoch:30 | flag:2 | value:32
  unsigned int* counters;      // [0] ticket, [1] workgroups done; zero between launches
  unsigned int epoch;
};
constexpr int64_t kScanChainTiles = (int64_t)1 << 20;   // 2^31 elements / kScanTile
scan_chain scan_chain_acquire(hipStream_t stream);


// ---- building blocks shared by the ABI ops and the no-sync walk (wg_fused.hip) -----------------
// uniform sampling of `n` targets (n.host = capacity): cnt/offsets have n.host+1 entries
// `loc` (nullable; needs row_start / row_deg and 0 < M <= 32): walk the seeds grouped by vertex-id range instead of in list
// order (wg_sample.hip, loc_rec) — same result, the picks of a vertex that occurs many times in the list share cache lines.
// Scratch: sample_locality_hist_ints() ints and 32 bytes per seed of capacity.
struct sample_locality {
  int* hist;     // [sample_locality_hist_ints()]
  void* recs;    // [n.host] 32-byte records, 32-byte aligned
  int shift;     // bucket of a vertex id = id >> shift (clamped): sample_locality_shift(id_bound)
};
int64_t sample_locality_hist_ints();
int sample_locality_shift(int64_t id_bound);
void uniform_sample_enqueue(const int64_t* row_ptr, const void* col, bool col64, const void* seeds, bool seeds64,
                            dev_count n, int M, rng_plan random_seed, const int* offsets, void* dst, int* src_lid,
                            int64_t* edge_gid, hipStream_t stream,
                            const int64_t* row_start = nullptr, const int* row_deg = nullptr,
                            const sample_locality* loc = nullptr);
// row_start / row_deg (nullable): first CSR slot and degree of every live seed, written next to the counts — the sampling
// kernel then reads them by seed index (coalesced) instead of chasing seeds -> row_ptr again
void sample_count_enqueue(const int64_t* row_ptr, const void* seeds, bool seeds64, dev_count n, int M, int* cnt,
                          int* big_deg, hipStream_t stream, int64_t* row_start = nullptr, int* row_deg = nullptr);
// biased (A-Res) sampling, 0 < M <= 256: `big_list` (weighted_list_ints(n.host) ints) receives the row lists by size class;
// `slab` = kWeightedBlocks slabs of slab_len keys for the rows longer than kWeightedLdsKeys candidates
// (slab_len >= the longest such row, e.g. the graph's maximum degree).  Weights FLOAT or DOUBLE.
constexpr int kWeightedBlocks  = 1024;
constexpr int kWeightedLdsKeys = 12288;
// row lists of a biased hop, built by the count kernel (block-aggregated appends): the rows that are sampled (deg > M)
// by size class — 0: deg <= 16, 1: <= 32, 2: <= 64 (one key per lane: 4 / 2 / 1 rows per wave), 3: <= 128, 4: <= 256,
// 5: <= 512, 6: <= 1024 (one wave per row, 2 / 4 / 8 / 16 keys per lane in registers), 7: <= 16384 and 8: more candidates
// (persistent workgroups, the huge rows first, dealt out through a queue head).  Rows copied whole (deg <= M) need no
// list.  Layout in ints:  [0 .. 8] list lengths | [9] queue head | [10] longest row that needs a key slab | pad to 16 |
// list c at 16 + c * cap
constexpr int kWeightedLists    = 9;
constexpr int kWeightedListHead = 16;
// one more list region behind the nine: rows a pruned kernel (wg_sample.hip, "threshold pruning") hands back to the exact
// workgroup kernel — [11] its length, [12] its queue head
constexpr int kWeightedRedoList  = 9;
constexpr int kWeightedRedoCount = 11;
constexpr int kWeightedRedoHead  = 12;
inline int64_t weighted_list_ints(int64_t cap)
{
  return kWeightedListHead + (int64_t)(kWeightedLists + 1) * (cap > 0 ? cap : 1);
}
void weighted_count_enqueue(const int64_t* row_ptr, const void* seeds, bool seeds64, dev_count n, int M, int* cnt,
                            int* big_list, hipStream_t stream);
void weighted_sample_enqueue(const int64_t* row_ptr, const void* col, bool col64, const void* weights, bool weights64,
                             const void* seeds, bool seeds64, dev_count n, int M, rng_plan random_seed, const int* offsets,
                             int* big_list, uint32_t* slab, int64_t slab_len, void* dst, int* src_lid,
                             int64_t* edge_gid, hipStream_t stream);
// renumbering: table of `slots` (power of two >= 2*(T.host+E.host)) entries, slot_of[T.host+E.host],
// rank[E.host+1], scan_tmp[scan_tmp_ints(E.host+1)].  unique_out may be NULL for `prepare`.
int64_t append_unique_slots(int64_t capacity);
// `bv` describes the call group (single batch: all-null / G = 1).  With G > 1 the table keys are
// (batch, id) pairs packed in int64, so `keys` must then hold int64 slots whatever the id type.
struct batch_view {
  const int* target_batch;  // [T]   batch of every target            (nullptr: one batch)
  const int* target_seg;    // [G+1] first target of every batch      (nullptr: one batch)
  const int* edge_row;      // [E]   target row of every sampled edge (needed when G > 1)
  const int* edge_offsets;  // [T+1] sample offsets (edge_seg[b] = edge_offsets[target_seg[b]])
  int G;
  int64_t id_bound;         // ids are < id_bound (e.g. the vertex count); 0 = unknown.  With G > 1 and a bound the renumber
                            // table packs (batch, id, first position) into ONE 64-bit word per slot when they fit
  int* unique_batch;        // out [T+E] batch of every unique entry  (nullable)
  int* unique_seg;          // out [G+1] first unique entry of every batch (nullable)
  // PyG-style walk ("expand only the vertices discovered by the previous hop"): the SAMPLED list
  // (frontier) is then not the renumber target list (all vertices so far).  Null = same lists.
  const int* sample_batch;  // [S]   batch of every sampled (frontier) vertex; edge_row indexes this list
  const int* sample_seg;    // [G+1] first frontier vertex of every batch; edge_offsets is over this list
  const int* sample_local0; // [G]   local id (inside its batch) of every batch's first frontier vertex
  void* frontier_out;       // out [E]   ids first seen in this hop, by (batch, first appearance) = next frontier
  int* frontier_batch_out;  // out [E]
  int* frontier_seg_out;    // out [G+1]
  int* frontier_local0_out; // out [G]   local id of the next frontier's first vertex = batch size before this hop
  int* neighbor_local_out;  // out [E]   per-batch LOCAL id of every edge's neighbour
  int* center_local_out;    // out [E]   per-batch LOCAL id of every edge's expanded vertex
  int no_pad;               // 1: leave the capacity slack of the unique list unwritten (WGAMD_HOP_NO_UNIQUE_PAD)
  __host__ __device__ const int* sbatch() const { return sample_batch ? sample_batch : target_batch; }
  __host__ __device__ const int* sseg() const { return sample_seg ? sample_seg : target_seg; }
};
// `tgt64` / `nbr64`: width of the target (= unique, frontier: the API's id type) and of the neighbour list (the sampled
// columns: INT when the CSR keeps 32-bit columns behind an INT64 API).  INT targets with INT64 neighbours are refused.
void append_unique_prepare_enqueue(const void* targets, dev_count T, bool tgt64, const void* neighbors, dev_count E, bool nbr64,
                                   batch_view bv, void* keys, int* minpos, int64_t slots, int* slot_of, int* rank,
                                   int* scan_tmp, hipStream_t stream);
void append_unique_emit_enqueue(const void* targets, dev_count T, bool tgt64, const void* neighbors, dev_count E, bool nbr64,
                                batch_view bv, const int* minpos, const int* slot_of, const int* rank, int64_t slots,
                                void* unique_out, int* map_out, int* counts_out /*nullable: {E, T+U}*/,
                                hipStream_t stream);


// ---- local row kernels (wg_gather.hip) and the DISTRIBUTED pipeline built on them (wg_comm.hip) ----
void local_rows_gather(const char* table, wholememory_matrix_description_t tm, const void* idx,
                       wholememory_dtype_t idx_dtype, int64_t n, char* out, wholememory_matrix_description_t om,
                       hipStream_t stream);
void local_rows_scatter(const char* in, wholememory_matrix_description_t im, const void* idx,
                        wholememory_dtype_t idx_dtype, int64_t n, char* table, wholememory_matrix_description_t tm,
                        hipStream_t stream);
// dst row dst_idx[i] <- src row src_idx[i] (same dtype; a negative index on either side skips the row)
void local_rows_permute(const char* src, wholememory_matrix_description_t sm, const int64_t* src_idx, const int64_t* dst_idx,
                        int64_t n, char* dst, wholememory_matrix_description_t dm, hipStream_t stream);
// Steps 1-4 of the DISTRIBUTED row exchange (wg_comm.hip), shared by gather / scatter and by the embedding gradient
// routing: who owns each id, how many ids every pair of ranks trades, the ids grouped by owner (the peers' buckets first
// in rank order, MY bucket last) and the ids the peers ask me for (localised to my partition).
struct id_exchange {
  wholememory_comm_t comm = nullptr;
  int W = 1, me = 0;
  int64_t local_start = 0, local_rows = 0;
  std::vector<size_t> send_cnt, recv_cnt;                // all ids per peer, self included
  std::vector<int64_t> bucket_start;                     // first grouped position of every owner
  std::vector<size_t> send_n, recv_n, send_at, recv_at;  // what really crosses the wire (ids), and where it sits
  int64_t n = 0, n_remote = 0, self_cnt = 0, recv_total = 0;
  bool self_direct = false;
  int64_t *d_grouped = nullptr, *d_pos = nullptr, *d_recv_ids = nullptr, *d_self_ids = nullptr, *d_self_pos = nullptr;
  temp_arena scratch;
  std::vector<size_t> so, sb, ro, rb;

  explicit id_exchange(wholememory_env_func_t* env) : scratch(env) {}
  // steps 1-3 (two host syncs, one on a single-rank communicator): fills every count above.  self_direct_: my own bucket
  // stays out of the exchange (exchange_ids localises it in place instead)
  void plan(wholememory_handle_t h, size_t entry_bytes, int64_t row0, const void* idx, wholememory_dtype_t idx_dtype,
            int64_t n_, bool self_direct_, hipStream_t stream);
  // step 4: ids all-to-all-v into `recv_ids` (room for recv_total ids), localised; enqueues only
  void exchange_ids(int64_t* recv_ids, hipStream_t stream);
  // rows travel WITH the ids (scatter direction): my grouped rows -> the owners' receive order
  void rows_to_owners(const char* send, char* recv, size_t row_bytes, hipStream_t stream);
  // rows travel BACK (gather direction): what I gathered for peer r -> peer r's grouped order
  void rows_to_askers(const char* send, char* recv, size_t row_bytes, hipStream_t stream);

 private:
  void scaled(size_t unit_send, size_t unit_recv, const std::vector<size_t>& s_n, const std::vector<size_t>& s_at,
              const std::vector<size_t>& r_n, const std::vector<size_t>& r_at);
};

// gather: dense = output rows; scatter: dense = input rows.  `tm` describes the GLOBAL table (sizes[0] = all rows).
void distributed_rows_op(bool scatter, wholememory_handle_t handle, wholememory_matrix_description_t tm, const void* idx,
                         wholememory_dtype_t idx_dtype, int64_t n, char* dense, wholememory_matrix_description_t dense_m,
                         wholememory_env_func_t* env, hipStream_t stream);

}  // namespace wgamd
