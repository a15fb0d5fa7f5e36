// Edge renumbering of a sampled hop (graph_append_unique) and csr_add_self_loop, for gfx950.
//
// Replaces /root/reference/cpp/include/wholememory/graph_op.h:27-48 (reference kernels
// cpp/src/graph_ops/append_unique_func.cuh:44-341, csr_add_self_loop_func.cuh:13-47).
//
// Design (not the reference's bucketed CAS table with slot-order ids):
//   one open-addressing table keyed by node id whose value is the MINIMUM position of that id in
//   the concatenation  targets ++ neighbours  (atomicMin => order-independent, deterministic).
//     position <  T  -> the id is a target, its local id is that position
//     position == T+e-> neighbour e is the FIRST occurrence of a new id
//   flag first occurrences, exclusive-scan the flags over the neighbour list, and
//   local id = T + rank.  The tail of `unique` therefore comes out in first-appearance order —
//   exactly what the reference's host oracle produces
//   (cpp/tests/graph_ops/append_unique_test_utils.cu:52-84) and a strict refinement of the
//   reference device op, which leaves that order to CAS races.
//
// Call groups (no-sync walk, G mini-batches per launch): the table key becomes the pair
// (batch, id) packed in an int64, so each mini-batch is renumbered on its own in the same pass;
// targets/edges of one batch are contiguous, hence "minimum position" is still "first
// appearance inside the batch".  Output rows are global rows of the concatenated per-batch
// unique lists:  row(target i of batch b) = i + rank[edge_seg[b]],
//                row(new node first seen at edge e of batch b) = target_seg[b+1] + rank[e].
#include <algorithm>
#include <type_traits>

#include "wg_common.hpp"

namespace wgamd {
namespace {

constexpr int kEmptyPos = 0x7fffffff;

template <typename KeyT>
struct key_traits;
template <>
struct key_traits<int32_t> {
  using cas_t                             = unsigned int;
  static constexpr wholememory_dtype_t dt = WHOLEMEMORY_DT_INT;
};
template <>
struct key_traits<int64_t> {
  using cas_t                             = unsigned long long;
  static constexpr wholememory_dtype_t dt = WHOLEMEMORY_DT_INT64;
};

__device__ __forceinline__ uint32_t hash_key(uint64_t k)
{
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 29;
  return (uint32_t)(k ^ (k >> 32));
}

// The table buffer is sized for the CAPACITY of the call (no host sync => worst case), but only a
// power-of-two prefix sized for the LIVE element count is used: a call group that fills a third of
// its capacity then touches a third of the memory (less to clear, and the random probes of the
// insert stay inside a footprint the 256 MB Infinity Cache holds).
// Slot count actually used: exactly 2x the live elements (load factor <= 0.5, no power-of-two
// rounding — that alone wasted ~1/3 of the clear traffic and of the probe footprint on average).
__device__ __forceinline__ uint32_t live_slot_count(int live_elements, int64_t capacity_slots)
{
  uint32_t want = 2u * (uint32_t)(live_elements > 512 ? live_elements : 512);
  return (int64_t)want > capacity_slots ? (uint32_t)capacity_slots : want;
}
// hash -> [0, slots) by multiply-shift (no modulo)
__device__ __forceinline__ uint32_t slot_for(uint32_t hash, uint32_t slots)
{
  return (uint32_t)(((uint64_t)hash * slots) >> 32);
}

// Fixed grid, 16 B per lane per store, grid-stride over the LIVE slot prefix only (the buffers come from the
// caller's allocator, 256-B aligned; both byte patterns are uniform, so whole int4 stores are valid).
template <typename TableKeyT>
__global__ void __launch_bounds__(256)
table_clear_kernel(TableKeyT* keys, int* minpos, int64_t capacity_slots, dev_count T_, dev_count E_)
{
  const int64_t slots  = (int64_t)live_slot_count(T_.get() + E_.get(), capacity_slots);
  const int64_t tid    = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  constexpr int KPV    = 16 / (int)sizeof(TableKeyT);  // keys per 16-byte store
  const int4 kfill = make_int4(-1, -1, -1, -1);
  const int4 pfill = make_int4(kEmptyPos, kEmptyPos, kEmptyPos, kEmptyPos);
  const int64_t kvec = (slots + KPV - 1) / KPV, pvec = (slots + 3) / 4;  // capacity is a multiple of 1024 slots
  int4* k4 = reinterpret_cast<int4*>(keys);
  int4* p4 = reinterpret_cast<int4*>(minpos);
  for (int64_t i = tid; i < kvec; i += stride) k4[i] = kfill;
  for (int64_t i = tid; i < pvec; i += stride) p4[i] = pfill;
}

// batch of position p (p < T: target p, else edge p - T)
__device__ __forceinline__ int batch_of(const batch_view& bv, int p, int T)
{
  if (bv.target_batch == nullptr) return 0;
  return p < T ? bv.target_batch[p] : bv.sbatch()[bv.edge_row[p - T]];
}

// id of position p of  targets ++ neighbours.  The two lists may differ in width: the API's ids (targets, unique) are
// INT64 while the sampled neighbours come out of a 32-bit column array (a graph with fewer than 2^31 vertices keeps its
// columns in 32 bits behind the int64 API: half the sector footprint for the sampler, half the bytes for every pass here).
template <typename TgtT, typename NbrT>
__device__ __forceinline__ int64_t id_at(const TgtT* __restrict__ targets, const NbrT* __restrict__ neighbors, int p, int T)
{
  return p < T ? (int64_t)targets[p] : (int64_t)neighbors[p - T];
}

// thread p < T inserts target p, thread T+e inserts neighbour e; remembers its slot.
template <typename TgtT, typename NbrT, typename TableKeyT>
__global__ void __launch_bounds__(256) table_insert_kernel(const TgtT* __restrict__ targets,
                                                           dev_count T_,
                                                           const NbrT* __restrict__ neighbors,
                                                           dev_count E_,
                                                           batch_view bv,
                                                           TableKeyT* keys,
                                                           int* minpos,
                                                           int64_t capacity_slots,
                                                           int* __restrict__ slot_of)
{
  using cas_t = typename key_traits<TableKeyT>::cas_t;
  const int T = T_.get(), E = E_.get();
  const uint32_t slots = live_slot_count(T + E, capacity_slots);
  int p       = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= T + E) return;
  const int64_t id = id_at(targets, neighbors, p, T);
  TableKeyT key    = (TableKeyT)id;
  if constexpr (sizeof(TableKeyT) == 8) {
    if (bv.target_batch != nullptr) key = (TableKeyT)(((int64_t)batch_of(bv, p, T) << 40) | (int64_t)id);
  }
  uint32_t h = slot_for(hash_key((uint64_t)(int64_t)key), slots);
  while (true) {
    TableKeyT cur = keys[h];
    if (cur == (TableKeyT)-1) {
      cas_t old = atomicCAS(reinterpret_cast<cas_t*>(keys + h), (cas_t)(TableKeyT)-1, (cas_t)key);
      cur       = (TableKeyT)old;
      if (cur == (TableKeyT)-1) cur = key;  // we own the slot now
    }
    if (cur == key) break;
    h = h + 1 == slots ? 0u : h + 1;
  }
  atomicMin(minpos + h, p);
  slot_of[p] = (int)h;
}

// flag[e] = 1 iff neighbour e is the first occurrence of an id that is not a target
__global__ void __launch_bounds__(256)
first_flag_kernel(const int* __restrict__ minpos, int* __restrict__ slot_of, dev_count T_, dev_count E_,
                  int* __restrict__ flag)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int E = E_.get();
  // the scan reads whole tiles: zero-fill the slack of the tile that holds the live end, skip the rest
  if (e >= E_.host || e >= (E / kScanTile + 1) * kScanTile) return;
  const int T = T_.get();
  int first = -1;
  if (e < E) {
    // the only random read of the table after the insert: remember the id's first position in place of the
    // slot number, so the emit kernel reads it coalesced
    first          = minpos[slot_of[T + e]];
    slot_of[T + e] = first;
  }
  flag[e] = (first == T + e) ? 1 : 0;
}

template <typename TgtT, typename NbrT>
__global__ void __launch_bounds__(256) renumber_emit_kernel(const TgtT* __restrict__ targets,
                                                            const NbrT* __restrict__ neighbors,
                                                            const int* __restrict__ minpos,
                                                            const int* __restrict__ slot_of,
                                                            const int* __restrict__ rank,  // exclusive scan of flag, [E.host+1]
                                                            dev_count T_,
                                                            dev_count E_,
                                                            batch_view bv,
                                                            TgtT* __restrict__ unique_out,
                                                            int* __restrict__ map_out,
                                                            int* __restrict__ counts_out)
{
  using KeyT = TgtT;
  const int T = T_.get(), E = E_.get();
  const int U = rank[E_.host];  // slack flags are 0, so the grand total sits at the capacity end
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0 && counts_out) {
    counts_out[0] = E;
    counts_out[1] = T + U;
  }
  const bool batched = bv.target_batch != nullptr;
  if (batched && bv.unique_seg && p <= bv.G) {
    // first unique row of batch p (p == G: one past the end)
    const int new_before = rank[bv.edge_offsets[bv.sseg()[p]]];   // new vertices of batches < p
    bv.unique_seg[p] = bv.target_seg[p] + new_before;
    if (bv.frontier_seg_out) bv.frontier_seg_out[p] = new_before;
    if (bv.frontier_local0_out && p < bv.G) bv.frontier_local0_out[p] = bv.target_seg[p + 1] - bv.target_seg[p];
  }
  // no-sync walk: pad the capacity slack of `unique` with -1 so that a capacity-sized feature
  // gather skips those rows (negative index => row untouched)
  if (counts_out && p >= T + U && p < T_.host + E_.host) unique_out[p] = (KeyT)-1;
  if (p >= T + E) return;
  const int b = batch_of(bv, p, T);
  // rows contributed by the new nodes of earlier batches / first row after my batch's targets
  const int shift    = batched ? rank[bv.edge_offsets[bv.sseg()[b]]] : 0;
  const int tail_row = batched ? bv.target_seg[b + 1] : T;
  if (p < T) {
    unique_out[p + shift] = targets[p];
    if (bv.unique_batch) bv.unique_batch[p + shift] = b;
    return;
  }
  const int e     = p - T;
  const int first = slot_of[p];  // first position of the id in targets ++ neighbours (memoised by the flag kernel)
  const int row   = first < T ? first + shift : tail_row + rank[first - T];
  if (first == p) {
    unique_out[row] = (KeyT)neighbors[e];
    if (bv.unique_batch) bv.unique_batch[row] = b;
    if (bv.frontier_out) {  // next frontier, ordered by (batch, first appearance) == by rank
      static_cast<KeyT*>(bv.frontier_out)[rank[e]] = (KeyT)neighbors[e];
      bv.frontier_batch_out[rank[e]]               = b;
    }
  }
  if (map_out) map_out[e] = row;
  if (bv.neighbor_local_out) bv.neighbor_local_out[e] = row - ((batched ? bv.target_seg[b] : 0) + shift);
  if (bv.center_local_out) {
    const int r = bv.edge_row[e];
    bv.center_local_out[e] = batched ? r - bv.sseg()[b] + (bv.sample_local0 ? bv.sample_local0[b] : 0) : r;
  }
}

template <typename KeyT>
void append_unique_impl(const KeyT* targets, int T, const KeyT* neighbors, int E, void* unique_ctx, int* map_out,
                        wholememory_env_func_t* env, hipStream_t stream)
{
  const int P         = T + E;
  const int64_t slots = append_unique_slots(P);
  // one scratch block from the caller's allocator instead of five (each request is four callbacks in the torch binding)
  temp_arena arena(env);
  const size_t o_keys = arena.add(sizeof(KeyT) * (size_t)slots), o_pos = arena.add(sizeof(int) * (size_t)slots),
               o_slot = arena.add(sizeof(int) * (size_t)P), o_rank = arena.add(sizeof(int) * ((size_t)E + 1)),
               o_tmp = arena.add(sizeof(int) * (size_t)scan_tmp_ints(E + 1));
  arena.commit();
  KeyT* keys   = arena.at<KeyT>(o_keys);
  int* minpos  = arena.at<int>(o_pos);
  int* slot_of = arena.at<int>(o_slot);
  int* rank    = arena.at<int>(o_rank);
  int* stmp    = arena.at<int>(o_tmp);
  const bool k64 = sizeof(KeyT) == 8;
  dev_count Tc{T, nullptr}, Ec{E, nullptr};
  batch_view one{};
  one.G = 1;

  append_unique_prepare_enqueue(targets, Tc, k64, neighbors, Ec, k64, one, keys, minpos, slots, slot_of, rank, stmp, stream);
  int U = 0;
  WG_HIP_CHECK(hipMemcpyAsync(&U, rank + E, sizeof(int), hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // output size

  KeyT* unique_out = static_cast<KeyT*>(output_alloc(env, unique_ctx, (int64_t)T + U, key_traits<KeyT>::dt));
  append_unique_emit_enqueue(targets, Tc, k64, neighbors, Ec, k64, one, minpos, slot_of, rank, slots, unique_out, map_out, nullptr,
                             stream);
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // scratch is released on return
}

__global__ void __launch_bounds__(256)
add_self_loop_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col, int rows, int* __restrict__ out_row_ptr,
                     int* __restrict__ out_col)
{
  // one wave per row
  int row  = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  int s = row_ptr[row], e = row_ptr[row + 1];
  if (lane == 0) {
    out_row_ptr[row] = s + row;
    if (row == rows - 1) out_row_ptr[rows] = e + rows;
  }
  for (int j = lane; j <= e - s; j += 64) out_col[s + row + j] = j == 0 ? row : col[s + j - 1];
}

// ---- packed table (call groups with a known id bound) ----------------------------------------------------------
// One 64-bit word per slot = [ batch | id | first position ], empty = all ones.  Inserting needs ONE memory-side atomic for
// a first occurrence (CAS empty -> word) and usually NONE for a repeat (positions are inserted roughly in increasing order,
// so the word read while probing already holds a smaller position; otherwise one atomicMin on the word, which keeps the
// key bits because they are equal) — against CAS + atomicMin on two arrays per key for the general table — and the slot is
// 8 instead of 12 bytes.
struct packed_layout {
  int pos_bits, id_bits;
  __host__ __device__ unsigned long long pos_mask() const { return (1ull << pos_bits) - 1ull; }
};

inline int bits_for(uint64_t v)  // smallest b with v < 2^b
{
  int b = 0;
  while (b < 64 && (v >> b) != 0) b++;
  return b;
}

inline bool packed_layout_for(int64_t capacity_positions, int G, int64_t id_bound, packed_layout& out)
{
  if (id_bound <= 0 || G < 1) return false;
  out.pos_bits = bits_for((uint64_t)capacity_positions);      // positions are < capacity, so never all ones
  out.id_bits  = bits_for((uint64_t)(id_bound - 1));
  const int batch_bits = bits_for((uint64_t)(G - 1));
  return out.pos_bits + out.id_bits + batch_bits <= 63;
}

__global__ void __launch_bounds__(256)
table_clear_packed_kernel(unsigned long long* table, int64_t capacity_slots, dev_count T_, dev_count E_)
{
  const int64_t slots  = (int64_t)live_slot_count(T_.get() + E_.get(), capacity_slots);
  const int64_t tid    = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int4 fill      = make_int4(-1, -1, -1, -1);
  int4* t4             = reinterpret_cast<int4*>(table);
  for (int64_t i = tid; i < (slots + 1) / 2; i += stride) t4[i] = fill;
}

template <typename TgtT, typename NbrT>
__global__ void __launch_bounds__(256)
table_insert_packed_kernel(const TgtT* __restrict__ targets, dev_count T_, const NbrT* __restrict__ neighbors, dev_count E_,
                           batch_view bv, unsigned long long* table, int64_t capacity_slots, packed_layout lay,
                           int* __restrict__ slot_of)
{
  const int T = T_.get(), E = E_.get();
  const uint32_t slots = live_slot_count(T + E, capacity_slots);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= T + E) return;
  const int64_t id = id_at(targets, neighbors, p, T);
  const unsigned long long key  = ((unsigned long long)batch_of(bv, p, T) << lay.id_bits) | (unsigned long long)id;
  const unsigned long long word = (key << lay.pos_bits) | (unsigned long long)p;
  const unsigned long long kEmpty = ~0ull;
  uint32_t h = slot_for(hash_key(key), slots);
  while (true) {
    // device-scope load: the slot is only ever changed by memory-side atomics, a stale L2 line must not be trusted
    unsigned long long cur = __hip_atomic_load(table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kEmpty) {
      cur = atomicCAS(table + h, kEmpty, word);
      if (cur == kEmpty) break;  // the slot is ours, with our position
    }
    if ((cur >> lay.pos_bits) == key) {
      if (word < cur) atomicMin(table + h, word);  // same key bits: the minimum is the smaller position
      break;
    }
    h = h + 1 == slots ? 0u : h + 1;
  }
  slot_of[p] = (int)h;
}

__global__ void __launch_bounds__(256)
first_flag_packed_kernel(const unsigned long long* __restrict__ table, int* __restrict__ slot_of, dev_count T_, dev_count E_,
                         packed_layout lay, int* __restrict__ flag)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int E = E_.get();
  if (e >= E_.host || e >= (E / kScanTile + 1) * kScanTile) return;
  const int T = T_.get();
  int first = -1;
  if (e < E) {
    first = (int)(__hip_atomic_load(table + slot_of[T + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & lay.pos_mask());
    slot_of[T + e] = first;
  }
  flag[e] = (first == T + e) ? 1 : 0;
}

template <typename TgtT, typename NbrT>
void prepare_packed_t(const TgtT* targets, dev_count T, const NbrT* neighbors, dev_count E, batch_view bv, void* keys,
                      int64_t slots, packed_layout lay, int* slot_of, int* rank, int* scan_tmp, hipStream_t stream)
{
  auto* table = static_cast<unsigned long long*>(keys);
  const int P = T.host + E.host;
  table_clear_packed_kernel<<<(int)std::min<int64_t>(ceil_div(slots, 1024), 4096), 256, 0, stream>>>(table, slots, T, E);
  if (P > 0)
    table_insert_packed_kernel<TgtT, NbrT><<<ceil_div(P, 256), 256, 0, stream>>>(targets, T, neighbors, E, bv, table, slots, lay,
                                                                                 slot_of);
  if (E.host > 0) first_flag_packed_kernel<<<ceil_div(E.host, 256), 256, 0, stream>>>(table, slot_of, T, E, lay, rank);
  WG_HIP_CHECK(hipGetLastError());
  exclusive_scan_i32(rank, rank, E.host, scan_tmp, stream, E.dev);
}

// ---- call groups, per-batch tables in LDS ----------------------------------------------------------------------------
// A mini-batch is renumbered on its own, so its table never has to be shared across the device — and a device-wide table
// costs one memory-side atomic per new key plus one fabric-side load per probe (gfx950: agent-scope atomics and sc1 loads
// are resolved beyond the XCD's L2), 24 G keys/s on an MI355X however it is laid out.  Instead, two kernels:
//   1. bucket_sort    block (batch, chunk): the chunk's (id, position) pairs SORTED BY HASH RANGE inside the chunk's own
//                     stretch of the pair array (R = positions / 3,000 ranges per batch: what one LDS table holds at load
//                     <= 0.3).  Histogram over the chunk, block-wide exclusive scan, scatter — the second read of the chunk's
//                     ids is an L2 hit (22-44 KB read a microsecond earlier by the same workgroup).  A pair is ONE 64-bit
//                     word [ id | position ] — the word the LDS table stores — and because every chunk sorts into its own
//                     stretch no block needs another block's counts: round 3 took a count kernel, a scatter kernel (the ids
//                     read from HBM twice) and 12-byte pairs for the same job.  Per chunk the range boundaries go to seg_off.
//   2. renumber_lds   workgroup (batch, range): the range's pairs — one segment per chunk — go into a 10,000-slot
//                     open-addressing table in LDS (ds_cmpst / ds_min on the pair word itself), then the range's neighbours
//                     look their first positions up.  The pairs stay IN REGISTERS between the two passes (one trip of
//                     kLdsUnroll pairs per thread covers a range; longer ranges reload).  No table in global memory,
//                     nothing to clear, no global atomics.  A range whose DISTINCT ids overfill the table (a hot id
//                     repeated is one key) is split in two and redone through a hash filter, so any id distribution
//                     terminates.
// Cost is linear in the positions whatever the batch size.  Scratch: the pair words live where the device-wide table would
// (8 B x slots >= 16 B per position), the segment boundaries where its positions array would.
#ifndef WG_LDS_SLOTS      // (tuning: -DWG_LDS_SLOTS / WG_LDS_KEYS / WG_LDS_THREADS / WG_LDS_UNROLL)
#define WG_LDS_SLOTS 10000
#define WG_LDS_KEYS 3000
#define WG_LDS_THREADS 512
#define WG_LDS_UNROLL 7
#endif
constexpr int kLdsSlots      = WG_LDS_SLOTS;     // 80,000 B: two workgroups per CU, one computes while the other waits on its loads
constexpr int kLdsKeysTarget = WG_LDS_KEYS;      // positions per range the range count is sized for (load <= 0.3: short probe chains)
constexpr int kLdsProbeLimit = 256;       // probes after which a range is declared overfull and split
constexpr int kLdsThreads    = WG_LDS_THREADS;   // one trip of kLdsUnroll pairs per thread covers a range (1024 threads: walk 1.227 -> 1.19 ms per call
                                          // group of 191; half / quarter-size tables with 512 / 256 threads: 1.28 / 1.48 — more ranges cost more to bucket)
constexpr int kLdsMaxRanges  = 2048;      // per batch; beyond, ranges simply start overfull and split
constexpr int kLdsStack      = 40;        // pending hash ranges of one workgroup (a split pushes two, pops one)
constexpr int kLdsChunks     = 16;        // blocks per batch in the bucketing kernel = segments of a range
constexpr int kBucketThreads = 256;
constexpr int kBucketUnroll  = 4;         // ids in flight per thread of the bucketing kernel
constexpr int kLdsUnroll     = WG_LDS_UNROLL;    // pairs per thread of the table kernel: 3,584 per trip (a range holds 3,000 on average)

__device__ __forceinline__ uint32_t hash_id32(uint32_t h)   // murmur3 finaliser
{
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  return h ^ (h >> 16);
}
// ID32: every id of the call fits 32 bits (the neighbours come from a 32-bit column array)
template <bool ID32>
__device__ __forceinline__ uint32_t hash_id(int64_t id)
{
  if constexpr (ID32) return hash_id32((uint32_t)id);
  else return hash_key((uint64_t)id);
}

// one mini-batch of the call group: its targets [t0, t0 + nT), its neighbours [e0, e0 + nE), its R hash ranges, where its
// boundary records start (rb: every batch gets floor(positions before it / kLdsKeysTarget) + 2 b, which leaves room for the
// R + 1 rows of the batches before it) and the positions one bucketing block takes
struct batch_part {
  int t0, nT, e0, nE, P, R, rb, chunk;
  // keys_target: kLdsKeysTarget, or the (larger) value of a test that wants ranges to overfill and split
  __device__ batch_part(const batch_view& bv, int b, int keys_target)
  {
    t0 = bv.target_seg[b];
    nT = bv.target_seg[b + 1] - t0;
    e0 = bv.edge_offsets[bv.sseg()[b]];
    nE = bv.edge_offsets[bv.sseg()[b + 1]] - e0;
    P  = nT + nE;
    R  = min(max((P + keys_target - 1) / keys_target, 1), kLdsMaxRanges);
    rb = (t0 + e0) / keys_target + 2 * b;
    chunk = (P + kLdsChunks - 1) / kLdsChunks;
  }
};

// Workgroup -> (batch, part).  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), and every kernel below
// writes a batch's stretch of some array in scattered 4-byte pieces: all parts of one batch go to ONE XCD, next to each
// other in dispatch order, so that the pieces meet in that XCD's L2 and leave it as whole lines (performance only — the
// result does not depend on where a workgroup runs).  Grid = 8 * ceil(G / 8) * parts; returns false for the padding.
__device__ __forceinline__ bool batch_of_block(int G, int parts, int& b, int& part)
{
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
  b    = (y / parts) * 8 + x;
  part = y % parts;
  return b < G;
}
inline int batch_grid(int G, int parts) { return 8 * ((G + 7) / 8) * parts; }

// scratch behind the two kernels
struct sort_scratch {
  int keys_target;
  int* seg_off;                 // [(rb + r) * kLdsChunks + c], r <= R: first pair of range r inside chunk c's stretch
  unsigned long long* words;    // [capacity positions]: batch b's chunk c owns [t0 + e0 + c * chunk, ...), sorted by range
};

template <typename TgtT, typename NbrT>
__global__ void __launch_bounds__(kBucketThreads)
bucket_sort_kernel(const TgtT* __restrict__ targets, dev_count T_, const NbrT* __restrict__ neighbors, batch_view bv,
                   packed_layout lay, sort_scratch sc)
{
  constexpr bool ID32  = sizeof(NbrT) == 4;
  constexpr int kItems = kLdsMaxRanges / kBucketThreads;   // histogram entries a thread scans
  __shared__ int hist[kLdsMaxRanges];   // counts of the chunk per range, then the write cursors
  __shared__ int wave_tot[kBucketThreads / 64];
  int b, c;
  if (!batch_of_block(bv.G, kLdsChunks, b, c)) return;
  const batch_part bp(bv, b, sc.keys_target);
  if (bp.nE <= 0) return;
  const int T = T_.get(), tid = threadIdx.x;
  for (int r = tid; r < bp.R; r += kBucketThreads) hist[r] = 0;
  __syncthreads();
  const int begin = min(c * bp.chunk, bp.P), end = min(bp.P, begin + bp.chunk);   // (a tiny batch leaves the last chunks empty)
  const TgtT* tg = targets + bp.t0;
  const NbrT* nb = neighbors + bp.e0 - bp.nT;   // neighbour of batch position i >= nT: nb[i]
  // ---- pass 1: the chunk's histogram over the batch's R hash ranges ------------------------------------------------
  for (int i0 = begin + tid; i0 < end; i0 += kBucketThreads * kBucketUnroll) {
    int64_t id[kBucketUnroll];   // all loads of the trip are in flight before the first histogram update
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++) {
      const int i = min(i0 + k * kBucketThreads, end - 1);
      id[k]       = i < bp.nT ? (int64_t)tg[i] : (int64_t)nb[i];
    }
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++)
      if (i0 + k * kBucketThreads < end) atomicAdd(&hist[__umulhi(hash_id<ID32>(id[k]), (uint32_t)bp.R)], 1);
  }
  __syncthreads();
  // ---- exclusive scan of the histogram (thread t owns entries [t * kItems, (t + 1) * kItems)) -----------------------
  {
    int v[kItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
      const int r = tid * kItems + k;
      v[k]        = r < bp.R ? hist[r] : 0;
      sum += v[k];
    }
    int inc = sum;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(inc, d, 64);
      if (lane >= d) inc += up;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; w++) run += wave_tot[w];
    int* bounds = sc.seg_off + (int64_t)bp.rb * kLdsChunks + c;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
      const int r = tid * kItems + k;
      if (r < bp.R) {
        hist[r]                         = run;   // from here on: where the next pair of range r goes
        bounds[(int64_t)r * kLdsChunks] = run;
      }
      run += v[k];
    }
    if (tid == 0) bounds[(int64_t)bp.R * kLdsChunks] = end - begin;
  }
  __syncthreads();
  // ---- pass 2: the pairs, grouped by range, into the chunk's own stretch (ids re-read: L2 hits) ----------------------
  unsigned long long* out = sc.words + (bp.t0 + bp.e0) + begin;
  for (int i0 = begin + tid; i0 < end; i0 += kBucketThreads * kBucketUnroll) {
    int64_t id[kBucketUnroll];
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++) {
      const int i = min(i0 + k * kBucketThreads, end - 1);
      id[k]       = i < bp.nT ? (int64_t)tg[i] : (int64_t)nb[i];
    }
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++) {
      const int i = i0 + k * kBucketThreads;
      if (i < end) {
        const int dst = atomicAdd(&hist[__umulhi(hash_id<ID32>(id[k]), (uint32_t)bp.R)], 1);
        const int pos = i < bp.nT ? bp.t0 + i : T + bp.e0 + (i - bp.nT);
        out[dst]      = ((unsigned long long)id[k] << lay.pos_bits) | (unsigned long long)pos;
      }
    }
  }
}

template <bool ID32>
__global__ void __launch_bounds__(kLdsThreads)
renumber_lds_kernel(dev_count T_, dev_count E_, batch_view bv, packed_layout lay, int wg_per_batch, sort_scratch sc,
                    int* __restrict__ slot_of, int* __restrict__ tile_sums, int n_tile_sums)
{
  __shared__ unsigned long long tbl[kLdsSlots];
  __shared__ unsigned long long st_lo[kLdsStack], st_span[kLdsStack];
  __shared__ int st_n, overfull;
  __shared__ int seg_base[kLdsChunks], seg_pre[kLdsChunks + 1];
  const unsigned long long kEmpty = ~0ull;
  const int T = T_.get();
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    // first_bits_kernel (the next launch) adds its popcounts into these with atomics: cleared here, one launch earlier
    for (int i = tid; i < n_tile_sums; i += kLdsThreads) tile_sums[i] = 0;
  }
  int b, j;
  if (!batch_of_block(bv.G, wg_per_batch, b, j)) return;
  const batch_part bp(bv, b, sc.keys_target);
  if (bp.nE <= 0) return;   // no neighbour needs a first position
  const unsigned long long pos_mask = lay.pos_mask();
  const volatile int* overfull_now  = &overfull;
  static_assert(kLdsChunks == 16, "the segment search below is written for 16 segments");

  for (int r = j; r < bp.R; r += wg_per_batch) {
    __syncthreads();   // the previous range's last look at the stack and the segment table is over
    if (tid < kLdsChunks) {
      // range r of the batch = one segment per bucketing chunk
      const int* row = sc.seg_off + (int64_t)(bp.rb + r) * kLdsChunks;
      const int lo = row[tid], n = row[kLdsChunks + tid] - lo;
      seg_base[tid] = bp.t0 + bp.e0 + tid * bp.chunk + lo;
      int inc = n;
#pragma unroll
      for (int d = 1; d < kLdsChunks; d <<= 1) {
        const int up = __shfl_up(inc, d, 64);
        if (tid >= d) inc += up;
      }
      seg_pre[tid + 1] = inc;
      if (tid == 0) {
        seg_pre[0] = 0;
        // mulhi(h, R) == r  <=>  h in [ceil(r 2^32 / R), ceil((r + 1) 2^32 / R))
        const unsigned long long lo_h = (((unsigned long long)r << 32) + bp.R - 1) / (unsigned)bp.R;
        st_lo[0]   = lo_h;
        st_span[0] = ((((unsigned long long)(r + 1) << 32) + bp.R - 1) / (unsigned)bp.R) - lo_h;
        st_n       = 1;
      }
    }
    __syncthreads();
    const int n_pairs = seg_pre[kLdsChunks];
    if (n_pairs == 0) continue;   // (the same value in every thread)
    // pair i of the range: the segment holding it by binary search over the 17 prefix values
    auto pair_at = [&](int i) -> unsigned long long {
      if (i >= n_pairs) return kEmpty;
      int c = i >= seg_pre[8] ? 8 : 0;
      c += i >= seg_pre[c + 4] ? 4 : 0;
      c += i >= seg_pre[c + 2] ? 2 : 0;
      c += i >= seg_pre[c + 1] ? 1 : 0;
      return sc.words[seg_base[c] + (i - seg_pre[c])];
    };
    // one trip of kLdsUnroll pairs per thread are in flight at once (one load at a time leaves a 8-wave workgroup waiting
    // on memory latency); a range that fits one trip — nearly all — keeps its pairs in registers for both passes
    const bool one_trip = n_pairs <= kLdsThreads * kLdsUnroll;
    unsigned long long w[kLdsUnroll];
    int memo[kLdsUnroll];   // one-trip ranges: the slot each pair's id ended up in (insert pass -> look-up pass)
    if (one_trip) {
#pragma unroll
      for (int k = 0; k < kLdsUnroll; k++) w[k] = pair_at(k * kLdsThreads + tid);
    }
    while (true) {
      const int n = st_n;
      if (n == 0) break;
      const unsigned long long lo = st_lo[n - 1], span = st_span[n - 1];
      __syncthreads();   // everyone has read the top of the stack
      {
        uint4* t4 = reinterpret_cast<uint4*>(tbl);
        const uint4 fill = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (int i = tid; i < kLdsSlots / 2; i += kLdsThreads) t4[i] = fill;
        if (tid == 0) {
          st_n     = n - 1;
          overfull = 0;
        }
      }
      __syncthreads();
      // ---- insert the pairs of the range (all of them unless the range was split) -----------------------------------
      for (int i0 = 0; i0 < n_pairs; i0 += kLdsThreads * kLdsUnroll) {
        if (!one_trip) {
#pragma unroll
          for (int k = 0; k < kLdsUnroll; k++) w[k] = pair_at(i0 + k * kLdsThreads + tid);
        }
#pragma unroll
        for (int k = 0; k < kLdsUnroll; k++) {
          const unsigned long long word = w[k];
          const unsigned long long id   = word >> lay.pos_bits;
          const uint32_t h              = hash_id<ID32>((int64_t)id);
          memo[k]                       = -1;
          if (word != kEmpty && (unsigned long long)h - lo < span && !*overfull_now) {
            uint32_t s = __umulhi(h * 0x9E3779B1u, (uint32_t)kLdsSlots);
            int probes = 0;
            while (true) {
              // one LDS round trip per probe: the compare-and-swap doubles as the read
              const unsigned long long cur = atomicCAS(&tbl[s], kEmpty, word);
              if (cur == kEmpty) break;
              if ((cur >> lay.pos_bits) == id) {
                if (word < cur) atomicMin(&tbl[s], word);
                break;
              }
              s = s + 1 == (uint32_t)kLdsSlots ? 0u : s + 1;
              if (++probes > kLdsProbeLimit) {   // never at the load a range is sized for: the table is (nearly) full
                overfull = 1;
                break;
              }
            }
            memo[k] = (int)s;   // the slot of this id (meaningless when the range overfilled: it is redone)
          }
        }
      }
      __syncthreads();
      if (overfull) {
        // split the range and redo both halves (a single hash value never holds a table full of distinct ids)
        if (tid == 0) {
          const unsigned long long half = span / 2;
          const int m = st_n;
          if (span < 2 || m + 2 > kLdsStack) __builtin_trap();
          st_lo[m]       = lo + half;
          st_span[m]     = span - half;
          st_lo[m + 1]   = lo;
          st_span[m + 1] = half;
          st_n           = m + 2;
        }
        __syncthreads();
        continue;
      }
      // ---- first position of every neighbour of the range ----------------------------------------------------------
      if (one_trip) {
        // the insert pass left every pair's slot in a register: one straight LDS read per neighbour, no hash, no probe loop
#pragma unroll
        for (int k = 0; k < kLdsUnroll; k++) {
          const int p = (int)(w[k] & pos_mask);
          if (memo[k] >= 0 && p >= T) slot_of[p] = (int)(tbl[memo[k]] & pos_mask);
        }
        __syncthreads();
        continue;
      }
      for (int i0 = 0; i0 < n_pairs; i0 += kLdsThreads * kLdsUnroll) {
        if (!one_trip) {
#pragma unroll
          for (int k = 0; k < kLdsUnroll; k++) w[k] = pair_at(i0 + k * kLdsThreads + tid);
        }
#pragma unroll
        for (int k = 0; k < kLdsUnroll; k++) {
          const unsigned long long word = w[k];
          const unsigned long long id   = word >> lay.pos_bits;
          const int p                   = (int)(word & pos_mask);
          const uint32_t h              = hash_id<ID32>((int64_t)id);
          // (a target: nobody asks for its first position)
          if (word != kEmpty && p >= T && (unsigned long long)h - lo < span) {
            uint32_t s = __umulhi(h * 0x9E3779B1u, (uint32_t)kLdsSlots);
            unsigned long long cur = tbl[s];
            while ((cur >> lay.pos_bits) != id) {
              s   = s + 1 == (uint32_t)kLdsSlots ? 0u : s + 1;
              cur = tbl[s];
            }
            slot_of[p] = (int)(cur & pos_mask);
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---- first-appearance ranks --------------------------------------------------------------------------------------------
// rank(q) = number of first occurrences among the neighbours before position q (q <= E).  Device-wide table paths keep it as
// a plain int array (flags -> exclusive scan).  The LDS path keeps ONE BIT per neighbour and a running count per 64-bit word:
// first_bits_kernel derives the bits from slot_of (neighbour e is a first occurrence iff slot_of[T + e] == T + e — the table
// kernel no longer writes a flag array), the scan runs over E / 64 word counts instead of E flags, and the emit kernel's
// random rank look-up lands in 12 bytes per 64 neighbours (2.4 MB for the 12.6 M neighbours of a products hop 2: it stays in
// L2) instead of a 4-byte entry per neighbour.  Hop 2 of the products call group: flag write 50 MB, scan 150 MB and the rank
// gather of the emit kernel are gone.
struct rank_array {
  const int* rank;
  __device__ __forceinline__ int at(int q) const { return rank[q]; }
  // slack flags are 0, so the grand total sits at the capacity end (where the scan publishes it)
  __device__ __forceinline__ int total(int /*e_live*/, int e_host) const { return rank[e_host]; }
};
struct rank_bits {
  const int* prefix;                  // [E.host / 64 + 2]: exclusive running count per word (+ the grand total behind the scan)
  const unsigned long long* words;    // [E.host / 64 + 1]: bit (e & 63) of word e >> 6 = neighbour e is a first occurrence
  __device__ __forceinline__ int at(int q) const
  {
    const int w = q >> 6;
    return prefix[w] + __popcll(words[w] & ((1ull << (q & 63)) - 1ull));
  }
  // only the live words are written: the total is the rank of the live end
  __device__ __forceinline__ int total(int e_live, int /*e_host*/) const { return at(e_live); }
};
// where the three pieces sit inside the hop's `rank` buffer of E.host + 1 ints
struct rank_bits_layout {
  int nw;              // words for the capacity: E.host / 64 + 1 (position E itself is looked up: the total)
  int words_at;        // int offset of the words (16-byte aligned)
  int live_at;         // int offset of one int: the number of LIVE words (device-side bound of the scan)
  __host__ __device__ explicit rank_bits_layout(int e_host)
  {
    nw       = e_host / 64 + 1;
    words_at = (nw + 1 + 3) / 4 * 4;
    live_at  = words_at + 2 * nw;
  }
  __host__ __device__ int ints() const { return live_at + 1; }
};
constexpr int kBitsWordsPerBlock = 64;   // 4,096 neighbours per block iteration: one atomic per iteration into the tile sums

__global__ void __launch_bounds__(256)
first_bits_kernel(const int* __restrict__ slot_of, dev_count T_, dev_count E_, int* __restrict__ rank_buf, int* __restrict__ tile_sums)
{
  __shared__ int wave_cnt[4];
  const int T = T_.get(), E = E_.get();
  const rank_bits_layout lay(E_.host);
  int* cnt                  = rank_buf;
  unsigned long long* words = reinterpret_cast<unsigned long long*>(rank_buf + lay.words_at);
  const int nw_live         = E / 64 + 1;
  // the scan reads whole tiles of word counts: the slack of the tile that holds the live end is zero-filled
  const int nw_end = min(lay.nw, (nw_live / kScanTile + 1) * kScanTile);
  if (blockIdx.x == 0 && threadIdx.x == 0) rank_buf[lay.live_at] = nw_live;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w0 = blockIdx.x * kBitsWordsPerBlock; w0 < nw_end; w0 += gridDim.x * kBitsWordsPerBlock) {
    // wave v takes words w0 + v, w0 + v + 4, ...: a block iteration reads 16 KB of slot_of, all sixteen loads of a lane in
    // flight before the first ballot
    bool f[kBitsWordsPerBlock / 4];
    int first[kBitsWordsPerBlock / 4];
#pragma unroll
    for (int k = 0; k < kBitsWordsPerBlock / 4; k++)   // unconditional loads (index clamped): all sixteen in flight at once
      first[k] = slot_of[max(T + min((w0 + wave + 4 * k) * 64 + lane, E - 1), 0)];
#pragma unroll
    for (int k = 0; k < kBitsWordsPerBlock / 4; k++) {
      const int e = (w0 + wave + 4 * k) * 64 + lane;
      f[k]        = e < E && first[k] == T + e;
    }
    int total = 0;
#pragma unroll
    for (int k = 0; k < kBitsWordsPerBlock / 4; k++) {
      const int w                    = w0 + wave + 4 * k;
      const unsigned long long word  = __ballot(f[k]);
      const int c                    = __popcll(word);
      total += c;
      if (lane == 0 && w < nw_end) {
        words[w] = word;
        cnt[w]   = c;
      }
    }
    // all words of a block iteration lie in ONE scan tile (kScanTile is a multiple of kBitsWordsPerBlock)
    if (lane == 0) wave_cnt[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      if (t) atomicAdd(&tile_sums[w0 / kScanTile], t);
    }
    __syncthreads();
  }
}
static_assert(kScanTile % kBitsWordsPerBlock == 0, "a block iteration of first_bits_kernel must stay inside one scan tile");

// Emit for call groups, one block per (batch, chunk): everything that depends on the batch only (its row shift, where
// its new vertices start, its local-id origin) is read once per block instead of chased through four dependent loads per
// edge, and the one random read left (the rank of the first occurrence) stays inside the batch's stretch of `rank`,
// which the XCD-affine block mapping keeps in one L2.  Same outputs as renumber_emit_kernel.  The unique / frontier lists
// are written in the TARGET (API) id type whatever the width of the sampled neighbours.
constexpr int kEmitChunks = 16;
constexpr int kEmitUnroll = 4;

template <typename TgtT, typename NbrT, typename RankT>
__global__ void __launch_bounds__(256)
renumber_emit_batched_kernel(const TgtT* __restrict__ targets, const NbrT* __restrict__ neighbors,
                             const int* __restrict__ slot_of, const RankT rank, dev_count T_, dev_count E_,
                             batch_view bv, TgtT* __restrict__ unique_out, int* __restrict__ map_out,
                             int* __restrict__ counts_out)
{
  using KeyT = TgtT;
  const int T = T_.get(), E = E_.get();
  const int U = rank.total(E, E_.host);
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsize = gridDim.x * blockDim.x;
  if (gtid == 0 && counts_out) {
    counts_out[0] = E;
    counts_out[1] = T + U;
  }
  if (bv.unique_seg && gtid <= bv.G) {
    const int new_before = rank.at(bv.edge_offsets[bv.sseg()[gtid]]);   // new vertices of batches < gtid
    bv.unique_seg[gtid] = bv.target_seg[gtid] + new_before;
    if (bv.frontier_seg_out) bv.frontier_seg_out[gtid] = new_before;
    if (bv.frontier_local0_out && gtid < bv.G) bv.frontier_local0_out[gtid] = bv.target_seg[gtid + 1] - bv.target_seg[gtid];
  }
  // no-sync walk: the capacity slack of `unique` is padded with -1 (a capacity-sized feature gather skips those rows) unless
  // the caller reads the sizes anyway (WGAMD_HOP_NO_UNIQUE_PAD): the capacity is the worst case — every seed with fan-out^hops
  // DISTINCT neighbours —, 3-4x the live size on the products-like graph: 347 MB of -1 per hop-2 call group of 191
  if (counts_out && !bv.no_pad)
    for (int p = T + U + gtid; p < T_.host + E_.host; p += gsize) unique_out[p] = (KeyT)-1;

  int b, c;
  if (!batch_of_block(bv.G, kEmitChunks, b, c)) return;
  const int t0 = bv.target_seg[b], nT = bv.target_seg[b + 1] - t0;
  const int s0 = bv.sseg()[b];
  const int e0 = bv.edge_offsets[s0], nE = bv.edge_offsets[bv.sseg()[b + 1]] - e0;
  const int shift    = rank.at(e0);          // rows contributed by the new vertices of earlier batches
  const int tail_row = t0 + nT;           // first row after my batch's targets (before the shift... see below)
  const int local0   = bv.sample_local0 ? bv.sample_local0[b] : 0;
  {
    const int chunk = (nT + kEmitChunks - 1) / kEmitChunks;
    const int end   = min(nT, (c + 1) * chunk);
    for (int i = c * chunk + threadIdx.x; i < end; i += blockDim.x) {
      unique_out[t0 + i + shift] = targets[t0 + i];
      if (bv.unique_batch) bv.unique_batch[t0 + i + shift] = b;
    }
  }
  const int chunk = (nE + kEmitChunks - 1) / kEmitChunks;
  const int end   = min(nE, (c + 1) * chunk);
  for (int i0 = c * chunk + threadIdx.x; i0 < end; i0 += 256 * kEmitUnroll) {
    // two dependent reads per edge (first position, then its rank): kEmitUnroll edges per thread keep both in flight
    int first[kEmitUnroll], row[kEmitUnroll];
#pragma unroll
    for (int k = 0; k < kEmitUnroll; k++) first[k] = slot_of[T + e0 + min(i0 + k * 256, end - 1)];
#pragma unroll
    for (int k = 0; k < kEmitUnroll; k++) row[k] = first[k] < T ? first[k] + shift : tail_row + rank.at(first[k] - T);
#pragma unroll
    for (int k = 0; k < kEmitUnroll; k++) {
      const int i = i0 + k * 256;
      if (i >= end) break;
      const int e = e0 + i;
      if (first[k] == T + e) {
        const KeyT id      = (KeyT)neighbors[e];
        unique_out[row[k]] = id;
        if (bv.unique_batch) bv.unique_batch[row[k]] = b;
        if (bv.frontier_out) {  // next frontier, ordered by (batch, first appearance) == by rank
          const int r = row[k] - tail_row;
          static_cast<KeyT*>(bv.frontier_out)[r] = id;
          bv.frontier_batch_out[r]               = b;
        }
      }
      if (map_out) map_out[e] = row[k];
      if (bv.neighbor_local_out) bv.neighbor_local_out[e] = row[k] - (t0 + shift);
      if (bv.center_local_out) bv.center_local_out[e] = bv.edge_row[e] - s0 + local0;
    }
  }
}

// boundary records the two kernels address: every batch starts at floor(positions before it / kLdsKeysTarget) + 2 b and
// owns R + 1 rows of kLdsChunks ints
inline int64_t lds_range_records(int64_t capacity_positions, int G) { return capacity_positions / kLdsKeysTarget + 3 * (int64_t)G + 2; }
// WGAMD_RENUMBER_KEYS_TARGET (tests): positions per hash range, >= kLdsKeysTarget; a value the table cannot hold makes
// every range overfill and exercises the split path (and ranges longer than one trip of the table kernel)
inline int lds_keys_target()
{
  static const int v = [] {
    const char* e = getenv("WGAMD_RENUMBER_KEYS_TARGET");
    const long t  = e ? atol(e) : 0;
    return t > kLdsKeysTarget ? (int)std::min<long>(t, 1 << 30) : kLdsKeysTarget;
  }();
  return v;
}
inline bool lds_scratch_fits(int64_t capacity_positions, int G, int64_t slots)
{
  // keys buffer: 8 B per slot holds one pair word per position; positions buffer: 4 B per slot holds the boundary rows
  return 8 * capacity_positions + 256 <= 8 * slots && lds_range_records(capacity_positions, G) * kLdsChunks <= slots;
}

// does a call take the per-batch LDS tables?  (prepare and emit must agree: the LDS path leaves the ranks as bits)
inline bool lds_path_taken(dev_count T, dev_count E, bool nbr64, const batch_view& bv, int64_t slots, packed_layout& lay)
{
  static const bool no_lds = getenv("WGAMD_RENUMBER_NO_LDS") != nullptr;
  // 32-bit neighbours promise ids below 2^31 whatever bound the caller stated
  const int64_t bound = bv.id_bound > 0 ? bv.id_bound : (nbr64 ? 0 : ((int64_t)1 << 31));
  const bool batched  = bv.target_batch != nullptr;
  return batched && !no_lds && bv.target_seg && bv.edge_offsets && bv.G > 1 && E.host >= 64 &&
         rank_bits_layout(E.host).ints() <= E.host + 1 &&
         packed_layout_for((int64_t)T.host + E.host, 1, bound, lay) && lds_scratch_fits((int64_t)T.host + E.host, bv.G, slots);
}

template <typename TgtT, typename NbrT>
void prepare_lds_t(const TgtT* targets, dev_count T, const NbrT* neighbors, dev_count E, batch_view bv, packed_layout lay,
                   void* keys, int* minpos, int* slot_of, int* rank, int* scan_tmp, hipStream_t stream)
{
  if (E.host > 0) {
    const int64_t cap = (int64_t)T.host + E.host;
    sort_scratch sc;
    sc.keys_target = lds_keys_target();
    sc.seg_off     = minpos;
    sc.words       = static_cast<unsigned long long*>(keys);
    bucket_sort_kernel<TgtT, NbrT><<<batch_grid(bv.G, kLdsChunks), kBucketThreads, 0, stream>>>(targets, T, neighbors, bv, lay, sc);
    const int64_t per_batch = (cap + bv.G - 1) / bv.G;
    const int wg_per_batch  = (int)std::max<int64_t>(1, std::min<int64_t>((per_batch + kLdsKeysTarget - 1) / kLdsKeysTarget, 32));
    const rank_bits_layout rl(E.host);
    const int n_tiles = (int)((rl.nw + kScanTile - 1) / kScanTile);
    renumber_lds_kernel<sizeof(NbrT) == 4><<<batch_grid(bv.G, wg_per_batch), kLdsThreads, 0, stream>>>(T, E, bv, lay, wg_per_batch, sc,
                                                                                                     slot_of, scan_tmp, n_tiles);
    // first occurrences as bits + per-word counts (+ their per-tile sums), then the counts' running sums in place
    const int bits_grid = (int)std::min<int64_t>(((int64_t)rl.nw + kBitsWordsPerBlock - 1) / kBitsWordsPerBlock, 256 * 8);
    first_bits_kernel<<<bits_grid, 256, 0, stream>>>(slot_of, T, E, rank, scan_tmp);
    WG_HIP_CHECK(hipGetLastError());
    exclusive_scan_i32_presummed(rank, rank, rl.nw, scan_tmp, stream, rank + rl.live_at);
  }
}

template <typename TgtT, typename NbrT, typename TableKeyT>
void prepare_t(const TgtT* targets, dev_count T, const NbrT* neighbors, dev_count E, batch_view bv, TableKeyT* keys,
               int* minpos, int64_t slots, int* slot_of, int* rank, int* scan_tmp, hipStream_t stream)
{
  const int P = T.host + E.host;
  table_clear_kernel<TableKeyT><<<(int)std::min<int64_t>(ceil_div(slots, 1024), 4096), 256, 0, stream>>>(keys, minpos, slots, T,
                                                                                                   E);
  if (P > 0)
    table_insert_kernel<TgtT, NbrT, TableKeyT><<<ceil_div(P, 256), 256, 0, stream>>>(targets, T, neighbors, E, bv, keys,
                                                                                    minpos, slots, slot_of);
  if (E.host > 0) first_flag_kernel<<<ceil_div(E.host, 256), 256, 0, stream>>>(minpos, slot_of, T, E, rank);
  WG_HIP_CHECK(hipGetLastError());
  exclusive_scan_i32(rank, rank, E.host, scan_tmp, stream, E.dev);  // flags -> ranks, rank[E.host] = #new nodes
}

// the three id-width combinations a call can have: both lists INT64, both INT, or INT64 targets with INT neighbours
template <typename F>
void with_id_types(bool tgt64, bool nbr64, const void* targets, const void* neighbors, F&& f)
{
  if (tgt64 && nbr64) f(static_cast<const int64_t*>(targets), static_cast<const int64_t*>(neighbors));
  else if (tgt64) f(static_cast<const int64_t*>(targets), static_cast<const int32_t*>(neighbors));
  else if (!nbr64) f(static_cast<const int32_t*>(targets), static_cast<const int32_t*>(neighbors));
  else throw logic_error("INT targets with INT64 neighbours: not a combination a hop produces");
}

}  // namespace

int64_t append_unique_slots(int64_t capacity)
{
  int64_t slots = 1024;
  while (slots < 2 * capacity) slots <<= 1;
  return slots;
}

void append_unique_prepare_enqueue(const void* targets, dev_count T, bool tgt64, const void* neighbors, dev_count E, bool nbr64,
                                   batch_view bv, void* keys, int* minpos, int64_t slots, int* slot_of, int* rank,
                                   int* scan_tmp, hipStream_t stream)
{
  const bool batched = bv.target_batch != nullptr;
  packed_layout lay{};
  const int64_t bound = bv.id_bound > 0 ? bv.id_bound : (nbr64 ? 0 : ((int64_t)1 << 31));
  if (lds_path_taken(T, E, nbr64, bv, slots, lay)) {
    with_id_types(tgt64, nbr64, targets, neighbors, [&](auto* t, auto* n) {
      prepare_lds_t(t, T, n, E, bv, lay, keys, minpos, slot_of, rank, scan_tmp, stream);
    });
    return;
  }
  if (batched && packed_layout_for((int64_t)T.host + E.host, bv.G, bound, lay)) {
    with_id_types(tgt64, nbr64, targets, neighbors, [&](auto* t, auto* n) {
      prepare_packed_t(t, T, n, E, bv, keys, slots, lay, slot_of, rank, scan_tmp, stream);
    });
    return;
  }
  with_id_types(tgt64, nbr64, targets, neighbors, [&](auto* t, auto* n) {
    using TgtT = std::remove_cv_t<std::remove_pointer_t<decltype(t)>>;
    if constexpr (sizeof(TgtT) == 4) {
      if (!batched) {
        prepare_t<int32_t, int32_t, int32_t>(t, T, n, E, bv, static_cast<int32_t*>(keys), minpos, slots, slot_of, rank, scan_tmp, stream);
        return;
      }
    }
    prepare_t(t, T, n, E, bv, static_cast<int64_t*>(keys), minpos, slots, slot_of, rank, scan_tmp, stream);
  });
}

void append_unique_emit_enqueue(const void* targets, dev_count T, bool tgt64, const void* neighbors, dev_count E, bool nbr64,
                                batch_view bv, const int* minpos, const int* slot_of, const int* rank, int64_t slots,
                                void* unique_out, int* map_out, int* counts_out, hipStream_t stream)
{
  packed_layout lay_unused{};
  const bool bits = lds_path_taken(T, E, nbr64, bv, slots, lay_unused);
  const int P    = T.host + E.host;
  const int grid = ceil_div(P > bv.G + 1 ? P : bv.G + 1, 256);  // thread 0 publishes the counts, threads <= G the segments
  const bool per_batch = bv.target_batch != nullptr && bv.target_seg && bv.edge_offsets && bv.G > 1;
  const int bgrid = std::max(batch_grid(bv.G, kEmitChunks), ceil_div(bv.G + 1, 256));
  with_id_types(tgt64, nbr64, targets, neighbors, [&](auto* t, auto* n) {
    using TgtT = std::remove_cv_t<std::remove_pointer_t<decltype(t)>>;
    using NbrT = std::remove_cv_t<std::remove_pointer_t<decltype(n)>>;
    if (per_batch && bits) {
      const rank_bits_layout rl(E.host);
      const rank_bits rb{rank, reinterpret_cast<const unsigned long long*>(rank + rl.words_at)};
      renumber_emit_batched_kernel<TgtT, NbrT, rank_bits><<<bgrid, 256, 0, stream>>>(t, n, slot_of, rb, T, E, bv,
                                                                                    static_cast<TgtT*>(unique_out), map_out, counts_out);
    } else if (per_batch)
      renumber_emit_batched_kernel<TgtT, NbrT, rank_array><<<bgrid, 256, 0, stream>>>(t, n, slot_of, rank_array{rank}, T, E, bv,
                                                                                     static_cast<TgtT*>(unique_out), map_out, counts_out);
    else
      renumber_emit_kernel<TgtT, NbrT><<<grid, 256, 0, stream>>>(t, n, minpos, slot_of, rank, T, E, bv, static_cast<TgtT*>(unique_out),
                                                                map_out, counts_out);
  });
  WG_HIP_CHECK(hipGetLastError());
}

}  // namespace wgamd

extern "C" {

wholememory_error_code_t graph_append_unique(wholememory_tensor_t target_nodes_tensor,
                                             wholememory_tensor_t neighbor_nodes_tensor,
                                             void* output_unique_node_memory_context,
                                             wholememory_tensor_t output_neighbor_raw_to_unique_mapping_tensor,
                                             wholememory_env_func_t* p_env_fns, void* stream)
{
  using namespace wgamd;
  return guarded("graph_append_unique", [&] {
    WG_REQUIRE_INPUT(target_nodes_tensor && neighbor_nodes_tensor && p_env_fns, "null tensor / env");
    WG_REQUIRE_INPUT(output_unique_node_memory_context != nullptr, "output_unique_node_memory_context is NULL");
    auto td = target_nodes_tensor->desc, nd = neighbor_nodes_tensor->desc;
    WG_REQUIRE_INPUT(td.dim == 1 && nd.dim == 1, "target / neighbor tensors must be 1-D");
    WG_REQUIRE_INPUT(td.dtype == nd.dtype, "target and neighbor dtypes differ");
    WG_REQUIRE_INPUT(td.dtype == WHOLEMEMORY_DT_INT || td.dtype == WHOLEMEMORY_DT_INT64, "node ids must be INT|INT64");
    WG_REQUIRE_INPUT(td.sizes[0] + nd.sizes[0] < ((int64_t)1 << 30), "too many nodes for one call");
    int* map_out = nullptr;
    auto mt      = output_neighbor_raw_to_unique_mapping_tensor;
    if (mt != nullptr && mt->desc.dim != 0 && tensor_data(mt) != nullptr) {
      WG_REQUIRE_INPUT(mt->desc.dim == 1 && mt->desc.dtype == WHOLEMEMORY_DT_INT, "mapping tensor must be 1-D INT");
      WG_REQUIRE_INPUT(mt->desc.sizes[0] == nd.sizes[0], "mapping tensor size != neighbor count");
      map_out = static_cast<int*>(tensor_data(mt));
    }
    auto s = static_cast<hipStream_t>(stream);
    if (td.dtype == WHOLEMEMORY_DT_INT) {
      append_unique_impl<int32_t>(static_cast<const int32_t*>(tensor_data(target_nodes_tensor)), (int)td.sizes[0],
                                  static_cast<const int32_t*>(tensor_data(neighbor_nodes_tensor)), (int)nd.sizes[0],
                                  output_unique_node_memory_context, map_out, p_env_fns, s);
    } else {
      append_unique_impl<int64_t>(static_cast<const int64_t*>(tensor_data(target_nodes_tensor)), (int)td.sizes[0],
                                  static_cast<const int64_t*>(tensor_data(neighbor_nodes_tensor)), (int)nd.sizes[0],
                                  output_unique_node_memory_context, map_out, p_env_fns, s);
    }
  });
}

wholememory_error_code_t csr_add_self_loop(wholememory_tensor_t csr_row_ptr_tensor,
                                           wholememory_tensor_t csr_col_ptr_tensor,
                                           wholememory_tensor_t output_csr_row_ptr_tensor,
                                           wholememory_tensor_t output_csr_col_ptr_tensor, void* stream)
{
  using namespace wgamd;
  return guarded("csr_add_self_loop", [&] {
    WG_REQUIRE_INPUT(csr_row_ptr_tensor && csr_col_ptr_tensor && output_csr_row_ptr_tensor && output_csr_col_ptr_tensor,
                     "null tensor");
    auto rd = csr_row_ptr_tensor->desc, cd = csr_col_ptr_tensor->desc;
    auto ord = output_csr_row_ptr_tensor->desc, ocd = output_csr_col_ptr_tensor->desc;
    WG_REQUIRE_INPUT(rd.dim == 1 && cd.dim == 1 && ord.dim == 1 && ocd.dim == 1, "all tensors must be 1-D");
    WG_REQUIRE_INPUT(rd.dtype == WHOLEMEMORY_DT_INT && cd.dtype == WHOLEMEMORY_DT_INT && ord.dtype == WHOLEMEMORY_DT_INT &&
                       ocd.dtype == WHOLEMEMORY_DT_INT,
                     "csr_add_self_loop supports INT only");
    WG_REQUIRE_INPUT(rd.sizes[0] >= 1 && ord.sizes[0] == rd.sizes[0], "output row_ptr size must equal input row_ptr size");
    WG_REQUIRE_INPUT(ocd.sizes[0] == cd.sizes[0] + rd.sizes[0] - 1, "output col size must be nnz + rows");
    int rows = (int)rd.sizes[0] - 1;
    if (rows == 0) {
      WG_HIP_CHECK(hipMemsetAsync(tensor_data(output_csr_row_ptr_tensor), 0, sizeof(int), static_cast<hipStream_t>(stream)));
      return;
    }
    add_self_loop_kernel<<<ceil_div((int64_t)rows * 64, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const int*>(tensor_data(csr_row_ptr_tensor)), static_cast<const int*>(tensor_data(csr_col_ptr_tensor)), rows,
      static_cast<int*>(tensor_data(output_csr_row_ptr_tensor)), static_cast<int*>(tensor_data(output_csr_col_ptr_tensor)));
    WG_HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"
