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

// thread p < T inserts target p, thread T+e inserts neighbour e; remembers its slot.
template <typename KeyT, typename TableKeyT>
__global__ void __launch_bounds__(256) table_insert_kernel(const KeyT* __restrict__ targets,
                                                           dev_count T_,
                                                           const KeyT* __restrict__ neighbors,
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
  KeyT id       = p < T ? targets[p] : neighbors[p - T];
  TableKeyT key = (TableKeyT)id;
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

template <typename KeyT>
__global__ void __launch_bounds__(256) renumber_emit_kernel(const KeyT* __restrict__ targets,
                                                            const KeyT* __restrict__ neighbors,
                                                            const int* __restrict__ minpos,
                                                            const int* __restrict__ slot_of,
                                                            const int* __restrict__ rank,  // exclusive scan of flag, [E.host+1]
                                                            dev_count T_,
                                                            dev_count E_,
                                                            batch_view bv,
                                                            KeyT* __restrict__ unique_out,
                                                            int* __restrict__ map_out,
                                                            int* __restrict__ counts_out)
{
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
    unique_out[row] = neighbors[e];
    if (bv.unique_batch) bv.unique_batch[row] = b;
    if (bv.frontier_out) {  // next frontier, ordered by (batch, first appearance) == by rank
      static_cast<KeyT*>(bv.frontier_out)[rank[e]] = neighbors[e];
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
  temp_buffer keys_b(env), pos_b(env), slot_b(env), flag_b(env), tmp_b(env);
  KeyT* keys   = static_cast<KeyT*>(keys_b.alloc(slots, key_traits<KeyT>::dt));
  int* minpos  = pos_b.device<int>(slots, WHOLEMEMORY_DT_INT);
  int* slot_of = slot_b.device<int>(P, WHOLEMEMORY_DT_INT);
  int* rank    = flag_b.device<int>(E + 1, WHOLEMEMORY_DT_INT);
  int* stmp    = tmp_b.device<int>(scan_tmp_ints(E + 1), WHOLEMEMORY_DT_INT);
  const bool k64 = sizeof(KeyT) == 8;
  dev_count Tc{T, nullptr}, Ec{E, nullptr};
  batch_view one{};
  one.G = 1;

  append_unique_prepare_enqueue(targets, Tc, neighbors, Ec, k64, one, keys, minpos, slots, slot_of, rank, stmp, stream);
  int U = 0;
  WG_HIP_CHECK(hipMemcpyAsync(&U, rank + E, sizeof(int), hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // output size

  KeyT* unique_out = static_cast<KeyT*>(output_alloc(env, unique_ctx, (int64_t)T + U, key_traits<KeyT>::dt));
  append_unique_emit_enqueue(targets, Tc, neighbors, Ec, k64, one, minpos, slot_of, rank, unique_out, map_out, nullptr,
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

template <typename KeyT>
__global__ void __launch_bounds__(256)
table_insert_packed_kernel(const KeyT* __restrict__ targets, dev_count T_, const KeyT* __restrict__ neighbors, dev_count E_,
                           batch_view bv, unsigned long long* table, int64_t capacity_slots, packed_layout lay,
                           int* __restrict__ slot_of)
{
  const int T = T_.get(), E = E_.get();
  const uint32_t slots = live_slot_count(T + E, capacity_slots);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= T + E) return;
  const KeyT id = p < T ? targets[p] : neighbors[p - T];
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

template <typename KeyT>
void prepare_packed_t(const KeyT* targets, dev_count T, const KeyT* neighbors, dev_count E, batch_view bv, void* keys,
                      int64_t slots, packed_layout lay, int* slot_of, int* rank, int* scan_tmp, hipStream_t stream)
{
  auto* table = static_cast<unsigned long long*>(keys);
  const int P = T.host + E.host;
  table_clear_packed_kernel<<<(int)std::min<int64_t>(ceil_div(slots, 1024), 4096), 256, 0, stream>>>(table, slots, T, E);
  if (P > 0)
    table_insert_packed_kernel<KeyT><<<ceil_div(P, 256), 256, 0, stream>>>(targets, T, neighbors, E, bv, table, slots, lay,
                                                                           slot_of);
  if (E.host > 0) first_flag_packed_kernel<<<ceil_div(E.host, 256), 256, 0, stream>>>(table, slot_of, T, E, lay, rank);
  WG_HIP_CHECK(hipGetLastError());
  exclusive_scan_i32(rank, rank, E.host, scan_tmp, stream, E.dev);
}

// ---- call groups, per-batch tables in LDS ----------------------------------------------------------------------------
// A mini-batch is renumbered on its own, so its table never has to be shared across the device — and a device-wide table
// costs one memory-side atomic per new key plus one fabric-side load per probe (gfx950: agent-scope atomics and sc1 loads
// are resolved beyond the XCD's L2), 24 G keys/s on an MI355X however it is laid out.  Instead:
//   1. bucket_count   block (batch, chunk): histogram of the chunk's ids over the batch's R hash ranges
//                     (R = positions / 5,500: what one LDS table holds at load <= 0.55) -> counts[range][chunk];
//   2. bucket_scatter same blocks: prefix of the batch's counts (no scan kernel: a batch's buckets fill exactly the
//                     batch's own stretch of the position space) -> (id, position) pairs grouped by range, streamed;
//   3. renumber_lds   workgroup (batch, range): its bucket goes into a 10,000-slot open-addressing table in LDS
//                     (word = [ id | first position ], ds_cmpst / ds_min), then the bucket's neighbours look their first
//                     positions up.  Every lane has work (the bucket is dense), no table in global memory, nothing to
//                     clear, no global atomics.  A range whose DISTINCT ids overfill the table (a hot id repeated is
//                     one key) is split in two and redone through a hash filter, so any id distribution terminates.
// Cost is linear in the positions whatever the batch size.  Scratch: the pairs live where the device-wide table would
// (8 B x slots >= 16 B per position), the counts where its positions array would.
constexpr int kLdsSlots      = 10000;     // 80,000 B: two workgroups per CU, one computes while the other waits on its loads
constexpr int kLdsKeysTarget = 3000;      // positions per range the range count is sized for (load <= 0.3: short probe chains)
constexpr int kLdsProbeLimit = 256;       // probes after which a range is declared overfull and split
constexpr int kLdsThreads    = 512;       // 6 pairs per thread = one trip of kLdsUnroll per range (1024: walk 1.227 -> 1.19 ms per call group of 191;
                                          // half / quarter-size tables with 512 / 256 threads: 1.28 / 1.48 — more ranges cost more to bucket)
constexpr int kLdsMaxRanges  = 2048;      // per batch; beyond, ranges simply start overfull and split
constexpr int kLdsStack      = 40;        // pending hash ranges of one workgroup (a split pushes two, pops one)
constexpr int kLdsChunks     = 16;        // blocks per batch in the two bucketing kernels
constexpr int kBucketThreads = 256;
constexpr int kBucketUnroll  = 4;         // ids in flight per thread of the two bucketing kernels
constexpr int kLdsUnroll     = 6;         // pairs in flight per thread of the table kernel

__device__ __forceinline__ uint32_t hash_id32(uint32_t h)   // murmur3 finaliser
{
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  return h ^ (h >> 16);
}
template <typename KeyT>
__device__ __forceinline__ uint32_t hash_id(KeyT id)
{
  if constexpr (sizeof(KeyT) == 4) return hash_id32((uint32_t)id);
  else return hash_key((uint64_t)id);
}

// one mini-batch of the call group: its targets [t0, t0 + nT), its neighbours [e0, e0 + nE), its R hash ranges and where its
// range records start (rb: every batch gets floor(positions before it / kLdsKeysTarget) + b, which leaves room for its R)
struct batch_part {
  int t0, nT, e0, nE, P, R, rb;
  // keys_target: kLdsKeysTarget, or the (larger) value of a test that wants ranges to overfill and split
  __device__ batch_part(const batch_view& bv, int b, int keys_target)
  {
    t0 = bv.target_seg[b];
    nT = bv.target_seg[b + 1] - t0;
    e0 = bv.edge_offsets[bv.sseg()[b]];
    nE = bv.edge_offsets[bv.sseg()[b + 1]] - e0;
    P  = nT + nE;
    R  = min(max((P + keys_target - 1) / keys_target, 1), kLdsMaxRanges);
    rb = (t0 + e0) / keys_target + b;
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

// scratch behind the three kernels: counts[(rb + r) * kLdsChunks + c], then per range {first pair, pairs}
struct bucket_scratch {
  int keys_target;
  int* counts;
  int* range_start;
  int* range_count;
  void* ids;    // KeyT[capacity positions]
  int* pos;     // int[capacity positions]
};

template <typename KeyT>
__global__ void __launch_bounds__(kBucketThreads)
bucket_count_kernel(const KeyT* __restrict__ targets, const KeyT* __restrict__ neighbors, batch_view bv, bucket_scratch sc)
{
  __shared__ int hist[kLdsMaxRanges];
  int b, c;
  if (!batch_of_block(bv.G, kLdsChunks, b, c)) return;
  const batch_part bp(bv, b, sc.keys_target);
  if (bp.nE <= 0) return;
  for (int r = threadIdx.x; r < bp.R; r += kBucketThreads) hist[r] = 0;
  __syncthreads();
  const int chunk = (bp.P + kLdsChunks - 1) / kLdsChunks;
  const int end   = min(bp.P, (c + 1) * chunk);
  for (int i0 = c * chunk + threadIdx.x; i0 < end; i0 += kBucketThreads * kBucketUnroll) {
    KeyT id[kBucketUnroll];   // all loads of the trip are in flight before the first histogram update
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++) {
      const int i = min(i0 + k * kBucketThreads, end - 1);
      id[k]       = i < bp.nT ? targets[bp.t0 + i] : neighbors[bp.e0 + i - bp.nT];
    }
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++)
      if (i0 + k * kBucketThreads < end) atomicAdd(&hist[__umulhi(hash_id<KeyT>(id[k]), (uint32_t)bp.R)], 1);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < bp.R; r += kBucketThreads) sc.counts[(bp.rb + r) * kLdsChunks + c] = hist[r];
}

template <typename KeyT>
__global__ void __launch_bounds__(kBucketThreads)
bucket_scatter_kernel(const KeyT* __restrict__ targets, dev_count T_, const KeyT* __restrict__ neighbors, batch_view bv,
                      bucket_scratch sc)
{
  __shared__ int cursor[kLdsMaxRanges];   // first: pairs of the range over all chunks; then: where my next pair goes
  __shared__ int before[kLdsMaxRanges];   // pairs of the range in the chunks before mine
  int b, c;
  if (!batch_of_block(bv.G, kLdsChunks, b, c)) return;
  const batch_part bp(bv, b, sc.keys_target);
  if (bp.nE <= 0) return;
  const int T = T_.get();
  for (int r = threadIdx.x; r < bp.R; r += kBucketThreads) {
    const int* row = sc.counts + (bp.rb + r) * kLdsChunks;
    int tot = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < kLdsChunks; k++) {
      const int v = row[k];
      mine += k < c ? v : 0;
      tot += v;
    }
    cursor[r] = tot;
    before[r] = mine;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // the batch's pairs fill [t0 + e0, t0 + e0 + P) of the pair arrays, range after range
    int at = bp.t0 + bp.e0;
    for (int r = 0; r < bp.R; r++) {
      const int tot = cursor[r];
      if (c == 0) {
        sc.range_start[bp.rb + r] = at;
        sc.range_count[bp.rb + r] = tot;
      }
      cursor[r] = at + before[r];
      at += tot;
    }
  }
  __syncthreads();
  KeyT* ids       = static_cast<KeyT*>(sc.ids);
  const int chunk = (bp.P + kLdsChunks - 1) / kLdsChunks;
  const int end   = min(bp.P, (c + 1) * chunk);
  for (int i0 = c * chunk + threadIdx.x; i0 < end; i0 += kBucketThreads * kBucketUnroll) {
    KeyT id[kBucketUnroll];
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++) {
      const int i = min(i0 + k * kBucketThreads, end - 1);
      id[k]       = i < bp.nT ? targets[bp.t0 + i] : neighbors[bp.e0 + i - bp.nT];
    }
#pragma unroll
    for (int k = 0; k < kBucketUnroll; k++) {
      const int i = i0 + k * kBucketThreads;
      if (i < end) {
        const int dst = atomicAdd(&cursor[__umulhi(hash_id<KeyT>(id[k]), (uint32_t)bp.R)], 1);
        ids[dst]      = id[k];
        sc.pos[dst]   = i < bp.nT ? bp.t0 + i : T + bp.e0 + (i - bp.nT);
      }
    }
  }
}

template <typename KeyT>
__global__ void __launch_bounds__(kLdsThreads)
renumber_lds_kernel(dev_count T_, dev_count E_, batch_view bv, packed_layout lay, int wg_per_batch, bucket_scratch sc,
                    int* __restrict__ slot_of, int* __restrict__ flag)
{
  __shared__ unsigned long long tbl[kLdsSlots];
  __shared__ unsigned long long st_lo[kLdsStack], st_span[kLdsStack];
  __shared__ int st_n, overfull;
  const unsigned long long kEmpty = ~0ull;
  const int T = T_.get(), E = E_.get();
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    // the scan reads whole tiles: zero-fill the slack of the tile that holds the live end
    const int end = min(E_.host, (E / kScanTile + 1) * kScanTile);
    for (int e = E + tid; e < end; e += kLdsThreads) flag[e] = 0;
  }
  int b, j;
  if (!batch_of_block(bv.G, wg_per_batch, b, j)) return;
  const batch_part bp(bv, b, sc.keys_target);
  if (bp.nE <= 0) return;   // no neighbour needs a first position
  const KeyT* ids = static_cast<const KeyT*>(sc.ids);
  const unsigned long long pos_mask = lay.pos_mask();
  const volatile int* overfull_now  = &overfull;

  for (int r = j; r < bp.R; r += wg_per_batch) {
    const int first_pair = sc.range_start[bp.rb + r], n_pairs = sc.range_count[bp.rb + r];
    if (n_pairs == 0) continue;
    __syncthreads();   // the previous range's last look at the stack is over
    if (tid == 0) {
      // mulhi(h, R) == r  <=>  h in [ceil(r 2^32 / R), ceil((r + 1) 2^32 / R))
      const unsigned long long lo = (((unsigned long long)r << 32) + bp.R - 1) / (unsigned)bp.R;
      st_lo[0]   = lo;
      st_span[0] = ((((unsigned long long)(r + 1) << 32) + bp.R - 1) / (unsigned)bp.R) - lo;
      st_n       = 1;
    }
    __syncthreads();
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
      // ---- insert the pairs of the range (all of the bucket unless the range was split) -----------------------------
      // kLdsUnroll pairs per thread are in flight before the first one is used: one load at a time leaves a
      // 16-wave workgroup waiting on memory latency
      for (int i0 = 0; i0 < n_pairs; i0 += kLdsThreads * kLdsUnroll) {
        KeyT id_k[kLdsUnroll];
        int pos_k[kLdsUnroll];
#pragma unroll
        for (int k = 0; k < kLdsUnroll; k++) {
          const int i = min(i0 + k * kLdsThreads + tid, n_pairs - 1);
          id_k[k]     = ids[first_pair + i];
          pos_k[k]    = sc.pos[first_pair + i];
        }
#pragma unroll
        for (int k = 0; k < kLdsUnroll; k++) {
          const KeyT id    = id_k[k];
          const uint32_t h = hash_id<KeyT>(id);
          if (i0 + k * kLdsThreads + tid < n_pairs && (unsigned long long)h - lo < span && !*overfull_now) {
            const unsigned long long word = ((unsigned long long)id << lay.pos_bits) | (unsigned long long)pos_k[k];
            uint32_t s = __umulhi(h * 0x9E3779B1u, (uint32_t)kLdsSlots);
            int probes = 0;
            while (true) {
              // one LDS round trip per probe: the compare-and-swap doubles as the read
              const unsigned long long cur = atomicCAS(&tbl[s], kEmpty, word);
              if (cur == kEmpty) break;
              if ((cur >> lay.pos_bits) == (unsigned long long)id) {
                if (word < cur) atomicMin(&tbl[s], word);
                break;
              }
              s = s + 1 == (uint32_t)kLdsSlots ? 0u : s + 1;
              if (++probes > kLdsProbeLimit) {   // never at the load a range is sized for: the table is (nearly) full
                overfull = 1;
                break;
              }
            }
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
      // ---- first position of every neighbour of the range --------------------------------------------------------
      for (int i0 = 0; i0 < n_pairs; i0 += kLdsThreads * kLdsUnroll) {
        KeyT id_k[kLdsUnroll];
        int pos_k[kLdsUnroll];
#pragma unroll
        for (int k = 0; k < kLdsUnroll; k++) {
          const int i = min(i0 + k * kLdsThreads + tid, n_pairs - 1);
          id_k[k]     = ids[first_pair + i];
          pos_k[k]    = sc.pos[first_pair + i];
        }
#pragma unroll
        for (int k = 0; k < kLdsUnroll; k++) {
          const KeyT id    = id_k[k];
          const int p      = pos_k[k];
          const uint32_t h = hash_id<KeyT>(id);
          // (a target: nobody asks for its first position)
          if (i0 + k * kLdsThreads + tid < n_pairs && p >= T && (unsigned long long)h - lo < span) {
            uint32_t s = __umulhi(h * 0x9E3779B1u, (uint32_t)kLdsSlots);
            unsigned long long cur = tbl[s];
            while ((cur >> lay.pos_bits) != (unsigned long long)id) {
              s   = s + 1 == (uint32_t)kLdsSlots ? 0u : s + 1;
              cur = tbl[s];
            }
            const int first = (int)(cur & pos_mask);
            slot_of[p]      = first;
            flag[p - T]     = first == p ? 1 : 0;
          }
        }
      }
      __syncthreads();
    }
  }
}

// Emit for call groups, one block per (batch, chunk): everything that depends on the batch only (its row shift, where
// its new vertices start, its local-id origin) is read once per block instead of chased through four dependent loads per
// edge, and the one random read left (the rank of the first occurrence) stays inside the batch's stretch of `rank`,
// which the XCD-affine block mapping keeps in one L2.  Same outputs as renumber_emit_kernel.
constexpr int kEmitChunks = 16;
constexpr int kEmitUnroll = 4;

template <typename KeyT>
__global__ void __launch_bounds__(256)
renumber_emit_batched_kernel(const KeyT* __restrict__ targets, const KeyT* __restrict__ neighbors,
                             const int* __restrict__ slot_of, const int* __restrict__ rank, dev_count T_, dev_count E_,
                             batch_view bv, KeyT* __restrict__ unique_out, int* __restrict__ map_out,
                             int* __restrict__ counts_out)
{
  const int T = T_.get(), E = E_.get();
  const int U = rank[E_.host];  // slack flags are 0, so the grand total sits at the capacity end
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsize = gridDim.x * blockDim.x;
  if (gtid == 0 && counts_out) {
    counts_out[0] = E;
    counts_out[1] = T + U;
  }
  if (bv.unique_seg && gtid <= bv.G) {
    const int new_before = rank[bv.edge_offsets[bv.sseg()[gtid]]];   // new vertices of batches < gtid
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
  const int shift    = rank[e0];          // rows contributed by the new vertices of earlier batches
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
    for (int k = 0; k < kEmitUnroll; k++) row[k] = first[k] < T ? first[k] + shift : tail_row + rank[first[k] - T];
#pragma unroll
    for (int k = 0; k < kEmitUnroll; k++) {
      const int i = i0 + k * 256;
      if (i >= end) break;
      const int e = e0 + i;
      if (first[k] == T + e) {
        const KeyT id      = neighbors[e];
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

// range records the three kernels address: every batch starts at floor(positions before it / kLdsKeysTarget) + b
inline int64_t lds_range_records(int64_t capacity_positions, int G) { return capacity_positions / kLdsKeysTarget + 2 * (int64_t)G + 2; }
// WGAMD_RENUMBER_KEYS_TARGET (tests): positions per hash range, >= kLdsKeysTarget; a value the table cannot hold makes
// every range overfill and exercises the split path
inline int lds_keys_target()
{
  static const int v = [] {
    const char* e = getenv("WGAMD_RENUMBER_KEYS_TARGET");
    const long t  = e ? atol(e) : 0;
    return t > kLdsKeysTarget ? (int)std::min<long>(t, 1 << 30) : kLdsKeysTarget;
  }();
  return v;
}
inline bool lds_scratch_fits(int64_t capacity_positions, int G, int64_t slots, size_t id_bytes)
{
  // keys buffer: 8 B per slot holds ids + positions; positions buffer: 4 B per slot holds counts + the two range arrays
  return (int64_t)(id_bytes + 4) * capacity_positions + 256 <= 8 * slots &&
         lds_range_records(capacity_positions, G) * (kLdsChunks + 2) <= slots;
}

template <typename KeyT>
void prepare_lds_t(const KeyT* targets, dev_count T, const KeyT* neighbors, dev_count E, batch_view bv, packed_layout lay,
                   void* keys, int* minpos, int* slot_of, int* rank, int* scan_tmp, hipStream_t stream)
{
  if (E.host > 0) {
    const int64_t cap = (int64_t)T.host + E.host;
    const int64_t rec = lds_range_records(cap, bv.G);
    bucket_scratch sc;
    sc.keys_target = lds_keys_target();
    sc.counts      = minpos;
    sc.range_start = minpos + rec * kLdsChunks;
    sc.range_count = sc.range_start + rec;
    sc.ids         = keys;
    sc.pos         = reinterpret_cast<int*>(static_cast<char*>(keys) + (((size_t)cap * sizeof(KeyT) + 255) / 256) * 256);
    bucket_count_kernel<KeyT><<<batch_grid(bv.G, kLdsChunks), kBucketThreads, 0, stream>>>(targets, neighbors, bv, sc);
    bucket_scatter_kernel<KeyT><<<batch_grid(bv.G, kLdsChunks), kBucketThreads, 0, stream>>>(targets, T, neighbors, bv, sc);
    const int64_t per_batch = (cap + bv.G - 1) / bv.G;
    const int wg_per_batch  = (int)std::max<int64_t>(1, std::min<int64_t>((per_batch + kLdsKeysTarget - 1) / kLdsKeysTarget, 32));
    renumber_lds_kernel<KeyT><<<batch_grid(bv.G, wg_per_batch), kLdsThreads, 0, stream>>>(T, E, bv, lay, wg_per_batch, sc, slot_of, rank);
    WG_HIP_CHECK(hipGetLastError());
  }
  exclusive_scan_i32(rank, rank, E.host, scan_tmp, stream, E.dev);
}

template <typename KeyT, typename TableKeyT>
void prepare_t(const KeyT* targets, dev_count T, const KeyT* neighbors, dev_count E, batch_view bv, TableKeyT* keys,
               int* minpos, int64_t slots, int* slot_of, int* rank, int* scan_tmp, hipStream_t stream)
{
  const int P = T.host + E.host;
  table_clear_kernel<TableKeyT><<<(int)std::min<int64_t>(ceil_div(slots, 1024), 4096), 256, 0, stream>>>(keys, minpos, slots, T,
                                                                                                   E);
  if (P > 0)
    table_insert_kernel<KeyT, TableKeyT><<<ceil_div(P, 256), 256, 0, stream>>>(targets, T, neighbors, E, bv, keys,
                                                                              minpos, slots, slot_of);
  if (E.host > 0) first_flag_kernel<<<ceil_div(E.host, 256), 256, 0, stream>>>(minpos, slot_of, T, E, rank);
  WG_HIP_CHECK(hipGetLastError());
  exclusive_scan_i32(rank, rank, E.host, scan_tmp, stream, E.dev);  // flags -> ranks, rank[E.host] = #new nodes
}

}  // namespace

int64_t append_unique_slots(int64_t capacity)
{
  int64_t slots = 1024;
  while (slots < 2 * capacity) slots <<= 1;
  return slots;
}

void append_unique_prepare_enqueue(const void* targets, dev_count T, const void* neighbors, dev_count E, bool ids64,
                                   batch_view bv, void* keys, int* minpos, int64_t slots, int* slot_of, int* rank,
                                   int* scan_tmp, hipStream_t stream)
{
  const bool batched = bv.target_batch != nullptr;
  packed_layout lay{};
  static const bool no_lds = getenv("WGAMD_RENUMBER_NO_LDS") != nullptr;
  if (batched && !no_lds && bv.target_seg && bv.edge_offsets && bv.G > 1 &&
      packed_layout_for((int64_t)T.host + E.host, 1, bv.id_bound, lay) &&
      lds_scratch_fits((int64_t)T.host + E.host, bv.G, slots, ids64 ? 8 : 4)) {
    if (ids64)
      prepare_lds_t<int64_t>(static_cast<const int64_t*>(targets), T, static_cast<const int64_t*>(neighbors), E, bv, lay, keys,
                             minpos, slot_of, rank, scan_tmp, stream);
    else
      prepare_lds_t<int32_t>(static_cast<const int32_t*>(targets), T, static_cast<const int32_t*>(neighbors), E, bv, lay, keys,
                             minpos, slot_of, rank, scan_tmp, stream);
    return;
  }
  if (batched && packed_layout_for((int64_t)T.host + E.host, bv.G, bv.id_bound, lay)) {
    if (ids64)
      prepare_packed_t<int64_t>(static_cast<const int64_t*>(targets), T, static_cast<const int64_t*>(neighbors), E, bv, keys,
                                slots, lay, slot_of, rank, scan_tmp, stream);
    else
      prepare_packed_t<int32_t>(static_cast<const int32_t*>(targets), T, static_cast<const int32_t*>(neighbors), E, bv, keys,
                                slots, lay, slot_of, rank, scan_tmp, stream);
    return;
  }
  if (ids64)
    prepare_t<int64_t, int64_t>(static_cast<const int64_t*>(targets), T, static_cast<const int64_t*>(neighbors), E, bv,
                                static_cast<int64_t*>(keys), minpos, slots, slot_of, rank, scan_tmp, stream);
  else if (batched)
    prepare_t<int32_t, int64_t>(static_cast<const int32_t*>(targets), T, static_cast<const int32_t*>(neighbors), E, bv,
                                static_cast<int64_t*>(keys), minpos, slots, slot_of, rank, scan_tmp, stream);
  else
    prepare_t<int32_t, int32_t>(static_cast<const int32_t*>(targets), T, static_cast<const int32_t*>(neighbors), E, bv,
                                static_cast<int32_t*>(keys), minpos, slots, slot_of, rank, scan_tmp, stream);
}

void append_unique_emit_enqueue(const void* targets, dev_count T, const void* neighbors, dev_count E, bool ids64,
                                batch_view bv, const int* minpos, const int* slot_of, const int* rank,
                                void* unique_out, int* map_out, int* counts_out, hipStream_t stream)
{
  const int P    = T.host + E.host;
  const int grid = ceil_div(P > bv.G + 1 ? P : bv.G + 1, 256);  // thread 0 publishes the counts, threads <= G the segments
  if (bv.target_batch != nullptr && bv.target_seg && bv.edge_offsets && bv.G > 1) {
    const int bgrid = std::max(batch_grid(bv.G, kEmitChunks), ceil_div(bv.G + 1, 256));
    if (ids64)
      renumber_emit_batched_kernel<int64_t><<<bgrid, 256, 0, stream>>>(static_cast<const int64_t*>(targets),
                                                                      static_cast<const int64_t*>(neighbors), slot_of, rank, T, E,
                                                                      bv, static_cast<int64_t*>(unique_out), map_out, counts_out);
    else
      renumber_emit_batched_kernel<int32_t><<<bgrid, 256, 0, stream>>>(static_cast<const int32_t*>(targets),
                                                                      static_cast<const int32_t*>(neighbors), slot_of, rank, T, E,
                                                                      bv, static_cast<int32_t*>(unique_out), map_out, counts_out);
    WG_HIP_CHECK(hipGetLastError());
    return;
  }
  if (ids64)
    renumber_emit_kernel<int64_t><<<grid, 256, 0, stream>>>(static_cast<const int64_t*>(targets),
                                                           static_cast<const int64_t*>(neighbors), minpos, slot_of, rank,
                                                           T, E, bv, static_cast<int64_t*>(unique_out), map_out, counts_out);
  else
    renumber_emit_kernel<int32_t><<<grid, 256, 0, stream>>>(static_cast<const int32_t*>(targets),
                                                           static_cast<const int32_t*>(neighbors), minpos, slot_of, rank,
                                                           T, E, bv, static_cast<int32_t*>(unique_out), map_out, counts_out);
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
