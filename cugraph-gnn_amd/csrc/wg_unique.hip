// De-duplication of an id list whose ids are known to lie below a bound (the vertex count of the table they index):
//     distinct = the distinct non-negative ids, ASCENDING;   inverse[i] = position of ids[i] in `distinct` (-1 for ids[i] < 0).
//
// Why it exists: a call group of G mini-batches fetches the features of every (batch, vertex) pair — 10.9 M rows for 191
// products batches — but the graph only HAS 2.45 M vertices, so at most 2.45 M distinct rows are behind them.  On one GPU the
// repeats are served by L2 / Infinity Cache and cost little; over xGMI every repeat is wire traffic (7 links x ~64 GB/s per
// direction per GPU).  The partitioned FeatureStore therefore fetches the DISTINCT rows through the all-to-all and expands
// them locally (wholegraph_amd/tensor.py: DistributedWholeMemoryTensor.gather(dedup=...)): 4.4x fewer bytes on the wire for
// the products call group.  The reference's NCCL gather (wholememory_gather_nccl, gather_op_impl_nccl.cu:23-171) exchanges
// every requested id; its embedding cache path de-duplicates for a different purpose (embedding_cache_func.cuh).
//
// Because the ids are bounded the job needs no sort and no hash table: mark -> pack + count -> scan over bound / 32 counts ->
// compact -> look up.  The compacted list comes out ascending, i.e. already grouped by owner rank of a range-partitioned table.
#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

// Round 6: the marks are one BYTE per possible id (2.4 MB for products: resident in every XCD's 4 MB L2, where the int flags'
// 10 MB were not), packed afterwards into one bit per id + a count per 32 ids; positions are prefix[id >> 5] + popc(bits below),
// kept side by side as {bits, prefix} so the look-up pass makes ONE scattered 8-byte load into 0.6 MB instead of two into 20 MB.
// For the 10.7 M listed rows of a products call group: mark 195 -> 134 us, look-up 114 -> 53 us (profiles/r06/README.md).  What is
// left of the mark is its stores (57 us without them, measured): every scattered byte store is one fabric write, and a mark set
// by one XCD is not seen by the other seven before the kernel ends, so an id listed by k batches is stored up to min(k, 8) times.
template <typename IdT>
__global__ void __launch_bounds__(256) unique_mark_kernel(const IdT* __restrict__ ids, int64_t n, int64_t bound, uint8_t* __restrict__ flags,
                                                          int* __restrict__ bad, const int* __restrict__ n_live)
{
  if (n_live) n = min(n, (int64_t)*n_live);   // (capacity-sized list of a no-sync walk: the live count is on the device)
  // four independent id loads, then four independent mark loads per trip
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4 * T) {
    int64_t id[4];
    uint8_t f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) id[k] = i + k * T < n ? (int64_t)ids[i + k * T] : -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (id[k] >= bound) {
        *bad  = 1;   // (every such lane writes the same value)
        id[k] = -1;
      }
      f[k] = id[k] >= 0 ? flags[id[k]] : 1;
    }
    // plain stores of one value: no atomics needed.  Test first: a call group names a hub thousands of times, and thousands of
    // stores to one address queue up in its L2 channel; the read of a byte that is already set is a broadcast hit
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (f[k] == 0) flags[id[k]] = 1;
  }
}

// 32 marks -> one word of bits + its population count (flags is padded with zeros to a multiple of 32)
__global__ void __launch_bounds__(256) unique_pack_kernel(const uint8_t* __restrict__ flags, int64_t n_words, uint2* __restrict__ rank,
                                                          int* __restrict__ cnt, int64_t cnt_padded)
{
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_words) {
    const uint4* p = reinterpret_cast<const uint4*>(flags + w * 32);
    const uint4 a = p[0], b = p[1];
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {   // bytes are 0 or 1: bit 0 of every byte, gathered
      const uint32_t x = v[k];
      word |= ((x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u)) << (4 * k);
    }
    rank[w].x = word;
    cnt[w]    = __popc(word);
  } else if (w < cnt_padded) {
    cnt[w] = 0;   // the scan reads whole tiles
  }
}

__global__ void __launch_bounds__(256) unique_compact_kernel(uint2* __restrict__ rank, const int* __restrict__ prefix, int64_t n_words,
                                                             int64_t* __restrict__ distinct, int* __restrict__ n_distinct)
{
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w == 0 && n_distinct) *n_distinct = prefix[n_words];
  if (w >= n_words) return;
  uint32_t word = rank[w].x;
  int at        = prefix[w];
  rank[w].y     = (uint32_t)at;   // {bits, ids set below this word} side by side: ONE scattered 8-byte load per look-up
  while (word) {
    const int k    = __ffs(word) - 1;
    distinct[at++] = w * 32 + k;
    word &= word - 1;
  }
}

template <typename IdT>
__global__ void __launch_bounds__(256) unique_inverse_kernel(const IdT* __restrict__ ids, int64_t n, int64_t bound, const uint2* __restrict__ rank,
                                                             int* __restrict__ inverse,
                                                             const int* __restrict__ n_live)
{
  if (n_live) n = min(n, (int64_t)*n_live);
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4 * T) {
    int64_t id[4];
    uint2 r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) id[k] = i + k * T < n ? (int64_t)ids[i + k * T] : -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool ok = id[k] >= 0 && id[k] < bound;
      r[k]          = ok ? rank[id[k] >> 5] : make_uint2(0u, 0xffffffffu);
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i + k * T < n) inverse[i + k * T] = (int)r[k].y + __popc(r[k].x & ((1u << (id[k] & 31)) - 1u));
  }
}

struct unique_plan {
  size_t flags, bits, cnt, tmp, bad, total;
  int64_t n_words, cnt_padded;
};
unique_plan plan_unique(int64_t bound)
{
  unique_plan p{};
  size_t at = 0;
  auto add  = [&](size_t bytes) {
    const size_t o = at;
    at += (bytes + 255) / 256 * 256;
    return o;
  };
  p.n_words    = (bound + 31) / 32;
  p.cnt_padded = (p.n_words + kScanTile) / kScanTile * kScanTile;   // counts are read by the scan in whole tiles
  p.flags = add((size_t)p.n_words * 32);
  p.bits  = add(sizeof(uint2) * (size_t)p.n_words);
  p.cnt   = add(sizeof(int) * (size_t)(p.cnt_padded + 1));           // scanned in place: prefix[n_words] = number of distinct ids
  p.tmp   = add(sizeof(int) * (size_t)scan_tmp_ints(p.n_words));
  p.bad   = add(sizeof(int));
  p.total = at;
  return p;
}

}  // namespace
}  // namespace wgamd

extern "C" {

size_t wgamd_unique_bounded_workspace_bytes(int64_t id_bound)
{
  if (id_bound <= 0 || id_bound >= ((int64_t)1 << 31) - 4096) return 0;
  return wgamd::plan_unique(id_bound).total;
}

wholememory_error_code_t wgamd_unique_bounded(const void* ids, wholememory_dtype_t id_dtype, int64_t n, int64_t id_bound,
                                              int64_t* distinct, int* inverse, int* n_distinct_dev, int* out_of_bound_dev,
                                              void* workspace, size_t workspace_bytes, void* stream)
{
  return wgamd_unique_bounded_live(ids, id_dtype, n, nullptr, id_bound, distinct, inverse, n_distinct_dev, out_of_bound_dev,
                                   workspace, workspace_bytes, stream);
}

wholememory_error_code_t wgamd_unique_bounded_live(const void* ids, wholememory_dtype_t id_dtype, int64_t n, const int* n_live_dev,
                                                   int64_t id_bound, int64_t* distinct, int* inverse, int* n_distinct_dev,
                                                   int* out_of_bound_dev, void* workspace, size_t workspace_bytes, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_unique_bounded", [&] {
    WG_REQUIRE_INPUT(id_dtype == WHOLEMEMORY_DT_INT || id_dtype == WHOLEMEMORY_DT_INT64, "id dtype must be INT|INT64");
    WG_REQUIRE_INPUT(n >= 0 && n < ((int64_t)1 << 31), "bad id count");
    WG_REQUIRE_INPUT(id_bound > 0 && id_bound < ((int64_t)1 << 31) - 4096, "id_bound must be in (0, 2^31)");
    WG_REQUIRE_INPUT(n_distinct_dev && workspace && (n == 0 || (ids && distinct && inverse)), "null pointer");
    const unique_plan p = plan_unique(id_bound);
    WG_REQUIRE_INPUT(workspace_bytes >= p.total, "workspace too small: need %zu bytes", p.total);
    WG_REQUIRE_INPUT((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* base     = static_cast<char*>(workspace);
    uint8_t* flags = reinterpret_cast<uint8_t*>(base + p.flags);
    uint2* rank    = reinterpret_cast<uint2*>(base + p.bits);
    int* cnt       = reinterpret_cast<int*>(base + p.cnt);
    int* tmp       = reinterpret_cast<int*>(base + p.tmp);
    int* bad       = reinterpret_cast<int*>(base + p.bad);
    WG_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)p.n_words * 32, st));
    WG_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), st));
    if (n > 0) {
      const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 256 * 32);
      if (id_dtype == WHOLEMEMORY_DT_INT) unique_mark_kernel<int32_t><<<grid, 256, 0, st>>>(static_cast<const int32_t*>(ids), n, id_bound, flags, bad, n_live_dev);
      else unique_mark_kernel<int64_t><<<grid, 256, 0, st>>>(static_cast<const int64_t*>(ids), n, id_bound, flags, bad, n_live_dev);
      WG_HIP_CHECK(hipGetLastError());
    }
    unique_pack_kernel<<<(int)ceil_div(p.cnt_padded, 256), 256, 0, st>>>(flags, p.n_words, rank, cnt, p.cnt_padded);
    exclusive_scan_i32(cnt, cnt, p.n_words, tmp, st);   // cnt[w] -> ids set below word w; cnt[n_words] = number of distinct ids
    unique_compact_kernel<<<(int)ceil_div(p.n_words, 256), 256, 0, st>>>(rank, cnt, p.n_words, distinct, n_distinct_dev);
    WG_HIP_CHECK(hipGetLastError());
    if (n > 0) {
      const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 256 * 32);
      if (id_dtype == WHOLEMEMORY_DT_INT) unique_inverse_kernel<int32_t><<<grid, 256, 0, st>>>(static_cast<const int32_t*>(ids), n, id_bound, rank, inverse, n_live_dev);
      else unique_inverse_kernel<int64_t><<<grid, 256, 0, st>>>(static_cast<const int64_t*>(ids), n, id_bound, rank, inverse, n_live_dev);
      WG_HIP_CHECK(hipGetLastError());
    }
    if (out_of_bound_dev) WG_HIP_CHECK(hipMemcpyAsync(out_of_bound_dev, bad, sizeof(int), hipMemcpyDeviceToDevice, st));
  });
}

}  // extern "C"
