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
// Because the ids are bounded the job needs no sort and no hash table: mark -> scan over the bound -> compact -> look up.
// The marks are one int per possible id (10 MB for products: L2-resident while 10.9 M lanes write into it), the compacted list
// comes out ascending, i.e. already grouped by owner rank of a range-partitioned table.
#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

template <typename IdT>
__global__ void __launch_bounds__(256) unique_mark_kernel(const IdT* __restrict__ ids, int64_t n, int64_t bound, int* __restrict__ flags,
                                                          int* __restrict__ bad, const int* __restrict__ n_live)
{
  if (n_live) n = min(n, (int64_t)*n_live);   // (capacity-sized list of a no-sync walk: the live count is on the device)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = (int64_t)ids[i];
    if (id < 0) continue;
    if (id >= bound) {
      *bad = 1;   // (every such lane writes the same value)
      continue;
    }
    flags[id] = 1;   // plain stores of one value: no atomics needed
  }
}

__global__ void __launch_bounds__(256) unique_compact_kernel(const int* __restrict__ flags, const int* __restrict__ pos, int64_t bound,
                                                             int64_t* __restrict__ distinct, int* __restrict__ n_distinct)
{
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id == 0 && n_distinct) *n_distinct = pos[bound];
  if (id < bound && flags[id]) distinct[pos[id]] = id;
}

template <typename IdT>
__global__ void __launch_bounds__(256) unique_inverse_kernel(const IdT* __restrict__ ids, int64_t n, int64_t bound, const int* __restrict__ pos,
                                                             int* __restrict__ inverse, const int* __restrict__ n_live)
{
  if (n_live) n = min(n, (int64_t)*n_live);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = (int64_t)ids[i];
    inverse[i]       = (id < 0 || id >= bound) ? -1 : pos[id];
  }
}

struct unique_plan {
  size_t flags, pos, tmp, bad, total;
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
  // flags are read by the scan in whole tiles: pad to the tile
  const int64_t padded = (bound + kScanTile) / kScanTile * kScanTile;
  p.flags = add(sizeof(int) * (size_t)padded);
  p.pos   = add(sizeof(int) * (size_t)(padded + 1));
  p.tmp   = add(sizeof(int) * (size_t)scan_tmp_ints(bound));
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
    int* flags     = reinterpret_cast<int*>(base + p.flags);
    int* pos       = reinterpret_cast<int*>(base + p.pos);
    int* tmp       = reinterpret_cast<int*>(base + p.tmp);
    int* bad       = reinterpret_cast<int*>(base + p.bad);
    WG_HIP_CHECK(hipMemsetAsync(flags, 0, p.pos - p.flags, st));
    WG_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), st));
    if (n > 0) {
      const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 256 * 32);
      if (id_dtype == WHOLEMEMORY_DT_INT) unique_mark_kernel<int32_t><<<grid, 256, 0, st>>>(static_cast<const int32_t*>(ids), n, id_bound, flags, bad, n_live_dev);
      else unique_mark_kernel<int64_t><<<grid, 256, 0, st>>>(static_cast<const int64_t*>(ids), n, id_bound, flags, bad, n_live_dev);
      WG_HIP_CHECK(hipGetLastError());
    }
    exclusive_scan_i32(flags, pos, id_bound, tmp, st);   // pos[id_bound] = number of distinct ids
    unique_compact_kernel<<<(int)ceil_div(id_bound, 256), 256, 0, st>>>(flags, pos, id_bound, distinct, n_distinct_dev);
    WG_HIP_CHECK(hipGetLastError());
    if (n > 0) {
      const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 256 * 32);
      if (id_dtype == WHOLEMEMORY_DT_INT) unique_inverse_kernel<int32_t><<<grid, 256, 0, st>>>(static_cast<const int32_t*>(ids), n, id_bound, pos, inverse, n_live_dev);
      else unique_inverse_kernel<int64_t><<<grid, 256, 0, st>>>(static_cast<const int64_t*>(ids), n, id_bound, pos, inverse, n_live_dev);
      WG_HIP_CHECK(hipGetLastError());
    }
    if (out_of_bound_dev) WG_HIP_CHECK(hipMemcpyAsync(out_of_bound_dev, bad, sizeof(int), hipMemcpyDeviceToDevice, st));
  });
}

}  // extern "C"
