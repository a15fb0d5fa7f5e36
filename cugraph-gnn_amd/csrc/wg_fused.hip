// No-host-sync hop: uniform sampling + renumbering with device-resident sizes (include/wgamd_ext.h).
//
// The reference's walk (GraphStructure.multilayer_sample_without_replacement,
// /root/reference/python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:136-196) pays at
// least five stream synchronisations per hop because every op returns an
// exact-size tensor.  Here the SAME kernels run with capacity-sized grids and read the live
// sizes from device memory, so a mini-batch is a fixed launch sequence with no D2H round trip.
#include "wg_common.hpp"

namespace wgamd {
namespace {

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct hop_workspace {
  size_t cnt, scan_tmp, nbr, keys, minpos, slot_of, rank, total;
  int64_t slots;
};

hop_workspace plan(int64_t target_cap, int64_t edge_cap, size_t id_bytes)
{
  hop_workspace w{};
  size_t off = 0;
  auto take  = [&](size_t bytes) {
    size_t at = off;
    off += align_up(bytes);
    return at;
  };
  w.slots         = append_unique_slots(target_cap + edge_cap);
  int64_t scan_n  = std::max(target_cap, edge_cap) + 1;
  w.cnt           = take(sizeof(int) * (size_t)(target_cap + 1));
  w.scan_tmp      = take(sizeof(int) * (size_t)scan_tmp_ints(scan_n));
  w.nbr           = take(id_bytes * (size_t)edge_cap);
  w.keys          = take(id_bytes * (size_t)w.slots);
  w.minpos        = take(sizeof(int) * (size_t)w.slots);
  w.slot_of       = take(sizeof(int) * (size_t)(target_cap + edge_cap));
  w.rank          = take(sizeof(int) * (size_t)(edge_cap + 1));
  w.total         = off;
  return w;
}

}  // namespace
}  // namespace wgamd

extern "C" {

size_t wgamd_sample_hop_workspace_bytes(int64_t target_cap, int64_t edge_cap, wholememory_dtype_t id_dtype)
{
  if (target_cap < 0 || edge_cap < 0) return 0;
  size_t idb = id_dtype == WHOLEMEMORY_DT_INT64 ? 8 : 4;
  return wgamd::plan(target_cap, edge_cap, idb).total;
}

wholememory_error_code_t wgamd_sample_hop_nosync(const int64_t* csr_row_ptr, const void* csr_col,
                                                 wholememory_dtype_t id_dtype, const void* targets,
                                                 const int* n_targets_dev, int64_t target_cap, int max_sample_count,
                                                 unsigned long long random_seed, int* offsets, int* neighbor_lid,
                                                 int* center_lid, int64_t* edge_gid, int64_t edge_cap, void* unique,
                                                 int* counts_dev, void* workspace, size_t workspace_bytes, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sample_hop_nosync", [&] {
    WG_REQUIRE_INPUT(id_dtype == WHOLEMEMORY_DT_INT || id_dtype == WHOLEMEMORY_DT_INT64, "id dtype must be INT|INT64");
    WG_REQUIRE_INPUT(csr_row_ptr && csr_col && targets && n_targets_dev && offsets && neighbor_lid && unique &&
                       counts_dev && workspace,
                     "null pointer");
    WG_REQUIRE_INPUT(max_sample_count > 0, "the no-sync walk needs a positive fan-out (capacity = targets * M)");
    WG_REQUIRE_INPUT(target_cap > 0 && edge_cap >= target_cap * (int64_t)max_sample_count, "edge_cap < target_cap * M");
    WG_REQUIRE_INPUT(target_cap + edge_cap < ((int64_t)1 << 30), "capacities too large for one call");
    const bool i64  = id_dtype == WHOLEMEMORY_DT_INT64;
    hop_workspace w = plan(target_cap, edge_cap, i64 ? 8 : 4);
    WG_REQUIRE_INPUT(workspace_bytes >= w.total, "workspace too small: need %zu bytes", w.total);
    WG_REQUIRE_INPUT((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
    auto st    = static_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    int* cnt      = reinterpret_cast<int*>(base + w.cnt);
    int* scan_tmp = reinterpret_cast<int*>(base + w.scan_tmp);
    void* nbr     = base + w.nbr;
    void* keys    = base + w.keys;
    int* minpos   = reinterpret_cast<int*>(base + w.minpos);
    int* slot_of  = reinterpret_cast<int*>(base + w.slot_of);
    int* rank     = reinterpret_cast<int*>(base + w.rank);

    dev_count T{(int)target_cap, n_targets_dev};
    sample_count_enqueue(csr_row_ptr, targets, i64, T, max_sample_count, cnt, nullptr, st);
    exclusive_scan_i32(cnt, offsets, target_cap, scan_tmp, st);  // slack rows add 0: offsets[cap] = #edges
    uniform_sample_enqueue(csr_row_ptr, csr_col, i64, targets, i64, T, max_sample_count, (uint64_t)random_seed,
                           offsets, nbr, center_lid, edge_gid, st);
    dev_count E{(int)edge_cap, offsets + target_cap};
    append_unique_prepare_enqueue(targets, T, nbr, E, i64, keys, minpos, w.slots, slot_of, rank, scan_tmp, st);
    append_unique_emit_enqueue(targets, T, nbr, E, i64, minpos, slot_of, rank, unique, neighbor_lid, counts_dev, st);
  });
}

}  // extern "C"
