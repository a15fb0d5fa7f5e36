// Uniform neighbour sampling WITH replacement over a CSR graph (include/wgamd_ext.h).
//
// cugraph_pyg exposes it as `replace=True` / `with_replacement` and hands it to libcugraph
// (/root/reference/python/cugraph-pyg/cugraph_pyg/sampler/distributed_sampler.py:775-792,864,
//  loader/neighbor_loader.py:118-120); the arithmetic is not in the reference tree, so the draw layout below is this
// library's own; the parity tests pin it with a CPU restatement, not with a reference vector:
//   a seed with N > 0 neighbours yields EXACTLY M picks (M > 0), pick t = col[start + G(seed64, i * M + t).i31() % N] —
//   one PCG32 stream per (seed index i, draw t), the op's usual generator (wg_rng.hpp) — emitted in draw order; a seed
//   without neighbours yields nothing.  sample_offset[i] = M x (number of seeds j < i with neighbours).
// One thread per pick: a table jump (no dependent multiply chain) + one draw, then one random 4/8-byte read of col.
#include "wg_common.hpp"
#include "wg_rng.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

template <typename SeedT>
__global__ void __launch_bounds__(256)
replace_count_kernel(const int64_t* __restrict__ row_ptr, const SeedT* __restrict__ seeds, int n, int M, int* __restrict__ cnt)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t nid = (int64_t)seeds[i];
  cnt[i]            = row_ptr[nid + 1] > row_ptr[nid] ? M : 0;
}

template <typename SeedT, typename ColT>
__global__ void __launch_bounds__(256)
replace_sample_kernel(const int64_t* __restrict__ row_ptr, const ColT* __restrict__ col, const SeedT* __restrict__ seeds, int n,
                      int M, uint64_t random_seed, const int* __restrict__ offsets, ColT* __restrict__ dst,
                      int* __restrict__ src_lid, int64_t* __restrict__ edge_gid)
{
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)n * M) return;
  const int i = (int)(p / M), t = (int)(p - (int64_t)i * M);
  // the draw depends only on (i, t): computed before the dependent seeds -> row_ptr loads are waited for
  int32_t r;
  if (p < (1ll << 31)) {
    Pcg32 g(random_seed, (uint32_t)p, Pcg32::table_tag{});
    r = g.next_i31();
  } else {
    Pcg32 g(random_seed, (uint64_t)p);
    r = g.next_i31();
  }
  const int64_t nid   = (int64_t)seeds[i];
  const int64_t start = row_ptr[nid];
  const int64_t N     = row_ptr[nid + 1] - start;
  if (N <= 0) return;
  const int64_t a   = start + (int64_t)(r % N);
  const int64_t out = (int64_t)offsets[i] + t;
  dst[out]          = col[a];
  if (src_lid) src_lid[out] = i;
  if (edge_gid) edge_gid[out] = a;
}

template <typename SeedT, typename ColT>
void run(const int64_t* row_ptr, const void* col_, const void* seeds_, int n, int M, int* offsets, void* dst_ctx, void* lid_ctx,
         void* gid_ctx, uint64_t random_seed, wholememory_dtype_t col_dt, wholememory_env_func_t* env, hipStream_t stream)
{
  const auto* col   = static_cast<const ColT*>(col_);
  const auto* seeds = static_cast<const SeedT*>(seeds_);
  temp_buffer cnt_buf(env), scan_tmp(env);
  int* cnt  = cnt_buf.device<int>(n + 1, WHOLEMEMORY_DT_INT);
  int* stmp = scan_tmp.device<int>(scan_tmp_ints(n + 1), WHOLEMEMORY_DT_INT);
  if (n > 0) replace_count_kernel<SeedT><<<ceil_div(n, 256), 256, 0, stream>>>(row_ptr, seeds, n, M, cnt);
  WG_HIP_CHECK(hipGetLastError());
  exclusive_scan_i32(cnt, offsets, n, stmp, stream);
  int total = 0;
  WG_HIP_CHECK(hipMemcpyAsync(&total, offsets + n, sizeof(int), hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // output sizes
  auto* dst = static_cast<ColT*>(output_alloc(env, dst_ctx, total, col_dt));
  int* lid  = lid_ctx ? static_cast<int*>(output_alloc(env, lid_ctx, total, WHOLEMEMORY_DT_INT)) : nullptr;
  auto* gid = gid_ctx ? static_cast<int64_t*>(output_alloc(env, gid_ctx, total, WHOLEMEMORY_DT_INT64)) : nullptr;
  if (n == 0 || total == 0) return;
  replace_sample_kernel<SeedT, ColT><<<ceil_div((int64_t)n * M, 256), 256, 0, stream>>>(row_ptr, col, seeds, n, M, random_seed,
                                                                                       offsets, dst, lid, gid);
  WG_HIP_CHECK(hipGetLastError());
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // outputs complete on return, scratch released (the ABI ops' contract)
}

}  // namespace
}  // namespace wgamd

extern "C" wholememory_error_code_t wgamd_csr_uniform_sample_with_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor, int sample_count, wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context, void* output_center_localid_memory_context, void* output_edge_gid_memory_context,
  unsigned long long random_seed, wholememory_env_func_t* p_env_fns, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_csr_uniform_sample_with_replacement", [&] {
    WG_REQUIRE_INPUT(wm_csr_row_ptr_tensor && wm_csr_col_ptr_tensor && center_nodes_tensor && output_sample_offset_tensor &&
                       p_env_fns,
                     "null tensor / env");
    WG_REQUIRE_INPUT(output_dest_memory_context != nullptr, "output_dest_memory_context must not be NULL");
    WG_REQUIRE_INPUT(sample_count > 0, "sampling with replacement needs a positive sample count");
    auto rd = wm_csr_row_ptr_tensor->desc, cd = wm_csr_col_ptr_tensor->desc, sd = center_nodes_tensor->desc,
         od = output_sample_offset_tensor->desc;
    WG_REQUIRE_INPUT(rd.dim == 1 && cd.dim == 1 && sd.dim == 1 && od.dim == 1, "all tensors must be 1-D");
    WG_REQUIRE_INPUT(!wm_csr_row_ptr_tensor->handle && !wm_csr_col_ptr_tensor->handle, "CSR tensors must wrap device pointers");
    WG_EXPECTS(rd.dtype == WHOLEMEMORY_DT_INT64, "csr_row_ptr dtype must be INT64, got %d", (int)rd.dtype);
    WG_EXPECTS(od.dtype == WHOLEMEMORY_DT_INT, "output_sample_offset dtype must be INT, got %d", (int)od.dtype);
    WG_REQUIRE_INPUT(cd.dtype == WHOLEMEMORY_DT_INT || cd.dtype == WHOLEMEMORY_DT_INT64, "csr_col dtype must be INT|INT64");
    WG_REQUIRE_INPUT(sd.dtype == WHOLEMEMORY_DT_INT || sd.dtype == WHOLEMEMORY_DT_INT64, "center_nodes dtype must be INT|INT64");
    WG_REQUIRE_INPUT(od.sizes[0] == sd.sizes[0] + 1, "output_sample_offset must have center_node_count+1 entries");
    WG_REQUIRE_INPUT(sd.sizes[0] * (int64_t)sample_count < ((int64_t)1 << 31), "too many samples for one call");
    const int n       = (int)sd.sizes[0];
    const auto* rp    = static_cast<const int64_t*>(tensor_data(wm_csr_row_ptr_tensor));
    const void* col   = tensor_data(wm_csr_col_ptr_tensor);
    const void* seeds = tensor_data(center_nodes_tensor);
    int* offsets      = static_cast<int*>(tensor_data(output_sample_offset_tensor));
    auto st           = static_cast<hipStream_t>(stream);
    const bool s64 = sd.dtype == WHOLEMEMORY_DT_INT64, c64 = cd.dtype == WHOLEMEMORY_DT_INT64;
#define WG_GO(S, C)                                                                                                        \
  run<S, C>(rp, col, seeds, n, sample_count, offsets, output_dest_memory_context, output_center_localid_memory_context,    \
            output_edge_gid_memory_context, (uint64_t)random_seed, cd.dtype, p_env_fns, st)
    if (s64 && c64) WG_GO(int64_t, int64_t);
    else if (s64) WG_GO(int64_t, int32_t);
    else if (c64) WG_GO(int32_t, int64_t);
    else WG_GO(int32_t, int32_t);
#undef WG_GO
  });
}
