// Communicator, DISTRIBUTED memory handles and the all-to-all feature fetch behind
// wholememory_gather / wholememory_scatter for handle-backed tensors (include/wgamd_comm.h).
//
// Algorithm of the row exchange = the reference's NCCL gather
// (/root/reference/cpp/src/wholememory_ops/gather_op_impl_nccl.cu:23-171, functions/bucket_ids_func.cu:20-129,
// functions/exchange_ids_nccl_func.cu:32-215, functions/exchange_embeddings_nccl_func.cu:23-65, collectives of
// cpp/src/wholememory/nccl_comms.cpp:345-426): owner rank of every index -> counts all-to-all -> indices
// all-to-all-v -> local gather -> rows all-to-all-v -> un-permute.  Differences by design:
//   * ids are grouped by owner with ONE radix pass over ceil(log2 W) key bits (stable: the caller's order survives
//     inside a bucket, so results are reproducible); the caller's own bucket is grouped last and never exchanged;
//   * RCCL is resolved with dlopen at run time — inside a PyTorch process this shares torch's librccl, and the
//     library still loads on a CPU-only box;
//   * only DISTRIBUTED/DEVICE memory exists: on an 8 x MI355X node every pair of GPUs has its own xGMI link,
//     a grouped send/recv all-to-all drives all 7 links at once, and 288 GB of HBM per GPU removes the need
//     for the host-pinned / VMM-mapped variants.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <rocprim/rocprim.hpp>
#include <vector>

#include "wg_common.hpp"
#include "wgamd_comm.h"

struct wholememory_comm_ {
  ncclComm_t nccl = nullptr;
  int rank = 0, size = 1;
};

struct wholememory_handle_ {
  wholememory_comm_t comm;
  wholememory_memory_type_t type;
  wholememory_memory_location_t location;
  size_t total_size, granularity;
  std::vector<size_t> byte_offsets;  // W+1, partition of [0, total_size) in bytes
  void* local_ptr;
};

namespace wgamd {
namespace {

// ---- RCCL, resolved lazily -------------------------------------------------------------------------
struct rccl_api {
  decltype(&ncclGetUniqueId) GetUniqueId   = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy   = nullptr;
  decltype(&ncclAllReduce) AllReduce       = nullptr;
  decltype(&ncclSend) Send                 = nullptr;
  decltype(&ncclRecv) Recv                 = nullptr;
  decltype(&ncclGroupStart) GroupStart     = nullptr;
  decltype(&ncclGroupEnd) GroupEnd         = nullptr;
  decltype(&ncclGetErrorString) ErrString  = nullptr;
  bool ok                                  = false;
};

rccl_api& rccl()
{
  static rccl_api api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    // WGAMD_RCCL_LIBRARY picks a specific RCCL build (the tests use it to load an in-process stand-in)
    if (const char* forced = getenv("WGAMD_RCCL_LIBRARY")) {
      h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
      if (!h) fprintf(stderr, "[wholegraph_amd] WGAMD_RCCL_LIBRARY=%s: %s\n", forced, dlerror());
    } else {
      for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    }
    if (!h) return;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    api.GetUniqueId  = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy  = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllReduce    = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.Send         = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv         = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart   = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd     = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.ErrString    = reinterpret_cast<decltype(api.ErrString)>(sym("ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Send && api.Recv &&
             api.GroupStart && api.GroupEnd;
  });
  return api;
}

#define WG_NCCL_CHECK(expr)                                                                                   \
  do {                                                                                                        \
    ncclResult_t r__ = (expr);                                                                                \
    if (r__ != ncclSuccess)                                                                                   \
      throw ::wgamd::comm_error(::wgamd::fmt("%s:%d RCCL error %d (%s) in %s", __FILE__, __LINE__, (int)r__,  \
                                             rccl().ErrString ? rccl().ErrString(r__) : "?", #expr));          \
  } while (0)

// variable all-to-all of bytes, explicit per-peer offsets and sizes in BYTES (nccl_comms.cpp:398-426: grouped ncclRecv x W
// then ncclSend x W); a peer with zero bytes is skipped on that side
void alltoallv_bytes(wholememory_comm_t comm, const char* send, const std::vector<size_t>& send_off,
                     const std::vector<size_t>& send_bytes, char* recv, const std::vector<size_t>& recv_off,
                     const std::vector<size_t>& recv_bytes, hipStream_t stream)
{
  auto& api = rccl();
  WG_NCCL_CHECK(api.GroupStart());
  for (int r = 0; r < comm->size; r++)
    if (recv_bytes[r]) WG_NCCL_CHECK(api.Recv(recv + recv_off[r], recv_bytes[r], ncclInt8, r, comm->nccl, stream));
  for (int r = 0; r < comm->size; r++)
    if (send_bytes[r]) WG_NCCL_CHECK(api.Send(send + send_off[r], send_bytes[r], ncclInt8, r, comm->nccl, stream));
  WG_NCCL_CHECK(api.GroupEnd());
}

// ---- kernels ---------------------------------------------------------------------------------------
constexpr int kMaxRanks = 1024;

// owner rank of an entry: last r with entry_offsets[r] <= id (negative ids -> last rank, they stay negative)
__device__ __forceinline__ int owner_of(int64_t id, const int64_t* entry_offsets, int W)
{
  if (id < 0) return W - 1;
  int lo = 0, hi = W;  // invariant: offsets[lo] <= id < offsets[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (entry_offsets[mid] <= id) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename IdxT>
__global__ void __launch_bounds__(256)
owner_histogram_kernel(const IdxT* __restrict__ idx, int64_t n, int64_t row0,
                       const int64_t* __restrict__ entry_offsets, int W, int* __restrict__ counts)
{
  __shared__ int local[kMaxRanks];
  for (int r = threadIdx.x; r < W; r += blockDim.x) local[r] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t id = (int64_t)idx[i];
    atomicAdd(&local[owner_of(id < 0 ? id : id + row0, entry_offsets, W)], 1);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < W; r += blockDim.x)
    if (local[r]) atomicAdd(&counts[r], local[r]);
}

// Grouping ids by owner is a STABLE partition: inside a bucket the ids keep the caller's order, so everything
// downstream (which duplicate wins a scatter, the order gradients of one row are summed in) is reproducible run to run.
// Sort key of id i = its owner rotated so that MY bucket comes last (me+1, me+2, ..., me-1, me); one radix pass over
// ceil(log2 W) bits sorts (key, position) pairs; a single-rank communicator needs no sort at all.
template <typename IdxT>
__global__ void __launch_bounds__(256)
owner_keys_kernel(const IdxT* __restrict__ idx, int64_t n, int64_t row0, const int64_t* __restrict__ entry_offsets, int W, int me,
                  uint32_t* __restrict__ keys, int* __restrict__ vals)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = (int64_t)idx[i];
  const int r      = owner_of(id < 0 ? id : id + row0, entry_offsets, W);
  keys[i]          = (uint32_t)((r - me - 1 + W) % W);
  vals[i]          = (int)i;
}

// grouped_ids[j] = id of the j-th pair of the sorted order (`order` == nullptr: identity), positions[j] = where it came from
template <typename IdxT>
__global__ void __launch_bounds__(256)
emit_grouped_kernel(const IdxT* __restrict__ idx, int64_t n, int64_t row0, const int* __restrict__ order,
                    int64_t* __restrict__ grouped_ids, int64_t* __restrict__ positions)
{
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int64_t i = order ? (int64_t)order[j] : j;
  int64_t id      = (int64_t)idx[i];
  if (id >= 0) id += row0;  // row 0 of a sub-tensor is entry `row0` of the handle
  grouped_ids[j] = id;
  positions[j]   = id < 0 ? -1 : i;  // a negative index leaves its dense row untouched
}

__global__ void __launch_bounds__(256) localize_ids_kernel(int64_t* ids, int64_t n, int64_t local_start)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ids[i] >= 0) ids[i] -= local_start;
}

}  // namespace

void id_exchange::plan(wholememory_handle_t h, size_t entry_bytes, int64_t row0, const void* idx, wholememory_dtype_t idx_dtype,
                       int64_t n_, bool self_direct_, hipStream_t stream)
{
  comm = h->comm;
  W    = comm->size;
  me   = comm->rank;
  n    = n_;
  self_direct = self_direct_;
  WG_EXPECTS(W <= kMaxRanks, "too many ranks");
  std::vector<int64_t> entry_offsets(W + 1);
  for (int r = 0; r <= W; r++) entry_offsets[r] = (int64_t)(h->byte_offsets[r] / entry_bytes);
  local_start = entry_offsets[me];
  local_rows  = entry_offsets[me + 1] - local_start;

  WG_EXPECTS(n < (int64_t)1 << 31, "too many indices in one call");
  unsigned key_bits = 0;
  while ((1 << key_bits) < W) key_bits++;
  size_t sort_bytes = 0;
  const int64_t n_sort = W > 1 ? n : 0;  // a single owner: the grouped order is the caller's order
  if (n_sort > 0)
    WG_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int*)nullptr,
                                           (int*)nullptr, (size_t)n_sort, 0u, key_bits, stream));
  const size_t o_offs = scratch.add(sizeof(int64_t) * (W + 1)), o_cnt = scratch.add(sizeof(int) * W),
               o_x = scratch.add(sizeof(int64_t) * 2 * W), o_gid = scratch.add(sizeof(int64_t) * n),
               o_pos = scratch.add(sizeof(int64_t) * n), o_k1 = scratch.add(sizeof(uint32_t) * n_sort),
               o_k2 = scratch.add(sizeof(uint32_t) * n_sort), o_v1 = scratch.add(sizeof(int) * n_sort),
               o_v2 = scratch.add(sizeof(int) * n_sort), o_tmp = scratch.add(sort_bytes);
  scratch.commit();
  // ---- 1. owners + counts --------------------------------------------------------------------
  auto* d_offsets = scratch.at<int64_t>(o_offs);
  int* d_counts   = scratch.at<int>(o_cnt);
  WG_HIP_CHECK(hipMemcpyAsync(d_offsets, entry_offsets.data(), sizeof(int64_t) * (W + 1), hipMemcpyHostToDevice, stream));
  WG_HIP_CHECK(hipMemsetAsync(d_counts, 0, sizeof(int) * W, stream));
  if (n > 0) {
    int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    if (idx_dtype == WHOLEMEMORY_DT_INT)
      owner_histogram_kernel<int32_t><<<grid, 256, 0, stream>>>(static_cast<const int32_t*>(idx), n, row0, d_offsets, W, d_counts);
    else
      owner_histogram_kernel<int64_t><<<grid, 256, 0, stream>>>(static_cast<const int64_t*>(idx), n, row0, d_offsets, W, d_counts);
    WG_HIP_CHECK(hipGetLastError());
  }
  std::vector<int> h_counts(W);
  WG_HIP_CHECK(hipMemcpyAsync(h_counts.data(), d_counts, sizeof(int) * W, hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));

  // ---- 2. counts all-to-all (W x int64; nothing to trade on a single-rank communicator) --------
  send_cnt.assign(W, 0);
  recv_cnt.assign(W, 0);
  for (int r = 0; r < W; r++) send_cnt[r] = (size_t)h_counts[r];
  if (W == 1) {
    recv_cnt[0] = send_cnt[0];
  } else {
    auto* d_x = scratch.at<int64_t>(o_x);
    std::vector<int64_t> tmp(send_cnt.begin(), send_cnt.end());
    WG_HIP_CHECK(hipMemcpyAsync(d_x, tmp.data(), sizeof(int64_t) * W, hipMemcpyHostToDevice, stream));
    std::vector<size_t> eight(W, sizeof(int64_t)), at(W);
    for (int r = 0; r < W; r++) at[r] = (size_t)r * sizeof(int64_t);
    alltoallv_bytes(comm, reinterpret_cast<const char*>(d_x), at, eight, reinterpret_cast<char*>(d_x + W), at, eight, stream);
    WG_HIP_CHECK(hipMemcpyAsync(tmp.data(), d_x + W, sizeof(int64_t) * W, hipMemcpyDeviceToHost, stream));
    WG_HIP_CHECK(hipStreamSynchronize(stream));
    for (int r = 0; r < W; r++) recv_cnt[r] = (size_t)tmp[r];
  }

  // ---- 3. group ids by owner, stable: the peers' buckets first (me+1, me+2, ... wrapping around), MY bucket last ----
  self_cnt = (int64_t)send_cnt[me];
  WG_EXPECTS(recv_cnt[me] == send_cnt[me], "self count mismatch");
  bucket_start.assign(W, 0);
  int64_t acc = 0;
  for (int k = 1; k <= W; k++) {
    const int r     = (me + k) % W;
    bucket_start[r] = acc;
    acc += (int64_t)send_cnt[r];
  }
  n_remote  = self_direct ? bucket_start[me] : n;  // leading rows of the grouped order that go through the exchange
  d_grouped = scratch.at<int64_t>(o_gid);
  d_pos     = scratch.at<int64_t>(o_pos);
  if (n > 0) {
    const int grid   = (int)((n + 255) / 256);
    const int* order = nullptr;
    if (n_sort > 0) {
      auto *k1 = scratch.at<uint32_t>(o_k1), *k2 = scratch.at<uint32_t>(o_k2);
      auto *v1 = scratch.at<int>(o_v1), *v2 = scratch.at<int>(o_v2);
      if (idx_dtype == WHOLEMEMORY_DT_INT)
        owner_keys_kernel<int32_t><<<grid, 256, 0, stream>>>(static_cast<const int32_t*>(idx), n, row0, d_offsets, W, me, k1, v1);
      else
        owner_keys_kernel<int64_t><<<grid, 256, 0, stream>>>(static_cast<const int64_t*>(idx), n, row0, d_offsets, W, me, k1, v1);
      WG_HIP_CHECK(hipGetLastError());
      WG_HIP_CHECK(rocprim::radix_sort_pairs(scratch.at<void>(o_tmp), sort_bytes, k1, k2, v1, v2, (size_t)n, 0u, key_bits, stream));
      order = v2;
    }
    if (idx_dtype == WHOLEMEMORY_DT_INT)
      emit_grouped_kernel<int32_t><<<grid, 256, 0, stream>>>(static_cast<const int32_t*>(idx), n, row0, order, d_grouped, d_pos);
    else
      emit_grouped_kernel<int64_t><<<grid, 256, 0, stream>>>(static_cast<const int64_t*>(idx), n, row0, order, d_grouped, d_pos);
    WG_HIP_CHECK(hipGetLastError());
  }
  // what crosses the wire: packed in rank order on the receiving side (without my own bucket when it stays home)
  send_n.assign(W, 0); recv_n.assign(W, 0); send_at.assign(W, 0); recv_at.assign(W, 0);
  recv_total = 0;
  for (int r = 0; r < W; r++) {
    const bool skip = self_direct && r == me;
    send_n[r]  = skip ? 0 : send_cnt[r];
    recv_n[r]  = skip ? 0 : recv_cnt[r];
    send_at[r] = (size_t)bucket_start[r];
    recv_at[r] = (size_t)recv_total;
    recv_total += (int64_t)recv_n[r];
  }
  so.resize(W); sb.resize(W); ro.resize(W); rb.resize(W);
  d_self_ids = d_grouped + bucket_start[me];  // not part of any send when self_direct
  d_self_pos = d_pos + bucket_start[me];
}

void id_exchange::exchange_ids(int64_t* recv_ids, hipStream_t stream)
{
  d_recv_ids = recv_ids;
  scaled(sizeof(int64_t), sizeof(int64_t), send_n, send_at, recv_n, recv_at);
  alltoallv_bytes(comm, reinterpret_cast<const char*>(d_grouped), so, sb, reinterpret_cast<char*>(d_recv_ids), ro, rb, stream);
  if (recv_total > 0) {
    localize_ids_kernel<<<(int)((recv_total + 255) / 256), 256, 0, stream>>>(d_recv_ids, recv_total, local_start);
    WG_HIP_CHECK(hipGetLastError());
  }
  if (self_direct && self_cnt > 0) {
    localize_ids_kernel<<<(int)((self_cnt + 255) / 256), 256, 0, stream>>>(d_self_ids, self_cnt, local_start);
    WG_HIP_CHECK(hipGetLastError());
  }
}

// byte offsets / sizes of one all-to-all-v from per-peer counts and positions (units of rows or ids)
void id_exchange::scaled(size_t unit_send, size_t unit_recv, const std::vector<size_t>& s_n, const std::vector<size_t>& s_at,
                         const std::vector<size_t>& r_n, const std::vector<size_t>& r_at)
{
  for (int r = 0; r < W; r++) {
    so[r] = s_at[r] * unit_send; sb[r] = s_n[r] * unit_send;
    ro[r] = r_at[r] * unit_recv; rb[r] = r_n[r] * unit_recv;
  }
}

void id_exchange::rows_to_owners(const char* send, char* recv, size_t row_bytes, hipStream_t stream)
{
  scaled(row_bytes, row_bytes, send_n, send_at, recv_n, recv_at);
  alltoallv_bytes(comm, send, so, sb, recv, ro, rb, stream);
}

void id_exchange::rows_to_askers(const char* send, char* recv, size_t row_bytes, hipStream_t stream)
{
  scaled(row_bytes, row_bytes, recv_n, recv_at, send_n, send_at);
  alltoallv_bytes(comm, send, so, sb, recv, ro, rb, stream);
}

void distributed_rows_op(bool scatter, wholememory_handle_t h, wholememory_matrix_description_t tm, const void* idx,
                         wholememory_dtype_t idx_dtype, int64_t n, char* dense, wholememory_matrix_description_t dense_m,
                         wholememory_env_func_t* env, hipStream_t stream)
{
  WG_EXPECTS(h->type == WHOLEMEMORY_MT_DISTRIBUTED || h->comm->size == 1, "unsupported memory type");
  const size_t tes         = dtype_size(tm.dtype);
  const size_t entry_bytes = (size_t)tm.stride * tes;
  WG_EXPECTS(h->granularity == entry_bytes, "tensor row stride (%zu B) != handle granularity (%zu B)", entry_bytes,
             h->granularity);
  const int64_t row0 = tm.storage_offset / tm.stride;  // sub-tensor views: first row / first column of the view
  const int64_t col0 = tm.storage_offset % tm.stride;
  WG_REQUIRE_INPUT(tm.storage_offset >= 0 && col0 + tm.sizes[1] <= tm.stride, "bad storage offset");

  // Rows I own never enter the exchange when no dtype conversion is asked for: one permuting copy moves them between my
  // partition and the dense rows (1/W of the traffic; all of it on a single-rank communicator).
  id_exchange x(env);
  x.plan(h, entry_bytes, row0, idx, idx_dtype, n, tm.dtype == dense_m.dtype, stream);
  const int64_t F   = tm.sizes[1];
  const size_t des  = dtype_size(dense_m.dtype);
  // gather: rows I look up for the peers (output dtype) + the rows that come back; scatter: my rows grouped by owner +
  // the rows the peers send me (table dtype)
  const size_t recv_row = (size_t)F * (scatter ? tes : des), send_row = recv_row;
  temp_arena arena(env);
  const size_t o_ids = arena.add(sizeof(int64_t) * x.recv_total), o_rows = arena.add(recv_row * x.recv_total),
               o_back = arena.add(send_row * x.n_remote);
  arena.commit();
  x.exchange_ids(arena.at<int64_t>(o_ids), stream);

  // local partition viewed as a matrix of its own rows
  wholememory_matrix_description_t local_m = tm;
  local_m.sizes[0]                         = x.local_rows;
  local_m.storage_offset                   = 0;  // the row kernels take a pointer to the first element
  const char* local_base                   = static_cast<const char*>(h->local_ptr) + (size_t)col0 * tes;
  wholememory_matrix_description_t dense0  = dense_m;
  dense0.storage_offset                    = 0;  // `dense` already points at the first element
  int64_t sz2[2];

  if (!scatter) {
    // ---- 5. local gather (table dtype -> output dtype), 6. rows back, 7. un-permute ------------
    if (x.self_direct) local_rows_permute(local_base, local_m, x.d_self_ids, x.d_self_pos, x.self_cnt, dense, dense0, stream);
    sz2[0] = x.recv_total; sz2[1] = F;
    wholememory_matrix_description_t rows_m = wholememory_create_matrix_desc(sz2, F, 0, dense_m.dtype);
    char* d_rows = arena.at<char>(o_rows);
    local_rows_gather(local_base, local_m, x.d_recv_ids, WHOLEMEMORY_DT_INT64, x.recv_total, d_rows, rows_m, stream);
    sz2[0] = x.n_remote;
    wholememory_matrix_description_t back_m = wholememory_create_matrix_desc(sz2, F, 0, dense_m.dtype);
    char* d_back = arena.at<char>(o_back);
    x.rows_to_askers(d_rows, d_back, (size_t)F * des, stream);
    local_rows_scatter(d_back, back_m, x.d_pos, WHOLEMEMORY_DT_INT64, x.n_remote, dense, dense0, stream);
  } else {
    // ---- scatter: permute my rows by owner, send ids + rows, owners write them -------------------
    if (x.self_direct)
      local_rows_permute(dense, dense0, x.d_self_pos, x.d_self_ids, x.self_cnt, const_cast<char*>(local_base), local_m, stream);
    sz2[0] = x.n_remote; sz2[1] = F;
    wholememory_matrix_description_t send_m = wholememory_create_matrix_desc(sz2, F, 0, tm.dtype);
    char* d_send = arena.at<char>(o_back);
    local_rows_gather(dense, dense0, x.d_pos, WHOLEMEMORY_DT_INT64, x.n_remote, d_send, send_m, stream);  // also converts
    sz2[0] = x.recv_total;
    wholememory_matrix_description_t recv_m = wholememory_create_matrix_desc(sz2, F, 0, tm.dtype);
    char* d_recv = arena.at<char>(o_rows);
    x.rows_to_owners(d_send, d_recv, (size_t)F * tes, stream);
    local_rows_scatter(d_recv, recv_m, x.d_recv_ids, WHOLEMEMORY_DT_INT64, x.recv_total, const_cast<char*>(local_base), local_m,
                       stream);
  }
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // scratch is released on return
}

}  // namespace wgamd

// ------------------------------------------------------------------------------------------------------
extern "C" {

using namespace wgamd;

static std::atomic<int> g_log_level{3};

wholememory_error_code_t wholememory_init(unsigned int /*flags*/, int log_level)
{
  g_log_level = log_level;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_finalize(void) { return WHOLEMEMORY_SUCCESS; }

wholememory_error_code_t wholememory_create_unique_id(wholememory_unique_id_t* unique_id)
{
  static_assert(sizeof(ncclUniqueId) <= WHOLEMEMORY_UNIQUE_ID_BYTES, "unique id does not fit");
  if (unique_id == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (!rccl().ok) {
    fprintf(stderr, "[wholegraph_amd] wholememory_create_unique_id: librccl.so not found\n");
    return WHOLEMEMORY_COMMUNICATION_ERROR;
  }
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return WHOLEMEMORY_COMMUNICATION_ERROR;
  memset(unique_id->internal, 0, WHOLEMEMORY_UNIQUE_ID_BYTES);
  memcpy(unique_id->internal, &id, sizeof(id));
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_create_communicator(wholememory_comm_t* comm, wholememory_unique_id_t unique_id,
                                                         int rank, int size)
{
  if (comm == nullptr || size < 1 || rank < 0 || rank >= size) return WHOLEMEMORY_INVALID_INPUT;
  if (!rccl().ok) {
    fprintf(stderr, "[wholegraph_amd] wholememory_create_communicator: librccl.so not found\n");
    return WHOLEMEMORY_COMMUNICATION_ERROR;
  }
  ncclUniqueId id;
  memcpy(&id, unique_id.internal, sizeof(id));
  auto* c = new wholememory_comm_;
  c->rank = rank;
  c->size = size;
  ncclResult_t r = rccl().CommInitRank(&c->nccl, size, id, rank);
  if (r != ncclSuccess) {
    fprintf(stderr, "[wholegraph_amd] ncclCommInitRank failed: %s\n", rccl().ErrString ? rccl().ErrString(r) : "?");
    delete c;
    return WHOLEMEMORY_COMMUNICATION_ERROR;
  }
  *comm = c;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_destroy_communicator(wholememory_comm_t comm)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (comm->nccl) rccl().CommDestroy(comm->nccl);
  delete comm;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_communicator_support_type_location(wholememory_comm_t comm,
                                                                        wholememory_memory_type_t memory_type,
                                                                        wholememory_memory_location_t memory_location)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (memory_location != WHOLEMEMORY_ML_DEVICE) return WHOLEMEMORY_NOT_SUPPORTED;
  if (memory_type == WHOLEMEMORY_MT_DISTRIBUTED) return WHOLEMEMORY_SUCCESS;
  if (comm->size == 1 && (memory_type == WHOLEMEMORY_MT_CONTINUOUS || memory_type == WHOLEMEMORY_MT_CHUNKED))
    return WHOLEMEMORY_SUCCESS;
  return WHOLEMEMORY_NOT_SUPPORTED;
}

wholememory_error_code_t wholememory_communicator_get_rank(int* rank, wholememory_comm_t comm)
{
  if (!rank || !comm) return WHOLEMEMORY_INVALID_INPUT;
  *rank = comm->rank;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_communicator_get_size(int* size, wholememory_comm_t comm)
{
  if (!size || !comm) return WHOLEMEMORY_INVALID_INPUT;
  *size = comm->size;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_communicator_barrier(wholememory_comm_t comm)
{
  // 1-int allreduce + sync, as nccl_comms.cpp:71-75
  return guarded("wholememory_communicator_barrier", [&] {
    WG_REQUIRE_INPUT(comm != nullptr, "null communicator");
    int* d = nullptr;
    WG_HIP_CHECK(hipMalloc(&d, sizeof(int)));
    WG_HIP_CHECK(hipMemset(d, 0, sizeof(int)));
    ncclResult_t r = rccl().AllReduce(d, d, 1, ncclInt32, ncclSum, comm->nccl, nullptr);
    hipError_t e   = hipStreamSynchronize(nullptr);
    (void)hipFree(d);
    if (r != ncclSuccess) throw comm_error("allreduce failed");
    WG_HIP_CHECK(e);
  });
}

wholememory_error_code_t wholememory_equal_entry_partition_plan(size_t* entry_per_rank, size_t total_entry_count,
                                                                int world_size)
{
  if (entry_per_rank == nullptr || world_size < 1) return WHOLEMEMORY_INVALID_INPUT;
  *entry_per_rank = (total_entry_count + (size_t)world_size - 1) / (size_t)world_size;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_malloc(wholememory_handle_t* handle_ptr, size_t total_size, wholememory_comm_t comm,
                                            wholememory_memory_type_t memory_type,
                                            wholememory_memory_location_t memory_location, size_t data_granularity,
                                            size_t* rank_entry_partition)
{
  return guarded("wholememory_malloc", [&] {
    WG_REQUIRE_INPUT(handle_ptr && comm && data_granularity > 0, "null argument / zero granularity");
    WG_REQUIRE_INPUT(total_size % data_granularity == 0, "total_size is not a multiple of data_granularity");
    if (wholememory_communicator_support_type_location(comm, memory_type, memory_location) != WHOLEMEMORY_SUCCESS)
      throw logic_error("memory type / location not supported: only DISTRIBUTED on DEVICE (see wgamd_comm.h)");
    const size_t entries = total_size / data_granularity;
    auto* h              = new wholememory_handle_;
    h->comm = comm; h->type = memory_type; h->location = memory_location;
    h->total_size = total_size; h->granularity = data_granularity; h->local_ptr = nullptr;
    h->byte_offsets.assign(comm->size + 1, 0);
    if (rank_entry_partition != nullptr) {
      size_t acc = 0;
      for (int r = 0; r < comm->size; r++) {
        h->byte_offsets[r] = acc * data_granularity;
        acc += rank_entry_partition[r];
      }
      if (acc != entries) {
        delete h;
        throw invalid_input("rank_entry_partition does not add up to the entry count");
      }
      h->byte_offsets[comm->size] = total_size;
    } else {
      size_t per = (entries + comm->size - 1) / comm->size;
      for (int r = 0; r <= comm->size; r++) h->byte_offsets[r] = std::min(entries, per * (size_t)r) * data_granularity;
    }
    size_t local = h->byte_offsets[comm->rank + 1] - h->byte_offsets[comm->rank];
    if (local > 0 && hipMalloc(&h->local_ptr, local) != hipSuccess) {
      delete h;
      throw std::bad_alloc();
    }
    *handle_ptr = h;
  });
}

wholememory_error_code_t wholememory_free(wholememory_handle_t h)
{
  if (h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (h->local_ptr) (void)hipFree(h->local_ptr);
  delete h;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_get_communicator(wholememory_comm_t* comm, wholememory_handle_t h)
{
  if (!comm || !h) return WHOLEMEMORY_INVALID_INPUT;
  *comm = h->comm;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_memory_type_t wholememory_get_memory_type(wholememory_handle_t h) { return h ? h->type : WHOLEMEMORY_MT_NONE; }
wholememory_memory_location_t wholememory_get_memory_location(wholememory_handle_t h)
{
  return h ? h->location : WHOLEMEMORY_ML_NONE;
}
size_t wholememory_get_total_size(wholememory_handle_t h) { return h ? h->total_size : 0; }
size_t wholememory_get_data_granularity(wholememory_handle_t h) { return h ? h->granularity : 0; }

wholememory_error_code_t wholememory_get_local_memory(void** local_ptr, size_t* local_size, size_t* local_offset,
                                                      wholememory_handle_t h)
{
  if (!h || !local_ptr || !local_size || !local_offset) return WHOLEMEMORY_INVALID_INPUT;
  *local_ptr    = h->local_ptr;
  *local_offset = h->byte_offsets[h->comm->rank];
  *local_size   = h->byte_offsets[h->comm->rank + 1] - h->byte_offsets[h->comm->rank];
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_get_rank_partition_sizes(size_t* sizes, wholememory_handle_t h)
{
  if (!h || !sizes) return WHOLEMEMORY_INVALID_INPUT;
  for (int r = 0; r < h->comm->size; r++) sizes[r] = h->byte_offsets[r + 1] - h->byte_offsets[r];
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_get_rank_partition_offsets(size_t* offsets, wholememory_handle_t h)
{
  if (!h || !offsets) return WHOLEMEMORY_INVALID_INPUT;
  for (int r = 0; r <= h->comm->size; r++) offsets[r] = h->byte_offsets[r];
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_make_tensor_from_handle(wholememory_tensor_t* out, wholememory_handle_t h,
                                                             wholememory_tensor_description_t* desc)
{
  if (!out || !h || !desc) return WHOLEMEMORY_INVALID_INPUT;
  if (desc->dim != 1 && desc->dim != 2) return WHOLEMEMORY_INVALID_VALUE;
  wholememory_tensor_t t = nullptr;
  auto rc                = wholememory_make_tensor_from_pointer(&t, nullptr, desc);
  if (rc != WHOLEMEMORY_SUCCESS) return rc;
  t->handle = h;
  *out      = t;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_create_tensor(wholememory_tensor_t* out, wholememory_tensor_description_t* desc,
                                                   wholememory_comm_t comm, wholememory_memory_type_t memory_type,
                                                   wholememory_memory_location_t memory_location,
                                                   size_t* tensor_entry_partition)
{
  if (!out || !desc || !comm) return WHOLEMEMORY_INVALID_INPUT;
  if (desc->dim != 1 && desc->dim != 2) {
    fprintf(stderr, "[wholegraph_amd] wholememory_create_tensor: only 1-D / 2-D tensors\n");
    return WHOLEMEMORY_INVALID_VALUE;
  }
  if (desc->storage_offset != 0) return WHOLEMEMORY_INVALID_VALUE;
  const size_t es      = wholememory_dtype_get_element_size(desc->dtype);
  const size_t rowsize = (size_t)(desc->dim == 2 ? desc->strides[0] : 1) * es;
  wholememory_handle_t h = nullptr;
  auto rc = wholememory_malloc(&h, (size_t)desc->sizes[0] * rowsize, comm, memory_type, memory_location, rowsize,
                               tensor_entry_partition);
  if (rc != WHOLEMEMORY_SUCCESS) return rc;
  rc = wholememory_make_tensor_from_handle(out, h, desc);
  if (rc != WHOLEMEMORY_SUCCESS) {
    wholememory_free(h);
    return rc;
  }
  (*out)->owns_handle = true;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_get_local_entry_count(size_t* count, wholememory_tensor_t t)
{
  if (!count || !t || !t->handle) return WHOLEMEMORY_INVALID_INPUT;
  auto* h = t->handle;
  *count  = (h->byte_offsets[h->comm->rank + 1] - h->byte_offsets[h->comm->rank]) / h->granularity;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_get_local_entry_start(size_t* start, wholememory_tensor_t t)
{
  if (!start || !t || !t->handle) return WHOLEMEMORY_INVALID_INPUT;
  *start = t->handle->byte_offsets[t->handle->comm->rank] / t->handle->granularity;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_map_local_tensor(wholememory_tensor_t t, wholememory_tensor_t* local_tensor)
{
  if (!t || !local_tensor || !t->handle) return WHOLEMEMORY_INVALID_INPUT;
  size_t n = 0;
  wholememory_tensor_get_local_entry_count(&n, t);
  wholememory_tensor_description_t d = t->desc;
  d.sizes[0]                         = (int64_t)n;
  return wholememory_make_tensor_from_pointer(local_tensor, t->handle->local_ptr, &d);
}

}  // extern "C"
