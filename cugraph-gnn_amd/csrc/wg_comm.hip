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
//   * the exchange costs ONE host synchronisation per call (both count vectors come back in one pinned read-back; the
//     reference synchronises three times, gather_op_impl_nccl.cu:60-150), and none at the end: scratch comes from the
//     caller's stream-ordered allocator;
//   * memory types: DISTRIBUTED (each rank's rows in its own HBM, rows travel by all-to-all-v) and the PEER-MAPPED types
//     CHUNKED / CONTINUOUS for ranks of one node (the reference's "vmm" fast path, gather_op_impl_mapped.cu:18-67,
//     device_reference.cuh:33-50): every rank exports its partition with hipIpcGetMemHandle, opens its peers', and
//     gather / scatter become ONE kernel whose loads / stores go straight over xGMI — no bucketing, no exchange, no
//     host synchronisation at all.  (No flat global pointer: a CONTINUOUS handle is addressed like a CHUNKED one.)
//     Host-pinned and HIERARCHY memory do not exist here: every table lives in HBM.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <rccl/rccl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <rocprim/rocprim.hpp>
#include <unordered_set>
#include <vector>

#include "wg_common.hpp"
#include "wgamd_comm.h"

struct wholememory_comm_ {
  ncclComm_t nccl = nullptr;
  int rank = 0, size = 1;
  bool intra_node = true;   // every rank runs on this host: peer-mapped memory types are available
  int* h_counts   = nullptr;  // pinned [2 * size]: send / receive counts of an exchange, read back together
  int distributed_backend = 1;   // WHOLEMEMORY_DB_NCCL (= RCCL here); NVSHMEM is refused
  std::mutex handles_mu;
  std::vector<wholememory_handle_*> live_handles;  // creation order; destroy_communicator releases what is left
};

namespace wgamd {
constexpr int kMaxMappedRanks = 64;
// what the peer-mapped kernels need, in device memory: base pointer of every rank's partition (mine: my own allocation,
// the others: their allocation opened through HIP IPC) and the entry partition
struct mapped_view {
  char* base[kMaxMappedRanks];
  int64_t entry_off[kMaxMappedRanks + 1];
  int W;
};
}  // namespace wgamd

struct wholememory_handle_ {
  uint64_t serial = 0;   // unique per allocation (never reused, unlike the address)
  wholememory_comm_t comm;
  wholememory_memory_type_t type;
  wholememory_memory_location_t location;
  size_t total_size, granularity;
  std::vector<size_t> byte_offsets;  // W+1, partition of [0, total_size) in bytes
  void* local_ptr;
  // peer-mapped types with more than one rank
  std::vector<void*> peer_ptr;         // [W]; peer_ptr[me] == local_ptr
  std::vector<char> peer_opened;       // [W]; 1 = came from hipIpcOpenMemHandle (closed by wholememory_free)
  wgamd::mapped_view* d_view = nullptr;
  // WHOLEMEMORY_ML_HOST: the partition is pinned host memory the GPU reads and writes in place over PCIe.  A partition
  // that other PROCESSES map (peer-mapped types) is a POSIX shared-memory segment registered with the runtime on
  // both sides; everything else is a hipHostMalloc block.
  bool host_shm = false;                // local_ptr is an mmap'ed + registered segment of `local_map_bytes`
  size_t local_map_bytes = 0;
  std::vector<void*> peer_host_map;     // [W]; the peers' segments as mapped into THIS process (unregistered + unmapped on free)
  std::vector<size_t> peer_map_bytes;   // [W]
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
  decltype(&ncclCommCount) CommCount       = nullptr;   // optional: only the rccl_info accessor uses it
  decltype(&ncclGetVersion) GetVersion     = nullptr;   // optional
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
      else fprintf(stderr, "[wholegraph_amd] WGAMD_RCCL_LIBRARY is set: collectives go through %s INSTEAD of librccl.so\n", forced);
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
    api.CommCount    = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.GetVersion   = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
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

// every rank contributes `bytes` host bytes, `all` receives the W records in rank order (a collective with one host
// synchronisation; setup paths only: communicator creation, peer-mapped allocation)
void allgather_host(wholememory_comm_t comm, const void* mine, size_t bytes, std::vector<char>& all)
{
  const int W = comm->size;
  all.assign((size_t)W * bytes, 0);
  if (W == 1) {
    memcpy(all.data(), mine, bytes);
    return;
  }
  char* d = nullptr;
  WG_HIP_CHECK(hipMalloc(&d, (size_t)(W + 1) * bytes));
  WG_HIP_CHECK(hipMemcpy(d, mine, bytes, hipMemcpyHostToDevice));
  std::vector<size_t> s_off(W, 0), s_b(W, bytes), r_off(W), r_b(W, bytes);
  for (int r = 0; r < W; r++) r_off[r] = (size_t)(r + 1) * bytes;
  try {
    alltoallv_bytes(comm, d, s_off, s_b, d, r_off, r_b, nullptr);
    WG_HIP_CHECK(hipMemcpy(all.data(), d + bytes, (size_t)W * bytes, hipMemcpyDeviceToHost));  // default stream: ordered + blocking
  } catch (...) {
    (void)hipFree(d);
    throw;
  }
  (void)hipFree(d);
}

// identity of this host: name + boot id (two containers of one machine share neither xGMI peers nor this value)
uint64_t host_identity()
{
  char buf[512] = {0};
  (void)gethostname(buf, 255);
  size_t len = strlen(buf);
  if (FILE* f = fopen("/proc/sys/kernel/random/boot_id", "r")) {
    len += fread(buf + len, 1, sizeof(buf) - 1 - len, f);
    fclose(f);
  }
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < len; i++) h = (h ^ (unsigned char)buf[i]) * 1099511628211ull;
  return h;
}

// ---- kernels ---------------------------------------------------------------------------------------
constexpr int kMaxRanks = 1024;

// owner rank of an entry: last r with entry_offsets[r] <= id.  Negative ids (rows to skip) stay negative and stay HOME:
// they land in the asker's own bucket, so neither the id nor a row for it crosses the wire.
__device__ __forceinline__ int owner_of(int64_t id, const int64_t* entry_offsets, int W, int me)
{
  if (id < 0) return me;
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
                       const int64_t* __restrict__ entry_offsets, int W, int me, int* __restrict__ counts)
{
  __shared__ int local[kMaxRanks];
  for (int r = threadIdx.x; r < W; r += blockDim.x) local[r] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t id = (int64_t)idx[i];
    atomicAdd(&local[owner_of(id < 0 ? id : id + row0, entry_offsets, W, me)], 1);
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
  const int r      = owner_of(id < 0 ? id : id + row0, entry_offsets, W, me);
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
               o_x = scratch.add(sizeof(int) * W), o_gid = scratch.add(sizeof(int64_t) * n),
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
      owner_histogram_kernel<int32_t><<<grid, 256, 0, stream>>>(static_cast<const int32_t*>(idx), n, row0, d_offsets, W, me, d_counts);
    else
      owner_histogram_kernel<int64_t><<<grid, 256, 0, stream>>>(static_cast<const int64_t*>(idx), n, row0, d_offsets, W, me, d_counts);
    WG_HIP_CHECK(hipGetLastError());
  }
  // ---- 2. counts all-to-all ON THE DEVICE (W x int; nothing to trade on a single-rank communicator), then BOTH count
  //         vectors come back in one pinned read-back: the only host synchronisation of the whole gather / scatter
  int* d_rcounts = scratch.at<int>(o_x);
  if (W == 1) {
    WG_HIP_CHECK(hipMemcpyAsync(d_rcounts, d_counts, sizeof(int), hipMemcpyDeviceToDevice, stream));
  } else {
    std::vector<size_t> four(W, sizeof(int)), at(W);
    for (int r = 0; r < W; r++) at[r] = (size_t)r * sizeof(int);
    alltoallv_bytes(comm, reinterpret_cast<const char*>(d_counts), at, four, reinterpret_cast<char*>(d_rcounts), at, four, stream);
  }
  WG_EXPECTS(comm->h_counts != nullptr, "communicator without its pinned count buffer");
  WG_HIP_CHECK(hipMemcpyAsync(comm->h_counts, d_counts, sizeof(int) * W, hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipMemcpyAsync(comm->h_counts + W, d_rcounts, sizeof(int) * W, hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));
  send_cnt.assign(W, 0);
  recv_cnt.assign(W, 0);
  for (int r = 0; r < W; r++) {
    send_cnt[r] = (size_t)comm->h_counts[r];
    recv_cnt[r] = (size_t)comm->h_counts[W + r];
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

// ---- peer-mapped gather / scatter: one kernel, loads / stores straight over xGMI ---------------------------------
namespace {
template <int V>
struct mvec;
template <> struct mvec<16> { using type = uint4; };
template <> struct mvec<8> { using type = uint2; };
template <> struct mvec<4> { using type = uint32_t; };
template <> struct mvec<2> { using type = uint16_t; };
template <> struct mvec<1> { using type = uint8_t; };
constexpr int kMappedRows = 4;   // rows in flight per lane group (a remote row is ~2-3 us away)

// SCATTER = false: dense row i <- table row idx[i];  true: table row idx[i] <- dense row i.  Negative index: row skipped.
template <int V, typename IdxT, bool SCATTER>
__global__ void __launch_bounds__(256)
mapped_rows_kernel(const mapped_view* __restrict__ view, int64_t row0, int64_t entry_bytes, int col0_bytes,
                   const IdxT* __restrict__ idx, int64_t n, int row_bytes, char* __restrict__ dense, int64_t dense_stride,
                   int log2_lanes)
{
  using vec_t = typename mvec<V>::type;
  __shared__ int64_t s_off[kMaxMappedRanks + 1];
  __shared__ char* s_base[kMaxMappedRanks];
  const int W = view->W;
  for (int r = threadIdx.x; r <= W; r += blockDim.x) {
    s_off[r] = view->entry_off[r];
    if (r < W) s_base[r] = view->base[r];
  }
  __syncthreads();
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t group   = tid >> log2_lanes;
  const int sub         = (int)(tid & (lanes - 1));
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const int step        = lanes * V;
  const int iters       = (row_bytes + step - 1) / step;
  for (int64_t r0 = group * kMappedRows; r0 < n; r0 += ngroups * kMappedRows) {
    char* tp[kMappedRows];
    bool all_ok = true;
#pragma unroll
    for (int k = 0; k < kMappedRows; k++) {
      const int64_t ri = r0 + k < n ? r0 + k : n - 1;   // unconditional index load
      int64_t id       = (int64_t)idx[ri];
      bool ok          = r0 + k < n && id >= 0 && id + row0 < s_off[W];   // past the last entry: skipped like a negative id
      id               = ok ? id + row0 : s_off[0];     // a dead slot points at some valid row and is never stored
      int lo = 0, hi = W;                               // owner: last r with entry_off[r] <= id
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= id) lo = mid; else hi = mid;
      }
      ok     = ok && s_base[lo] != nullptr;             // an owner without an allocation cannot hold the row
      tp[k]  = ok ? s_base[lo] + (id - s_off[lo]) * entry_bytes + col0_bytes : nullptr;
      all_ok = all_ok && ok;
    }
    if (__all(all_ok)) {
      // fast path: nothing under a per-lane branch, so the kMappedRows remote fetches really are in flight together
      for (int it = 0; it < iters; it++) {
        int off = sub * V + it * step;
        off     = off + V <= row_bytes ? off : row_bytes - V;   // lanes past the row re-copy its last V bytes
        vec_t v[kMappedRows];
#pragma unroll
        for (int k = 0; k < kMappedRows; k++)
          v[k] = *reinterpret_cast<const vec_t*>(SCATTER ? dense + (r0 + k) * dense_stride + off : tp[k] + off);
#pragma unroll
        for (int k = 0; k < kMappedRows; k++)
          *reinterpret_cast<vec_t*>(SCATTER ? tp[k] + off : dense + (r0 + k) * dense_stride + off) = v[k];
      }
      continue;
    }
    for (int off = sub * V; off + V <= row_bytes; off += step) {
#pragma unroll
      for (int k = 0; k < kMappedRows; k++) {
        if (tp[k] != nullptr) {
          char* d = dense + (r0 + k) * dense_stride + off;
          if (SCATTER) *reinterpret_cast<vec_t*>(tp[k] + off) = *reinterpret_cast<const vec_t*>(d);
          else *reinterpret_cast<vec_t*>(d) = *reinterpret_cast<const vec_t*>(tp[k] + off);
        }
      }
    }
  }
}

template <typename IdxT, bool SCATTER>
void mapped_launch(const mapped_view* view, int64_t row0, int64_t entry_bytes, int col0_bytes, const IdxT* idx, int64_t n,
                   int row_bytes, char* dense, int64_t dense_stride, hipStream_t stream)
{
  int V = 16;
  while (V > 1 && ((row_bytes | entry_bytes | col0_bytes | dense_stride | (int64_t)reinterpret_cast<uintptr_t>(dense)) & (V - 1)) != 0)
    V >>= 1;
  int l2 = 0;
  while ((1 << l2) * V < row_bytes && l2 < 6) l2++;
  const int64_t groups = (n + kMappedRows - 1) / kMappedRows;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(((groups << l2) + 255) / 256, 256 * 16));
#define WG_MAPPED(VV)                                                                                                    \
  mapped_rows_kernel<VV, IdxT, SCATTER><<<grid, 256, 0, stream>>>(view, row0, entry_bytes, col0_bytes, idx, n, row_bytes, dense, \
                                                                  dense_stride, l2)
  switch (V) {
    case 16: WG_MAPPED(16); break;
    case 8: WG_MAPPED(8); break;
    case 4: WG_MAPPED(4); break;
    case 2: WG_MAPPED(2); break;
    default: WG_MAPPED(1); break;
  }
#undef WG_MAPPED
  WG_HIP_CHECK(hipGetLastError());
}
}  // namespace

void distributed_rows_op(bool scatter, wholememory_handle_t h, wholememory_matrix_description_t tm, const void* idx,
                         wholememory_dtype_t idx_dtype, int64_t n, char* dense, wholememory_matrix_description_t dense_m,
                         wholememory_env_func_t* env, hipStream_t stream)
{
  const size_t tes         = dtype_size(tm.dtype);
  const size_t entry_bytes = (size_t)tm.stride * tes;
  WG_EXPECTS(h->granularity == entry_bytes, "tensor row stride (%zu B) != handle granularity (%zu B)", entry_bytes,
             h->granularity);
  const int64_t row0 = tm.storage_offset / tm.stride;  // sub-tensor views: first row / first column of the view
  const int64_t col0 = tm.storage_offset % tm.stride;
  WG_REQUIRE_INPUT(tm.storage_offset >= 0 && col0 + tm.sizes[1] <= tm.stride, "bad storage offset");

  // Peer-mapped handle (CHUNKED / CONTINUOUS over HIP IPC): every partition is addressable from this GPU — one kernel,
  // no exchange, no host synchronisation.  (A dtype-converting call takes the exchange below: any handle has a communicator.)
  if (h->d_view != nullptr && tm.dtype == dense_m.dtype) {
    if (n == 0) return;
    const int64_t dstride = dense_m.stride * (int64_t)dtype_size(dense_m.dtype);
    const int row_bytes   = (int)(tm.sizes[1] * (int64_t)tes);
#define WG_MAPPED_GO(IDX, SC)                                                                                           \
  mapped_launch<IDX, SC>(h->d_view, row0, (int64_t)entry_bytes, (int)(col0 * (int64_t)tes), static_cast<const IDX*>(idx), n, \
                         row_bytes, dense, dstride, stream)
    if (idx_dtype == WHOLEMEMORY_DT_INT) {
      if (scatter) WG_MAPPED_GO(int32_t, true); else WG_MAPPED_GO(int32_t, false);
    } else {
      if (scatter) WG_MAPPED_GO(int64_t, true); else WG_MAPPED_GO(int64_t, false);
    }
#undef WG_MAPPED_GO
    return;
  }
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
  // No synchronisation here: the scratch goes back to the caller's allocator, whose contract (env_func_ptrs.h) is that a
  // temporary block is not reused before the work enqueued on `stream` ahead of its release has run — true of torch's
  // caching allocator on the calling stream and of the default hipMalloc / hipFree pair (hipFree synchronises).
}

}  // namespace wgamd

// ------------------------------------------------------------------------------------------------------
// ids of a peer-mapped table -> where their rows start, as byte offsets from the lowest partition base
namespace wgamd {
namespace {
template <typename IdxT>
__global__ void mapped_offsets_kernel(const mapped_view* __restrict__ view, const char* base0, int64_t row0, int64_t entry_bytes,
                                      int64_t col0_bytes, const IdxT* __restrict__ idx, int64_t n, int64_t* __restrict__ out)
{
  __shared__ int64_t s_off[kMaxMappedRanks + 1];
  __shared__ char* s_base[kMaxMappedRanks];
  const int W = view->W;
  for (int r = threadIdx.x; r <= W; r += blockDim.x) {
    s_off[r] = view->entry_off[r];
    if (r < W) s_base[r] = view->base[r];
  }
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t id    = (int64_t)idx[i] + row0;
    const bool ok = idx[i] >= 0 && id < s_off[W];
    id            = ok ? id : s_off[0];
    int lo = 0, hi = W;   // owner: last r with entry_off[r] <= id
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_off[mid] <= id) lo = mid;
      else hi = mid;
    }
    // an id outside the table (the -1 slack of a padded unique list) resolves to row 0 of the lowest partition: the readers of
    // these offsets (sage_layer_mfma_kernel, compose_self_kernel) dereference base + offset unconditionally with 16-byte
    // loads, so the answer must be an aligned, mapped address; no edge references such a row
    out[i] = ok ? (s_base[lo] - base0) + (id - s_off[lo]) * entry_bytes + col0_bytes : col0_bytes;
  }
}
}  // namespace
}  // namespace wgamd

extern "C" {

using namespace wgamd;

static std::atomic<int> g_log_level{3};

// every handle wholememory_malloc has returned and nobody has released yet: wholememory_free of anything else (a handle
// already released together with its communicator) is refused instead of touching freed memory
static std::mutex g_handles_mu;
static std::unordered_set<wholememory_handle_t> g_handles;


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
  if (hipHostMalloc(reinterpret_cast<void**>(&c->h_counts), sizeof(int) * 2 * (size_t)size, hipHostMallocDefault) != hipSuccess) {
    rccl().CommDestroy(c->nccl);
    delete c;
    return WHOLEMEMORY_OUT_OF_MEMORY;
  }
  // do all ranks share this host?  (decides whether the peer-mapped memory types are offered)
  auto rc = guarded("wholememory_create_communicator", [&] {
    const uint64_t mine = host_identity();
    std::vector<char> all;
    allgather_host(c, &mine, sizeof(mine), all);
    for (int r = 0; r < size; r++) {
      uint64_t v;
      memcpy(&v, all.data() + (size_t)r * sizeof(v), sizeof(v));
      if (v != mine) c->intra_node = false;
    }
  });
  if (rc != WHOLEMEMORY_SUCCESS) {
    (void)hipHostFree(c->h_counts);
    rccl().CommDestroy(c->nccl);
    delete c;
    return rc;
  }
  *comm = c;
  return WHOLEMEMORY_SUCCESS;
}

static void release_handle(wholememory_handle_t h, bool collective);

wholememory_error_code_t wholememory_destroy_communicator(wholememory_comm_t comm)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  // handles still alive are released first, oldest first (memory_handle.cpp destroy_all_wholememory): every rank created
  // them in the same order, so the collective frees of peer-mapped handles pair up; a handle can therefore never outlive
  // its communicator and dereference a freed one
  for (;;) {
    wholememory_handle_t h = nullptr;
    {
      std::lock_guard<std::mutex> g(comm->handles_mu);
      if (!comm->live_handles.empty()) h = comm->live_handles.front();
    }
    if (!h) break;
    fprintf(stderr, "[wholegraph_amd] wholememory_destroy_communicator: releasing a handle that was never freed\n");
    release_handle(h, /*collective=*/true);
  }
  if (comm->nccl) rccl().CommDestroy(comm->nccl);
  if (comm->h_counts) (void)hipHostFree(comm->h_counts);
  delete comm;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_communicator_support_type_location(wholememory_comm_t comm,
                                                                        wholememory_memory_type_t memory_type,
                                                                        wholememory_memory_location_t memory_location)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (memory_location != WHOLEMEMORY_ML_DEVICE && memory_location != WHOLEMEMORY_ML_HOST) return WHOLEMEMORY_NOT_SUPPORTED;
  if (memory_type == WHOLEMEMORY_MT_DISTRIBUTED) return WHOLEMEMORY_SUCCESS;
  // peer-mapped types: all ranks on one node (HIP IPC + xGMI peer access), at most kMaxMappedRanks of them
  if ((memory_type == WHOLEMEMORY_MT_CONTINUOUS || memory_type == WHOLEMEMORY_MT_CHUNKED) &&
      (comm->size == 1 || (comm->intra_node && comm->size <= wgamd::kMaxMappedRanks)))
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

/* What the collective library itself says about this communicator: ncclCommCount (number of ranks RCCL joined) and
 * ncclGetVersion.  bench.py prints both so a scaling line shows that RCCL — not a stand-in — carried the exchange. */
wholememory_error_code_t wgamd_communicator_rccl_info(wholememory_comm_t comm, int* rccl_ranks, int* rccl_version)
{
  return guarded("wgamd_communicator_rccl_info", [&] {
    WG_REQUIRE_INPUT(comm != nullptr && rccl_ranks != nullptr, "null argument");
    *rccl_ranks = -1;
    if (rccl_version) *rccl_version = -1;
    auto& api = rccl();
    if (!api.ok) throw comm_error("RCCL not loaded");
    if (api.CommCount) WG_NCCL_CHECK(api.CommCount(comm->nccl, rccl_ranks));
    if (rccl_version && api.GetVersion) WG_NCCL_CHECK(api.GetVersion(rccl_version));
  });
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

static void release_handle(wholememory_handle_t h, bool collective);

wholememory_error_code_t wholememory_malloc(wholememory_handle_t* handle_ptr, size_t total_size, wholememory_comm_t comm,
                                            wholememory_memory_type_t memory_type,
                                            wholememory_memory_location_t memory_location, size_t data_granularity,
                                            size_t* rank_entry_partition)
{
  return guarded("wholememory_malloc", [&] {
    WG_REQUIRE_INPUT(handle_ptr && comm && data_granularity > 0, "null argument / zero granularity");
    WG_REQUIRE_INPUT(total_size % data_granularity == 0, "total_size is not a multiple of data_granularity");
    if (wholememory_communicator_support_type_location(comm, memory_type, memory_location) != WHOLEMEMORY_SUCCESS)
      throw logic_error("memory type / location not supported: DISTRIBUTED on DEVICE, or the peer-mapped CHUNKED / CONTINUOUS "
                        "types when all ranks share a node (see wgamd_comm.h)");
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
    const bool host   = memory_location == WHOLEMEMORY_ML_HOST;
    const bool mapped = memory_type != WHOLEMEMORY_MT_DISTRIBUTED && comm->size > 1;
    char shm_name[48] = {0};
    // A PEER-MAPPED allocation is collective: a rank whose own partition cannot be allocated must still meet its peers in
    // the exchange below and fail WITH them (throwing here would leave them waiting in the allgather for ever).
    const bool mapped_multi = memory_type != WHOLEMEMORY_MT_DISTRIBUTED && comm->size > 1;
    bool alloc_failed       = false;
    // (test hook: WGAMD_TEST_FAIL_MALLOC_RANK=<r> makes rank r's partition of a peer-mapped allocation "fail")
    const char* inject_env = getenv("WGAMD_TEST_FAIL_MALLOC_RANK");
    const bool inject      = inject_env != nullptr && mapped_multi && atoi(inject_env) == comm->rank;
    if (local > 0 && !host && (inject || hipMalloc(&h->local_ptr, local) != hipSuccess)) {
      (void)hipGetLastError();
      h->local_ptr = nullptr;
      if (!mapped_multi) {
        delete h;
        throw std::bad_alloc();
      }
      alloc_failed = true;
    }
    if (local > 0 && host) {
      // memory_handle.cpp:432-520 (host memory of the mapped types = one shared segment every process maps) /
      // :233-262 (distributed host memory = pinned memory of the owning process).  Pinned either way, so the GPU's loads
      // and stores reach it in place.
      bool ok = !inject;
      if (!ok) {
      } else if (mapped) {
        static std::atomic<unsigned> seq{0};
        timespec now{};
        clock_gettime(CLOCK_MONOTONIC, &now);   // pid + counter + time: containers that share /dev/shm may share pids
        snprintf(shm_name, sizeof(shm_name), "/wgamd.%ld.%u.%lx", (long)getpid(), seq.fetch_add(1),
                 (unsigned long)now.tv_nsec ^ ((unsigned long)now.tv_sec << 20));
        const int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        ok           = fd >= 0 && ftruncate(fd, (off_t)local) == 0;
        void* m      = ok ? mmap(nullptr, local, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
        if (fd >= 0) close(fd);
        ok = ok && m != MAP_FAILED;
        if (ok && hipHostRegister(m, local, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {
          munmap(m, local);
          ok = false;
        }
        if (ok) {
          h->local_ptr       = m;
          h->host_shm        = true;
          h->local_map_bytes = local;
        } else {
          shm_unlink(shm_name);
        }
      } else {
        ok = hipHostMalloc(&h->local_ptr, local, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess;
      }
      if (!ok) {
        (void)hipGetLastError();
        h->local_ptr = nullptr;
        if (!mapped_multi) {
          delete h;
          throw std::bad_alloc();
        }
        alloc_failed = true;
      }
    }
    if (mapped_multi) {
      // PEER MAPPING (collective): every rank publishes {HIP IPC handle, pid, pointer} of its partition and opens the
      // others'.  Ranks that live in this very process (threads as ranks) use the pointer as it is — an IPC handle cannot
      // be opened by the process that exported it.
      struct record {
        hipIpcMemHandle_t ipc;
        int64_t pid;
        uint64_t ptr;
        uint64_t bytes;
        int64_t device;
        char shm[48];   // host location: name of the rank's shared-memory segment
        int64_t failed;  // this rank could not allocate its partition: every rank gives up together
      } mine{};
      mine.failed = alloc_failed ? 1 : 0;
      memcpy(mine.shm, shm_name, sizeof(shm_name));
      int my_device = 0;
      WG_HIP_CHECK(hipGetDevice(&my_device));
      mine.device = my_device;
      mine.pid   = (int64_t)getpid();
      mine.ptr   = reinterpret_cast<uint64_t>(h->local_ptr);
      mine.bytes = local;
      try {
        if (local > 0 && !host && !alloc_failed && hipIpcGetMemHandle(&mine.ipc, h->local_ptr) != hipSuccess) {
          (void)hipGetLastError();
          mine.failed = 1;
        }
        std::vector<char> all;
        allgather_host(comm, &mine, sizeof(mine), all);
        for (int r = 0; r < comm->size; r++) {
          record rec;
          memcpy(&rec, all.data() + (size_t)r * sizeof(rec), sizeof(rec));
          if (rec.failed) throw std::bad_alloc();   // (every rank sees the same records: all of them leave here)
        }
        h->peer_ptr.assign(comm->size, nullptr);
        h->peer_opened.assign(comm->size, 0);
        h->peer_host_map.assign(comm->size, nullptr);
        h->peer_map_bytes.assign(comm->size, 0);
        mapped_view view{};
        view.W = comm->size;
        std::exception_ptr map_error;   // a mapping that fails HERE is reported after every rank has said how it went
        try {
        for (int r = 0; r < comm->size; r++) {
          record rec;
          memcpy(&rec, all.data() + (size_t)r * sizeof(rec), sizeof(rec));
          if (r == comm->rank || rec.bytes == 0) {
            h->peer_ptr[r] = r == comm->rank ? h->local_ptr : nullptr;
          } else if (rec.pid == mine.pid && host && getenv("WGAMD_HOST_SHM_MAP_ALWAYS") == nullptr) {
            // registered portable by its owner, in this process.  (WGAMD_HOST_SHM_MAP_ALWAYS: a test switch — ranks that are
            // threads of one process map each other's segments the way separate processes do, branch below.)
            h->peer_ptr[r] = reinterpret_cast<void*>(rec.ptr);
          } else if (host) {
            // the peer's segment, mapped and registered here as well
            const int fd = shm_open(rec.shm, O_RDWR, 0600);
            void* m      = fd >= 0 ? mmap(nullptr, rec.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
            if (fd >= 0) close(fd);
            if (m == MAP_FAILED) throw logic_error("host peer mapping: the shared-memory segment of a peer cannot be mapped");
            h->peer_host_map[r]  = m;
            h->peer_map_bytes[r] = rec.bytes;
            h->peer_opened[r]    = 2;   // (set with the mapping: a failing registration below still gets its munmap on release)
            WG_HIP_CHECK(hipHostRegister(m, rec.bytes, hipHostRegisterPortable | hipHostRegisterMapped));
            void* dp          = nullptr;
            WG_HIP_CHECK(hipHostGetDevicePointer(&dp, m, 0));
            h->peer_ptr[r] = dp;
          } else if (rec.pid == mine.pid) {
            // a rank of this very process: its pointer is valid here, but if it sits on ANOTHER device the kernel's loads
            // need peer access between the two devices
            if ((int)rec.device != my_device) {
              int can = 0;
              WG_HIP_CHECK(hipDeviceCanAccessPeer(&can, my_device, (int)rec.device));
              if (!can) throw logic_error("peer-mapped memory type: no peer access between two devices of this process");
              hipError_t pe = hipDeviceEnablePeerAccess((int)rec.device, 0);
              if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
              else WG_HIP_CHECK(pe);
            }
            h->peer_ptr[r] = reinterpret_cast<void*>(rec.ptr);
          } else {
            WG_HIP_CHECK(hipIpcOpenMemHandle(&h->peer_ptr[r], rec.ipc, hipIpcMemLazyEnablePeerAccess));
            h->peer_opened[r] = 1;
            // prove the mapping now (a 4-byte read of the peer's partition) rather than in the first gather kernel: a
            // mapping that cannot be read fails HERE with an error code the caller can fall back from
            uint32_t probe = 0;
            WG_HIP_CHECK(hipMemcpy(&probe, h->peer_ptr[r], rec.bytes >= 4 ? 4 : (size_t)rec.bytes, hipMemcpyDeviceToHost));
          }
          view.base[r]      = static_cast<char*>(h->peer_ptr[r]);
          view.entry_off[r] = (int64_t)(h->byte_offsets[r] / data_granularity);
        }
        view.entry_off[comm->size] = (int64_t)(h->byte_offsets[comm->size] / data_granularity);
        WG_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_view), sizeof(mapped_view)));
        WG_HIP_CHECK(hipMemcpy(h->d_view, &view, sizeof(view), hipMemcpyHostToDevice));
        } catch (...) {
          map_error = std::current_exception();
          (void)hipGetLastError();
        }
        {
          // every rank says whether it holds all its mappings; one failure fails the allocation everywhere (and only after
          // this exchange may the shared-memory names go: the memory lives until the last unmap)
          std::vector<char> seen;
          char done = map_error ? 0 : 1;
          allgather_host(comm, &done, 1, seen);
          if (shm_name[0]) shm_unlink(shm_name);
          shm_name[0] = 0;
          if (map_error) std::rethrow_exception(map_error);
          for (int r = 0; r < comm->size; r++)
            if (!seen[r]) throw logic_error("peer-mapped allocation: another rank could not map a partition");
        }
      } catch (...) {
        if (shm_name[0]) shm_unlink(shm_name);
        release_handle(h, /*collective=*/false);
        throw;
      }
    }
    {
      std::lock_guard<std::mutex> g(comm->handles_mu);
      comm->live_handles.push_back(h);
    }
    {
      static std::atomic<uint64_t> next_serial{1};
      h->serial = next_serial.fetch_add(1);
      std::lock_guard<std::mutex> g(g_handles_mu);
      g_handles.insert(h);
    }
    *handle_ptr = h;
  });
}

/* Releases a handle.  For a PEER-MAPPED handle (CHUNKED / CONTINUOUS over more than one rank) the release is COLLECTIVE when
 * `collective` is set, as in the reference (memory_handle.cpp:1007-1029: barrier, unmap, barrier, release, barrier): a peer's
 * mapped_rows_kernel may still be reading or writing THIS rank's partition over xGMI, so every rank first drains its own
 * device, all ranks meet, the mappings are closed, all ranks meet again, and only then is the partition handed back.
 * `collective = false` is for the error path of wholememory_malloc, where the other ranks cannot be assumed to follow. */
static void release_handle(wholememory_handle_t h, bool collective)
{
  {
    std::lock_guard<std::mutex> g(g_handles_mu);
    g_handles.erase(h);
  }
  {
    std::lock_guard<std::mutex> g(h->comm->handles_mu);
    auto& v = h->comm->live_handles;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i] == h) {
        v.erase(v.begin() + (long)i);
        break;
      }
  }
  const bool mapped = !h->peer_ptr.empty();
  auto meet = [&](const char* when) {
    auto rc = wholememory_communicator_barrier(h->comm);
    if (rc != WHOLEMEMORY_SUCCESS)
      fprintf(stderr, "[wholegraph_amd] wholememory_free: barrier %s failed (%d): a peer may still be using this partition\n",
              when, (int)rc);
  };
  if (mapped) {
    (void)hipDeviceSynchronize();
    if (collective) meet("before closing the peer mappings");
  }
  for (size_t r = 0; r < h->peer_ptr.size(); r++) {
    if (h->peer_opened[r] == 1 && h->peer_ptr[r]) (void)hipIpcCloseMemHandle(h->peer_ptr[r]);
    if (h->peer_opened[r] == 2 && h->peer_host_map[r]) {
      (void)hipHostUnregister(h->peer_host_map[r]);
      munmap(h->peer_host_map[r], h->peer_map_bytes[r]);
    }
  }
  if (h->d_view) (void)hipFree(h->d_view);
  if (mapped && collective) meet("before releasing the partition");
  if (h->local_ptr) {
    if (h->host_shm) {
      (void)hipHostUnregister(h->local_ptr);
      munmap(h->local_ptr, h->local_map_bytes);
    } else if (h->location == WHOLEMEMORY_ML_HOST) {
      (void)hipHostFree(h->local_ptr);
    } else {
      (void)hipFree(h->local_ptr);
    }
  }
  delete h;
}

wholememory_error_code_t wholememory_free(wholememory_handle_t h)
{
  if (h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  {
    std::lock_guard<std::mutex> g(g_handles_mu);
    if (!g_handles.count(h)) {
      fprintf(stderr, "[wholegraph_amd] wholememory_free: not a live handle (already released with its communicator?)\n");
      return WHOLEMEMORY_INVALID_INPUT;
    }
  }
  release_handle(h, /*collective=*/true);
  return WHOLEMEMORY_SUCCESS;
}

uint64_t wgamd_handle_serial(wholememory_handle_t h)
{
  std::lock_guard<std::mutex> g(g_handles_mu);
  return (h != nullptr && g_handles.count(h)) ? h->serial : 0;
}

wholememory_error_code_t wgamd_free_if_serial(wholememory_handle_t h, uint64_t serial)
{
  if (h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  {
    std::lock_guard<std::mutex> g(g_handles_mu);
    if (!g_handles.count(h) || h->serial != serial) return WHOLEMEMORY_INVALID_INPUT;   // gone, or another handle at this address
  }
  release_handle(h, /*collective=*/true);
  return WHOLEMEMORY_SUCCESS;
}

/* chunked view of a peer-mapped handle: pointer of every rank's partition as seen from THIS process (the role of
 * wholememory_get_global_reference, cpp/include/wholememory/wholememory.h); NULL entries = ranks without rows */
wholememory_error_code_t wgamd_get_peer_pointers(void** pointers, wholememory_handle_t h)
{
  if (!h || !pointers) return WHOLEMEMORY_INVALID_INPUT;
  if (h->comm->size == 1) {
    pointers[0] = h->local_ptr;
    return WHOLEMEMORY_SUCCESS;
  }
  if (h->peer_ptr.empty()) return WHOLEMEMORY_NOT_SUPPORTED;
  for (int r = 0; r < h->comm->size; r++) pointers[r] = h->peer_ptr[r];
  return WHOLEMEMORY_SUCCESS;
}

/* Row addresses of a peer-mapped (CHUNKED / CONTINUOUS) 2-D table for a kernel that reads the rows ITSELF — the one-kernel
 * SAGE layer with the feature fetch folded in, now over xGMI (the reference's mapped gather reads the partitions through
 * global references the same way: wholememory_ops/functions/gather_scatter_func.cuh:242-505, gather_op_impl_mapped.cu):
 * offsets[i] = byte distance of row ids[i] from *base (the lowest partition base of this process's mapping); an id that is
 * negative or past the last row gets the offset of the first row of the lowest partition (a readable, aligned address).  A single-rank handle answers with its own partition.  Not collective. */
wholememory_error_code_t wgamd_mapped_row_offsets(wholememory_tensor_t table, const void* ids, wholememory_dtype_t ids_dtype,
                                                  int64_t n, int64_t* offsets, void** base, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_mapped_row_offsets", [&] {
    WG_REQUIRE_INPUT(table && base && n >= 0 && (n == 0 || (ids && offsets)), "null pointer");
    WG_REQUIRE_INPUT(ids_dtype == WHOLEMEMORY_DT_INT || ids_dtype == WHOLEMEMORY_DT_INT64, "ids must be INT or INT64");
    wholememory_handle_t h = static_cast<wholememory_handle_t>(wholememory_tensor_get_memory_handle(table));
    if (h == nullptr) throw invalid_input("not a handle-backed tensor");
    const wholememory_tensor_description_t* d = wholememory_tensor_get_tensor_description(table);
    WG_REQUIRE_INPUT(d->dim == 2, "a 2-D table");
    const int64_t tes = (int64_t)dtype_size(d->dtype), stride = d->strides[0];
    const int64_t entry_bytes = stride * tes;
    WG_EXPECTS((int64_t)h->granularity == entry_bytes, "tensor row stride != handle granularity");
    const int64_t row0 = d->storage_offset / stride, col0 = d->storage_offset % stride;
    auto st = static_cast<hipStream_t>(stream);
    const mapped_view* d_view = h->d_view;
    const char* base0         = nullptr;
    if (d_view == nullptr) {
      if (h->comm->size != 1 && h->peer_ptr.empty()) throw logic_error("not a peer-mapped handle (DISTRIBUTED rows are not addressable)");
      throw logic_error("single-partition handle: read the local tensor directly");
    }
    for (int r = 0; r < h->comm->size; r++) {
      const char* p = static_cast<const char*>(h->peer_ptr[r]);
      if (p != nullptr && (base0 == nullptr || p < base0)) base0 = p;
    }
    *base = const_cast<char*>(base0);
    if (n == 0) return;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    if (ids_dtype == WHOLEMEMORY_DT_INT)
      mapped_offsets_kernel<int32_t><<<grid, 256, 0, st>>>(d_view, base0, row0, entry_bytes, col0 * tes, static_cast<const int32_t*>(ids), n, offsets);
    else
      mapped_offsets_kernel<int64_t><<<grid, 256, 0, st>>>(d_view, base0, row0, entry_bytes, col0 * tes, static_cast<const int64_t*>(ids), n, offsets);
    WG_HIP_CHECK(hipGetLastError());
  });
}

/* the two HIP IPC steps on their own (what wholememory_malloc does per peer): a 64-byte handle of a hipMalloc'ed block,
 * and mapping such a handle exported by ANOTHER process of this node */
wholememory_error_code_t wgamd_ipc_export(void* device_ptr, void* handle64)
{
  return guarded("wgamd_ipc_export", [&] {
    WG_REQUIRE_INPUT(device_ptr && handle64, "null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handles are 64 bytes");
    hipIpcMemHandle_t ipc;
    WG_HIP_CHECK(hipIpcGetMemHandle(&ipc, device_ptr));
    memcpy(handle64, &ipc, sizeof(ipc));
  });
}
wholememory_error_code_t wgamd_ipc_open(const void* handle64, void** device_ptr)
{
  return guarded("wgamd_ipc_open", [&] {
    WG_REQUIRE_INPUT(device_ptr && handle64, "null pointer");
    hipIpcMemHandle_t ipc;
    memcpy(&ipc, handle64, sizeof(ipc));
    WG_HIP_CHECK(hipIpcOpenMemHandle(device_ptr, ipc, hipIpcMemLazyEnablePeerAccess));
  });
}
wholememory_error_code_t wgamd_ipc_close(void* device_ptr)
{
  return guarded("wgamd_ipc_close", [&] { WG_HIP_CHECK(hipIpcCloseMemHandle(device_ptr)); });
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
    release_handle(h, /*collective=*/false);  // local failure: the peers are not in a matching free
    return rc;
  }
  (*out)->owns_handle   = true;
  (*out)->handle_serial = wgamd_handle_serial(h);
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


// ---- the rest of the surface the reference's Cython binding links against (wholememory_binding.pyx:31-262,501-565): on this
// design — one node-local RCCL communicator per group, no NVSHMEM, no HIERARCHY type — most have one-line answers ------------

wholememory_error_code_t wholememory_get_local_size(size_t* local_size, wholememory_handle_t h)
{
  if (!h || !local_size) return WHOLEMEMORY_INVALID_INPUT;
  *local_size = h->byte_offsets[h->comm->rank + 1] - h->byte_offsets[h->comm->rank];
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_get_local_offset(size_t* local_offset, wholememory_handle_t h)
{
  if (!h || !local_offset) return WHOLEMEMORY_INVALID_INPUT;
  *local_offset = h->byte_offsets[h->comm->rank];
  return WHOLEMEMORY_SUCCESS;
}

/* memory of rank `rank` as THIS process can address it (memory_handle.cpp:2052-2069): always for the caller's own rank; for
 * a peer only through a peer-mapped (CHUNKED / CONTINUOUS) handle — a DISTRIBUTED handle has no pointer to a peer's rows. */
wholememory_error_code_t wholememory_get_rank_memory(void** rank_memory_ptr, size_t* rank_memory_size,
                                                     size_t* rank_memory_offset, int rank, wholememory_handle_t h)
{
  if (!h || !rank_memory_ptr || !rank_memory_size || !rank_memory_offset) return WHOLEMEMORY_INVALID_INPUT;
  if (rank < 0 || rank >= h->comm->size) return WHOLEMEMORY_INVALID_INPUT;
  void* p = nullptr;
  if (rank == h->comm->rank) p = h->local_ptr;
  else if (!h->peer_ptr.empty()) p = h->peer_ptr[rank];
  else return WHOLEMEMORY_INVALID_INPUT;
  *rank_memory_ptr    = p;
  *rank_memory_offset = h->byte_offsets[rank];
  *rank_memory_size   = h->byte_offsets[rank + 1] - h->byte_offsets[rank];
  return WHOLEMEMORY_SUCCESS;
}

/* one flat pointer over ALL ranks' rows (memory_handle.cpp:2071-2079; the reference maps the partitions back to back in a
 * reserved virtual range).  Here partitions are mapped one by one (wgamd_get_peer_pointers), so the flat pointer exists
 * exactly when one rank holds everything; otherwise INVALID_INPUT, the reference's answer for "no continuous mapping". */
wholememory_error_code_t wholememory_get_global_pointer(void** global_ptr, wholememory_handle_t h)
{
  if (!h || !global_ptr) return WHOLEMEMORY_INVALID_INPUT;
  *global_ptr = nullptr;
  if (h->type != WHOLEMEMORY_MT_CONTINUOUS && h->type != WHOLEMEMORY_MT_CHUNKED) return WHOLEMEMORY_INVALID_INPUT;
  const size_t mine = h->byte_offsets[h->comm->rank + 1] - h->byte_offsets[h->comm->rank];
  if (h->comm->size == 1 || mine == h->total_size) *global_ptr = h->local_ptr;
  return *global_ptr ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_INVALID_INPUT;
}

/* HIERARCHY handles only (memory_handle.cpp:1983-2011) — that type does not exist here */
wholememory_error_code_t wholememory_get_local_communicator(wholememory_comm_t* comm, wholememory_handle_t h)
{
  if (!h || !comm) return WHOLEMEMORY_INVALID_INPUT;
  return WHOLEMEMORY_NOT_SUPPORTED;
}
wholememory_error_code_t wholememory_get_cross_communicator(wholememory_comm_t* comm, wholememory_handle_t h)
{
  if (!h || !comm) return WHOLEMEMORY_INVALID_INPUT;
  return WHOLEMEMORY_NOT_SUPPORTED;
}

wholememory_error_code_t wholememory_communicator_get_local_size(int* local_size, wholememory_comm_t comm)
{
  if (!local_size || !comm) return WHOLEMEMORY_INVALID_INPUT;
  *local_size = comm->intra_node ? comm->size : 1;  // ranks of this communicator on the caller's node
  return WHOLEMEMORY_SUCCESS;
}

/* MNNVL cliques (multi-node NVLink domains) have no counterpart on an xGMI node: nobody is in one (communicator.cpp:920-925) */
wholememory_error_code_t wholememory_communicator_get_clique_info(clique_info_t* clique_info, wholememory_comm_t comm)
{
  if (!clique_info || !comm) return WHOLEMEMORY_INVALID_INPUT;
  clique_info->is_in_clique      = 0;
  clique_info->clique_first_rank = -1;
  clique_info->clique_rank       = -1;
  clique_info->clique_rank_num   = 0;
  clique_info->clique_id         = -1;
  clique_info->clique_num        = 0;
  return WHOLEMEMORY_SUCCESS;
}

bool wholememory_communicator_is_bind_to_nvshmem(wholememory_comm_t) { return false; }

/* communicator.cpp:1045-1076 — only the collective-library backend (NCCL there, RCCL here) exists */
wholememory_error_code_t wholememory_communicator_set_distributed_backend(wholememory_comm_t comm,
                                                                          wholememory_distributed_backend_t backend)
{
  if (!comm) return WHOLEMEMORY_INVALID_INPUT;
  if (backend == WHOLEMEMORY_DB_NCCL) {
    comm->distributed_backend = (int)backend;
    return WHOLEMEMORY_SUCCESS;
  }
  fprintf(stderr, "[wholegraph_amd] wholememory_communicator_set_distributed_backend: only WHOLEMEMORY_DB_NCCL (RCCL)\n");
  return backend == WHOLEMEMORY_DB_NVSHMEM ? WHOLEMEMORY_NOT_SUPPORTED : WHOLEMEMORY_INVALID_INPUT;
}
wholememory_distributed_backend_t wholememory_communicator_get_distributed_backend(wholememory_comm_t comm)
{
  return comm ? (wholememory_distributed_backend_t)comm->distributed_backend : WHOLEMEMORY_DB_NONE;
}
wholememory_distributed_backend_t wholememory_get_distributed_backend(wholememory_handle_t h)
{
  return h ? (wholememory_distributed_backend_t)h->comm->distributed_backend : WHOLEMEMORY_DB_NONE;
}

bool wholememory_is_intranode_communicator(wholememory_comm_t comm) { return comm ? comm->intra_node : false; }
bool wholememory_is_intra_mnnvl_communicator(wholememory_comm_t) { return false; }
bool wholememory_is_build_with_nvshmem(void) { return false; }

/* Device count asked of a CHILD process, so the caller's process has not initialised the runtime when it forks its workers
 * afterwards (system_info.cpp ForkGetDeviceCount); -1 on error. */
int fork_get_device_count(void)
{
  int fd[2];
  if (pipe(fd) != 0) return -1;
  const pid_t pid = fork();
  if (pid < 0) {
    close(fd[0]);
    close(fd[1]);
    return -1;
  }
  if (pid == 0) {
    close(fd[0]);
    int n = -1;
    if (hipGetDeviceCount(&n) != hipSuccess) n = -1;
    ssize_t w = write(fd[1], &n, sizeof(n));
    (void)w;
    close(fd[1]);
    _exit(0);
  }
  close(fd[1]);
  int n          = -1;
  const ssize_t r = read(fd[0], &n, sizeof(n));
  close(fd[0]);
  int status = 0;
  (void)waitpid(pid, &status, 0);
  return r == (ssize_t)sizeof(n) ? n : -1;
}

/* communicator.cpp:753-830 (ncclCommSplit there).  Built from calls every RCCL has: the members of the parent trade
 * (color, key), the member of every colour that comes first in (key, parent rank) order makes a unique id, the ids are traded,
 * and every colour runs its own ncclCommInitRank.  COLLECTIVE over the parent; color < 0 (WHOLEMEMORY_SPLIT_NOCOLOR) takes
 * part in the two trades and gets NULL. */
wholememory_error_code_t wholememory_split_communicator(wholememory_comm_t* new_comm, wholememory_comm_t comm, int color,
                                                        int key)
{
  if (!new_comm || !comm) return WHOLEMEMORY_INVALID_INPUT;
  *new_comm = nullptr;
  return guarded("wholememory_split_communicator", [&] {
    const int W = comm->size;
    struct ck { int color, key; } mine{color, key};
    std::vector<char> all;
    allgather_host(comm, &mine, sizeof(mine), all);
    std::vector<ck> v(W);
    memcpy(v.data(), all.data(), sizeof(ck) * (size_t)W);
    // my group in (key, parent rank) order
    std::vector<int> members;
    if (color >= 0) {
      for (int r = 0; r < W; r++)
        if (v[r].color == color) members.push_back(r);
      std::stable_sort(members.begin(), members.end(), [&](int a, int b) { return v[a].key < v[b].key; });
    }
    const bool leader = color >= 0 && members[0] == comm->rank;
    wholememory_unique_id_t uid;
    memset(&uid, 0, sizeof(uid));
    if (leader && wholememory_create_unique_id(&uid) != WHOLEMEMORY_SUCCESS) throw comm_error("unique id for the split");
    std::vector<char> ids;
    allgather_host(comm, &uid, sizeof(uid), ids);
    if (color < 0) return;
    memcpy(&uid, ids.data() + (size_t)members[0] * sizeof(uid), sizeof(uid));
    int new_rank = 0;
    while (members[new_rank] != comm->rank) new_rank++;
    auto rc = wholememory_create_communicator(new_comm, uid, new_rank, (int)members.size());
    if (rc != WHOLEMEMORY_SUCCESS) throw comm_error("communicator of the split");
  });
}

/* wholememory_tensor.cpp:285-346 — partition of a tensor's dim 0 over the ranks, in ENTRIES (rows); a tensor over plain
 * memory is one partition */
wholememory_error_code_t wholememory_tensor_get_entry_offsets(size_t* entry_offsets, wholememory_tensor_t t)
{
  if (!entry_offsets || !t) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_tensor_t root = wholememory_tensor_get_root(t);
  if (root->desc.dim != 1 && root->desc.dim != 2) return WHOLEMEMORY_INVALID_VALUE;
  if (!root->handle) {
    entry_offsets[0] = 0;
    entry_offsets[1] = (size_t)root->desc.sizes[0];
    return WHOLEMEMORY_SUCCESS;
  }
  auto* h = root->handle;
  for (int r = 0; r <= h->comm->size; r++) entry_offsets[r] = h->byte_offsets[r] / h->granularity;
  return WHOLEMEMORY_SUCCESS;
}
wholememory_error_code_t wholememory_tensor_get_entry_partition_sizes(size_t* entry_partition, wholememory_tensor_t t)
{
  if (!entry_partition || !t) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_tensor_t root = wholememory_tensor_get_root(t);
  if (root->desc.dim != 1 && root->desc.dim != 2) return WHOLEMEMORY_INVALID_VALUE;
  if (!root->handle) {
    entry_partition[0] = (size_t)root->desc.sizes[0];
    return WHOLEMEMORY_SUCCESS;
  }
  auto* h = root->handle;
  for (int r = 0; r < h->comm->size; r++) entry_partition[r] = (h->byte_offsets[r + 1] - h->byte_offsets[r]) / h->granularity;
  return WHOLEMEMORY_SUCCESS;
}

}  // extern "C"
