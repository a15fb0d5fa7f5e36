import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_comm.hip')
s = open(p).read()

# ---- header comment
s = s.replace('''//   * only DISTRIBUTED/DEVICE memory exists: on an 8 x MI355X node every pair of GPUs has its own xGMI link,
//     a grouped send/recv all-to-all drives all 7 links at once, and 288 GB of HBM per GPU removes the need
//     for the host-pinned / VMM-mapped variants.''', '''//   * the exchange costs ONE host synchronisation per call (both count vectors come back in one pinned read-back; the
//     reference synchronises three times, gather_op_impl_nccl.cu:60-150), and none at the end: scratch comes from the
//     caller's stream-ordered allocator;
//   * memory types: DISTRIBUTED (each rank's rows in its own HBM, rows travel by all-to-all-v) and the PEER-MAPPED types
//     CHUNKED / CONTINUOUS for ranks of one node (the reference's "vmm" fast path, gather_op_impl_mapped.cu:18-67,
//     device_reference.cuh:33-50): every rank exports its partition with hipIpcGetMemHandle, opens its peers', and
//     gather / scatter become ONE kernel whose loads / stores go straight over xGMI — no bucketing, no exchange, no
//     host synchronisation at all.  (No flat global pointer: a CONTINUOUS handle is addressed like a CHUNKED one.)
//     Host-pinned and HIERARCHY memory do not exist here: every table lives in HBM.''')
s = s.replace('#include <dlfcn.h>\n#include <rccl/rccl.h>\n', '#include <dlfcn.h>\n#include <rccl/rccl.h>\n#include <unistd.h>\n')

# ---- structs
s = s.replace('''struct wholememory_comm_ {
  ncclComm_t nccl = nullptr;
  int rank = 0, size = 1;
};
''', '''struct wholememory_comm_ {
  ncclComm_t nccl = nullptr;
  int rank = 0, size = 1;
  bool intra_node = true;   // every rank runs on this host: peer-mapped memory types are available
  int* h_counts   = nullptr;  // pinned [2 * size]: send / receive counts of an exchange, read back together
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
''')
s = s.replace('''  std::vector<size_t> byte_offsets;  // W+1, partition of [0, total_size) in bytes
  void* local_ptr;
};''', '''  std::vector<size_t> byte_offsets;  // W+1, partition of [0, total_size) in bytes
  void* local_ptr;
  // peer-mapped types with more than one rank
  std::vector<void*> peer_ptr;         // [W]; peer_ptr[me] == local_ptr
  std::vector<char> peer_opened;       // [W]; 1 = came from hipIpcOpenMemHandle (closed by wholememory_free)
  wgamd::mapped_view* d_view = nullptr;
};''')

# ---- all-gather helper + host identity, placed after alltoallv_bytes
anchor = '''// ---- kernels ---------------------------------------------------------------------------------------
constexpr int kMaxRanks = 1024;
'''
helper = '''// every rank contributes `bytes` host bytes, `all` receives the W records in rank order (a collective with one host
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

'''
assert anchor in s
s = s.replace(anchor, helper + anchor)

# ---- plan(): one synchronisation
old = s[s.index("  std::vector<int> h_counts(W);\n  WG_HIP_CHECK(hipMemcpyAsync(h_counts.data(), d_counts"):s.index("  // ---- 3. group ids by owner, stable")]
new = '''  // ---- 2. counts all-to-all ON THE DEVICE (W x int; nothing to trade on a single-rank communicator), then BOTH count
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

'''
s = s.replace(old, new)
s = s.replace("o_x = scratch.add(sizeof(int64_t) * 2 * W),", "o_x = scratch.add(sizeof(int) * W),")

# ---- mapped kernels + op, before distributed_rows_op
anchor = "void distributed_rows_op(bool scatter, wholememory_handle_t h, wholememory_matrix_description_t tm, const void* idx,"
mapped = '''// ---- peer-mapped gather / scatter: one kernel, loads / stores straight over xGMI ---------------------------------
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
      const bool ok    = r0 + k < n && id >= 0;
      id               = ok ? id + row0 : s_off[0];     // a dead slot points at some valid row and is never stored
      int lo = 0, hi = W;                               // owner: last r with entry_off[r] <= id
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= id) lo = mid; else hi = mid;
      }
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
#define WG_MAPPED(VV)                                                                                                    \\
  mapped_rows_kernel<VV, IdxT, SCATTER><<<grid, 256, 0, stream>>>(view, row0, entry_bytes, col0_bytes, idx, n, row_bytes, dense, \\
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

'''
assert anchor in s
s = s.replace(anchor, mapped + anchor, 1)

# ---- distributed_rows_op: mapped fast path + no final sync
old = '''  WG_EXPECTS(h->type == WHOLEMEMORY_MT_DISTRIBUTED || h->comm->size == 1, "unsupported memory type");
  const size_t tes         = dtype_size(tm.dtype);'''
new = '''  const size_t tes         = dtype_size(tm.dtype);'''
assert old in s
s = s.replace(old, new)
old = '''  // Rows I own never enter the exchange when no dtype conversion is asked for: one permuting copy moves them between my
  // partition and the dense rows (1/W of the traffic; all of it on a single-rank communicator).
  id_exchange x(env);'''
new = '''  // Peer-mapped handle (CHUNKED / CONTINUOUS over HIP IPC): every partition is addressable from this GPU — one kernel,
  // no exchange, no host synchronisation.  (A dtype-converting call takes the exchange below: any handle has a communicator.)
  if (h->d_view != nullptr && tm.dtype == dense_m.dtype) {
    if (n == 0) return;
    const int64_t dstride = dense_m.stride * (int64_t)dtype_size(dense_m.dtype);
    const int row_bytes   = (int)(tm.sizes[1] * (int64_t)tes);
#define WG_MAPPED_GO(IDX, SC)                                                                                           \\
  mapped_launch<IDX, SC>(h->d_view, row0, (int64_t)entry_bytes, (int)(col0 * (int64_t)tes), static_cast<const IDX*>(idx), n, \\
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
  id_exchange x(env);'''
assert old in s
s = s.replace(old, new)
old = '''    local_rows_scatter(d_recv, recv_m, x.d_recv_ids, WHOLEMEMORY_DT_INT64, x.recv_total, const_cast<char*>(local_base), local_m,
                       stream);
  }
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // scratch is released on return
}'''
new = '''    local_rows_scatter(d_recv, recv_m, x.d_recv_ids, WHOLEMEMORY_DT_INT64, x.recv_total, const_cast<char*>(local_base), local_m,
                       stream);
  }
  // No synchronisation here: the scratch goes back to the caller's allocator, whose contract (env_func_ptrs.h) is that a
  // temporary block is not reused before the work enqueued on `stream` ahead of its release has run — true of torch's
  // caching allocator on the calling stream and of the default hipMalloc / hipFree pair (hipFree synchronises).
}'''
assert old in s
s = s.replace(old, new)

# ---- communicator: pinned counts + intra-node
old = '''  *comm = c;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_destroy_communicator(wholememory_comm_t comm)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (comm->nccl) rccl().CommDestroy(comm->nccl);
  delete comm;
  return WHOLEMEMORY_SUCCESS;
}'''
new = '''  if (hipHostMalloc(reinterpret_cast<void**>(&c->h_counts), sizeof(int) * 2 * (size_t)size, hipHostMallocDefault) != hipSuccess) {
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

wholememory_error_code_t wholememory_destroy_communicator(wholememory_comm_t comm)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (comm->nccl) rccl().CommDestroy(comm->nccl);
  if (comm->h_counts) (void)hipHostFree(comm->h_counts);
  delete comm;
  return WHOLEMEMORY_SUCCESS;
}'''
assert old in s
s = s.replace(old, new)
old = '''  if (memory_type == WHOLEMEMORY_MT_DISTRIBUTED) return WHOLEMEMORY_SUCCESS;
  if (comm->size == 1 && (memory_type == WHOLEMEMORY_MT_CONTINUOUS || memory_type == WHOLEMEMORY_MT_CHUNKED))
    return WHOLEMEMORY_SUCCESS;
  return WHOLEMEMORY_NOT_SUPPORTED;'''
new = '''  if (memory_type == WHOLEMEMORY_MT_DISTRIBUTED) return WHOLEMEMORY_SUCCESS;
  // peer-mapped types: all ranks on one node (HIP IPC + xGMI peer access), at most kMaxMappedRanks of them
  if ((memory_type == WHOLEMEMORY_MT_CONTINUOUS || memory_type == WHOLEMEMORY_MT_CHUNKED) &&
      (comm->size == 1 || (comm->intra_node && comm->size <= wgamd::kMaxMappedRanks)))
    return WHOLEMEMORY_SUCCESS;
  return WHOLEMEMORY_NOT_SUPPORTED;'''
assert old in s
s = s.replace(old, new)

# ---- malloc: IPC exchange
old = '''    size_t local = h->byte_offsets[comm->rank + 1] - h->byte_offsets[comm->rank];
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
}'''
new = '''    size_t local = h->byte_offsets[comm->rank + 1] - h->byte_offsets[comm->rank];
    if (local > 0 && hipMalloc(&h->local_ptr, local) != hipSuccess) {
      delete h;
      throw std::bad_alloc();
    }
    if (memory_type != WHOLEMEMORY_MT_DISTRIBUTED && comm->size > 1) {
      // PEER MAPPING (collective): every rank publishes {HIP IPC handle, pid, pointer} of its partition and opens the
      // others'.  Ranks that live in this very process (threads as ranks) use the pointer as it is — an IPC handle cannot
      // be opened by the process that exported it.
      struct record {
        hipIpcMemHandle_t ipc;
        int64_t pid;
        uint64_t ptr;
        uint64_t bytes;
      } mine{};
      mine.pid   = (int64_t)getpid();
      mine.ptr   = reinterpret_cast<uint64_t>(h->local_ptr);
      mine.bytes = local;
      try {
        if (local > 0) WG_HIP_CHECK(hipIpcGetMemHandle(&mine.ipc, h->local_ptr));
        std::vector<char> all;
        allgather_host(comm, &mine, sizeof(mine), all);
        h->peer_ptr.assign(comm->size, nullptr);
        h->peer_opened.assign(comm->size, 0);
        mapped_view view{};
        view.W = comm->size;
        for (int r = 0; r < comm->size; r++) {
          record rec;
          memcpy(&rec, all.data() + (size_t)r * sizeof(rec), sizeof(rec));
          if (r == comm->rank || rec.bytes == 0) {
            h->peer_ptr[r] = r == comm->rank ? h->local_ptr : nullptr;
          } else if (rec.pid == mine.pid) {
            h->peer_ptr[r] = reinterpret_cast<void*>(rec.ptr);
          } else {
            WG_HIP_CHECK(hipIpcOpenMemHandle(&h->peer_ptr[r], rec.ipc, hipIpcMemLazyEnablePeerAccess));
            h->peer_opened[r] = 1;
          }
          view.base[r]      = static_cast<char*>(h->peer_ptr[r]);
          view.entry_off[r] = (int64_t)(h->byte_offsets[r] / data_granularity);
        }
        view.entry_off[comm->size] = (int64_t)(h->byte_offsets[comm->size] / data_granularity);
        WG_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_view), sizeof(mapped_view)));
        WG_HIP_CHECK(hipMemcpy(h->d_view, &view, sizeof(view), hipMemcpyHostToDevice));
      } catch (...) {
        wholememory_free(h);
        throw;
      }
    }
    *handle_ptr = h;
  });
}

wholememory_error_code_t wholememory_free(wholememory_handle_t h)
{
  if (h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  for (size_t r = 0; r < h->peer_ptr.size(); r++)
    if (h->peer_opened[r] && h->peer_ptr[r]) (void)hipIpcCloseMemHandle(h->peer_ptr[r]);
  if (h->d_view) (void)hipFree(h->d_view);
  if (h->local_ptr) (void)hipFree(h->local_ptr);
  delete h;
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
}'''
assert old in s
s = s.replace(old, new)
old = '''    if (wholememory_communicator_support_type_location(comm, memory_type, memory_location) != WHOLEMEMORY_SUCCESS)
      throw logic_error("memory type / location not supported: only DISTRIBUTED on DEVICE (see wgamd_comm.h)");'''
new = '''    if (wholememory_communicator_support_type_location(comm, memory_type, memory_location) != WHOLEMEMORY_SUCCESS)
      throw logic_error("memory type / location not supported: DISTRIBUTED on DEVICE, or the peer-mapped CHUNKED / CONTINUOUS "
                        "types when all ranks share a node (see wgamd_comm.h)");'''
assert old in s
s = s.replace(old, new)
open(p, 'w').write(s)

# header
p = os.path.join(ROOT, 'include/wgamd_comm.h')
s = open(p).read()
print([l for l in s.split('\n') if 'NOT_SUPPORTED' in l or 'wholememory_free' in l][:10])
