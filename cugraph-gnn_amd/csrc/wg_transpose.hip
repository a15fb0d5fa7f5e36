// Source-major view of a sampled-hop CSR (include/wgamd_ext.h: wgamd_csr_transpose_i32) for the backward passes of the
// aggregation kernels (wg_aggregate.hip, wg_gat_bwd.hip), which gather over the transposed structure instead of
// scattering with atomics.  A stable LSD radix sort of (source row, edge id) pairs restricted to the bits a source row
// needs (3.65 M sources -> 22 bits -> 3 passes; a generic 64-bit sort of the same keys runs 8), then
//   * row_ptr_t from the run boundaries of the sorted keys (every thread fills the offsets of the sources between its
//     key and the previous one — empty sources included),
//   * the destination row of an edge by binary search in row_ptr (L2-resident).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

__global__ void __launch_bounds__(256) iota_kernel(int* v, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (int)i;
}

// sorted keys -> CSR offsets: offsets[s] = first position whose key is >= s.  Thread j owns the sources between key j - 1 and
// key j.  A SHORT stretch it writes itself; a long one — the rows of a trimmed layer's input that this hop never touches: millions
// of sources behind the last key, which one thread used to write one by one (12 ms of a 20 ms training step at F = 256, three
// layers) — goes onto a list that fill_listed_gaps_kernel spreads over the whole grid.  The list lives in the sort's value
// buffer (dead once the keys are sorted): [0] = count, entries (lo, hi, j) from word 4 on.
constexpr int kGapShort = 32;
constexpr int kGapListMax = 4096;
__global__ void __launch_bounds__(256) run_offsets_kernel(const unsigned* __restrict__ keys, int64_t n, int64_t n_src,
                                                          int* __restrict__ offsets, int* __restrict__ gaps, int gap_cap)
{
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n) return;
  const int64_t lo = j == 0 ? 0 : (int64_t)keys[j - 1] + 1;     // sources after the previous key ...
  int64_t hi       = j == n ? n_src : (int64_t)keys[j];          // ... up to my key start at position j
  hi               = hi < n_src ? hi : n_src;                    // an id outside [0, n_src) must not write out of bounds
  if (hi - lo >= kGapShort && gap_cap > 0) {
    const int at = atomicAdd(gaps, 1);
    if (at < gap_cap) {
      gaps[4 + 3 * at]     = (int)lo;
      gaps[4 + 3 * at + 1] = (int)hi;
      gaps[4 + 3 * at + 2] = (int)j;
      return;
    }
  }
  for (int64_t s = lo; s <= hi; s++) offsets[s] = (int)j;
}

__global__ void __launch_bounds__(256) fill_listed_gaps_kernel(const int* __restrict__ gaps, int gap_cap, int* __restrict__ offsets)
{
  const int count = min(gaps[0], gap_cap);
  for (int g = 0; g < count; g++) {
    const int64_t lo = gaps[4 + 3 * g], hi = gaps[4 + 3 * g + 1];
    const int j      = gaps[4 + 3 * g + 2];
    for (int64_t s = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= hi; s += (int64_t)gridDim.x * blockDim.x) offsets[s] = j;
  }
}

// both launches, `list` = the sort's value buffer of n ints
inline void run_offsets(const unsigned* keys, int64_t n, int64_t n_src, int* offsets, int* list, hipStream_t st)
{
  const int cap = (int)std::max<int64_t>(0, std::min<int64_t>(kGapListMax, (n - 4) / 3));
  if (cap > 0) WG_HIP_CHECK(hipMemsetAsync(list, 0, sizeof(int), st));
  run_offsets_kernel<<<(int)((n + 1 + 255) / 256), 256, 0, st>>>(keys, n, n_src, offsets, list, cap);
  if (cap > 0) fill_listed_gaps_kernel<<<1024, 256, 0, st>>>(list, cap, offsets);
}

__device__ __forceinline__ int row_of_edge(const int* __restrict__ row_ptr, int n_rows, int e)
{
  int lo = 0, hi = n_rows;  // invariant: row_ptr[lo] <= e < row_ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (row_ptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) edge_rows_kernel(const int* __restrict__ row_ptr, int n_rows, int64_t n_edges,
                                                        const int* __restrict__ perm, int* __restrict__ edge_dst,
                                                        int* __restrict__ col_t)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  if (edge_dst) edge_dst[k] = row_of_edge(row_ptr, n_rows, (int)k);
  if (col_t) col_t[k] = row_of_edge(row_ptr, n_rows, perm[k]);
}

// COO -> CSR helpers: 32-bit sort keys from 64-bit destination ids, and the sources in destination-major order
__global__ void __launch_bounds__(256) coo_keys_kernel(const int64_t* __restrict__ dst, int64_t n, unsigned* __restrict__ keys,
                                                       int* __restrict__ iota)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = (unsigned)dst[i];
  iota[i] = (int)i;
}

__global__ void __launch_bounds__(256) permute_sources_kernel(const int64_t* __restrict__ src, const int* __restrict__ perm,
                                                              int64_t n, int* __restrict__ col)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) col[i] = (int)src[perm[i]];
}

// ---- a SMALL hop (one mini-batch of a per-mini-batch training step: 11 k edges, 11 k sources) in ONE launch --------------------
// The radix-sort pipeline above is nine launches; at this size each of them is its launch latency and the transpose is 45 us of
// a 390-us step.  Here one workgroup keeps the source counters AND the permutation in LDS: destination row of every edge (by
// rows, 16 bits) -> count (LDS atomics) -> scan -> place every edge, packed as dst:15 | edge:16, into its source's segment (LDS
// atomics: arbitrary order inside a segment) -> order every segment by edge id, which IS the stable order (= ascending
// destination row: the gradient sums stay run-to-run deterministic) -> write.  Segments of up to 8 entries: a sorting network
// in the registers of the thread that owns the source (an insertion sort in LDS was 10 us of dependent LDS round trips); up to
// 512: rank sort by one wave (d^2 / 64 steps); longer: one wave re-reads the edge list and ranks the source's edges by ballots
// (E / 64 steps, whatever the degree).  Per phase at the products hop (us): 1.8 / 1.9 / 1.7 / 3.3 / sort / 0 / write.
constexpr int kSmallThreads    = 1024;
constexpr int kSmallHubMax     = 4096;
constexpr int kSmallThreadSort = 8;
constexpr int kSmallLdsInts    = 32 * 1024;   // n_src + 1 + n_edges + ceil(n_edges / 2) <= this (128 KB of the CU's 160 KB; the
                                              // list of long segments takes 16)
constexpr int kSmallBatch      = 8;           // independent loads of the edge list in flight per thread
constexpr int kSmallEdgeMask   = 0xffff;      // packed entry: bit 31 = written by a long-segment pass, 30..16 dst, 15..0 edge

__device__ __forceinline__ void cswap(int& a, int& b)
{
  const int lo = min(a, b), hi = max(a, b);   // (edge ids sit in the LOW bits, but equal edges do not exist and the destination
  a = lo, b = hi;                             //  row is monotone in the edge id: the packed words order like the edge ids)
}

__global__ void __launch_bounds__(kSmallThreads) csr_transpose_small_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                                            int n_rows, int n_edges, int n_src,
                                                                            int* __restrict__ row_ptr_t, int* __restrict__ edge_perm,
                                                                            int* __restrict__ edge_dst, int* __restrict__ col_t)
{
  extern __shared__ int lds[];
  int* start = lds;                 // [n_src + 1]: counts -> offsets -> (after the placement) the END of every segment
  int* perm  = lds + n_src + 1;     // [n_edges]: packed entries by source
  unsigned short* dst16 = reinterpret_cast<unsigned short*>(perm + n_edges);   // [n_edges]: destination row of every edge
  __shared__ int wave_sums[kSmallThreads / 64];
  __shared__ int n_hubs;
  __shared__ int hubs[kSmallHubMax];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  auto emit = [&](int p, int v) {
    if (edge_perm) edge_perm[p] = v & kSmallEdgeMask;
    if (col_t) col_t[p] = (v >> 16) & 0x7fff;
  };
  for (int i = t; i <= n_src; i += kSmallThreads) start[i] = 0;
  // (edges outside [row_ptr[0], row_ptr[n_rows]) — a malformed CSR — read as row 0 instead of whatever the LDS held)
  for (int i = t; i < (n_edges + 1) / 2; i += kSmallThreads) reinterpret_cast<int*>(dst16)[i] = 0;
  __syncthreads();
  for (int r = t; r < n_rows; r += kSmallThreads) {
    const int e1 = min(row_ptr[r + 1], n_edges);
    for (int e = max(row_ptr[r], 0); e < e1; e++) dst16[e] = (unsigned short)r;
  }
  if (t == 0) n_hubs = 0;
  __syncthreads();
  // (an id outside [0, n_src) is the caller's error; it is counted on the last source so that nothing is written out of bounds)
  for (int e0 = t; e0 < n_edges; e0 += kSmallThreads * kSmallBatch) {
    int s[kSmallBatch];
#pragma unroll
    for (int k = 0; k < kSmallBatch; k++) {
      const int e = e0 + k * kSmallThreads;
      s[k]        = e < n_edges ? (int)min((unsigned)col[e], (unsigned)(n_src - 1)) : -1;
    }
#pragma unroll
    for (int k = 0; k < kSmallBatch; k++)
      if (s[k] >= 0) atomicAdd(&start[s[k]], 1);
  }
  __syncthreads();
  {  // exclusive scan of start[0..n_src]: a contiguous chunk per thread, wave scans, 16 wave sums
    const int chunk = (n_src + kSmallThreads) / kSmallThreads, lo = min(t * chunk, n_src + 1), hi = min(lo + chunk, n_src + 1);
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += start[i];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_sums[w] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int k = 0; k < w; k++) run += wave_sums[k];
    for (int i = lo; i < hi; i++) {
      const int c = start[i];
      start[i]    = run;
      run += c;
    }
  }
  __syncthreads();
  for (int i = t; i <= n_src; i += kSmallThreads) row_ptr_t[i] = start[i];
  if (edge_dst)
    for (int e = t; e < n_edges; e += kSmallThreads) edge_dst[e] = dst16[e];
  __syncthreads();
  for (int e0 = t; e0 < n_edges; e0 += kSmallThreads * kSmallBatch) {
    int s[kSmallBatch];
#pragma unroll
    for (int k = 0; k < kSmallBatch; k++) {
      const int e = e0 + k * kSmallThreads;
      s[k]        = e < n_edges ? (int)min((unsigned)col[e], (unsigned)(n_src - 1)) : -1;
    }
#pragma unroll
    for (int k = 0; k < kSmallBatch; k++) {
      const int e = e0 + k * kSmallThreads;
      if (s[k] >= 0) perm[atomicAdd(&start[s[k]], 1)] = e | ((int)dst16[e] << 16);
    }
  }
  __syncthreads();
  for (int s = t; s < n_src; s += kSmallThreads) {
    const int b = s ? start[s - 1] : 0, d = start[s] - b;
    if (d < 2) continue;
    if (d > kSmallThreadSort) {
      const int h = atomicAdd(&n_hubs, 1);
      if (h < kSmallHubMax) {
        hubs[h] = s;
        continue;
      }
      for (int i = 1; i < d; i++) {   // (a long segment that did not fit the list — 32 k / 9 < 4096: cannot happen; slow, correct)
        const int v = perm[b + i];
        int j       = i - 1;
        while (j >= 0 && perm[b + j] > v) {
          perm[b + j + 1] = perm[b + j];
          j--;
        }
        perm[b + j + 1] = v;
      }
      continue;
    }
    int v[kSmallThreadSort];
#pragma unroll
    for (int i = 0; i < kSmallThreadSort; i++) v[i] = i < d ? perm[b + i] : 0x7fffffff;
    // Batcher's odd-even merge sort of 8 (19 compare-exchanges, static register indices)
    cswap(v[0], v[1]); cswap(v[2], v[3]); cswap(v[4], v[5]); cswap(v[6], v[7]);
    cswap(v[0], v[2]); cswap(v[1], v[3]); cswap(v[4], v[6]); cswap(v[5], v[7]);
    cswap(v[1], v[2]); cswap(v[5], v[6]);
    cswap(v[0], v[4]); cswap(v[1], v[5]); cswap(v[2], v[6]); cswap(v[3], v[7]);
    cswap(v[2], v[4]); cswap(v[3], v[5]);
    cswap(v[1], v[2]); cswap(v[3], v[4]); cswap(v[5], v[6]);
#pragma unroll
    for (int i = 0; i < kSmallThreadSort; i++)
      if (i < d) perm[b + i] = v[i];
  }
  __syncthreads();
  const int nh = min(n_hubs, kSmallHubMax);
  for (int h = w; h < nh; h += kSmallThreads / 64) {
    const int s = hubs[h], b = s ? start[s - 1] : 0, d = start[s] - b;
    if (d <= 512) {
      for (int i = lane; i < d; i += 64) {
        const int v = perm[b + i] & 0x7fffffff;
        int r       = 0;
        for (int j = 0; j < d; j++) r += (perm[b + j] & 0x7fffffff) < v;
        emit(b + r, v);
      }
    } else {
      int seen = 0;
      for (int e0 = 0; e0 < n_edges; e0 += 64) {
        const int e                 = e0 + lane;
        const bool mine             = e < n_edges && (int)min((unsigned)col[e], (unsigned)(n_src - 1)) == s;
        const unsigned long long bm = __ballot(mine);
        if (mine) emit(b + seen + __popcll(bm & ((1ull << lane) - 1ull)), e | ((int)dst16[e] << 16));
        seen += __popcll(bm);
      }
    }
    for (int i = lane; i < d; i += 64) atomicOr(&perm[b + i], (int)0x80000000);   // (entries another lane may still be comparing: the flag is masked there)
  }
  __syncthreads();
  for (int p = t; p < n_edges; p += kSmallThreads) {
    const int v = perm[p];
    if (v >= 0) emit(p, v);
  }
}

bool small_transpose_enabled()
{
  static const bool on = [] {
    const char* v = getenv("WGAMD_TRANSPOSE_SMALL");
    return !(v && v[0] == '0');
  }();
  return on;
}

unsigned key_bits(int64_t n_src)
{
  unsigned bits = 1;
  while (((int64_t)1 << bits) < n_src && bits < 31) bits++;
  return bits;
}

size_t sort_bytes(int64_t n_edges, int64_t n_src)
{
  size_t bytes = 0;
  if (n_edges > 0)
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr,
                                    (size_t)n_edges, 0u, key_bits(n_src), (hipStream_t) nullptr);
  return bytes;
}

inline size_t pad(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace
}  // namespace wgamd

extern "C" size_t wgamd_csr_transpose_workspace_bytes(int64_t n_edges, int64_t n_src)
{
  using namespace wgamd;
  if (n_edges < 0 || n_src < 0) return 0;
  // sorted keys + iota + (perm when the caller does not want it) + rocprim scratch
  return 3 * pad(sizeof(int) * (size_t)n_edges) + pad(sort_bytes(n_edges, n_src)) + 256;
}

extern "C" wholememory_error_code_t wgamd_csr_transpose_i32(const int* row_ptr, const int* col, int64_t n_rows, int64_t n_edges,
                                                            int64_t n_src, int* row_ptr_t, int* edge_perm, int* edge_dst,
                                                            int* col_t, void* workspace, size_t workspace_bytes, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_csr_transpose_i32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && n_edges >= 0 && n_src >= 0 && n_edges < ((int64_t)1 << 31) && n_rows < ((int64_t)1 << 31),
                     "bad sizes");
    WG_REQUIRE_INPUT(row_ptr && row_ptr_t && (n_edges == 0 || col), "null pointer");
    WG_REQUIRE_INPUT(workspace_bytes >= wgamd_csr_transpose_workspace_bytes(n_edges, n_src) && (workspace || n_edges == 0),
                     "workspace too small");
    auto st = static_cast<hipStream_t>(stream);
    if (n_edges == 0) {
      WG_HIP_CHECK(hipMemsetAsync(row_ptr_t, 0, sizeof(int) * (size_t)(n_src + 1), st));
      return;
    }
    if (n_src >= 1 && n_src + 1 + n_edges + (n_edges + 1) / 2 <= kSmallLdsInts && n_rows < 32768 && small_transpose_enabled()) {
      const size_t lds = sizeof(int) * (size_t)(n_src + 1 + n_edges + (n_edges + 1) / 2);
      WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(csr_transpose_small_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * kSmallLdsInts)));
      csr_transpose_small_kernel<<<1, kSmallThreads, lds, st>>>(row_ptr, col, (int)n_rows, (int)n_edges, (int)n_src, row_ptr_t, edge_perm,
                                                               edge_dst, col_t);
      WG_HIP_CHECK(hipGetLastError());
      return;
    }
    char* ws        = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    const size_t nb = pad(sizeof(int) * (size_t)n_edges);
    auto* keys_out  = reinterpret_cast<unsigned*>(ws);
    auto* iota      = reinterpret_cast<int*>(ws + nb);
    int* perm       = edge_perm ? edge_perm : reinterpret_cast<int*>(ws + 2 * nb);
    void* tmp       = ws + 3 * nb;
    size_t tmp_b    = sort_bytes(n_edges, n_src);
    const int grid  = (int)((n_edges + 255) / 256);
    iota_kernel<<<grid, 256, 0, st>>>(iota, n_edges);
    WG_HIP_CHECK(hipGetLastError());
    WG_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_b, reinterpret_cast<const unsigned*>(col), keys_out, iota, perm,
                                           (size_t)n_edges, 0u, key_bits(n_src), st));
    run_offsets(keys_out, n_edges, n_src, row_ptr_t, iota, st);
    WG_HIP_CHECK(hipGetLastError());
    if (edge_dst || col_t) {
      edge_rows_kernel<<<grid, 256, 0, st>>>(row_ptr, (int)n_rows, n_edges, perm, edge_dst, col_t);
      WG_HIP_CHECK(hipGetLastError());
    }
  });
}

extern "C" size_t wgamd_coo_to_csr_workspace_bytes(int64_t n_edges, int64_t n_dst)
{
  using namespace wgamd;
  if (n_edges < 0 || n_dst < 0) return 0;
  return 4 * pad(sizeof(int) * (size_t)n_edges) + pad(sort_bytes(n_edges, n_dst)) + 256;
}

extern "C" wholememory_error_code_t wgamd_coo_to_csr_i64(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_dst,
                                                         int* row_ptr, int* col, int* edge_perm, void* workspace,
                                                         size_t workspace_bytes, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_coo_to_csr_i64", [&] {
    WG_REQUIRE_INPUT(n_edges >= 0 && n_dst >= 0 && n_edges < ((int64_t)1 << 31) && n_dst < ((int64_t)1 << 31), "bad sizes");
    WG_REQUIRE_INPUT(row_ptr && (n_edges == 0 || (src && dst && col)), "null pointer");
    WG_REQUIRE_INPUT(workspace_bytes >= wgamd_coo_to_csr_workspace_bytes(n_edges, n_dst) && (workspace || n_edges == 0),
                     "workspace too small");
    auto st = static_cast<hipStream_t>(stream);
    if (n_edges == 0) {
      WG_HIP_CHECK(hipMemsetAsync(row_ptr, 0, sizeof(int) * (size_t)(n_dst + 1), st));
      return;
    }
    char* ws        = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    const size_t nb = pad(sizeof(int) * (size_t)n_edges);
    auto* keys_in   = reinterpret_cast<unsigned*>(ws);
    auto* keys_out  = reinterpret_cast<unsigned*>(ws + nb);
    auto* iota      = reinterpret_cast<int*>(ws + 2 * nb);
    int* perm       = edge_perm ? edge_perm : reinterpret_cast<int*>(ws + 3 * nb);
    void* tmp       = ws + 4 * nb;
    size_t tmp_b    = sort_bytes(n_edges, n_dst);
    const int grid  = (int)((n_edges + 255) / 256);
    coo_keys_kernel<<<grid, 256, 0, st>>>(dst, n_edges, keys_in, iota);
    WG_HIP_CHECK(hipGetLastError());
    WG_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_b, keys_in, keys_out, iota, perm, (size_t)n_edges, 0u, key_bits(n_dst), st));
    run_offsets(keys_out, n_edges, n_dst, row_ptr, iota, st);
    permute_sources_kernel<<<grid, 256, 0, st>>>(src, perm, n_edges, col);
    WG_HIP_CHECK(hipGetLastError());
  });
}
