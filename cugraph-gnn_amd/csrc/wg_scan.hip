// int32 exclusive scan over up to 2^31 elements, wave64-native.
//   n <= TILE        : one launch (single workgroup).
//   otherwise        : tile sums -> tile rescans, every workgroup adding up the sums before its own tiles (2 launches).
//   WGAMD_SCAN_CHAINED=1: ONE launch instead, single pass with decoupled look-back over library-owned tile states
//                      (wg_scan_chain.hpp).  Built for round 3's "fewer launches per walk" and MEASURED SLOWER on gfx950: the
//                      ~3000 tiles of a hop-2 scan all run at once, so hardly any tile finds a finished prefix nearby and the
//                      look-back walks back 64 agent-scope words at a time (each a trip beyond the XCD's L2): 40 us per scan
//                      against 14 us for the two launches, walk 0.51 -> 0.62 ms per call group, -5 % end to end
//                      (A/B on one box).  Kept as an OPT-IN switch; the tests run both.  NOT graph-capture
//                      safe (first use allocates, and the epoch travels as a kernel argument, so a replayed graph would
//                      reuse an epoch): a capturing stream always takes the two-launch path.  State: one 8 MiB buffer per
//                      (device, stream), at most kChainStates of them alive — older ones are freed (hipFree synchronises).
// Used for sample offsets (role of thrust::exclusive_scan in
// /root/reference/cpp/src/wholegraph_ops/unweighted_sample_without_replacement_func.cuh:323-326)
// and for the first-appearance ranks of append_unique.
#include <map>

#include "wg_common.hpp"
#include "wg_scan_chain.hpp"

namespace wgamd {

scan_chain scan_chain_acquire(hipStream_t stream)
{
  struct state {
    unsigned long long* tiles = nullptr;
    unsigned int* counters    = nullptr;
    unsigned int epoch        = 0;
  };
  static std::mutex m;
  static std::map<std::pair<int, void*>, state> table;
  constexpr size_t kChainStates = 8;
  int dev = 0;
  WG_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(m);
  const std::pair<int, void*> key{dev, static_cast<void*>(stream)};
  if (table.find(key) == table.end() && table.size() >= kChainStates) {
    // streams come and go (handles are recycled by the runtime): never more than kChainStates buffers alive.  hipFree waits
    // for the device, so no scan still reads the state it frees; the survivor set starts over.
    for (auto& kv : table)
      if (kv.second.tiles) (void)hipFree(kv.second.tiles);
    table.clear();
  }
  state& s = table[key];
  if (s.tiles == nullptr) {
    // first scan on this stream: 8 MiB of tile words (any n up to 2^31) + the two counters, zeroed once
    void* p = nullptr;
    WG_HIP_CHECK(hipMalloc(&p, sizeof(unsigned long long) * (size_t)kScanChainTiles + 256));
    WG_HIP_CHECK(hipMemsetAsync(p, 0, sizeof(unsigned long long) * (size_t)kScanChainTiles + 256, stream));
    s.tiles    = static_cast<unsigned long long*>(p);
    s.counters = reinterpret_cast<unsigned int*>(static_cast<char*>(p) + sizeof(unsigned long long) * (size_t)kScanChainTiles);
  }
  if (++s.epoch >= (1u << 30)) {   // the epoch field is 30 bits: start over on a cleared buffer (stream-ordered)
    WG_HIP_CHECK(hipMemsetAsync(s.tiles, 0, sizeof(unsigned long long) * (size_t)kScanChainTiles, stream));
    s.epoch = 1;
  }
  return scan_chain{s.tiles, s.counters, s.epoch};
}

namespace {

constexpr int kThreads = 256;
constexpr int kItems   = 8;
constexpr int kTile    = kThreads * kItems;  // 2048 ints = 8 KiB per workgroup
constexpr int kScanGrid = 2048;              // workgroups of the two tile kernels (8 per CU), striding over the tiles
static_assert(kTile == kScanTile, "wg_common.hpp publishes the tile size to the kernels that zero-fill scan inputs");

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int up = __shfl_up(v, d, 64);
    if (lane >= d) v += up;
  }
  return v;
}

// exclusive scan of one value per thread over the workgroup; returns exclusive prefix, total in *total
__device__ __forceinline__ int block_exclusive_scan(int v, int* total)
{
  __shared__ int wave_sums[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = wave_inclusive_scan(v);
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  int wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) {
    int s = wave_sums[w];
    if (w < wave) wave_off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wave_off + inc - v;
}

// Thread t owns items [t*kItems, (t+1)*kItems) of the tile (blocked arrangement: each lane reads
// 32 contiguous bytes -> two dwordx4 loads).
__device__ __forceinline__ void load_tile(const int* in, int64_t base, int64_t n, int (&x)[kItems])
{
  int64_t p = base + (int64_t)threadIdx.x * kItems;
  if (p + kItems <= n && ((reinterpret_cast<uintptr_t>(in + p) & 15) == 0)) {
    const int4* v = reinterpret_cast<const int4*>(in + p);
    int4 a = v[0], b = v[1];
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < kItems; k++) x[k] = (p + k < n) ? in[p + k] : 0;
  }
}

__global__ void __launch_bounds__(kThreads) scan_single_kernel(const int* in, int* out, int64_t n)
{
  int x[kItems];
  load_tile(in, 0, n, x);
  int s = 0;
#pragma unroll
  for (int k = 0; k < kItems; k++) s += x[k];
  int total;
  int run = block_exclusive_scan(s, &total);
  int64_t p = (int64_t)threadIdx.x * kItems;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    if (p + k < n) out[p + k] = run;
    run += x[k];
  }
  if (threadIdx.x == 0) out[n] = total;
}

// `live` (device count, may be null): only tiles that hold live elements do any memory traffic;
// slack tiles contribute 0 without being read (the no-sync walk scans capacity-sized arrays).
__device__ __forceinline__ bool tile_is_live(dev_count live, unsigned tile)
{
  return live.dev == nullptr || (int64_t)tile * kTile <= (int64_t)live.get();
}

// Both tile kernels run a bounded grid that strides over the tiles: a capacity-sized scan has thousands of tiles of which a
// third hold live elements, and a workgroup per tile means thousands of dispatches that each wait for a wave slot when
// another stream keeps the chip full.
__global__ void __launch_bounds__(kThreads) scan_tile_sums_kernel(const int* in, int64_t n, int* sums, int64_t m, dev_count live)
{
  for (int64_t tile = blockIdx.x; tile < m; tile += gridDim.x) {
    if (!tile_is_live(live, (unsigned)tile)) break;   // dead tiles are never read (scan_sums stops at the live count)
    int x[kItems];
    load_tile(in, tile * kTile, n, x);
    int s = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) s += x[k];
    int total;
    (void)block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) sums[tile] = total;
  }
}

// sum of v over the workgroup (every thread gets it)
__device__ __forceinline__ int block_sum(int v)
{
  int total;
  (void)block_exclusive_scan(v, &total);
  return total;
}

// Second (last) launch: workgroup b owns a CONTIGUOUS run of live tiles, adds up the tile sums before its run itself
// (a few thousand ints out of L2 — cheaper than a third launch that scans them), then rescans its tiles with a running
// carry.  Workgroup 0 also publishes the grand total at out[n].
__global__ void __launch_bounds__(kThreads)
scan_tile_final_kernel(const int* in, int* out, int64_t n, const int* sums, int64_t m, dev_count live)
{
  const int64_t m_live = live.dev == nullptr ? m : min(m, (int64_t)live.get() / kTile + 1);
  const int64_t per    = (m_live + gridDim.x - 1) / gridDim.x;
  const int64_t first  = (int64_t)blockIdx.x * per, last = min(m_live, first + per);
  if (blockIdx.x == 0) {
    int part = 0;
    for (int64_t t = threadIdx.x; t < m_live; t += kThreads) part += sums[t];
    const int total = block_sum(part);
    if (threadIdx.x == 0) out[n] = total;
  }
  if (first >= last) return;
  int part = 0;
  for (int64_t t = threadIdx.x; t < first; t += kThreads) part += sums[t];
  int carry = block_sum(part);
  for (int64_t tile = first; tile < last; tile++) {
    int x[kItems];
    const int64_t base = tile * kTile;
    load_tile(in, base, n, x);
    int s = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) s += x[k];
    int total;
    int run   = block_exclusive_scan(s, &total) + carry;
    int64_t p = base + (int64_t)threadIdx.x * kItems;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
      if (p + k < n) out[p + k] = run;
      run += x[k];
    }
    carry += total;
  }
}

// ONE launch: workgroups take tiles by ticket, reduce, chain the prefixes through the library-owned state words and write
// their tile.  `live` as above: tiles past the live count are never touched; the last live tile publishes the total at out[n].
__global__ void __launch_bounds__(kThreads)
scan_chained_kernel(const int* in, int* out, int64_t n, int64_t m, dev_count live, scan_chain chain)
{
  const int64_t m_live = live.dev == nullptr ? m : min(m, (int64_t)live.get() / kTile + 1);
  while (true) {
    const unsigned tile = chain_next_tile(chain);
    if ((int64_t)tile >= m_live) break;
    int x[kItems];
    const int64_t base = (int64_t)tile * kTile;
    load_tile(in, base, n, x);
    int s = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) s += x[k];
    int total;
    int run = chain_block_exclusive_scan(s, &total);
    const int prefix = chain_exclusive_prefix(chain, tile, total);
    run += prefix;
    int64_t p = base + (int64_t)threadIdx.x * kItems;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
      if (p + k < n) out[p + k] = run;
      run += x[k];
    }
    if ((int64_t)tile == m_live - 1 && threadIdx.x == 0) out[n] = prefix + total;
  }
  chain_finish(chain);
}

}  // namespace

int64_t scan_tmp_ints(int64_t n) { return (n + kTile - 1) / kTile + 2; }

void exclusive_scan_i32(const int* in, int* out, int64_t n, int* tmp, hipStream_t stream, const int* n_live_dev)
{
  dev_count live{(int)n, n_live_dev};
  if (n <= kTile) {
    scan_single_kernel<<<1, kThreads, 0, stream>>>(in, out, n);
  } else {
    int64_t m = (n + kTile - 1) / kTile;
    const unsigned grid = (unsigned)std::min<int64_t>(m, kScanGrid);
    static const bool chained = getenv("WGAMD_SCAN_CHAINED") != nullptr;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (chained && hipStreamIsCapturing(stream, &capturing) != hipSuccess) (void)hipGetLastError();
    if (chained && capturing == hipStreamCaptureStatusNone && m <= kScanChainTiles) {
      scan_chained_kernel<<<grid, kThreads, 0, stream>>>(in, out, n, m, live, scan_chain_acquire(stream));
      WG_HIP_CHECK(hipGetLastError());
      return;
    }
    scan_tile_sums_kernel<<<grid, kThreads, 0, stream>>>(in, n, tmp, m, live);
    // in-place is safe: every tile reads its inputs into registers before writing them back,
    // and out[n] is written from tmp, not from `in`.
    scan_tile_final_kernel<<<grid, kThreads, 0, stream>>>(in, out, n, tmp, m, live);
  }
  WG_HIP_CHECK(hipGetLastError());
}

void exclusive_scan_i32_presummed(const int* in, int* out, int64_t n, const int* sums, hipStream_t stream, const int* n_live_dev)
{
  dev_count live{(int)n, n_live_dev};
  if (n <= kTile) {
    scan_single_kernel<<<1, kThreads, 0, stream>>>(in, out, n);
  } else {
    const int64_t m     = (n + kTile - 1) / kTile;
    const unsigned grid = (unsigned)std::min<int64_t>(m, kScanGrid);
    scan_tile_final_kernel<<<grid, kThreads, 0, stream>>>(in, out, n, sums, m, live);
  }
  WG_HIP_CHECK(hipGetLastError());
}

}  // namespace wgamd
