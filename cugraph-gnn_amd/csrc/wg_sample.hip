// One-hop neighbour sampling without replacement over a CSR graph (uniform + weighted A-Res),
// hand-written for gfx950 (wave64).
//
// Replaces wholegraph_csr_{un,}weighted_sample_without_replacement
// (/root/reference/cpp/include/wholememory/wholegraph_op.h:31-73; reference kernels
// cpp/src/wholegraph_ops/unweighted_sample_without_replacement_func.cuh:28-271,
// weighted_sample_without_replacement_func.cuh:33-281).  The RESULT is bit-identical to the
// reference's host oracle (cpp/tests/wholegraph_ops/graph_sampling_test_utils.cu:312-401); the
// way it is computed is not the reference's:
//
//  * M <= 32 (every fan-out of the BASELINE configs): the reference runs one 32-thread block per
//    seed = half an idle wave64.  Here a wave carries TWO seeds, one per 32-lane half; lane t
//    draws r_t from PCG stream (i*32+t) — the stream numbering the reference fixes — and the
//    sequential Fisher-Yates table is resolved in registers with wave ballots: step t needs
//    Q[r_t] and Q[N-t-1], each "the value written by the latest earlier step that touched that
//    position, else the position itself", found with one ballot + one bpermute.  No LDS, no
//    radix sort, no pointer jumping.
//  * 32 < M <= 1024: one workgroup per seed; draws in parallel with the reference's
//    (block-size, items-per-lane) stream layout, then one lane walks the swap table through a
//    small LDS hash (only <= M of the N table positions are ever touched).
//  * M > 1024: reservoir with a max-reduction per slot, as the reference's large kernel.
//  * weighted: keys key_e = log2(u_e)/w_e with the reference's per-lane stream layout, then an
//    exact top-M by 4-pass radix select on the key bits, emitted in CSR order.
#include <algorithm>
#include <cmath>

#include "wg_common.hpp"
#include "wg_rng.hpp"

namespace wgamd {
namespace {

template <typename T>
struct dtype_of;
template <>
struct dtype_of<int32_t> {
  static constexpr wholememory_dtype_t value = WHOLEMEMORY_DT_INT;
};
template <>
struct dtype_of<int64_t> {
  static constexpr wholememory_dtype_t value = WHOLEMEMORY_DT_INT64;
};

// launch table of the reference (…_func.cuh:412-446): PCG stream of lane j of seed i is i*B+j,
// its k-th draw belongs to sample position k*B+j.
__host__ __device__ constexpr int ref_block_threads(int M)
{
  int f = (M - 1) / 32;
  return f < 3 ? 32 : f < 6 ? 64 : f < 12 ? 128 : 256;
}
__host__ __device__ constexpr int ref_items_per_thread(int M)
{
  constexpr int t[32] = {1, 2, 3, 2, 3, 3, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2,
                         3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4};
  return t[(M - 1) / 32];
}

// the reference computes `int gidx = threadIdx.x + blockIdx.x*blockDim.x` and widens it
__device__ __forceinline__ uint64_t stream_id(int64_t seed_index, int B, int lane)
{
  return (uint64_t)(int64_t)(int32_t)(seed_index * B + lane);
}

// generator of stream seed_index*B + lane: table jump for indices below 2^31, the generic loop (with the reference's
// sign extension of the 32-bit index) beyond
__device__ __forceinline__ Pcg32 stream_generator(uint64_t random_seed, int64_t seed_index, int B, int lane)
{
  const int64_t sid = seed_index * B + lane;
  if (sid < (1ll << 31)) return Pcg32(random_seed, (uint32_t)sid, Pcg32::table_tag{});
  return Pcg32(random_seed, stream_id(seed_index, B, lane));
}

// ------------------------------------------------------------------------------------------
constexpr int kWaveRowCapDecl = 1024;  // == kWaveRowCap (rows the one-wave weighted kernel keeps in registers)
constexpr int kHugeRow = 16384;   // candidates; rows above go first (list 4), one workgroup each like the long ones

template <typename SeedT>
__global__ void __launch_bounds__(256) sample_count_kernel(const int64_t* __restrict__ row_ptr,
                                                           const SeedT* __restrict__ seeds,
                                                           dev_count n_,
                                                           int M,
                                                           int* __restrict__ cnt,
                                                           int* __restrict__ big_deg /*nullable*/,
                                                           int* __restrict__ lists = nullptr /*biased hop: see wg_common.hpp*/,
                                                           int list_cap            = 0,
                                                           int scratch_threshold   = 0,
                                                           int64_t* __restrict__ row_start = nullptr,
                                                           int* __restrict__ row_deg       = nullptr)
{
  const int n_live = n_.get();
  if (lists == nullptr) {
    // uniform hop of the no-sync walk: a bounded grid strides over the live seeds and the slack of their last scan tile
    // (a capacity-sized grid is 2/3 workgroups with nothing to do: the dispatcher, not the loads, then bounds the launch)
    const int end = min(n_.host, (n_live / kScanTile + 1) * kScanTile);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
      int deg = 0;
      if (i < n_live) {
        const int64_t nid   = (int64_t)seeds[i];
        const int64_t first = row_ptr[nid];
        deg                 = (int)(row_ptr[nid + 1] - first);
        if (row_start) {
          row_start[i] = first;
          row_deg[i]   = deg;
        }
        deg = (M > 0 && deg > M) ? M : deg;
      }
      cnt[i] = deg;
      if (big_deg) big_deg[i] = 0;
    }
    return;
  }
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int cls          = -1;
  int deg          = 0;
  if (i < n_.host) {
    if (i >= n_live) {  // capacity slack of the no-sync walk: zero the rest of the live scan tile only
      if (i < (n_live / kScanTile + 1) * kScanTile) {
        cnt[i] = 0;
        if (big_deg) big_deg[i] = 0;
      }
    } else {
      int64_t nid         = (int64_t)seeds[i];
      const int64_t first = row_ptr[nid];
      deg                 = (int)(row_ptr[nid + 1] - first);
      cnt[i]              = (M > 0 && deg > M) ? M : deg;
      if (row_start) {   // what the sampling kernel needs of this row, by seed index
        row_start[i] = first;
        row_deg[i]   = deg;
      }
      if (big_deg) big_deg[i] = 0;
      if (lists != nullptr && M > 0 && deg > M) {
        cls = deg <= 16 ? 0 : deg <= 32 ? 1 : deg <= 64 ? 2 : deg <= 128 ? 3 : deg <= 256 ? 4 : deg <= 512 ? 5
              : deg <= kWaveRowCapDecl ? 6 : deg <= kHugeRow ? 7 : 8;
        if (cls >= 7 && deg > scratch_threshold) atomicMax(lists + 10, deg);  // longest row that needs a key slab
      }
    }
  }
  if (lists == nullptr) return;
  // block-aggregated append: ranks inside the block from LDS counters, ONE global atomic per block and list (every seed
  // hitting the same five words directly costs more than the whole count: a word takes ~90 atomics per microsecond)
  __shared__ int blk_cnt[kWeightedLists], blk_base[kWeightedLists];
  if (threadIdx.x < kWeightedLists) blk_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int my_rank    = 0;
#pragma unroll
  for (int c = 0; c < kWeightedLists; c++) {
    const uint64_t m = __ballot(cls == c);
    if (m == 0ull) continue;
    const int leader = __ffsll((long long)m) - 1;
    int base         = 0;
    if (lane == leader) base = atomicAdd(&blk_cnt[c], __popcll(m));
    base = __shfl(base, leader, 64);
    if (cls == c) my_rank = base + __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (threadIdx.x < kWeightedLists && blk_cnt[threadIdx.x] > 0) blk_base[threadIdx.x] = atomicAdd(lists + threadIdx.x, blk_cnt[threadIdx.x]);
  __syncthreads();
  if (cls >= 0) lists[kWeightedListHead + (int64_t)cls * list_cap + blk_base[cls] + my_rank] = i;
}

// `col` may be NULL: the CSR columns live on other GPUs, the kernel then only produces the picked CSR positions
// (edge_gid) and the caller fetches the columns afterwards (the distributed-CSR path of the ABI op)
template <typename ColT>
__device__ __forceinline__ ColT col_at(const ColT* __restrict__ col, int64_t at)
{
  return col ? col[at] : (ColT)0;
}

constexpr int kWalkGrid = 256 * 16;   // workgroups of the walk's bounded, grid-striding launches (16 per CU: 8 resident, 8 queued)

template <typename ColT>
__device__ __forceinline__ void emit(ColT* dst, int* src_lid, int64_t* edge_gid, int64_t out, ColT v,
                                     int seed_index, int64_t gid)
{
  if (dst) dst[out] = v;
  if (src_lid) src_lid[out] = seed_index;
  if (edge_gid) edge_gid[out] = gid;
}

// ---- M <= 32: LANES (32, or 16 when M <= 16) lanes per seed = 2 or 4 seeds per wave64, everything in registers ----
// (a hop is a chain of three dependent memory latencies per seed — seeds -> row_ptr -> col — so what matters is how
//  many seeds a wave keeps in flight; with fan-out 10 a 32-lane group would idle 22 lanes)
// ---- vertex-grouped order for the long hops of a call group --------------------------------------------------------------
// The frontier of a deep hop of a call group holds the same vertices again and again (products, hop 2 of 191 mini-batches:
// 1.5 M entries over 0.52 M vertices, the hubs in every batch), and a fan-out-10 pick from a long row fetches one 128-byte
// line per pick: in frontier (batch-major) order the picks of one hub are spread over the whole launch and every one of them
// misses — 9.2 M line fetches where the entries together touch 3.2 M distinct lines.  So the sampling kernel walks the
// frontier GROUPED BY VERTEX RANGE instead: a counting sort of the entry indices over kLocBuckets vertex-id ranges (histogram
// per block, then every block derives its own write cursors from the histogram matrix and scatters 32-byte records holding
// everything the sampling kernel needs of an entry), and XCD x takes the x-th eighth of the grouped order, so that the picks
// of one vertex meet in ONE L2 within microseconds of each other.  Every entry still draws from its own PCG streams and
// writes its own output positions: the result does not depend on the order, it is bit-identical.
struct loc_rec {
  int64_t start;   // first CSR slot of the row
  int i;           // entry index (output position base, src_lid)
  int deg;
  int base;        // offsets[i]
  int i_local;     // entry index inside its batch (PCG stream numbering)
  int batch;
  int pad;
};
static_assert(sizeof(loc_rec) == 32, "one 32-byte sector per record");
constexpr int kLocBuckets = 4096;
constexpr int kLocBlocks  = 128;    // blocks of both kernels: block b owns entries [b per, (b + 1) per)
constexpr int kLocThreads = 1024;

template <typename SeedT>
__global__ void __launch_bounds__(kLocThreads)
locality_hist_kernel(const SeedT* __restrict__ seeds, dev_count n_, int shift, int* __restrict__ hist /*[kLocBlocks][kLocBuckets]*/)
{
  __shared__ int h[kLocBuckets];
  const int n = n_.get(), per = (n + kLocBlocks - 1) / kLocBlocks;
  const int lo = min(n, (int)blockIdx.x * per), hi = min(n, lo + per);
  for (int k = threadIdx.x; k < kLocBuckets; k += kLocThreads) h[k] = 0;
  __syncthreads();
  for (int i = lo + threadIdx.x; i < hi; i += kLocThreads)
    atomicAdd(&h[min((int)((uint64_t)seeds[i] >> shift), kLocBuckets - 1)], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < kLocBuckets; k += kLocThreads) hist[(int64_t)blockIdx.x * kLocBuckets + k] = h[k];
}

template <typename SeedT>
__global__ void __launch_bounds__(kLocThreads)
locality_scatter_kernel(const SeedT* __restrict__ seeds, dev_count n_, int shift, const int* __restrict__ hist,
                        const int64_t* __restrict__ row_start, const int* __restrict__ row_deg, const int* __restrict__ offsets,
                        rng_plan rng, loc_rec* __restrict__ recs)
{
  __shared__ int cursor[kLocBuckets];
  __shared__ int wave_tot[kLocThreads / 64];
  static_assert(kLocBuckets == 4 * kLocThreads, "thread t owns buckets 4 t .. 4 t + 3");
  const int n = n_.get(), per = (n + kLocBlocks - 1) / kLocBlocks;
  const int lo = min(n, (int)blockIdx.x * per), hi = min(n, lo + per);
  // where my entries of bucket k go: all entries of the buckets before k, plus bucket k's entries of the blocks before me
  int tot[4] = {0, 0, 0, 0}, before[4] = {0, 0, 0, 0};
  const int4* h4 = reinterpret_cast<const int4*>(hist);
  for (int b = 0; b < kLocBlocks; b++) {
    const int4 v = h4[(int64_t)b * (kLocBuckets / 4) + threadIdx.x];
    tot[0] += v.x; tot[1] += v.y; tot[2] += v.z; tot[3] += v.w;
    if (b < (int)blockIdx.x) { before[0] += v.x; before[1] += v.y; before[2] += v.z; before[3] += v.w; }
  }
  const int sum = tot[0] + tot[1] + tot[2] + tot[3];
  int inc = sum;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(inc, d, 64);
    if (lane >= d) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int run = inc - sum;
  for (int w = 0; w < wave; w++) run += wave_tot[w];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    cursor[4 * threadIdx.x + k] = run + before[k];
    run += tot[k];
  }
  __syncthreads();
  for (int i = lo + threadIdx.x; i < hi; i += kLocThreads) {
    const int j = atomicAdd(&cursor[min((int)((uint64_t)seeds[i] >> shift), kLocBuckets - 1)], 1);
    loc_rec r;
    r.start = row_start[i];
    r.i     = i;
    r.deg   = row_deg[i];
    r.base  = offsets[i];
    r.i_local = i;
    r.batch   = 0;
    if (rng.target_batch) {
      r.batch   = rng.target_batch[i];
      r.i_local = i - rng.target_seg[r.batch];
    }
    r.pad   = 0;
    recs[j] = r;
  }
}

template <typename SeedT, typename ColT, int LANES>
__global__ void __launch_bounds__(256) sample_uniform_halfwave_kernel(const int64_t* __restrict__ row_ptr,
                                                                      const ColT* __restrict__ col,
                                                                      const SeedT* __restrict__ seeds,
                                                                      dev_count n_,
                                                                      int M,
                                                                      rng_plan rng,
                                                                      const int* __restrict__ offsets,
                                                                      ColT* __restrict__ dst,
                                                                      int* __restrict__ src_lid,
                                                                      int64_t* __restrict__ edge_gid,
                                                                      const int64_t* __restrict__ row_start,
                                                                      const int* __restrict__ row_deg,
                                                                      const loc_rec* __restrict__ recs = nullptr)
{
  const int n    = n_.get();
  const int lane = threadIdx.x & 63;
  const int hl   = lane & (LANES - 1);   // lane inside my group
  const int hb   = lane & ~(LANES - 1);  // first lane of my group
  int i          = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES);  // seed index
  loc_rec rec{};
  if (recs) {
    // vertex-grouped order: XCD x (workgroups are dealt to the XCDs round-robin) walks the x-th eighth of the records
    constexpr int kGroups = 256 / LANES;   // entries per workgroup
    const int per_x = ((n + 7) / 8 + kGroups - 1) / kGroups * kGroups;
    const int k     = (int)(blockIdx.x >> 3) * kGroups + (int)threadIdx.x / LANES;
    const int j     = (int)(blockIdx.x & 7) * per_x + k;
    i               = n;   // (nothing to do)
    if (k < per_x && j < n) {
      rec = recs[j];
      i   = rec.i;
    }
  }
  // The draw of lane t depends only on (seed index, t), not on the row: it is computed FIRST, so that the ~60 ALU
  // instructions of the table jump run under the latency of the dependent seeds -> row_ptr loads issued right after
  // (rows that turn out to be copied whole waste the draw, which is cheaper than putting it on the critical path).
  int32_t r_draw = 0;
  if (i < n && hl < M) {
    uint64_t random_seed;
    int i_local;
    if (recs) {
      i_local     = rec.i_local;
      random_seed = rng.seeds_dev ? rng.seeds_dev[rec.batch] : rng.seed;
    } else
      rng.resolve(i, random_seed, i_local);
    // stream index = i_local*32 + lane; the table jump covers every index below 2^31, the generic
    // loop keeps the reference's sign-extension semantics beyond that
    if (i_local < (1 << 26)) {
      Pcg32 g(random_seed, (uint32_t)(i_local * 32 + hl), Pcg32::table_tag{});   // stream layout: 32 per seed, always
      r_draw = g.next_i31();
    } else {
      Pcg32 g(random_seed, stream_id(i_local, 32, hl));
      r_draw = g.next_i31();
    }
  }
  int64_t start = 0;
  int N = 0, base = 0;
  if (i < n && recs) {
    start = rec.start;
    N     = rec.deg;
    base  = rec.base;
  } else if (i < n) {
    if (row_start) {   // written by the count kernel: one coalesced read instead of the seeds -> row_ptr chain
      start = row_start[i];
      N     = row_deg[i];
    } else {
      int64_t nid = (int64_t)seeds[i];
      start       = row_ptr[nid];
      N           = (int)(row_ptr[nid + 1] - start);
    }
    base = offsets[i];
  }
  const bool pick = N > M;  // uniform inside a half
  // Is any half of this wave sampling?  (wave-uniform branch around the resolve loop)
  if (__ballot(pick) == 0ull) {
    if (hl < N) emit<ColT>(dst, src_lid, edge_gid, (int64_t)base + hl, col_at<ColT>(col, start + hl), i, start + hl);
    return;
  }
  int r = 0;
  if (pick && hl < M) r = r_draw % (N - hl);
  // Fisher-Yates:  a[t] = Q[r_t];  Q[r_t] = Q[N-t-1], resolved WITHOUT walking the M steps in order.
  // Step t writes position r_t with the value it found at position N-t-1.  For lane t let
  //   p1 = latest step s < t that wrote position r_t      (r_s == r_t),
  //   p2 = latest step s < t that wrote position N-t-1    (r_s == N-t-1).
  // Then  val_t = (p2 exists) ? val_{p2} : N-t-1  is a chain that always ends at a step with no
  // earlier write, so val_t = N - root(t) - 1 with root found by pointer jumping (ceil(log2 M)
  // rounds), and  a_t = (p1 exists) ? val_{p1} : r_t.  All-pairs compare = M-1 independent
  // bpermutes instead of 3M dependent ones.
  const int tail = N - hl - 1;
  int p1 = -1, p2 = -1;
  for (int s = 0; s + 1 < M; s++) {
    const int rs    = __shfl(r, hb | s, 64);
    const bool prev = s < hl;
    p1 = (prev && rs == r) ? s : p1;
    p2 = (prev && rs == tail) ? s : p2;
  }
  int root = p2 >= 0 ? p2 : hl;
#pragma unroll
  for (int round = 0; round < (LANES == 32 ? 5 : 4); round++) root = __shfl(root, hb | root, 64);
  const int val = N - root - 1;
  const int vp1 = __shfl(val, hb | (p1 & (LANES - 1)), 64);
  const int a   = p1 >= 0 ? vp1 : r;
  if (pick) {
    if (hl < M) emit<ColT>(dst, src_lid, edge_gid, (int64_t)base + hl, col_at<ColT>(col, start + a), i, start + a);
  } else if (hl < N) {
    emit<ColT>(dst, src_lid, edge_gid, (int64_t)base + hl, col_at<ColT>(col, start + hl), i, start + hl);
  }
}

// ---- the same hop for the no-sync walk: lane groups as wide as the fan-out, K seeds per group ----------------------------
// (row_start / row_deg come from the count kernel.)  What round 6 measured on the products hop 2 (1.6 M seeds x fan-out 10):
// the launch is bound by VALU issue, not by memory — the vertex-grouped order below cut the lines it fetches 2.7x (PMC) and
// its duration by 2 %, four seeds in flight per group instead of one (K) bought 14 %; a wave spends ~2,000 cycles on four
// seeds, a third of them in the quarter-rate 64-bit multiplies of the PCG jump.  So the lanes are what is scarce: a group is
// GW = the fan-out lanes wide (10 -> six seeds per wave where the 16-lane groups of sample_uniform_halfwave_kernel hold
// four with six lanes idle; fan-out 5 -> twelve), and takes K consecutive seeds whose loads are issued stage by stage.
// Same draws (stream 32 i_local + t), same output positions.
template <typename SeedT, typename ColT, int GW, int K, bool EXACT>
__global__ void __launch_bounds__(256)
sample_uniform_multi_kernel(const ColT* __restrict__ col, dev_count n_, int M_rt, rng_plan rng, const int* __restrict__ offsets,
                            ColT* __restrict__ dst, int* __restrict__ src_lid, int64_t* __restrict__ edge_gid,
                            const int64_t* __restrict__ row_start, const int* __restrict__ row_deg,
                            const loc_rec* __restrict__ recs)
{
  constexpr int kGPW    = 64 / GW;        // lane groups per wave (the lanes past kGPW * GW idle)
  constexpr int kGroups = 4 * kGPW;       // lane groups per workgroup
  constexpr int kRounds = GW <= 2 ? 1 : GW <= 4 ? 2 : GW <= 8 ? 3 : GW <= 16 ? 4 : 5;   // pointer jumping: ceil(log2 GW)
  // EXACT: the group is exactly as wide as the fan-out — M is a compile-time constant (the all-pairs loop unrolls, every lane
  // of a group draws)
  const int M    = EXACT ? GW : M_rt;
  const int n    = n_.get();
  const int lane = threadIdx.x & 63;
  const int grp  = lane / GW;             // (GW is a compile-time constant: a multiply and a shift)
  const int hl   = lane - grp * GW;       // lane inside my group
  const int hb   = grp * GW;              // first lane of my group
  const bool live_group = grp < kGPW;
  const int wg_group    = (int)(threadIdx.x >> 6) * kGPW + grp;
  // The grid is BOUNDED (a few workgroups per CU) and strides over the live seeds: the walk's buffers are sized for the worst
  // case, 3-4x the live count, and a grid sized for the capacity spends its time dispatching workgroups that find nothing
  // to do — measured on the products hop 2 (PMC): 53 k workgroups of which 17 k had seeds, 1.8 waves resident per SIMD on
  // average instead of 8, the launch bound by the dispatcher.
  const int per_x = ((n + 7) / 8 + kGroups * K - 1) / (kGroups * K) * (kGroups * K);   // an XCD's share of the grouped order
  const int64_t n_chunks = recs ? (int64_t)(per_x / (kGroups * K)) * 8 : ((int64_t)n + kGroups * K - 1) / (kGroups * K);
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
  int i[K], N[K], base[K], i_local[K], batch[K];
  int64_t start[K];
  // ---- stage 1: the row records of my K seeds ----------------------------------------------------------------------
  if (recs) {
    // vertex-grouped order: XCD x (workgroups are dealt to the XCDs round-robin; the grid is a multiple of 8) walks the x-th
    // eighth of the records
    const int k0    = ((int)(chunk >> 3) * kGroups + wg_group) * K;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int j = (int)(chunk & 7) * per_x + k0 + k;
      i[k] = n, N[k] = 0, base[k] = 0, start[k] = 0, i_local[k] = 0, batch[k] = 0;
      if (live_group && k0 + k < per_x && j < n) {
        const loc_rec r = recs[j];
        i[k] = r.i, N[k] = r.deg, base[k] = r.base, start[k] = r.start, i_local[k] = r.i_local, batch[k] = r.batch;
      }
    }
  } else {
    const int64_t g = chunk * kGroups + wg_group;
#pragma unroll
    for (int k = 0; k < K; k++) {
      i[k] = live_group ? (int)min(g * K + k, (int64_t)n) : n;
      N[k] = 0, base[k] = 0, start[k] = 0, i_local[k] = i[k], batch[k] = 0;
      if (i[k] < n) {
        start[k] = row_start[i[k]];
        N[k]     = row_deg[i[k]];
        base[k]  = offsets[i[k]];
        if (rng.target_batch) batch[k] = rng.target_batch[i[k]];
      }
    }
    if (rng.target_batch) {
#pragma unroll
      for (int k = 0; k < K; k++)
        if (i[k] < n) i_local[k] = i[k] - rng.target_seg[batch[k]];
    }
  }
  // ---- stage 2: the draws (ALU work under the latency of stage 1) -----------------------------------------------------
  int32_t r_draw[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    r_draw[k] = 0;
    if (i[k] < n && hl < M) {
      const uint64_t random_seed = rng.seeds_dev ? rng.seeds_dev[rng.target_batch ? batch[k] : 0] : rng.seed;
      if (i_local[k] < (1 << 26)) {
        Pcg32 g(random_seed, (uint32_t)(i_local[k] * 32 + hl), Pcg32::table_tag{});   // stream layout: 32 per seed, always
        r_draw[k] = g.next_i31();
      } else {
        Pcg32 g(random_seed, stream_id(i_local[k], 32, hl));
        r_draw[k] = g.next_i31();
      }
    }
  }
  // ---- stage 3: Fisher-Yates resolved with shuffles (see sample_uniform_halfwave_kernel), then the K picks in flight ----
  int a[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const bool pick = N[k] > M;
    a[k]            = hl;
    if (__ballot(pick) != 0ull) {   // (wave-uniform)
      int r = 0;
      if (pick && hl < M) r = r_draw[k] % (N[k] - hl);
      const int tail = N[k] - hl - 1;
      int p1 = -1, p2 = -1;
      auto step = [&](int s) {
        const int rs    = __shfl(r, hb + s, 64);
        const bool prev = s < hl;
        p1 = (prev && rs == r) ? s : p1;
        p2 = (prev && rs == tail) ? s : p2;
      };
      if constexpr (EXACT) {
#pragma unroll
        for (int s = 0; s + 1 < GW; s++) step(s);
      } else {
        for (int s = 0; s + 1 < M; s++) step(s);
      }
      int root = p2 >= 0 ? p2 : hl;
#pragma unroll
      for (int round = 0; round < kRounds; round++) root = __shfl(root, hb + root, 64);
      const int val = N[k] - root - 1;
      const int vp1 = __shfl(val, hb + max(p1, 0), 64);
      if (pick) a[k] = p1 >= 0 ? vp1 : r;
    }
  }
  ColT v[K];
  bool on[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    on[k] = live_group && hl < (N[k] > M ? M : N[k]);
    v[k]  = on[k] ? col_at<ColT>(col, start[k] + a[k]) : (ColT)0;
  }
#pragma unroll
  for (int k = 0; k < K; k++)
    if (on[k]) emit<ColT>(dst, src_lid, edge_gid, (int64_t)base[k] + hl, v[k], i[k], start[k] + a[k]);
  }   // chunk
}

// ---- 32 < M <= 1024: one workgroup per seed ------------------------------------------------
constexpr int kHashSlots = 4096;  // >= 2 * (M + M) touched positions, power of two

__device__ __forceinline__ int lds_hash_find(const int* keys, int pos)
{
  uint32_t h = ((uint32_t)pos * 2654435761u) & (kHashSlots - 1);
  while (true) {
    int k = keys[h];
    if (k == pos || k == -1) return (int)h;
    h = (h + 1) & (kHashSlots - 1);
  }
}

template <typename SeedT, typename ColT>
__global__ void __launch_bounds__(256) sample_uniform_block_kernel(const int64_t* __restrict__ row_ptr,
                                                                   const ColT* __restrict__ col,
                                                                   const SeedT* __restrict__ seeds,
                                                                   dev_count n_,
                                                                   int M,
                                                                   int B,
                                                                   int items,
                                                                   rng_plan rng,
                                                                   const int* __restrict__ offsets,
                                                                   ColT* __restrict__ dst,
                                                                   int* __restrict__ src_lid,
                                                                   int64_t* __restrict__ edge_gid)
{
  __shared__ int r[1024];
  __shared__ int a[1024];
  __shared__ int hkeys[kHashSlots];
  __shared__ int hvals[kHashSlots];
  const int i = blockIdx.x;
  if (i >= n_.get()) return;
  const int64_t nid   = (int64_t)seeds[i];
  const int64_t start = row_ptr[nid];
  const int N         = (int)(row_ptr[nid + 1] - start);
  if (N <= 0) return;
  const int base = offsets[i];
  if (N <= M) {
    for (int j = threadIdx.x; j < N; j += blockDim.x)
      emit<ColT>(dst, src_lid, edge_gid, (int64_t)base + j, col_at<ColT>(col, start + j), i, start + j);
    return;
  }
  for (int h = threadIdx.x; h < kHashSlots; h += blockDim.x) hkeys[h] = -1;
  if ((int)threadIdx.x < B) {
    uint64_t random_seed;
    int i_local;
    rng.resolve(i, random_seed, i_local);
    Pcg32 g(random_seed, stream_id(i_local, B, threadIdx.x));
    for (int k = 0; k < items; k++) {
      int id = k * B + threadIdx.x;
      int v  = g.next_i31();
      if (id < M) r[id] = v % (N - id);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 0; t < M; t++) {
      const int rt = r[t], tail = N - t - 1;
      int s1     = lds_hash_find(hkeys, rt);
      int q_rt   = hkeys[s1] == rt ? hvals[s1] : rt;
      int s2     = lds_hash_find(hkeys, tail);
      int q_tail = hkeys[s2] == tail ? hvals[s2] : tail;
      a[t]       = q_rt;
      hkeys[s1]  = rt;  // s1 is either rt's slot or the empty slot its probe ended on
      hvals[s1]  = q_tail;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < M; t += blockDim.x)
    emit<ColT>(dst, src_lid, edge_gid, (int64_t)base + t, col_at<ColT>(col, start + a[t]), i, start + a[t]);
}

// ---- M > 1024: reservoir, slot s keeps max{idx : draw % (idx+1) == s} ----------------------
template <typename SeedT, typename ColT>
__global__ void __launch_bounds__(64) sample_uniform_reservoir_kernel(const int64_t* __restrict__ row_ptr,
                                                                      const ColT* __restrict__ col,
                                                                      const SeedT* __restrict__ seeds,
                                                                      dev_count n_,
                                                                      int M,
                                                                      rng_plan rng,
                                                                      const int* __restrict__ offsets,
                                                                      ColT* __restrict__ dst,
                                                                      int* __restrict__ src_lid,
                                                                      int64_t* __restrict__ edge_gid)
{
  const int i = blockIdx.x;
  if (i >= n_.get()) return;
  const int64_t nid   = (int64_t)seeds[i];
  const int64_t start = row_ptr[nid];
  const int N         = (int)(row_ptr[nid + 1] - start);
  if (N <= 0) return;
  const int64_t base = offsets[i];
  if (N <= M) {
    for (int j = threadIdx.x; j < N; j += blockDim.x)
      emit<ColT>(dst, src_lid, edge_gid, base + j, col_at<ColT>(col, start + j), i, start + j);
    return;
  }
  for (int s = threadIdx.x; s < M; s += blockDim.x) dst[base + s] = (ColT)s;
  __syncthreads();
  if (threadIdx.x < 32) {
    uint64_t random_seed;
    int i_local;
    rng.resolve(i, random_seed, i_local);
    Pcg32 g(random_seed, stream_id(i_local, 32, threadIdx.x));
    for (int idx = M + threadIdx.x; idx < N; idx += 32) {
      int rn = g.next_i31() % (idx + 1);
      if (rn < M) {
        if constexpr (sizeof(ColT) == 8)
          atomicMax(reinterpret_cast<long long*>(dst + base + rn), (long long)idx);
        else
          atomicMax(reinterpret_cast<int*>(dst + base + rn), idx);
      }
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < M; s += blockDim.x) {
    // device-scope load: the slot was updated by L2 atomics, never trust this CU's L1 copy
    int sel = (int)__hip_atomic_load(dst + base + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    emit<ColT>(dst, src_lid, edge_gid, base + s, col_at<ColT>(col, start + sel), i, start + sel);
  }
}

// ---- weighted --------------------------------------------------------------------------------
__device__ __forceinline__ float ares_key(float w, Pcg32& g, bool* redrawn = nullptr)
{
  float u = g.next_f32();
  u       = (float)(-(0.5 + 0.5 * (double)u));
  uint64_t x;
  int zero_draws = -1;
  do {
    x = g.next_u64();
    zero_draws++;
  } while (!x);
  if (redrawn != nullptr && zero_draws > 0) *redrawn = true;   // more than three draws for this key
  int one_bit = __clzll((long long)x) + zero_draws * 64;
  u *= exp2f((float)(-one_bit));
#ifdef WG_TUNE_CHEAP_KEY   // sizing experiment only: what the hop costs without log1pf and the two IEEE divisions
  return u * 1.442695f * __builtin_amdgcn_rcpf(w);
#endif
  return (log1pf(u) / logf(2.0f)) * (1.0f / w);
}

constexpr int kWaveRowCap = 1024;  // rows the one-wave weighted kernel keeps in registers (16 keys per lane)
static_assert(kWaveRowCap == kWaveRowCapDecl, "keep the two in step");

// order-preserving float -> uint (larger key <=> larger uint)
__device__ __forceinline__ uint32_t key_bits(float k)
{
  uint32_t b = __float_as_uint(k);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// One workgroup per seed with deg > M: keys, exact top-M via radix select, emitted in CSR order (ties on the threshold
// key: lowest neighbour index first).  B = the reference's stream layout (lane j of its B-thread block owns neighbours
// j, j+B, ... and draws their keys one after the other from stream seed*B+j); T >= B = threads really used.  With
// T > B thread t starts at neighbour t — stream t % B positioned 3*(t/B) draws in (one key = 3 draws unless a 64-bit
// draw comes back 0, probability 2^-64) — and hops T/B keys at a time with a table jump, so a 100k-candidate row is
// keyed by 1024 threads instead of 128.  A thread that does see a zero draw raises a flag and the row is redone the
// sequential way, which keeps the result exact.  Keys live in LDS when the row fits (kLdsKeys), else in scratch.
constexpr int kLdsKeys = kWeightedLdsKeys;  // 48 KB of the 64 KB static LDS a workgroup may declare

template <typename SeedT, typename ColT, typename WeightT, int B, int T>
__global__ void __launch_bounds__(T) sample_weighted_kernel(const int64_t* __restrict__ row_ptr,
                                                            const ColT* __restrict__ col,
                                                            const WeightT* __restrict__ weight,
                                                            const SeedT* __restrict__ seeds,
                                                            dev_count n_,
                                                            int M,
                                                            rng_plan rng,
                                                            const int* __restrict__ offsets,
                                                            uint32_t* __restrict__ slab,
                                                            int64_t slab_len,
                                                            ColT* __restrict__ dst,
                                                            int* __restrict__ src_lid,
                                                            int64_t* __restrict__ edge_gid,
                                                            int* __restrict__ lists /*nullable*/,
                                                            int list_cap,
                                                            int redo = 0 /*1: the rows the pruned kernels handed back*/)
{
  static_assert(T % B == 0 && T % 64 == 0, "threads must be a multiple of the stream layout and of the wave");
  __shared__ uint32_t lds_keys[kLdsKeys];
  __shared__ int hist[256];
  __shared__ int sh_digit, sh_need, sh_redo;
  __shared__ int wave_cnt[2][T / 64];
  // persistent workgroups: the rows to do (all seeds, or the listed long ones) are dealt out round-robin; a workgroup
  // owns ONE scratch slab of slab_len keys (>= the longest row) for the rows that do not fit LDS
  // `lists` (biased hop with 0 < M <= 256): the huge rows (list 4) then the long ones (list 3), dealt out through a queue
  // head so that a workgroup stuck on a 150k-candidate hub does not hold back the rows a static deal would have given it
  __shared__ int sh_li;
  const int n_live  = n_.get();
  const int n_huge  = lists && !redo ? lists[8] : 0;
  const int count   = lists ? (redo ? lists[kWeightedRedoCount] : n_huge + lists[7]) : n_live;
  uint32_t* gkeys   = slab + (int64_t)blockIdx.x * slab_len;
  int li            = blockIdx.x;
  while (true) {
  __syncthreads();  // the previous row's readers of the shared counters are done
  if (lists) {
    if (threadIdx.x == 0) sh_li = atomicAdd(lists + (redo ? kWeightedRedoHead : 9), 1);
    __syncthreads();
    li = sh_li;
  }
  if (li >= count) break;
  const int i = lists ? (redo ? lists[kWeightedListHead + (int64_t)kWeightedRedoList * list_cap + li]
                         : li < n_huge ? lists[kWeightedListHead + 8 * (int64_t)list_cap + li]
                                       : lists[kWeightedListHead + 7 * (int64_t)list_cap + (li - n_huge)])
                      : li;
  if (!lists) li += gridDim.x;
  if (i >= n_live) continue;
  uint64_t random_seed;
  int i_rng;
  rng.resolve(i, random_seed, i_rng);
  const int64_t nid   = (int64_t)seeds[i];
  const int64_t start = row_ptr[nid];
  const int N         = (int)(row_ptr[nid + 1] - start);
  if (N <= 0) continue;
  const int64_t base = offsets[i];
  if (M <= 0 || N <= M) {
    for (int j = threadIdx.x; j < N; j += T)
      emit<ColT>(dst, src_lid, edge_gid, base + j, col_at<ColT>(col, start + j), i, start + j);
    continue;
  }
  const bool in_lds = N <= kLdsKeys;
  // keys written to global scratch are re-read by other lanes of this workgroup, and neighbouring workgroups' key
  // segments share cache lines: those re-reads are device-scope loads so they never hit a stale line in this CU's L1
  auto put = [&](int id, uint32_t k) { if (in_lds) lds_keys[id] = k; else gkeys[id] = k; };
  auto get = [&](int id) -> uint32_t {
    return in_lds ? lds_keys[id] : __hip_atomic_load(gkeys + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (threadIdx.x == 0) sh_redo = 0;
  __syncthreads();
  {
    // stream `lane` (of B) owns neighbours lane, lane + B, ... = L keys drawn one after the other; its T / B threads take
    // CONTIGUOUS shares of that sequence: ONE jump to the share's first draw (3 draws per key), then plain sequential
    // draws — no per-key jump.  A key that needed more than three draws (a 64-bit draw of 0, p = 2^-64) invalidates the
    // positions after it: every thread reports it and the row is redone the sequential way below.
    constexpr int hop = T / B;
    const int lane = threadIdx.x % B, part = threadIdx.x / B;
    const int L     = lane < N ? (N - lane + B - 1) / B : 0;          // keys of this stream
    const int share = (L + hop - 1) / hop;
    const int m0 = part * share, m1 = min(L, m0 + share);
    bool redrawn = false;
    if (m0 < m1) {
      const int64_t sid = (int64_t)i_rng * B + lane;
      Pcg32 g = (sid < (1ll << 31)) ? Pcg32(random_seed, (uint32_t)sid, Pcg32::table_tag{}, 3u * (uint32_t)m0)
                                    : Pcg32(random_seed, stream_id(i_rng, B, lane));
      if (sid >= (1ll << 31)) g.skipahead(3u * (uint64_t)m0);
      for (int m = m0; m < m1; m++) {
        const int id = lane + m * B;
        put(id, key_bits(ares_key((float)weight[start + id], g, &redrawn)));
      }
    }
    if (redrawn) sh_redo = 1;
  }
  __syncthreads();
  if (T > B && sh_redo) {  // a 64-bit draw was 0 somewhere: redo the row with the sequential stream walk
    if (threadIdx.x < B) {
      Pcg32 g = stream_generator(random_seed, i_rng, B, threadIdx.x);
      for (int id = threadIdx.x; id < N; id += B) put(id, key_bits(ares_key((float)weight[start + id], g)));
    }
    __syncthreads();
  }
  uint32_t prefix = 0, prefix_mask = 0;
  int need = M;  // how many keys we still have to take among those matching `prefix`
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int h = threadIdx.x; h < 256; h += T) hist[h] = 0;
    __syncthreads();
    for (int id = threadIdx.x; id < N; id += T) {
      uint32_t k = get(id);
      if ((k & prefix_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      // highest digit d with  #(digits > d) < need <= #(digits >= d):  lane l owns bins 4l..4l+3, a suffix sum over
      // the lanes finds the owning lane (exactly one: at least `need` keys match the prefix), which scans its 4 bins
      const int l  = threadIdx.x;
      const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
      const int own = h0 + h1 + h2 + h3;
      int inc = own;
      for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_down(inc, off);
        if (l + off < 64) inc += v;
      }
      int acc = inc - own;  // keys in the bins of higher lanes
      if (acc < need && need <= inc) {
        int d = 4 * l + 3;
        if (acc + h3 < need) {
          acc += h3;
          d = 4 * l + 2;
          if (acc + h2 < need) {
            acc += h2;
            d = 4 * l + 1;
            if (acc + h1 < need) {
              acc += h1;
              d = 4 * l;
            }
          }
        }
        sh_digit = d;
        sh_need  = need - acc;
      }
    }
    __syncthreads();
    prefix |= (uint32_t)sh_digit << shift;
    prefix_mask |= 255u << shift;
    need = sh_need;
    __syncthreads();
  }
  // prefix == the M-th largest key; take every key > prefix and the first `need` keys == prefix.
  const uint32_t thr = prefix;
  int out_run = 0, tie_run = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int chunk = 0; chunk < N; chunk += T) {
    int id     = chunk + threadIdx.x;
    uint32_t k = id < N ? get(id) : 0u;
    bool gt    = id < N && k > thr;
    bool eq    = id < N && k == thr;
    uint64_t meq = __ballot(eq);
    int eq_before_in_wave = __popcll(meq & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[1][wave] = __popcll(meq);
    __syncthreads();
    int eq_before = tie_run + eq_before_in_wave, eq_total = 0;
    for (int w = 0; w < T / 64; w++) {
      if (w < wave) eq_before += wave_cnt[1][w];
      eq_total += wave_cnt[1][w];
    }
    bool take      = gt || (eq && eq_before < need);
    uint64_t mtake = __ballot(take);
    int before_in_wave = __popcll(mtake & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[0][wave] = __popcll(mtake);
    __syncthreads();
    int before = out_run + before_in_wave, total = 0;
    for (int w = 0; w < T / 64; w++) {
      if (w < wave) before += wave_cnt[0][w];
      total += wave_cnt[0][w];
    }
    if (take) emit<ColT>(dst, src_lid, edge_gid, base + before, col_at<ColT>(col, start + id), i, start + id);
    out_run += total;
    tie_run += eq_total;
    __syncthreads();
  }
  }  // persistent loop over rows
}

// Rows of at most LANES (16 / 32 / 64) candidates: ONE key per lane, 64 / LANES rows per wave (the rows of a size class,
// listed by the count kernel).  Neighbour j < 64 of the reference's 128-thread block is drawn from stream seed*128 + j, so
// lane hl of a group draws exactly that key; the 31-step bitwise search for the M-th largest key and the ballot prefix sums
// of the emission are shared by the rows of the wave (masked to the lane group), which is what the short rows of a
// mini-batch frontier were paying a whole wave each for.
template <typename SeedT, typename ColT, typename WeightT, int LANES>
__global__ void __launch_bounds__(256) sample_weighted_group_kernel(const int64_t* __restrict__ row_ptr,
                                                                    const ColT* __restrict__ col,
                                                                    const WeightT* __restrict__ weight,
                                                                    const SeedT* __restrict__ seeds,
                                                                    int M,
                                                                    rng_plan rng,
                                                                    const int* __restrict__ offsets,
                                                                    ColT* __restrict__ dst,
                                                                    int* __restrict__ src_lid,
                                                                    int64_t* __restrict__ edge_gid,
                                                                    const int* __restrict__ lists,
                                                                    int list_cap,
                                                                    int cls)
{
  const int lane = threadIdx.x & 63;
  const int hl = lane & (LANES - 1), hb = lane & ~(LANES - 1);
  const int64_t li  = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES);
  const bool active = li < (int64_t)lists[cls];
  if (__ballot(active) == 0ull) return;
  int i = 0, N = 0;
  int64_t start = 0, base = 0;
  uint32_t k = 0u;  // below every real key
  if (active) {
    i = lists[kWeightedListHead + (int64_t)cls * list_cap + li];
    uint64_t random_seed;
    int i_rng;
    rng.resolve(i, random_seed, i_rng);
    const int64_t nid = (int64_t)seeds[i];
    start             = row_ptr[nid];
    N                 = (int)(row_ptr[nid + 1] - start);   // M < N <= LANES by construction of the list
    base              = offsets[i];
    if (hl < N) {
      Pcg32 g = stream_generator(random_seed, i_rng, 128, hl);
      k       = key_bits(ares_key((float)weight[start + hl], g));
    }
  }
  const uint64_t gm = LANES == 64 ? ~0ull : (((1ull << LANES) - 1ull) << hb);
  uint32_t prefix = 0;
  int need        = M;
  for (int bit = 31; bit >= 0; bit--) {
    const uint32_t cand = prefix | (1u << bit);
    const uint32_t hi   = ~((1u << bit) - 1u);
    const int cnt       = __popcll(__ballot((k & hi) == cand) & gm);
    if (cnt >= need) prefix = cand; else need -= cnt;
  }
  // prefix == the M-th largest key of my row: every key above it and the first `need` equal to it (index order)
  const uint64_t below = ((1ull << lane) - 1ull) & gm;
  const bool eq        = active && hl < N && k == prefix;
  const uint64_t meq   = __ballot(eq) & gm;
  const bool take      = active && hl < N && (k > prefix || (eq && __popcll(meq & below) < need));
  const uint64_t mt    = __ballot(take) & gm;
  if (take) emit<ColT>(dst, src_lid, edge_gid, base + __popcll(mt & below), col_at<ColT>(col, start + hl), i, start + hl);
}

// One WAVE per row for rows of up to 64*KMAX candidates (size classes 3 .. 6 of the count kernel's lists: KMAX = 2, 4, 8,
// 16; B = 128 stream layout, i.e. M <= 256): the keys stay in registers (slot s of lane l = neighbour (s/2)*128 + (s%2)*64
// + l, drawn from stream l or l+64 exactly as lane l / l+64 of the reference's 128-thread block would), the M-th largest
// key is found by a bitwise search whose counts are wave ballots, and the picks are emitted in CSR order with ballot
// prefix sums.  No LDS, no scratch, no barrier.  The slot count is a compile-time constant per class, so nothing in the
// search is under a per-row condition (an empty slot holds 0, which is below every real key and matches no candidate).
template <typename SeedT, typename ColT, typename WeightT, int KMAX>
__global__ void __launch_bounds__(256) sample_weighted_wave_kernel(const int64_t* __restrict__ row_ptr,
                                                                   const ColT* __restrict__ col,
                                                                   const WeightT* __restrict__ weight,
                                                                   const SeedT* __restrict__ seeds,
                                                                   int M,
                                                                   rng_plan rng,
                                                                   const int* __restrict__ offsets,
                                                                   ColT* __restrict__ dst,
                                                                   int* __restrict__ src_lid,
                                                                   int64_t* __restrict__ edge_gid,
                                                                   const int* __restrict__ lists,
                                                                   int list_cap,
                                                                   int cls)
{
  const int lane   = threadIdx.x & 63;
  const int64_t li = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (li >= (int64_t)lists[cls]) return;
  const int i = lists[kWeightedListHead + (int64_t)cls * list_cap + li];
  uint64_t random_seed;
  int i_rng;
  rng.resolve(i, random_seed, i_rng);
  const int64_t nid   = (int64_t)seeds[i];
  const int64_t start = row_ptr[nid];
  // (one row per wave: make that visible to the compiler, so that `s * 64 < N` below is a scalar branch)
  const int N         = __builtin_amdgcn_readfirstlane((int)(row_ptr[nid + 1] - start));   // M < N <= 64 * KMAX (list)
  const int64_t base  = offsets[i];
  uint32_t k[KMAX];
  {
    Pcg32 ga = stream_generator(random_seed, i_rng, 128, lane);
    Pcg32 gb = stream_generator(random_seed, i_rng, 128, lane + 64);
#pragma unroll
    for (int s = 0; s < KMAX; s++) {
      const int id = (s >> 1) * 128 + (s & 1) * 64 + lane;
      k[s]         = 0u;  // below every real key (key_bits of any float, -inf and NaN included, is > 0)
      // the wave-uniform branch keeps the KMAX key computations in separate blocks: as straight-line predicated code the
      // scheduler interleaves them all (238 VGPRs at KMAX = 16, one wave per SIMD)
      if (s * 64 < N) {
        if (id < N) k[s] = key_bits(ares_key((float)weight[start + id], (s & 1) ? gb : ga));
      }
    }
  }
  // Bitwise search for the M-th largest key, shortened at both ends.  (1) Leading bits on which ALL keys of the row agree
  // (sign, most of the exponent: the keys are log2(u)/w of one row) need no counting: the search starts below them.
  // (2) It stops as soon as exactly `need` keys match the decided bits: those and everything above them ARE the top M,
  // whatever the undecided low bits say (no tie can straddle the cut).  Same selection as the full 32-step search.
  uint32_t k_or = 0u, k_and = ~0u;
#pragma unroll
  for (int s = 0; s < KMAX; s++) {
    k_or |= k[s];
    k_and &= k[s] != 0u ? k[s] : ~0u;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    k_or |= __shfl_xor(k_or, d, 64);
    k_and &= __shfl_xor(k_and, d, 64);
  }
  k_or  = __builtin_amdgcn_readfirstlane(k_or);    // wave-uniform after the butterfly: keep the loop scalar
  k_and = __builtin_amdgcn_readfirstlane(k_and);
  const uint32_t differ = k_or ^ k_and;
  const int top         = differ ? 31 - __clz(differ) : -1;   // highest bit on which two keys differ (-1: all equal)
  uint32_t hi     = top < 0 ? ~0u : top >= 31 ? 0u : ~((2u << top) - 1u);   // decided bits = the common leading bits
  uint32_t prefix = k_and & hi;
  int need        = M;
  int match       = N;   // keys that carry `prefix` in the decided bits (every live key, so far)
#pragma unroll 1
  for (int bit = top; bit >= 0 && match != need; bit--) {
    const uint32_t cand = prefix | (1u << bit);
    hi |= 1u << bit;
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < KMAX; s++) cnt += __popcll(__ballot((k[s] & hi) == cand));
    if (cnt >= need) {
      prefix = cand;
      match  = cnt;
    } else {
      need -= cnt;
      match -= cnt;
    }
  }
  // decided bits `hi`, value `prefix`: take every key above it (in the decided bits) and the first `need` equal to it in
  // index order (after a full search hi == ~0 and this is "the M-th largest key and its ties")
  const uint64_t below = (1ull << lane) - 1ull;
  int out_run = 0, tie_run = 0;
#pragma unroll
  for (int s = 0; s < KMAX; s++) {
    const int id       = (s >> 1) * 128 + (s & 1) * 64 + lane;
    const uint32_t kd  = k[s] & hi;
    const bool eq      = k[s] != 0u && kd == prefix;
    const uint64_t meq = __ballot(eq);
    const bool take    = (k[s] != 0u && kd > prefix) || (eq && tie_run + __popcll(meq & below) < need);
    const uint64_t mt  = __ballot(take);
    if (take) emit<ColT>(dst, src_lid, edge_gid, base + out_run + __popcll(mt & below), col_at<ColT>(col, start + id), i, start + id);
    out_run += __popcll(mt);
    tie_run += __popcll(meq);
    __builtin_amdgcn_sched_barrier(0);   // (16 unrolled slots x three 64-bit output addresses hoisted together: 238 VGPRs)
  }
}

// ---- threshold pruning (M <= 32) ---------------------------------------------------------------------------------------
// The key of a candidate is  K = log1pf(-a) / logf(2) * (1 / w)  with  a = |u| 2^-one_bit  taken from draws 1 and 3 of
// the candidate's three (the middle draw only matters when the third one is 0, p = 2^-32).  K costs ~130 instructions
// (log1pf, two IEEE divisions); what decides whether a candidate can be among the M largest keys does not:
//     |K| >= a / (ln 2 * w)            (log1p(x) <= x),
// so  magU = a * rcp(w) * log2(e) * (1 - 2^-16)  is a lower bound of |K| AS COMPUTED (the margin covers v_rcp_f32's ulp
// and the <= 4 ulp of the exact expression).  A row is then sampled in three moves:
//   1. magU of every candidate (two independent 64-bit multiplies per key: state -> state of draw 3 and -> next key,
//      A^2 and A^3 jumps with per-stream constants, instead of three dependent steps);
//   2. the ~M candidates of smallest magU (bitwise search that stops as soon as <= 64 are left) get their EXACT keys, one
//      per lane; the M-th largest of those, m*, is a key that M candidates reach;
//   3. every candidate with magU > |m*| is out (|K| >= magU > |m*|: strictly below M others, ties included); the few
//      left — almost always the same ones — are selected exactly as before (M largest keys, ties to the lowest index,
//      CSR order).
// Same result as computing every key.  Rows where this does not apply — a third draw of 0, a weight that is not a
// positive normal float, more than 64 candidates left — are appended to the redo list and done by the exact workgroup
// kernel afterwards.
constexpr uint64_t kPcgA  = Pcg32::kMult;
constexpr uint64_t kPcgA2 = kPcgA * kPcgA;
constexpr uint64_t kPcgA3 = kPcgA2 * kPcgA;
constexpr float kMagScale = 1.4426950408889634f * (1.0f - 1.0f / 65536.0f);

__device__ __forceinline__ uint32_t pcg_output(uint64_t old)
{
  const uint32_t x   = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  const uint32_t rot = (uint32_t)(old >> 59u);
  return (x >> rot) | (x << ((0u - rot) & 31u));
}

struct KeyStream {   // a PCG32 stream consumed one A-Res key (three draws) at a time
  uint64_t s, c2, c3;
  __device__ __forceinline__ explicit KeyStream(const Pcg32& g)
    : s(g.state), c2(g.inc * (kPcgA + 1u)), c3(g.inc * (kPcgA2 + kPcgA + 1u))
  {
  }
  // a = |u| 2^-one_bit of the next key; `rare` is raised when the third draw is 0 (the exact path must look at the
  // second draw, and at further ones if that is 0 too)
  __device__ __forceinline__ float next_mag(bool& rare)
  {
    const uint64_t s0 = s;
    const uint64_t s2 = s0 * kPcgA2 + c2;
    s                 = s0 * kPcgA3 + c3;
    const uint32_t d1 = pcg_output(s0), d3 = pcg_output(s2);
    rare |= d3 == 0u;
    // (float)(-(0.5 + 0.5 * (double)f)), f = (d1 >> 8) / 2^24  ==  -(float)(2^24 + (d1 >> 8)) / 2^25 (one RNE rounding of
    // the same 25-bit integer), then the exact scaling by 2^-clz
    return ldexpf((float)((d1 >> 8u) + (1u << 24)), -25 - __clz((int)d3));
  }
};

__device__ __forceinline__ float ares_key_exact(float a, float w) { return (log1pf(-a) / logf(2.0f)) * (1.0f / w); }
// weights the bound arithmetic is safe for: positive, normal, far from overflow of a * rcp(w)
__device__ __forceinline__ bool weight_is_odd(float w) { return ((__float_as_uint(w) >> 23u) - 27u) > 200u; }

__device__ __forceinline__ void redo_append(int* lists, int list_cap, int i)
{
  const int p = atomicAdd(lists + kWeightedRedoCount, 1);
  lists[kWeightedListHead + (int64_t)kWeightedRedoList * list_cap + p] = i;
}

// M-th largest of one key per lane (0 = no key): decided bits `hi`, their value `prefix`, and how many keys equal to it
// (in the decided bits) are still to be taken; stops as soon as that is all of them
__device__ __forceinline__ void lane_select(uint32_t kb, int n_keys, int M, uint32_t& prefix, uint32_t& hi, int& need)
{
  prefix    = 0u;
  hi        = 0u;
  need      = M;
  int match = n_keys;
#pragma unroll 1
  for (int bit = 31; bit >= 0 && match != need; bit--) {
    const uint32_t cand = prefix | (1u << bit);
    hi |= 1u << bit;
    const int cnt = __popcll(__ballot(kb != 0u && (kb & hi) == cand));
    if (cnt >= need) {
      prefix = cand;
      match  = cnt;
    } else {
      need -= cnt;
      match -= cnt;
    }
  }
}

// One WAVE per row of up to 64 * KMAX candidates, M <= 32 (same size classes and stream layout as
// sample_weighted_wave_kernel: slot s of lane l = neighbour 64 s + l, drawn from stream l or l + 64).
template <typename SeedT, typename ColT, typename WeightT, int KMAX>
__global__ void __launch_bounds__(256) sample_weighted_wave_pruned_kernel(const int64_t* __restrict__ row_ptr,
                                                                          const ColT* __restrict__ col,
                                                                          const WeightT* __restrict__ weight,
                                                                          const SeedT* __restrict__ seeds,
                                                                          int M,
                                                                          rng_plan rng,
                                                                          const int* __restrict__ offsets,
                                                                          ColT* __restrict__ dst,
                                                                          int* __restrict__ src_lid,
                                                                          int64_t* __restrict__ edge_gid,
                                                                          int* __restrict__ lists,
                                                                          int list_cap,
                                                                          int cls,
                                                                          int force_redo)
{
  __shared__ int sh_id[4][64];
  __shared__ float sh_a[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t li = (int64_t)blockIdx.x * 4 + wv;
  if (li >= (int64_t)lists[cls]) return;
  const int i = lists[kWeightedListHead + (int64_t)cls * list_cap + li];
  uint64_t random_seed;
  int i_rng;
  rng.resolve(i, random_seed, i_rng);
  const int64_t nid   = (int64_t)seeds[i];
  const int64_t start = row_ptr[nid];
  const int N         = __builtin_amdgcn_readfirstlane((int)(row_ptr[nid + 1] - start));   // M < N <= 64 * KMAX (list)
  const int64_t base  = offsets[i];
  uint32_t k[KMAX];   // magU bits (positive floats order like their bits); ~0 = no candidate
  float av[KMAX];
  bool rare = force_redo != 0;
  {
    // all weights of the row first, as one batch of unconditional loads (a load under the per-slot branch is waited for
    // before the next one is issued: KMAX memory latencies in a row, which is what the kernel then takes)
    float wv[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; s++) wv[s] = (float)weight[start + min(s * 64 + lane, N - 1)];
    KeyStream ga(stream_generator(random_seed, i_rng, 128, lane));
    KeyStream gb(stream_generator(random_seed, i_rng, 128, lane + 64));
#pragma unroll
    for (int s = 0; s < KMAX; s++) {
      const int id = s * 64 + lane;
      k[s]         = ~0u;
      av[s]        = 0.f;
      if (s * 64 < N) {   // wave-uniform
        if (id < N) {
          const float w = wv[s];
          const float a = ((s & 1) ? gb : ga).next_mag(rare);
          rare |= weight_is_odd(w);
          av[s] = a;
          k[s]  = __float_as_uint(a * __builtin_amdgcn_rcpf(w) * kMagScale);
        }
      }
    }
  }
  if (__ballot(rare) != 0ull) {
    if (lane == 0) redo_append(lists, list_cap, i);
    return;
  }
  const uint64_t below = (1ull << lane) - 1ull;
  // (1) the candidates of smallest magU: bitwise search for the M-th smallest, from the highest bit on which two keys
  // differ, until at most 64 candidates are at or below the decided prefix
  uint32_t k_or = 0u, k_and = ~0u;
#pragma unroll
  for (int s = 0; s < KMAX; s++) {
    k_or |= k[s] != ~0u ? k[s] : 0u;
    k_and &= k[s];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    k_or |= __shfl_xor(k_or, d, 64);
    k_and &= __shfl_xor(k_and, d, 64);
  }
  k_or  = __builtin_amdgcn_readfirstlane(k_or);
  k_and = __builtin_amdgcn_readfirstlane(k_and);
  const uint32_t differ = k_or ^ k_and;
  const int top         = differ ? 31 - __clz(differ) : -1;
  uint32_t hi     = top < 0 ? ~0u : top >= 31 ? 0u : ~((2u << top) - 1u);
  uint32_t prefix = k_and & hi;
  int need = M, match = N, under = 0;   // under: candidates strictly below the prefix range
#pragma unroll 1
  for (int bit = top; bit >= 0 && match != need && under + match > 64; bit--) {
    hi |= 1u << bit;
    int cnt0 = 0;   // candidates in the prefix range with this bit clear
#pragma unroll
    for (int s = 0; s < KMAX; s++) cnt0 += __popcll(__ballot((k[s] & hi) == prefix));
    if (cnt0 >= need) {
      match = cnt0;
    } else {
      need -= cnt0;
      under += cnt0;
      match -= cnt0;
      prefix |= 1u << bit;
    }
  }
  if (under + match > 64) {   // (equal bounds en masse)
    if (lane == 0) redo_append(lists, list_cap, i);
    return;
  }
  uint32_t bound = prefix | ~hi;   // candidates with magU <= bound: under + match of them, at least M
  uint32_t kb = 0u, sel_prefix = 0u, sel_hi = 0u;
  int cid = 0, sel_need = 0, n_sel = 0;
#pragma unroll 1
  for (int round = 0; round < 2; round++) {
    // compaction in CSR order: slot-major == neighbour index order
    n_sel = 0;
#pragma unroll
    for (int s = 0; s < KMAX; s++) {
      const bool in    = k[s] <= bound && k[s] != ~0u;
      const uint64_t m = __ballot(in);
      if (in) {
        const int p  = n_sel + __popcll(m & below);
        sh_id[wv][p] = s * 64 + lane;
        sh_a[wv][p]  = av[s];
      }
      n_sel += __popcll(m);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its own LDS writes are visible once they are done
    kb = 0u;
    if (lane < n_sel) {
      cid           = sh_id[wv][lane];
      const float a = sh_a[wv][lane];
      kb            = key_bits(ares_key_exact(a, (float)weight[start + cid]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next round overwrites the scratch)
    lane_select(kb, n_sel, M, sel_prefix, sel_hi, sel_need);
    if (round == 1) break;
    // (2) |m*|: magnitude of the smallest key the selection takes (the M-th largest, or a tie of it); keys are negative
    // and key_bits = ~bits, so it is ~(smallest taken key_bits) without the sign.  (3) Who else could reach it?
    uint32_t kmin = (kb != 0u && (kb & sel_hi) >= sel_prefix) ? kb : ~0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64));
    const uint32_t mstar = (~kmin) & 0x7fffffffu;
    int extra = 0;
#pragma unroll
    for (int s = 0; s < KMAX; s++) extra += __popcll(__ballot(k[s] > bound && k[s] <= mstar));
    if (extra == 0) break;   // nobody: the selection over the first set stands
    int total = 0;
#pragma unroll
    for (int s = 0; s < KMAX; s++) total += __popcll(__ballot(k[s] <= mstar));   // (~0 is above every magnitude)
    if (total > 64) {
      if (lane == 0) redo_append(lists, list_cap, i);
      return;
    }
    bound = mstar;
  }
  const uint32_t kd  = kb & sel_hi;
  const bool eq      = kb != 0u && kd == sel_prefix;
  const uint64_t meq = __ballot(eq);
  const bool take    = (kb != 0u && kd > sel_prefix) || (eq && __popcll(meq & below) < sel_need);
  const uint64_t mt  = __ballot(take);
  if (take) emit<ColT>(dst, src_lid, edge_gid, base + __popcll(mt & below), col_at<ColT>(col, start + cid), i, start + cid);
}

// Rows of more than 1024 candidates, M <= 32 (size classes 7 and 8: persistent 512-thread workgroups fed by the queue
// head, the huge rows first; stream layout B = 128, a stream's four threads take contiguous shares of its key sequence as
// in sample_weighted_kernel).  The same three moves as the one-wave kernel, streamed: a CANDIDATE LIST in LDS holds
// (neighbour, a) of everything that can still be among the M largest keys, `thr` is the magnitude of a key that M
// candidates are known to reach.  Round 0 (one key per thread): every wave finds the M-th smallest bound among its own 64,
// the smallest of those eight values admits the first candidates, their exact keys give thr.  Then the rest of the row
// streams by — bound, compare, append the rare survivor — and whenever the list could overflow it is COMPRESSED: exact
// keys of the listed candidates, the M largest stay, thr tightens.  The last compression emits.  No key slab, no pass
// over all N keys, ~60 instructions per candidate instead of ~290.
constexpr int kCandCap     = 4096;   // candidate list entries (id, a, key bits: 48 KB of LDS)
constexpr int kChunkRounds = 8;      // rounds of 512 candidates between two looks at the list's fill level; the weights
                                     // of a chunk are requested together (one memory latency per chunk, not per key)
constexpr int kEqCap       = 64;     // candidates tied with the M-th key that the tie-break can rank

template <typename SeedT, typename ColT, typename WeightT>
__global__ void __launch_bounds__(512) sample_weighted_block_pruned_kernel(const int64_t* __restrict__ row_ptr,
                                                                           const ColT* __restrict__ col,
                                                                           const WeightT* __restrict__ weight,
                                                                           const SeedT* __restrict__ seeds,
                                                                           dev_count n_,
                                                                           int M,
                                                                           rng_plan rng,
                                                                           const int* __restrict__ offsets,
                                                                           ColT* __restrict__ dst,
                                                                           int* __restrict__ src_lid,
                                                                           int64_t* __restrict__ edge_gid,
                                                                           int* __restrict__ lists,
                                                                           int list_cap,
                                                                           int force_redo)
{
  constexpr int T = 512, B = 128, W = T / 64;
  __shared__ int c_id[kCandCap];
  __shared__ float c_a[kCandCap];
  __shared__ uint32_t c_kb[kCandCap];
  __shared__ int hist[256];
  __shared__ int sh_li, sh_count, sh_rare, sh_digit, sh_need, sh_keep, sh_eq;
  __shared__ uint32_t sh_thr, sh_kmin, sh_sel[2];
  __shared__ uint32_t sh_wthr[W];
  __shared__ int keep_id[32], eq_id[kEqCap];
  __shared__ float keep_a[32], eq_a[kEqCap];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_live = n_.get();
  const int n_huge = lists[8];
  const int count  = n_huge + lists[7];

  auto append = [&](int id, float a) {
    const int p = atomicAdd(&sh_count, 1);
    if (p < kCandCap) {
      c_id[p] = id;
      c_a[p]  = a;
    } else {
      sh_rare = 1;   // overflow between two looks at the fill level: the row goes to the exact kernel.  (After round 0
                     // the threshold is a key M of the first 512 candidates reach: a chunk of 4096 more adds ~2 M.)
    }
  };

  while (true) {
    __syncthreads();   // the previous row's readers of the shared words are done
    if (tid == 0) sh_li = atomicAdd(lists + 9, 1);
    __syncthreads();
    const int li = sh_li;
    if (li >= count) break;
    const int i = li < n_huge ? lists[kWeightedListHead + 8 * (int64_t)list_cap + li]
                              : lists[kWeightedListHead + 7 * (int64_t)list_cap + (li - n_huge)];
    if (i >= n_live) continue;
    uint64_t random_seed;
    int i_rng;
    rng.resolve(i, random_seed, i_rng);
    const int64_t nid   = (int64_t)seeds[i];
    const int64_t start = row_ptr[nid];
    const int N         = (int)(row_ptr[nid + 1] - start);   // > 1024 >= M (list)
    const int64_t base  = offsets[i];
    if (tid == 0) {
      sh_count = 0;
      sh_rare  = force_redo;
      sh_thr   = 0x7f800000u;
    }
    // COMPRESS: exact keys of the listed candidates, keep the M largest (ties: lowest neighbour index), thr = |smallest
    // kept key|; `last` emits them in CSR order instead of restarting the list with them
    auto compress = [&](bool last) {
      __syncthreads();
      const int n = min(sh_count, kCandCap);   // >= M
      for (int e = tid; e < n; e += T) c_kb[e] = key_bits(ares_key_exact(c_a[e], (float)weight[start + c_id[e]]));
      if (tid == 0) {
        sh_keep = 0;
        sh_eq   = 0;
        sh_kmin = ~0u;
      }
      __syncthreads();
      // the M-th largest key: decided bits `hi`, their value `prefix`, `need` of the keys equal to it (in those bits)
      uint32_t prefix = 0u, hi = 0u;
      int need = M;
      if (n <= 256) {   // one wave, four keys per lane, ballots
        if (wave == 0) {
          uint32_t kb[4];
#pragma unroll
          for (int j = 0; j < 4; j++) kb[j] = lane + 64 * j < n ? c_kb[lane + 64 * j] : 0u;
          int match = n;
#pragma unroll 1
          for (int bit = 31; bit >= 0 && match != need; bit--) {
            const uint32_t cand = prefix | (1u << bit);
            hi |= 1u << bit;
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += __popcll(__ballot(kb[j] != 0u && (kb[j] & hi) == cand));
            if (cnt >= need) {
              prefix = cand;
              match  = cnt;
            } else {
              need -= cnt;
              match -= cnt;
            }
          }
          if (lane == 0) {
            sh_sel[0] = prefix;
            sh_sel[1] = hi;
            sh_need   = need;
          }
        }
        __syncthreads();
        prefix = sh_sel[0];
        hi     = sh_sel[1];
        need   = sh_need;
      } else {          // 4-pass radix select over the list
        for (int shift = 24; shift >= 0; shift -= 8) {
          for (int h = tid; h < 256; h += T) hist[h] = 0;
          __syncthreads();
          for (int e = tid; e < n; e += T) {
            const uint32_t kk = c_kb[e];
            if ((kk & hi) == prefix) atomicAdd(&hist[(kk >> shift) & 255], 1);
          }
          __syncthreads();
          if (tid < 64) {
            const int l  = tid;
            const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            const int own = h0 + h1 + h2 + h3;
            int inc = own;
            for (int off = 1; off < 64; off <<= 1) {
              int v = __shfl_down(inc, off);
              if (l + off < 64) inc += v;
            }
            int acc = inc - own;
            if (acc < need && need <= inc) {
              int d = 4 * l + 3;
              if (acc + h3 < need) {
                acc += h3;
                d = 4 * l + 2;
                if (acc + h2 < need) {
                  acc += h2;
                  d = 4 * l + 1;
                  if (acc + h1 < need) {
                    acc += h1;
                    d = 4 * l;
                  }
                }
              }
              sh_digit = d;
              sh_need  = need - acc;
            }
          }
          __syncthreads();
          prefix |= (uint32_t)sh_digit << shift;
          hi |= 255u << shift;
          need = sh_need;
          __syncthreads();
        }
      }
      // keep what is above the cut; what sits on it goes to the tie list
      for (int e = tid; e < n; e += T) {
        const uint32_t kk = c_kb[e], kd = kk & hi;
        if (kk != 0u && kd > prefix) {
          const int p = atomicAdd(&sh_keep, 1);
          keep_id[p]  = c_id[e];
          keep_a[p]   = c_a[e];
          atomicMin(&sh_kmin, kk);
        } else if (kk != 0u && kd == prefix) {
          const int q = atomicAdd(&sh_eq, 1);
          if (q < kEqCap) {
            eq_id[q] = c_id[e];
            eq_a[q]  = c_a[e];
            atomicMin(&sh_kmin, kk);   // (an untaken tie has the same decided bits: still a key M candidates reach)
          } else {
            sh_rare = 1;
          }
        }
      }
      __syncthreads();
      const int eqc = min(sh_eq, kEqCap);
      if (tid < eqc) {   // the `need` ties of lowest neighbour index
        int rank = 0;
        for (int j = 0; j < eqc; j++) rank += eq_id[j] < eq_id[tid] ? 1 : 0;
        if (rank < need) {
          const int p = atomicAdd(&sh_keep, 1);
          keep_id[p]  = eq_id[tid];
          keep_a[p]   = eq_a[tid];
        }
      }
      __syncthreads();   // sh_keep == M
      if (last) {
        if (tid < M && !sh_rare) {
          const int mine = keep_id[tid];
          int rank = 0;
          for (int j = 0; j < M; j++) rank += keep_id[j] < mine ? 1 : 0;
          emit<ColT>(dst, src_lid, edge_gid, base + rank, col_at<ColT>(col, start + mine), i, start + mine);
        }
      } else {
        if (tid < M) {
          c_id[tid] = keep_id[tid];
          c_a[tid]  = keep_a[tid];
        }
        if (tid == 0) {
          sh_count = M;
          sh_thr   = (~sh_kmin) & 0x7fffffffu;
        }
      }
      __syncthreads();
    };

    // this thread's keys: m0 .. m1 of stream `sl` (neighbours sl + m * B)
    const int sl = tid % B, part = tid / B;
    const int L     = (N - sl + B - 1) / B;
    const int share = (L + 3) / 4;
    const int m0 = part * share, m1 = min(L, m0 + share);
    const int rounds = (((N + B - 1) / B) + 3) / 4;   // the longest share
    const int64_t sid = (int64_t)i_rng * B + sl;
    Pcg32 g0 = (sid < (1ll << 31)) ? Pcg32(random_seed, (uint32_t)sid, Pcg32::table_tag{}, 3u * (uint32_t)m0)
                                   : Pcg32(random_seed, stream_id(i_rng, B, sl));
    if (sid >= (1ll << 31)) g0.skipahead(3u * (uint64_t)m0);
    KeyStream g(g0);
    bool rare = false;
    auto next = [&](float w, float& a, uint32_t& k) {   // bound of the stream's next key, candidate weight w
      a = g.next_mag(rare);
      rare |= weight_is_odd(w);
      k = __float_as_uint(a * __builtin_amdgcn_rcpf(w) * kMagScale);
    };
    auto weight_of = [&](int m) { return (float)weight[start + sl + (int64_t)min(m, m1 - 1) * B]; };   // (m1 >= 1)
    // ---- round 0 ----
    const bool has0 = m0 < m1;
    float a0    = 0.f;
    uint32_t k0 = ~0u;
    if (has0) next(weight_of(m0), a0, k0);
    {
      // M-th smallest bound of this wave's keys (or +inf when it has fewer than M)
      const int nvalid = __popcll(__ballot(has0));
      uint32_t v = 0x7f800000u;
      if (nvalid >= M) {
        uint32_t prefix = 0u, hi = 0u;
        int need = M, match = nvalid;
#pragma unroll 1
        for (int bit = 31; bit >= 0 && match != need; bit--) {
          hi |= 1u << bit;
          const int cnt0 = __popcll(__ballot(has0 && (k0 & hi) == prefix));
          if (cnt0 >= need) {
            match = cnt0;
          } else {
            need -= cnt0;
            match -= cnt0;
            prefix |= 1u << bit;
          }
        }
        v = prefix | ~hi;
      }
      if (lane == 0) sh_wthr[wave] = v;
    }
    __syncthreads();
    uint32_t thr0 = sh_wthr[0];
#pragma unroll
    for (int w = 1; w < W; w++) thr0 = min(thr0, sh_wthr[w]);
    if (has0 && k0 <= thr0) append(m0 * B + sl, a0);
    compress(false);
    uint32_t thr = sh_thr;
    if (has0 && k0 > thr0 && k0 <= thr) append(m0 * B + sl, a0);
    // ---- the rest of the row ----
    for (int r = 1; r < rounds;) {
      // (one decision for the workgroup: a thread that is through the barrier may already be appending again)
      if (__syncthreads_or(sh_count > kCandCap / 2)) compress(false);
      thr = sh_thr;
      float wv[kChunkRounds];
#pragma unroll
      for (int c = 0; c < kChunkRounds; c++) wv[c] = weight_of(m0 + r + c);
#pragma unroll
      for (int c = 0; c < kChunkRounds; c++) {
        const int m = m0 + r + c;
        if (m < m1) {
          float a;
          uint32_t k;
          next(wv[c], a, k);
          if (k <= thr) append(m * B + sl, a);
        }
      }
      r += kChunkRounds;
    }
    if (rare) sh_rare = 1;
    compress(true);
    if (tid == 0 && sh_rare) redo_append(lists, list_cap, i);
  }
}

// rows that are copied whole (deg <= M, or sample-all): 16 lanes per seed, all seeds
template <typename SeedT, typename ColT>
__global__ void __launch_bounds__(256) copy_short_rows_kernel(const int64_t* __restrict__ row_ptr, const ColT* __restrict__ col,
                                                              const SeedT* __restrict__ seeds, dev_count n_, int M,
                                                              const int* __restrict__ offsets, ColT* __restrict__ dst,
                                                              int* __restrict__ src_lid, int64_t* __restrict__ edge_gid)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(t >> 4), hl = (int)(t & 15);
  if (i >= n_.get()) return;
  const int64_t nid   = (int64_t)seeds[i];
  const int64_t start = row_ptr[nid];
  const int N         = (int)(row_ptr[nid + 1] - start);
  if (N <= 0 || N > M) return;
  const int64_t base = offsets[i];
  for (int j = hl; j < N; j += 16) emit<ColT>(dst, src_lid, edge_gid, base + j, col_at<ColT>(col, start + j), i, start + j);
}

template <typename SeedT, typename ColT>
void uniform_launch(const int64_t* row_ptr, const ColT* col, const SeedT* seeds, dev_count n, int M,
                    rng_plan random_seed, const int* offsets, ColT* dst, int* lid, int64_t* gid, hipStream_t stream,
                    const int64_t* row_start = nullptr, const int* row_deg = nullptr, const sample_locality* loc = nullptr)
{
  const int cap = n.host;
  if (cap <= 0) return;
  if (M > 0 && M <= 32 && row_start != nullptr) {
    // the no-sync walk: K seeds per lane group (WGAMD_SAMPLE_K = 1 | 2 | 4 | 8, default 4), optionally in vertex-grouped
    // order (see loc_rec): two short launches build the records, the sampling kernel walks them
    static const int K = [] {
      const char* e = getenv("WGAMD_SAMPLE_K");
      const int k   = e ? atoi(e) : 4;
      return k == 1 || k == 2 ? k : 4;
    }();
    const loc_rec* recs = nullptr;
    if (loc != nullptr) {
      auto* r = static_cast<loc_rec*>(loc->recs);
      locality_hist_kernel<SeedT><<<kLocBlocks, kLocThreads, 0, stream>>>(seeds, n, loc->shift, loc->hist);
      locality_scatter_kernel<SeedT><<<kLocBlocks, kLocThreads, 0, stream>>>(seeds, n, loc->shift, loc->hist, row_start, row_deg,
                                                                            offsets, random_seed, r);
      recs = r;
    }
    // (grid: a multiple of 8 — the grouped order deals chunk c to XCD c % 8 — bounded by kWalkGrid, striding over the seeds)
#define WG_MULTI(GW, KK)                                                                                             \
  sample_uniform_multi_kernel<SeedT, ColT, GW, KK, WG_EXACT>                                                          \
    <<<(int)std::min<int64_t>((ceil_div((int64_t)cap, 4 * (64 / GW) * KK) + 15) / 8 * 8, kWalkGrid), 256, 0, stream>>>(           \
      col, n, M, random_seed, offsets, dst, lid, gid, row_start, row_deg, recs)
#define WG_MULTI_K(GW)                                                                                               \
  do {                                                                                                               \
    if (K == 1) WG_MULTI(GW, 1); else if (K == 2) WG_MULTI(GW, 2); else WG_MULTI(GW, 4);                              \
  } while (0)
    // group width = the fan-out where a kernel is built for it (the BASELINE fan-outs), else the next of 8 / 16 / 32
    static const bool exact = getenv("WGAMD_SAMPLE_EXACT_WIDTH") == nullptr || atoi(getenv("WGAMD_SAMPLE_EXACT_WIDTH")) != 0;
#define WG_EXACT true
    if (exact && M == 5) WG_MULTI_K(5);
    else if (exact && M == 10) WG_MULTI_K(10);
    else if (exact && M == 15) WG_MULTI_K(15);
    else if (exact && M == 25) WG_MULTI_K(25);
    else {
#undef WG_EXACT
#define WG_EXACT false
      if (M <= 8) WG_MULTI_K(8);
      else if (M <= 16) WG_MULTI_K(16);
      else WG_MULTI_K(32);
    }
#undef WG_EXACT
#undef WG_MULTI_K
#undef WG_MULTI
    WG_HIP_CHECK(hipGetLastError());
    return;
  }
  if (M <= 0) {
    // sample-all: every row is copied whole (rows can be long -> workgroup copy path)
    sample_uniform_block_kernel<SeedT, ColT><<<cap, 64, 0, stream>>>(row_ptr, col, seeds, n, 0x7fffffff, 32, 1,
                                                                     random_seed, offsets, dst, lid, gid);
  } else if (M <= 16) {
    sample_uniform_halfwave_kernel<SeedT, ColT, 16><<<ceil_div((int64_t)cap * 16, 256), 256, 0, stream>>>(
      row_ptr, col, seeds, n, M, random_seed, offsets, dst, lid, gid, row_start, row_deg);
  } else if (M <= 32) {
    sample_uniform_halfwave_kernel<SeedT, ColT, 32><<<ceil_div((int64_t)cap * 32, 256), 256, 0, stream>>>(
      row_ptr, col, seeds, n, M, random_seed, offsets, dst, lid, gid, row_start, row_deg);
  } else if (M <= 1024) {
    const int B = ref_block_threads(M);
    sample_uniform_block_kernel<SeedT, ColT><<<cap, B < 64 ? 64 : B, 0, stream>>>(
      row_ptr, col, seeds, n, M, B, ref_items_per_thread(M), random_seed, offsets, dst, lid, gid);
  } else {
    sample_uniform_reservoir_kernel<SeedT, ColT><<<cap, 64, 0, stream>>>(row_ptr, col, seeds, n, M, random_seed,
                                                                         offsets, dst, lid, gid);
  }
  WG_HIP_CHECK(hipGetLastError());
}

// Weighted hop over n seeds (live count n.get()).  M in 1..256: the one-wave kernel covers rows of up to kWaveRowCap
// candidates, the persistent workgroup kernel the listed longer ones (long rows first: their tail then drains while the
// wave kernel fills the GPU); otherwise (256-thread stream layout / sample-all) the workgroup kernel walks all seeds.
// `slab` holds `blocks` slabs of slab_len keys (slab_len >= the longest row above kLdsKeys candidates).
// switches of the biased sampler (wgamd_set_weighted_sampling_mode; initial values from the environment, read once)
inline int& weighted_mode(int which)
{
  static int mode[2] = {[] { const char* e = getenv("WGAMD_WEIGHTED_PRUNING"); return e && e[0] == '0' ? 0 : 1; }(),
                        [] { const char* e = getenv("WGAMD_WEIGHTED_FORCE_REDO"); return e && e[0] == '1' ? 1 : 0; }()};
  return mode[which];
}
inline bool weighted_pruning_off() { return weighted_mode(0) == 0; }
inline int weighted_force_redo() { return weighted_mode(1); }

// A side stream for the persistent workgroup kernel of the long rows: its last hub row (one workgroup walking 100k+
// candidates) keeps the kernel alive long after the other workgroups have run dry, and in stream order everything behind
// it would wait.  Forked off the caller's stream and joined before the redo pass, the one-wave kernels fill the machine
// under that tail.  One per host thread and device (ranks of a test may be threads sharing a GPU).
struct side_stream {
  int device         = -1;
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  void ensure()
  {
    int dev = 0;
    WG_HIP_CHECK(hipGetDevice(&dev));
    if (stream != nullptr && dev == device) return;
    device = dev;   // (a thread that moves to another device gets fresh objects; the old ones live as long as the process)
    WG_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    WG_HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    WG_HIP_CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  }
};
inline side_stream& weighted_side_stream()
{
  static thread_local side_stream s;
  s.ensure();
  return s;
}

template <typename SeedT, typename ColT, typename WeightT>
void weighted_sample_launch(const int64_t* row_ptr, const ColT* col, const WeightT* weights, const SeedT* seeds, dev_count n,
                            int M, rng_plan rng, const int* offsets, int* lists, int blocks, uint32_t* slab,
                            int64_t slab_len, ColT* dst, int* lid, int64_t* gid, hipStream_t stream)
{
  if (n.host <= 0) return;
  const int cap = n.host;
  if (M <= 0 || M > 256) {
    sample_weighted_kernel<SeedT, ColT, WeightT, 256, 256><<<std::max(blocks, 1), 256, 0, stream>>>(
      row_ptr, col, weights, seeds, n, M, rng, offsets, slab, slab_len, dst, lid, gid, nullptr, 0);
    return;
  }
  const bool pruned = M <= 32 && !weighted_pruning_off();   // threshold pruning: exact keys only for the candidates near the cut
  bool forked = false;
  side_stream* side = nullptr;
  if (blocks > 0 && pruned) {
    side = &weighted_side_stream();
    WG_HIP_CHECK(hipEventRecord(side->fork, stream));
    WG_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
    sample_weighted_block_pruned_kernel<SeedT, ColT, WeightT><<<blocks, 512, 0, side->stream>>>(
      row_ptr, col, weights, seeds, n, M, rng, offsets, dst, lid, gid, lists, cap, weighted_force_redo());
    WG_HIP_CHECK(hipEventRecord(side->join, side->stream));
    forked = true;
  } else if (blocks > 0)
    sample_weighted_kernel<SeedT, ColT, WeightT, 128, 512><<<blocks, 512, 0, stream>>>(
      row_ptr, col, weights, seeds, n, M, rng, offsets, slab, slab_len, dst, lid, gid, lists, cap);
  // short rows, one key per lane: 4 / 2 / 1 rows per wave (grids sized for the capacity; waves past the list end exit)
  if (M < 16)
    sample_weighted_group_kernel<SeedT, ColT, WeightT, 16><<<ceil_div((int64_t)cap * 16, 256), 256, 0, stream>>>(
      row_ptr, col, weights, seeds, M, rng, offsets, dst, lid, gid, lists, cap, 0);
  if (M < 32)
    sample_weighted_group_kernel<SeedT, ColT, WeightT, 32><<<ceil_div((int64_t)cap * 32, 256), 256, 0, stream>>>(
      row_ptr, col, weights, seeds, M, rng, offsets, dst, lid, gid, lists, cap, 1);
  if (M < 64)
    sample_weighted_group_kernel<SeedT, ColT, WeightT, 64><<<ceil_div((int64_t)cap * 64, 256), 256, 0, stream>>>(
      row_ptr, col, weights, seeds, M, rng, offsets, dst, lid, gid, lists, cap, 2);
  // 65 .. 1024 candidates: one wave per row, 2 / 4 / 8 / 16 keys per lane in registers
#define WG_WAVE(KM, CLS)                                                                                               \
  do {                                                                                                                 \
    if (pruned)                                                                                                        \
      sample_weighted_wave_pruned_kernel<SeedT, ColT, WeightT, KM><<<ceil_div(cap, 4), 256, 0, stream>>>(              \
        row_ptr, col, weights, seeds, M, rng, offsets, dst, lid, gid, lists, cap, CLS, weighted_force_redo());          \
    else                                                                                                               \
      sample_weighted_wave_kernel<SeedT, ColT, WeightT, KM><<<ceil_div(cap, 4), 256, 0, stream>>>(                     \
        row_ptr, col, weights, seeds, M, rng, offsets, dst, lid, gid, lists, cap, CLS);                                \
  } while (0)
  if (M < 128) WG_WAVE(2, 3);
  WG_WAVE(4, 4);
  WG_WAVE(8, 5);
  WG_WAVE(16, 6);
#undef WG_WAVE
  // rows copied whole
  copy_short_rows_kernel<SeedT, ColT><<<ceil_div((int64_t)cap * 16, 256), 256, 0, stream>>>(row_ptr, col, seeds, n, M, offsets,
                                                                                          dst, lid, gid);
  // rows the pruned kernels handed back (a third draw of 0, odd weights, too many candidates near the cut): exact
  // workgroup kernel, fed from the redo list; a slab exists whenever a row longer than the LDS key array does
  if (forked) WG_HIP_CHECK(hipStreamWaitEvent(stream, side->join, 0));
  if (pruned)
    sample_weighted_kernel<SeedT, ColT, WeightT, 128, 512><<<blocks > 0 ? std::min(blocks, 128) : 64, 512, 0, stream>>>(
      row_ptr, col, weights, seeds, n, M, rng, offsets, slab, slab_len, dst, lid, gid, lists, cap, 1);
}

// ------------------------------------------------------------------------------------------
struct sample_args {
  wholememory_tensor_t row_ptr, col, weight, seeds, out_offsets;
  int M;
  void *dst_ctx, *lid_ctx, *gid_ctx;
  uint64_t random_seed;
  wholememory_env_func_t* env;
  hipStream_t stream;
};

void validate(const sample_args& a, bool weighted)
{
  WG_REQUIRE_INPUT(a.row_ptr && a.col && a.seeds && a.out_offsets && a.env, "null tensor / env");
  WG_REQUIRE_INPUT(a.dst_ctx != nullptr, "output_dest_memory_context must not be NULL");
  auto rd = a.row_ptr->desc, cd = a.col->desc, sd = a.seeds->desc, od = a.out_offsets->desc;
  WG_REQUIRE_INPUT(rd.dim == 1 && cd.dim == 1 && sd.dim == 1 && od.dim == 1, "all tensors must be 1-D");
  WG_REQUIRE_INPUT(!weighted || (!a.row_ptr->handle && !a.col->handle),
                   "weighted sampling needs CSR tensors that wrap device pointers (a partitioned CSR is served for the "
                   "unweighted op only, as in the reference)");
  WG_EXPECTS(rd.dtype == WHOLEMEMORY_DT_INT64, "csr_row_ptr dtype must be INT64, got %d", (int)rd.dtype);
  WG_EXPECTS(od.dtype == WHOLEMEMORY_DT_INT, "output_sample_offset dtype must be INT, got %d", (int)od.dtype);
  WG_REQUIRE_INPUT(cd.dtype == WHOLEMEMORY_DT_INT || cd.dtype == WHOLEMEMORY_DT_INT64, "csr_col dtype must be INT|INT64");
  WG_REQUIRE_INPUT(sd.dtype == WHOLEMEMORY_DT_INT || sd.dtype == WHOLEMEMORY_DT_INT64, "center_nodes dtype must be INT|INT64");
  WG_REQUIRE_INPUT(od.sizes[0] == sd.sizes[0] + 1, "output_sample_offset must have center_node_count+1 entries");
  WG_REQUIRE_INPUT(sd.sizes[0] < (int64_t)1 << 31, "too many center nodes");
  if (weighted) {
    WG_REQUIRE_INPUT(a.weight && !a.weight->handle && a.weight->desc.dim == 1, "csr_weight must be a 1-D device tensor");
    WG_REQUIRE_INPUT(a.weight->desc.dtype == WHOLEMEMORY_DT_FLOAT || a.weight->desc.dtype == WHOLEMEMORY_DT_DOUBLE,
                     "csr_weight dtype must be FLOAT|DOUBLE");
    WG_REQUIRE_INPUT(a.weight->desc.sizes[0] == cd.sizes[0], "csr_weight and csr_col sizes differ");
  }
}

template <typename SeedT>
__global__ void __launch_bounds__(256)
max_degree_kernel(const int64_t* __restrict__ row_ptr, const SeedT* __restrict__ seeds, int n, int threshold, int* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t nid = (int64_t)seeds[i];
  const int deg     = (int)(row_ptr[nid + 1] - row_ptr[nid]);
  if (deg > threshold) atomicMax(out, deg);
}

template <typename SeedT, typename ColT, typename WeightT>
void run(const sample_args& a, bool weighted)
{
  const int n            = (int)a.seeds->desc.sizes[0];
  const int M            = a.M;
  hipStream_t stream     = a.stream;
  const auto* row_ptr    = static_cast<const int64_t*>(tensor_data(a.row_ptr));
  const auto* col        = static_cast<const ColT*>(tensor_data(a.col));
  const auto* seeds      = static_cast<const SeedT*>(tensor_data(a.seeds));
  int* offsets           = static_cast<int*>(tensor_data(a.out_offsets));
  const WeightT* weights = weighted ? static_cast<const WeightT*>(tensor_data(a.weight)) : nullptr;

  temp_buffer cnt_buf(a.env), scan_tmp(a.env), list_buf(a.env);
  int* cnt  = cnt_buf.device<int>(n + 1, WHOLEMEMORY_DT_INT);
  int* stmp = scan_tmp.device<int>(scan_tmp_ints(n + 1), WHOLEMEMORY_DT_INT);
  // weighted: list[0] = number of long rows, list[1..] = their seed indices, list[n+1] = longest row that needs a slab
  const bool wave_path = weighted && M > 0 && M <= 256;
  int* big_list        = weighted ? list_buf.device<int>(weighted_list_ints(n), WHOLEMEMORY_DT_INT) : nullptr;
  int h_tot[3]         = {0, 0, 0};  // total samples, long rows, longest slab row
  int h_head[kWeightedListHead] = {0};

  if (weighted) {
    WG_HIP_CHECK(hipMemsetAsync(big_list, 0, kWeightedListHead * sizeof(int), stream));
    if (n > 0 && wave_path)
      sample_count_kernel<SeedT><<<ceil_div(n, 256), 256, 0, stream>>>(row_ptr, seeds, dev_count{n, nullptr}, M, cnt, nullptr,
                                                                      big_list, n, kLdsKeys);
    else if (n > 0)
      sample_count_kernel<SeedT><<<ceil_div(n, 256), 256, 0, stream>>>(row_ptr, seeds, dev_count{n, nullptr}, M, cnt, nullptr);
    WG_HIP_CHECK(hipGetLastError());
    WG_HIP_CHECK(hipMemcpyAsync(h_head, big_list, kWeightedListHead * sizeof(int), hipMemcpyDeviceToHost, stream));
  } else {
    sample_count_enqueue(row_ptr, seeds, sizeof(SeedT) == 8, dev_count{n, nullptr}, M, cnt, nullptr, stream);
  }
  exclusive_scan_i32(cnt, offsets, n, stmp, stream);
  WG_HIP_CHECK(hipMemcpyAsync(&h_tot[0], offsets + n, sizeof(int), hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // the one unavoidable sync: output sizes
  const int total = h_tot[0];
  h_tot[1]        = h_head[7] + h_head[8];   // rows for the persistent workgroups
  h_tot[2]        = h_head[10];              // longest row that needs a key slab
  if (weighted && !wave_path) {
    // 256-thread stream layout / sample-all: the workgroup kernel walks all seeds; any row may need the slab
    h_tot[2] = 0;
    if (n > 0) {
      // longest row among the seeds (one more small pass; this path is the rare M > 256 case)
      temp_buffer deg_buf(a.env);
      int* dmax = deg_buf.device<int>(1, WHOLEMEMORY_DT_INT);
      WG_HIP_CHECK(hipMemsetAsync(dmax, 0, sizeof(int), stream));
      max_degree_kernel<SeedT><<<ceil_div(n, 256), 256, 0, stream>>>(row_ptr, seeds, n, kLdsKeys, dmax);
      WG_HIP_CHECK(hipMemcpyAsync(&h_tot[2], dmax, sizeof(int), hipMemcpyDeviceToHost, stream));
      WG_HIP_CHECK(hipStreamSynchronize(stream));
    }
  }

  auto* dst = static_cast<ColT*>(output_alloc(a.env, a.dst_ctx, total, dtype_of<ColT>::value));
  int* lid  = a.lid_ctx ? static_cast<int*>(output_alloc(a.env, a.lid_ctx, total, WHOLEMEMORY_DT_INT)) : nullptr;
  auto* gid = a.gid_ctx ? static_cast<int64_t*>(output_alloc(a.env, a.gid_ctx, total, WHOLEMEMORY_DT_INT64)) : nullptr;
  if (n == 0 || total == 0) return;

  if (weighted) {
    const int rows_for_blocks = wave_path ? h_tot[1] : n;
    const int blocks          = std::min(rows_for_blocks, kWeightedBlocks);
    const int64_t slab_len    = std::max(h_tot[2], 1);
    temp_buffer key_buf(a.env);
    uint32_t* slab = key_buf.device<uint32_t>(h_tot[2] > 0 ? (int64_t)blocks * slab_len : 1, WHOLEMEMORY_DT_INT);
    weighted_sample_launch<SeedT, ColT, WeightT>(row_ptr, col, weights, seeds, dev_count{n, nullptr}, M,
                                                 rng_plan{a.random_seed, nullptr, nullptr, nullptr}, offsets, big_list, blocks,
                                                 slab, slab_len, dst, lid, gid, stream);
    WG_HIP_CHECK(hipGetLastError());
    WG_HIP_CHECK(hipStreamSynchronize(stream));  // scratch is released on return
    return;
  }
  uniform_sample_enqueue(row_ptr, col, sizeof(ColT) == 8, seeds, sizeof(SeedT) == 8, dev_count{n, nullptr}, M,
                         rng_plan{a.random_seed, nullptr, nullptr, nullptr}, offsets, dst, lid, gid, stream);
  WG_HIP_CHECK(hipGetLastError());
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // outputs complete on return (reference contract)
}

// ---- CSR partitioned over the GPUs of a communicator (DISTRIBUTED / CHUNKED handles) ------------------------------
// What the reference does (unweighted_sample_without_replacement_nccl_func.cuh:213-372): fetch row_ptr[v] and
// row_ptr[v + 1] of every centre with one gather, pick POSITIONS locally (same generator, same order as the local
// kernel), fetch the columns at the picked positions with a second gather.  Here the two gathers are the library's own
// (all-to-all for DISTRIBUTED, direct loads for peer-mapped handles), and the picks come from the ordinary sampling
// kernels run on a two-entry-per-centre row_ptr with `col` = NULL — so a partitioned CSR samples exactly what a
// replicated one does.  Collective: every rank of the CSR's communicator makes the call (with its own centres).
__global__ void __launch_bounds__(256) centre_pairs_kernel(const void* __restrict__ seeds, bool seeds64, int n,
                                                           int64_t* __restrict__ ids, int* __restrict__ twice)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = seeds64 ? static_cast<const int64_t*>(seeds)[i] : (int64_t)static_cast<const int32_t*>(seeds)[i];
  ids[2 * i]     = v;
  ids[2 * i + 1] = v + 1;
  twice[i]       = 2 * i;   // "centre" i of the fetched row_ptr pairs
}

struct borrowed_tensor {   // a wholememory_tensor_t over caller-owned device memory, destroyed on scope exit
  wholememory_tensor_t t = nullptr;
  borrowed_tensor(void* p, int64_t n, wholememory_dtype_t dt)
  {
    wholememory_tensor_description_t d;
    wholememory_initialize_tensor_desc(&d);
    d.dim        = 1;
    d.dtype      = dt;
    d.sizes[0]   = n;
    d.strides[0] = 1;
    WG_EXPECTS(wholememory_make_tensor_from_pointer(&t, p, &d) == WHOLEMEMORY_SUCCESS, "tensor over scratch");
  }
  ~borrowed_tensor() { wholememory_destroy_tensor(t); }
  borrowed_tensor(const borrowed_tensor&)            = delete;
  borrowed_tensor& operator=(const borrowed_tensor&) = delete;
};

template <typename ColT>
void run_partitioned(const sample_args& a)
{
  const int n        = (int)a.seeds->desc.sizes[0];
  const int M        = a.M;
  hipStream_t stream = a.stream;
  const bool seeds64 = a.seeds->desc.dtype == WHOLEMEMORY_DT_INT64;
  int* offsets       = static_cast<int*>(tensor_data(a.out_offsets));
  auto fetch = [&](wholememory_tensor_t table, wholememory_tensor_t idx, wholememory_tensor_t out, const char* what) {
    const auto rc = wholememory_gather(table, idx, out, a.env, stream, -1);
    if (rc != WHOLEMEMORY_SUCCESS) throw logic_error(fmt("gather of %s failed (%d)", what, (int)rc));
  };
  temp_arena arena(a.env);
  const size_t o_ids = arena.add(sizeof(int64_t) * 2 * n), o_ptr = arena.add(sizeof(int64_t) * 2 * n),
               o_twice = arena.add(sizeof(int) * n), o_cnt = arena.add(sizeof(int) * (n + 1)),
               o_scan = arena.add(sizeof(int) * scan_tmp_ints(n + 1));
  arena.commit();
  int64_t* ids  = arena.at<int64_t>(o_ids);
  int64_t* ptrs = arena.at<int64_t>(o_ptr);
  int* twice    = arena.at<int>(o_twice);
  int* cnt      = arena.at<int>(o_cnt);
  // (1) row_ptr[v], row_ptr[v + 1] of every centre
  if (n > 0) {
    centre_pairs_kernel<<<ceil_div(n, 256), 256, 0, stream>>>(tensor_data(a.seeds), seeds64, n, ids, twice);
    WG_HIP_CHECK(hipGetLastError());
  }
  {
    borrowed_tensor t_ids(ids, 2 * (int64_t)n, WHOLEMEMORY_DT_INT64), t_ptrs(ptrs, 2 * (int64_t)n, WHOLEMEMORY_DT_INT64);
    fetch(a.row_ptr, t_ids.t, t_ptrs.t, "csr_row_ptr");
  }
  // (2) sample counts, offsets, total (the one host sync the output sizes need)
  sample_count_enqueue(ptrs, twice, false, dev_count{n, nullptr}, M, cnt, nullptr, stream);
  exclusive_scan_i32(cnt, offsets, n, arena.at<int>(o_scan), stream);
  int total = 0;
  WG_HIP_CHECK(hipMemcpyAsync(&total, offsets + n, sizeof(int), hipMemcpyDeviceToHost, stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));
  auto* dst = static_cast<ColT*>(output_alloc(a.env, a.dst_ctx, total, dtype_of<ColT>::value));
  int* lid  = a.lid_ctx ? static_cast<int*>(output_alloc(a.env, a.lid_ctx, total, WHOLEMEMORY_DT_INT)) : nullptr;
  temp_buffer gid_buf(a.env);
  auto* gid = a.gid_ctx ? static_cast<int64_t*>(output_alloc(a.env, a.gid_ctx, total, WHOLEMEMORY_DT_INT64))
                        : gid_buf.device<int64_t>(total, WHOLEMEMORY_DT_INT64);
  // (3) the picks, as CSR positions
  if (n > 0 && total > 0)
    uniform_sample_enqueue(ptrs, nullptr, sizeof(ColT) == 8, twice, false, dev_count{n, nullptr}, M,
                           rng_plan{a.random_seed, nullptr, nullptr, nullptr}, offsets, nullptr, lid, gid, stream);
  // (4) the columns at the picked positions (every rank takes part, also with nothing to fetch)
  {
    borrowed_tensor t_gid(gid, total, WHOLEMEMORY_DT_INT64), t_dst(dst, total, dtype_of<ColT>::value);
    fetch(a.col, t_gid.t, t_dst.t, "csr_col");
  }
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // outputs complete on return (reference contract); scratch is released
}

template <typename WeightT>
void dispatch(const sample_args& a, bool weighted)
{
  const bool s64 = a.seeds->desc.dtype == WHOLEMEMORY_DT_INT64;
  const bool c64 = a.col->desc.dtype == WHOLEMEMORY_DT_INT64;
  if (!weighted && (a.row_ptr->handle || a.col->handle)) return c64 ? run_partitioned<int64_t>(a) : run_partitioned<int32_t>(a);
  if (s64 && c64) return run<int64_t, int64_t, WeightT>(a, weighted);
  if (s64 && !c64) return run<int64_t, int32_t, WeightT>(a, weighted);
  if (!s64 && c64) return run<int32_t, int64_t, WeightT>(a, weighted);
  return run<int32_t, int32_t, WeightT>(a, weighted);
}

}  // namespace

void sample_count_enqueue(const int64_t* row_ptr, const void* seeds, bool seeds64, dev_count n, int M, int* cnt,
                          int* big_deg, hipStream_t stream, int64_t* row_start, int* row_deg)
{
  if (n.host <= 0) return;
  if (seeds64)
    sample_count_kernel<int64_t><<<std::min(ceil_div(n.host, 256), kWalkGrid), 256, 0, stream>>>(
      row_ptr, static_cast<const int64_t*>(seeds), n, M, cnt, big_deg, nullptr, 0, 0, row_start, row_deg);
  else
    sample_count_kernel<int32_t><<<std::min(ceil_div(n.host, 256), kWalkGrid), 256, 0, stream>>>(
      row_ptr, static_cast<const int32_t*>(seeds), n, M, cnt, big_deg, nullptr, 0, 0, row_start, row_deg);
  WG_HIP_CHECK(hipGetLastError());
}

void weighted_count_enqueue(const int64_t* row_ptr, const void* seeds, bool seeds64, dev_count n, int M, int* cnt,
                            int* big_list, hipStream_t stream)
{
  if (n.host <= 0) return;
  WG_HIP_CHECK(hipMemsetAsync(big_list, 0, kWeightedListHead * sizeof(int), stream));
  if (seeds64)
    sample_count_kernel<int64_t><<<ceil_div(n.host, 256), 256, 0, stream>>>(row_ptr, static_cast<const int64_t*>(seeds), n, M,
                                                                           cnt, nullptr, big_list, n.host, kLdsKeys);
  else
    sample_count_kernel<int32_t><<<ceil_div(n.host, 256), 256, 0, stream>>>(row_ptr, static_cast<const int32_t*>(seeds), n, M,
                                                                           cnt, nullptr, big_list, n.host, kLdsKeys);
  WG_HIP_CHECK(hipGetLastError());
}

void weighted_sample_enqueue(const int64_t* row_ptr, const void* col, bool col64, const void* weights, bool weights64,
                             const void* seeds, bool seeds64, dev_count n, int M, rng_plan random_seed, const int* offsets,
                             int* big_list, uint32_t* slab, int64_t slab_len, void* dst, int* src_lid,
                             int64_t* edge_gid, hipStream_t stream)
{
  WG_REQUIRE_INPUT(M > 0 && M <= 256, "the no-sync biased hop needs 0 < fan-out <= 256");
#define WG_W(ST, CT, WT)                                                                                                  \
  weighted_sample_launch<ST, CT, WT>(row_ptr, static_cast<const CT*>(col), static_cast<const WT*>(weights),              \
                                     static_cast<const ST*>(seeds), n, M, random_seed, offsets, big_list, kWeightedBlocks, \
                                     slab, slab_len, static_cast<CT*>(dst), src_lid, edge_gid, stream)
#define WG_WW(ST, CT) do { if (weights64) WG_W(ST, CT, double); else WG_W(ST, CT, float); } while (0)
  if (seeds64 && col64) WG_WW(int64_t, int64_t);
  else if (seeds64) WG_WW(int64_t, int32_t);
  else if (col64) WG_WW(int32_t, int64_t);
  else WG_WW(int32_t, int32_t);
#undef WG_WW
#undef WG_W
  WG_HIP_CHECK(hipGetLastError());
}

int64_t sample_locality_hist_ints() { return (int64_t)kLocBlocks * kLocBuckets; }
int sample_locality_shift(int64_t id_bound)
{
  int bits = 0;
  while (bits < 62 && ((int64_t)1 << bits) < id_bound) bits++;
  int buckets_log2 = 0;
  while ((1 << buckets_log2) < kLocBuckets) buckets_log2++;
  return bits > buckets_log2 ? bits - buckets_log2 : 0;
}

void uniform_sample_enqueue(const int64_t* row_ptr, const void* col, bool col64, const void* seeds, bool seeds64,
                            dev_count n, int M, rng_plan random_seed, const int* offsets, void* dst, int* src_lid,
                            int64_t* edge_gid, hipStream_t stream, const int64_t* row_start,
                            const int* row_deg, const sample_locality* loc)
{
#define WG_U(ST, CT)                                                                                               \
  uniform_launch<ST, CT>(row_ptr, static_cast<const CT*>(col), static_cast<const ST*>(seeds), n, M, random_seed, \
                         offsets, static_cast<CT*>(dst), src_lid, edge_gid, stream, row_start, row_deg, loc)
  if (seeds64 && col64) WG_U(int64_t, int64_t);
  else if (seeds64) WG_U(int64_t, int32_t);
  else if (col64) WG_U(int32_t, int64_t);
  else WG_U(int32_t, int32_t);
#undef WG_U
}

}  // namespace wgamd

extern "C" {

void wgamd_set_weighted_sampling_mode(int pruning, int force_redo)
{
  if (pruning >= 0) wgamd::weighted_mode(0) = pruning ? 1 : 0;
  if (force_redo >= 0) wgamd::weighted_mode(1) = force_redo ? 1 : 0;
}

wholememory_error_code_t wholegraph_csr_unweighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor, int max_sample_count,
  wholememory_tensor_t output_sample_offset_tensor, void* output_dest_memory_context,
  void* output_center_localid_memory_context, void* output_edge_gid_memory_context,
  unsigned long long random_seed, wholememory_env_func_t* p_env_fns, void* stream)
{
  return wgamd::guarded("wholegraph_csr_unweighted_sample_without_replacement", [&] {
    wgamd::sample_args a{wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, nullptr, center_nodes_tensor,
                         output_sample_offset_tensor, max_sample_count, output_dest_memory_context,
                         output_center_localid_memory_context, output_edge_gid_memory_context,
                         (uint64_t)random_seed, p_env_fns, static_cast<hipStream_t>(stream)};
    wgamd::validate(a, false);
    wgamd::dispatch<float>(a, false);
  });
}

wholememory_error_code_t wholegraph_csr_weighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t wm_csr_weight_ptr_tensor, wholememory_tensor_t center_nodes_tensor,
  int max_sample_count, wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context, void* output_center_localid_memory_context,
  void* output_edge_gid_memory_context, unsigned long long random_seed,
  wholememory_env_func_t* p_env_fns, void* stream)
{
  return wgamd::guarded("wholegraph_csr_weighted_sample_without_replacement", [&] {
    wgamd::sample_args a{wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, wm_csr_weight_ptr_tensor,
                         center_nodes_tensor, output_sample_offset_tensor, max_sample_count,
                         output_dest_memory_context, output_center_localid_memory_context,
                         output_edge_gid_memory_context, (uint64_t)random_seed, p_env_fns,
                         static_cast<hipStream_t>(stream)};
    wgamd::validate(a, true);
    if (a.weight->desc.dtype == WHOLEMEMORY_DT_DOUBLE)
      wgamd::dispatch<double>(a, true);
    else
      wgamd::dispatch<float>(a, true);
  });
}

wholememory_error_code_t generate_random_positive_int_cpu(int64_t random_seed, int64_t subsequence,
                                                          wholememory_tensor_t output)
{
  if (output == nullptr || output->desc.dim != 1) {
    fprintf(stderr, "[wholegraph_amd] generate_random_positive_int_cpu: output should be 1D tensor.\n");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (output->desc.dtype != WHOLEMEMORY_DT_INT && output->desc.dtype != WHOLEMEMORY_DT_INT64) {
    fprintf(stderr, "[wholegraph_amd] generate_random_positive_int_cpu: output should be int64 or int32 tensor.\n");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  wgamd::Pcg32 g((uint64_t)random_seed, (uint64_t)subsequence);
  void* p = wgamd::tensor_data(output);
  for (int64_t k = 0; k < output->desc.sizes[0]; k++) {
    if (output->desc.dtype == WHOLEMEMORY_DT_INT)
      static_cast<int32_t*>(p)[k] = g.next_i31();
    else
      static_cast<int64_t*>(p)[k] = g.next_i63();
  }
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t generate_exponential_distribution_negative_float_cpu(int64_t random_seed,
                                                                              int64_t subsequence,
                                                                              wholememory_tensor_t output)
{
  if (output == nullptr || output->desc.dim != 1) {
    fprintf(stderr, "[wholegraph_amd] generate_exponential_distribution_negative_float_cpu: output should be 1D tensor.\n");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (output->desc.dtype != WHOLEMEMORY_DT_FLOAT) {
    fprintf(stderr, "[wholegraph_amd] generate_exponential_distribution_negative_float_cpu: output should be float.\n");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  wgamd::Pcg32 g((uint64_t)random_seed, (uint64_t)subsequence);
  float* p = static_cast<float*>(wgamd::tensor_data(output));
  for (int64_t k = 0; k < output->desc.sizes[0]; k++) {
    float u = g.next_f32();
    u       = (float)(-(0.5 + 0.5 * (double)u));
    uint64_t x;
    int zero_draws = -1;
    do {
      x = g.next_u64();
      zero_draws++;
    } while (!x);
    int lz = 0;
    for (uint64_t probe = x; !(probe >> 63); probe <<= 1) lz++;
    u    = (float)((double)u * std::pow(2.0, -(lz + zero_draws * 64)));
    p[k] = (float)(std::log1p((double)u) / std::log(2.0));
  }
  return WHOLEMEMORY_SUCCESS;
}

}  // extern "C"
