import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_common.hpp')
s = open(p).read()
old = "constexpr int kWeightedBlocks  = 1024;"
new = '''constexpr int kWeightedBlocks  = 1024;
// row lists of a biased hop, built by the count kernel (wave-aggregated appends): rows that are sampled (deg > M) by size
// class — 0: deg <= 16, 1: <= 32, 2: <= 64 (one key per lane: 4 / 2 / 1 rows per wave), 3: 1025 .. 16384 and 4: > 16384
// candidates (persistent workgroups, the huge rows first, dealt out through a queue head).  Rows of 65 .. 1024 candidates
// and rows copied whole need no list: the one-wave kernel walks all seeds.  Layout in ints:
//   [0 .. 4] list lengths | [5] queue head | [6] longest row that needs a key slab | [7] pad | list c at 8 + c * cap
constexpr int kWeightedLists = 5;
inline int64_t weighted_list_ints(int64_t cap) { return 8 + (int64_t)kWeightedLists * (cap > 0 ? cap : 1); }'''
assert old in s
s = s.replace(old, new, 1)
open(p, 'w').write(s)

p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_fused.hip')
s = open(p).read()
old = "    w.big_list = take(sizeof(int) * (size_t)(target_cap + 2));"
new = "    w.big_list = take(sizeof(int) * (size_t)weighted_list_ints(target_cap));"
assert old in s
s = s.replace(old, new)
open(p, 'w').write(s)

p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sample.hip')
s = open(p).read()

# ---- count kernel: size-class lists
old = s[s.index("template <typename SeedT>\n__global__ void __launch_bounds__(256) sample_count_kernel("):s.index("template <typename ColT>\n__device__ __forceinline__ void emit(")]
new = '''constexpr int kHugeRow = 16384;   // candidates; rows above go first (list 4), one workgroup each like the long ones

template <typename SeedT>
__global__ void __launch_bounds__(256) sample_count_kernel(const int64_t* __restrict__ row_ptr,
                                                           const SeedT* __restrict__ seeds,
                                                           dev_count n_,
                                                           int M,
                                                           int* __restrict__ cnt,
                                                           int* __restrict__ big_deg /*nullable*/,
                                                           int* __restrict__ lists = nullptr /*biased hop: see wg_common.hpp*/,
                                                           int list_cap            = 0,
                                                           int scratch_threshold   = 0)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_live = n_.get();
  int cls          = -1;
  int deg          = 0;
  if (i < n_.host) {
    if (i >= n_live) {  // capacity slack of the no-sync walk: zero the rest of the live scan tile only
      if (i < (n_live / kScanTile + 1) * kScanTile) {
        cnt[i] = 0;
        if (big_deg) big_deg[i] = 0;
      }
    } else {
      int64_t nid = (int64_t)seeds[i];
      deg         = (int)(row_ptr[nid + 1] - row_ptr[nid]);
      cnt[i]      = (M > 0 && deg > M) ? M : deg;
      if (big_deg) big_deg[i] = 0;
      if (lists != nullptr && M > 0 && deg > M) {
        cls = deg <= 16 ? 0 : deg <= 32 ? 1 : deg <= 64 ? 2 : deg <= kWaveRowCapDecl ? -1 : deg <= kHugeRow ? 3 : 4;
        if (cls >= 3 && deg > scratch_threshold) atomicMax(lists + 6, deg);  // longest row that needs a key slab
      }
    }
  }
  if (lists == nullptr) return;
  // wave-aggregated append: one atomic per wave and list
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < kWeightedLists; c++) {
    const uint64_t m = __ballot(cls == c);
    if (m == 0ull) continue;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(lists + c, __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    if (cls == c) lists[8 + (int64_t)c * list_cap + base + __popcll(m & ((1ull << lane) - 1ull))] = i;
  }
}

'''
s = s.replace(old, new)
s = s.replace("// ------------------------------------------------------------------------------------------\ntemplate <typename SeedT>\n__global__ void __launch_bounds__(256) sample_count_kernel(",
              "// ------------------------------------------------------------------------------------------\nconstexpr int kWaveRowCapDecl = 1024;  // == kWaveRowCap (rows the one-wave weighted kernel keeps in registers)\ntemplate <typename SeedT>\n__global__ void __launch_bounds__(256) sample_count_kernel(", 1)
if "kWaveRowCapDecl = 1024" not in s:
    s = s.replace("constexpr int kHugeRow = 16384;", "constexpr int kWaveRowCapDecl = 1024;  // == kWaveRowCap (rows the one-wave weighted kernel keeps in registers)\nconstexpr int kHugeRow = 16384;", 1)

# ---- ares_key reports redraws
old = '''__device__ __forceinline__ float ares_key(float w, Pcg32& g)
{
  float u = g.next_f32();
  u       = (float)(-(0.5 + 0.5 * (double)u));
  uint64_t x;
  int zero_draws = -1;
  do {
    x = g.next_u64();
    zero_draws++;
  } while (!x);'''
new = '''__device__ __forceinline__ float ares_key(float w, Pcg32& g, bool* redrawn = nullptr)
{
  float u = g.next_f32();
  u       = (float)(-(0.5 + 0.5 * (double)u));
  uint64_t x;
  int zero_draws = -1;
  do {
    x = g.next_u64();
    zero_draws++;
  } while (!x);
  if (redrawn != nullptr && zero_draws > 0) *redrawn = true;   // more than three draws for this key'''
assert old in s
s = s.replace(old, new)
s = s.replace("constexpr int kWaveRowCap = 1024;  // rows the one-wave weighted kernel keeps in registers (16 keys per lane)",
              "constexpr int kWaveRowCap = 1024;  // rows the one-wave weighted kernel keeps in registers (16 keys per lane)\nstatic_assert(kWaveRowCap == kWaveRowCapDecl, \"keep the two in step\");")

# ---- long-row kernel: queue + two lists, cheaper three-draw check
old = '''  const int n_live = n_.get();
  const int count  = seed_list ? seed_list[0] : n_live;
  uint32_t* gkeys  = slab + (int64_t)blockIdx.x * slab_len;
  for (int li = blockIdx.x; li < count; li += gridDim.x) {
  __syncthreads();  // the previous row's readers of the shared counters are done
  const int i = seed_list ? seed_list[1 + li] : li;
  if (i >= n_live) continue;'''
new = '''  // `lists` (biased hop with 0 < M <= 256): the huge rows (list 4) then the long ones (list 3), dealt out through a queue
  // head so that a workgroup stuck on a 150k-candidate hub does not hold back the rows a static deal would have given it
  __shared__ int sh_li;
  const int n_live  = n_.get();
  const int n_huge  = lists ? lists[4] : 0;
  const int count   = lists ? n_huge + lists[3] : n_live;
  uint32_t* gkeys   = slab + (int64_t)blockIdx.x * slab_len;
  int li            = blockIdx.x;
  while (true) {
  __syncthreads();  // the previous row's readers of the shared counters are done
  if (lists) {
    if (threadIdx.x == 0) sh_li = atomicAdd(lists + 5, 1);
    __syncthreads();
    li = sh_li;
  }
  if (li >= count) break;
  const int i = lists ? (li < n_huge ? lists[8 + 4 * (int64_t)list_cap + li] : lists[8 + 3 * (int64_t)list_cap + (li - n_huge)]) : li;
  if (!lists) li += gridDim.x;
  if (i >= n_live) continue;'''
assert old in s
s = s.replace(old, new)
s = s.replace('''                                                            int64_t* __restrict__ edge_gid,
                                                            const int* __restrict__ seed_list /*nullable: [0] = count*/)
{
  static_assert(T % B == 0 && T % 64 == 0''', '''                                                            int64_t* __restrict__ edge_gid,
                                                            int* __restrict__ lists /*nullable*/,
                                                            int list_cap)
{
  static_assert(T % B == 0 && T % 64 == 0''')
old = '''    for (int id = threadIdx.x; id < N; id += T) {
      const uint64_t before = g.state;
      put(id, key_bits(ares_key((float)weight[start + id], g)));
      if (hop > 1) {
        // exactly three draws? (state after 3 steps is a fixed affine map of the state before)
        Pcg32 chk = g;
        chk.state = before;
        chk.jump_table(3u);
        redrawn |= chk.state != g.state;
        g.jump_table(3u * (uint32_t)(hop - 1));
      }
    }'''
new = '''    for (int id = threadIdx.x; id < N; id += T) {
      put(id, key_bits(ares_key((float)weight[start + id], g, &redrawn)));   // exactly three draws unless it says so
      if (hop > 1) g.jump_table(3u * (uint32_t)(hop - 1));
    }'''
assert old in s
s = s.replace(old, new)

# ---- group kernel (one key per lane) before the wave kernel
anchor = "// One WAVE per seed for rows of up to 64*KMAX candidates"
group = '''// Rows of at most LANES (16 / 32 / 64) candidates: ONE key per lane, 64 / LANES rows per wave (the rows of a size class,
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
    i = lists[8 + (int64_t)cls * list_cap + li];
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
  if (take) emit<ColT>(dst, src_lid, edge_gid, base + __popcll(mt & below), col[start + hl], i, start + hl);
}

'''
assert anchor in s
s = s.replace(anchor, group + anchor, 1)

# ---- wave kernel: skip the rows the group kernels take
old = "  if (N > 64 * KMAX) return;  // the workgroup kernel takes these (seed list built by the count kernel)"
new = "  if (N > 64 * KMAX || (skip_short && N <= 64)) return;  // the workgroup kernel / the group kernels take these (listed by the count kernel)"
assert old in s
s = s.replace(old, new)
s = s.replace('''                                                                   int* __restrict__ src_lid,
                                                                   int64_t* __restrict__ edge_gid)
{
  const int lane = threadIdx.x & 63;
  const int i    = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_.get()) return;''', '''                                                                   int* __restrict__ src_lid,
                                                                   int64_t* __restrict__ edge_gid,
                                                                   bool skip_short)
{
  const int lane = threadIdx.x & 63;
  const int i    = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_.get()) return;''')

# ---- launch
old = s[s.index("template <typename SeedT, typename ColT, typename WeightT>\nvoid weighted_sample_launch("):s.index("// ------------------------------------------------------------------------------------------\nstruct sample_args {")]
new = '''template <typename SeedT, typename ColT, typename WeightT>
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
  if (blocks > 0)
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
  // everything else the wave holds in registers (65 .. 1024 candidates) + the rows copied whole: all seeds
  sample_weighted_wave_kernel<SeedT, ColT, WeightT, kWaveRowCap / 64><<<ceil_div(cap, 4), 256, 0, stream>>>(
    row_ptr, col, weights, seeds, n, M, rng, offsets, dst, lid, gid, true);
}

'''
s = s.replace(old, new)

# ---- run(): lists
old = '''  const bool wave_path = weighted && M > 0 && M <= 256;
  int* big_list        = weighted ? list_buf.device<int>(n + 2, WHOLEMEMORY_DT_INT) : nullptr;
  int h_tot[3]         = {0, 0, 0};  // total samples, long rows, longest slab row

  if (weighted) {
    WG_HIP_CHECK(hipMemsetAsync(big_list, 0, sizeof(int), stream));
    WG_HIP_CHECK(hipMemsetAsync(big_list + n + 1, 0, sizeof(int), stream));
    if (n > 0)
      sample_count_kernel<SeedT><<<ceil_div(n, 256), 256, 0, stream>>>(
        row_ptr, seeds, dev_count{n, nullptr}, M, cnt, nullptr, wave_path ? kWaveRowCap : 0, wave_path ? big_list : nullptr,
        big_list + n + 1, kLdsKeys);
    WG_HIP_CHECK(hipGetLastError());
    WG_HIP_CHECK(hipMemcpyAsync(&h_tot[1], big_list, sizeof(int), hipMemcpyDeviceToHost, stream));
    WG_HIP_CHECK(hipMemcpyAsync(&h_tot[2], big_list + n + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
  } else {'''
new = '''  const bool wave_path = weighted && M > 0 && M <= 256;
  int* big_list        = weighted ? list_buf.device<int>(weighted_list_ints(n), WHOLEMEMORY_DT_INT) : nullptr;
  int h_tot[3]         = {0, 0, 0};  // total samples, long rows, longest slab row
  int h_head[8]        = {0, 0, 0, 0, 0, 0, 0, 0};

  if (weighted) {
    WG_HIP_CHECK(hipMemsetAsync(big_list, 0, 8 * sizeof(int), stream));
    if (n > 0 && wave_path)
      sample_count_kernel<SeedT><<<ceil_div(n, 256), 256, 0, stream>>>(row_ptr, seeds, dev_count{n, nullptr}, M, cnt, nullptr,
                                                                      big_list, n, kLdsKeys);
    else if (n > 0)
      sample_count_kernel<SeedT><<<ceil_div(n, 256), 256, 0, stream>>>(row_ptr, seeds, dev_count{n, nullptr}, M, cnt, nullptr);
    WG_HIP_CHECK(hipGetLastError());
    WG_HIP_CHECK(hipMemcpyAsync(h_head, big_list, 8 * sizeof(int), hipMemcpyDeviceToHost, stream));
  } else {'''
assert old in s
s = s.replace(old, new)
old = '''  const int total = h_tot[0];
'''
new = '''  const int total = h_tot[0];
  h_tot[1]        = h_head[3] + h_head[4];   // rows for the persistent workgroups
  h_tot[2]        = h_head[6];               // longest row that needs a key slab
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
'''
assert old in s
s = s.replace(old, new, 1)

# max_degree_kernel helper before run()
anchor = "template <typename SeedT, typename ColT, typename WeightT>\nvoid run(const sample_args& a, bool weighted)"
helper = '''template <typename SeedT>
__global__ void __launch_bounds__(256)
max_degree_kernel(const int64_t* __restrict__ row_ptr, const SeedT* __restrict__ seeds, int n, int threshold, int* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t nid = (int64_t)seeds[i];
  const int deg     = (int)(row_ptr[nid + 1] - row_ptr[nid]);
  if (deg > threshold) atomicMax(out, deg);
}

'''
assert anchor in s
s = s.replace(anchor, helper + anchor, 1)

# no-sync enqueue
old = s[s.index("void weighted_count_enqueue("):s.index("void weighted_sample_enqueue(")]
new = '''void weighted_count_enqueue(const int64_t* row_ptr, const void* seeds, bool seeds64, dev_count n, int M, int* cnt,
                            int* big_list, hipStream_t stream)
{
  if (n.host <= 0) return;
  WG_HIP_CHECK(hipMemsetAsync(big_list, 0, 8 * sizeof(int), stream));
  if (seeds64)
    sample_count_kernel<int64_t><<<ceil_div(n.host, 256), 256, 0, stream>>>(row_ptr, static_cast<const int64_t*>(seeds), n, M,
                                                                           cnt, nullptr, big_list, n.host, kLdsKeys);
  else
    sample_count_kernel<int32_t><<<ceil_div(n.host, 256), 256, 0, stream>>>(row_ptr, static_cast<const int32_t*>(seeds), n, M,
                                                                           cnt, nullptr, big_list, n.host, kLdsKeys);
  WG_HIP_CHECK(hipGetLastError());
}

'''
s = s.replace(old, new)
s = s.replace('''                             const void* seeds, bool seeds64, dev_count n, int M, rng_plan random_seed, const int* offsets,
                             const int* big_list, uint32_t* slab, int64_t slab_len, void* dst, int* src_lid,''', '''                             const void* seeds, bool seeds64, dev_count n, int M, rng_plan random_seed, const int* offsets,
                             int* big_list, uint32_t* slab, int64_t slab_len, void* dst, int* src_lid,''')
open(p, 'w').write(s)

p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_common.hpp')
s = open(p).read()
s = s.replace('''                             const void* seeds, bool seeds64, dev_count n, int M, rng_plan random_seed, const int* offsets,
                             const int* big_list,''', '''                             const void* seeds, bool seeds64, dev_count n, int M, rng_plan random_seed, const int* offsets,
                             int* big_list,''')
open(p, 'w').write(s)
print("ok")
