import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_common.hpp')
s = open(p).read()
old = s[s.index("// row lists of a biased hop, built by the count kernel"):s.index("void weighted_count_enqueue(")]
new = '''// row lists of a biased hop, built by the count kernel (block-aggregated appends): the rows that are sampled (deg > M)
// by size class — 0: deg <= 16, 1: <= 32, 2: <= 64 (one key per lane: 4 / 2 / 1 rows per wave), 3: <= 128, 4: <= 256,
// 5: <= 512, 6: <= 1024 (one wave per row, 2 / 4 / 8 / 16 keys per lane in registers), 7: <= 16384 and 8: more candidates
// (persistent workgroups, the huge rows first, dealt out through a queue head).  Rows copied whole (deg <= M) need no
// list.  Layout in ints:  [0 .. 8] list lengths | [9] queue head | [10] longest row that needs a key slab | pad to 16 |
// list c at 16 + c * cap
constexpr int kWeightedLists    = 9;
constexpr int kWeightedListHead = 16;
inline int64_t weighted_list_ints(int64_t cap) { return kWeightedListHead + (int64_t)kWeightedLists * (cap > 0 ? cap : 1); }
'''
s = s.replace(old, new)
open(p, 'w').write(s)

p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sample.hip')
s = open(p).read()
# count kernel classes + layout
s = s.replace("        cls = deg <= 16 ? 0 : deg <= 32 ? 1 : deg <= 64 ? 2 : deg <= kWaveRowCapDecl ? -1 : deg <= kHugeRow ? 3 : 4;\n        if (cls >= 3 && deg > scratch_threshold) atomicMax(lists + 6, deg);  // longest row that needs a key slab",
              "        cls = deg <= 16 ? 0 : deg <= 32 ? 1 : deg <= 64 ? 2 : deg <= 128 ? 3 : deg <= 256 ? 4 : deg <= 512 ? 5\n              : deg <= kWaveRowCapDecl ? 6 : deg <= kHugeRow ? 7 : 8;\n        if (cls >= 7 && deg > scratch_threshold) atomicMax(lists + 10, deg);  // longest row that needs a key slab")
s = s.replace("  if (cls >= 0) lists[8 + (int64_t)cls * list_cap + blk_base[cls] + my_rank] = i;",
              "  if (cls >= 0) lists[kWeightedListHead + (int64_t)cls * list_cap + blk_base[cls] + my_rank] = i;")
# long-row kernel
s = s.replace("  const int n_huge  = lists ? lists[4] : 0;\n  const int count   = lists ? n_huge + lists[3] : n_live;",
              "  const int n_huge  = lists ? lists[8] : 0;\n  const int count   = lists ? n_huge + lists[7] : n_live;")
s = s.replace("    if (threadIdx.x == 0) sh_li = atomicAdd(lists + 5, 1);", "    if (threadIdx.x == 0) sh_li = atomicAdd(lists + 9, 1);")
s = s.replace("  const int i = lists ? (li < n_huge ? lists[8 + 4 * (int64_t)list_cap + li] : lists[8 + 3 * (int64_t)list_cap + (li - n_huge)]) : li;",
              "  const int i = lists ? (li < n_huge ? lists[kWeightedListHead + 8 * (int64_t)list_cap + li]\n                                    : lists[kWeightedListHead + 7 * (int64_t)list_cap + (li - n_huge)])\n                      : li;")
# group kernel
s = s.replace("    i = lists[8 + (int64_t)cls * list_cap + li];", "    i = lists[kWeightedListHead + (int64_t)cls * list_cap + li];")

# wave kernel: list driven, fixed KMAX, no slot guards
i0 = s.index("// One WAVE per seed for rows of up to 64*KMAX candidates")
i1 = s.index("template <typename SeedT, typename ColT>\nvoid uniform_launch(")
wave = '''// One WAVE per row for rows of up to 64*KMAX candidates (size classes 3 .. 6 of the count kernel's lists: KMAX = 2, 4, 8,
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
  const int N         = (int)(row_ptr[nid + 1] - start);   // M < N <= 64 * KMAX by construction of the list
  const int64_t base  = offsets[i];
  uint32_t k[KMAX];
  {
    Pcg32 ga = stream_generator(random_seed, i_rng, 128, lane);
    Pcg32 gb = stream_generator(random_seed, i_rng, 128, lane + 64);
#pragma unroll
    for (int s = 0; s < KMAX; s++) {
      const int id = (s >> 1) * 128 + (s & 1) * 64 + lane;
      k[s]         = 0u;  // below every real key (key_bits of any float, -inf and NaN included, is > 0)
      if (id < N) k[s] = key_bits(ares_key((float)weight[start + id], (s & 1) ? gb : ga));
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
    if (take) emit<ColT>(dst, src_lid, edge_gid, base + out_run + __popcll(mt & below), col[start + id], i, start + id);
    out_run += __popcll(mt);
    tie_run += __popcll(meq);
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
  for (int j = hl; j < N; j += 16) emit<ColT>(dst, src_lid, edge_gid, base + j, col[start + j], i, start + j);
}

'''
s = s[:i0] + wave + s[i1:]

# launch
old = s[s.index("  // short rows, one key per lane: 4 / 2 / 1 rows per wave"):s.index("// ------------------------------------------------------------------------------------------\nstruct sample_args {")]
new = '''  // short rows, one key per lane: 4 / 2 / 1 rows per wave (grids sized for the capacity; waves past the list end exit)
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
#define WG_WAVE(KM, CLS)                                                                                               \\
  sample_weighted_wave_kernel<SeedT, ColT, WeightT, KM><<<ceil_div(cap, 4), 256, 0, stream>>>(                           \\
    row_ptr, col, weights, seeds, M, rng, offsets, dst, lid, gid, lists, cap, CLS)
  if (M < 128) WG_WAVE(2, 3);
  WG_WAVE(4, 4);
  WG_WAVE(8, 5);
  WG_WAVE(16, 6);
#undef WG_WAVE
  // rows copied whole
  copy_short_rows_kernel<SeedT, ColT><<<ceil_div((int64_t)cap * 16, 256), 256, 0, stream>>>(row_ptr, col, seeds, n, M, offsets,
                                                                                          dst, lid, gid);
}

'''
s = s.replace(old, new)

# run(): header size and indices
s = s.replace("  int h_head[8]        = {0, 0, 0, 0, 0, 0, 0, 0};", "  int h_head[kWeightedListHead] = {0};")
s = s.replace("    WG_HIP_CHECK(hipMemsetAsync(big_list, 0, 8 * sizeof(int), stream));\n    if (n > 0 && wave_path)",
              "    WG_HIP_CHECK(hipMemsetAsync(big_list, 0, kWeightedListHead * sizeof(int), stream));\n    if (n > 0 && wave_path)")
s = s.replace("    WG_HIP_CHECK(hipMemcpyAsync(h_head, big_list, 8 * sizeof(int), hipMemcpyDeviceToHost, stream));",
              "    WG_HIP_CHECK(hipMemcpyAsync(h_head, big_list, kWeightedListHead * sizeof(int), hipMemcpyDeviceToHost, stream));")
s = s.replace("  h_tot[1]        = h_head[3] + h_head[4];   // rows for the persistent workgroups\n  h_tot[2]        = h_head[6];               // longest row that needs a key slab",
              "  h_tot[1]        = h_head[7] + h_head[8];   // rows for the persistent workgroups\n  h_tot[2]        = h_head[10];              // longest row that needs a key slab")
s = s.replace("  WG_HIP_CHECK(hipMemsetAsync(big_list, 0, 8 * sizeof(int), stream));\n  if (seeds64)",
              "  WG_HIP_CHECK(hipMemsetAsync(big_list, 0, kWeightedListHead * sizeof(int), stream));\n  if (seeds64)")
open(p, 'w').write(s)
print("ok", "8 * sizeof(int)" in s)
