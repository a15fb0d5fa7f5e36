// One kernel for a whole SAGEConv layer over a sampled hop:
//     out[i, :] = act( [ mean_{j in N(i)} x[j] | x[self(i)] ] @ [W_l | W_r]^T + b )
// = feature fetch (optional id indirection into the global table) -> neighbour aggregation -> dense transform.
// The aggregated operand never goes to HBM.  A TEAM of N/64 waves builds the [64 x 2F] operand tile of 64 destination
// rows in LDS (gather phase: CSR bounds, neighbour ids and the neighbour rows of a lane group's 8 rows are fetched with
// branch-free, software-pipelined 16-B loads; fp32 sums in CSR order, bit-identical to wgamd_sage_aggregate_f32) and then
// multiplies it by the weight with fp32 MFMA (v_mfma_f32_16x16x4_f32, exact f32): every wave owns 64 output columns =
// 4 x 4 accumulator tiles, A fragments from LDS, B fragments (the weight, L2-resident) straight from global memory one
// k-step ahead.  A workgroup holds TWO teams with one tile each that run one phase apart — while team 0 gathers (HBM
// bound), team 1 multiplies (MFMA bound), then they swap at a workgroup barrier — so both machines of the CU stay busy
// by construction instead of by luck of the workgroup scheduler.  When two tiles do not fit the 160 KB of LDS (F > 152) a
// single-team workgroup is used.  Replaces "SpMM kernel -> [agg | x_self] matrix in HBM -> hipBLASLt GEMM"
// (DESIGN.md §3.5); semantics of torch_geometric.nn.SAGEConv as the reference uses it
// (python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTileRows = 64;

template <typename IdT>
__device__ __forceinline__ int64_t table_row(const IdT* ids, int64_t local)
{
  if constexpr (std::is_same<IdT, void>::value) return local;
  else return (int64_t)ids[local];
}

struct layer_args {
  const int* row_ptr;
  const int* col;
  int64_t n_rows;
  const float* x;
  int64_t ldx;
  int64_t x_rows;
  int F;
  const void* src_ids;
  const int64_t* self_rows;
  int mean;
  const float* w_t;
  int64_t ldw;
  const float* bias;
  int relu;
  float* out;
  int64_t ldo;
};

// ---- gather phase: the team's waves fill a_tile[64][S] = [ mean/sum of neighbour rows | self row ] -------------------
// Only ~2 waves share a SIMD, so latency is hidden inside the wave, not by occupancy: (1) CSR bounds of all IT rows of
// this lane group, (2) their first LG neighbour ids (+ table indirection) and self ids, (3) the neighbour rows — kNb
// (+ self) 16-B loads per lane of row it+1 are in flight while row it is summed.  Every load is UNCONDITIONAL (slots
// past a row's degree / idle lanes read row 0 or the last 16 B of the row, which stay in L1, and are masked by a select):
// a load under a branch costs the branch and makes the compiler fall back to s_waitcnt vmcnt(0), which would put the
// whole phase in series.
// OFF32: every byte offset into x fits 32 bits (x smaller than 4 GB) -> one shuffle per neighbour instead of two and
// `uniform base + 32-bit lane offset` addressing, i.e. no 64-bit select/add per neighbour row
template <typename IdT, int LG, int WAVES, bool OFF32>
__device__ __forceinline__ void gather_tile(const layer_args& a, int64_t tile, float* a_tile, int S, int tw, int lane)
{
  using off_t = typename std::conditional<OFF32, uint32_t, int64_t>::type;
  constexpr int kGroupsPerWave = 64 / LG;
  constexpr int kGroups        = kGroupsPerWave * WAVES;
  constexpr int IT             = kTileRows / kGroups;  // rows of the tile per lane group
  constexpr int kNb            = LG < 10 ? LG : 10;    // neighbour rows prefetched per destination row (fan-out 10)
  static_assert(kTileRows % kGroups == 0, "lane groups must tile the 64 rows evenly");
  const IdT* src_ids = static_cast<const IdT*>(a.src_ids);
  const float* x     = a.x;
  const int64_t ldx = a.ldx, n_rows = a.n_rows;
  const int F       = a.F;
  const int sub = lane & (LG - 1), gbase = lane & ~(LG - 1);
  const int group   = tw * kGroupsPerWave + lane / LG;
  const int f0      = sub * 4;
  const bool live   = f0 < F;
  const int f0c     = live ? f0 : F - 4;
  const int64_t row0 = tile * kTileRows;

  int s_[IT], d_[IT], lcol_[IT];
  int64_t lself_[IT];
  off_t src_[IT], self_[IT];   // BYTE offsets of the neighbour / self rows (incl. the lane's column offset for self)
  bool has_self_[IT];
  const char* xb = reinterpret_cast<const char*>(x);
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int64_t row  = row0 + group + it * kGroups;
    const int64_t rowc = row < n_rows ? row : n_rows - 1;
    s_[it]             = a.row_ptr[rowc];
    d_[it]             = a.row_ptr[rowc + 1];
  }
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int64_t row = row0 + group + it * kGroups;
    d_[it]            = row < n_rows ? d_[it] - s_[it] : 0;
  }
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int64_t row = row0 + group + it * kGroups;
    const int* pc     = (sub < d_[it]) ? a.col + s_[it] + sub : a.row_ptr;  // row_ptr[0] == 0: a valid local row
    lcol_[it]         = *pc;
    lself_[it]        = a.self_rows[row < n_rows ? row : n_rows - 1];
  }
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int64_t row = row0 + group + it * kGroups;
    src_[it]          = (off_t)(table_row<IdT>(src_ids, (int64_t)lcol_[it]) * ldx * 4);
    self_[it]         = (off_t)(table_row<IdT>(src_ids, lself_[it]) * ldx * 4);
    has_self_[it]     = row < n_rows;
  }
  // ring of kDepth row buffers: the fetches of rows it+1 .. it+kDepth-1 are in flight while row it is summed (the
  // accumulator tiles of the transform phase are not live here, so the gather may spend ~180 VGPRs on this)
  constexpr int kDepth = IT < 3 ? IT : 3;
  f32x4 buf[kDepth][kNb + 1];
  auto issue = [&](int it, f32x4* v) {
#pragma unroll
    for (int k = 0; k < kNb; k++) {
      const int src_lane = gbase | (k & (LG - 1));
      off_t off;
      if constexpr (OFF32) {
        off = (uint32_t)__shfl((int)src_[it], src_lane, 64);
      } else {
        const int lo = __shfl((int)(src_[it] & 0xffffffff), src_lane, 64);
        const int hi = __shfl((int)(src_[it] >> 32), src_lane, 64);
        off          = ((int64_t)hi << 32) | (uint32_t)lo;
      }
      off  = k < d_[it] ? off : (off_t)0;
      v[k] = *reinterpret_cast<const f32x4*>(xb + off + (off_t)(f0c * 4));
    }
    v[kNb] = *reinterpret_cast<const f32x4*>(xb + (has_self_[it] ? self_[it] : (off_t)0) + (off_t)(f0c * 4));
  };
#pragma unroll
  for (int it = 0; it < kDepth - 1; it++) issue(it, buf[it]);
#pragma unroll
  for (int it = 0; it < IT; it++) {
    if (it + kDepth - 1 < IT) issue(it + kDepth - 1, buf[(it + kDepth - 1) % kDepth]);
    const f32x4* v = buf[it % kDepth];
    const int deg  = d_[it];
    f32x4 acc      = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kNb; k++) acc += k < deg ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};  // select, never multiply by 0
    if (live) {
      // rows longer than the prefetched window are finished by the second pass below (their partial sum waits in LDS)
      if (a.mean && deg > 0 && deg <= kNb) acc /= (float)deg;
      const int r = group + it * kGroups;
      *reinterpret_cast<f32x4*>(a_tile + r * S + f0)     = acc;
      *reinterpret_cast<f32x4*>(a_tile + r * S + F + f0) = has_self_[it] ? v[kNb] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // second pass, kept OUT of the pipelined loop (a branch with loads in it would make every join a vmcnt(0) wait): rows
  // longer than the prefetched window continue in CSR order, chunk by chunk (wave-uniform branches)
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int deg = d_[it];
    if (__ballot(deg > kNb) == 0ull) continue;
    const int r = group + it * kGroups;
    f32x4 acc   = {0.f, 0.f, 0.f, 0.f};
    if (live && deg > kNb) acc = *reinterpret_cast<const f32x4*>(a_tile + r * S + f0);
    int maxdeg = deg;
#pragma unroll
    for (int d = 32; d >= LG; d >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, d, 64));
    for (int c0 = 0; c0 < maxdeg; c0 += LG) {
      const int64_t my_src = (c0 + sub < deg) ? table_row<IdT>(src_ids, (int64_t)a.col[s_[it] + c0 + sub]) : 0;
      const int chunk      = min(LG, maxdeg - c0);
      for (int j = (c0 == 0 ? kNb : 0); j < chunk; j++) {
        const int src_lane = gbase | (j & (LG - 1));
        const int lo       = __shfl((int)(my_src & 0xffffffff), src_lane, 64);
        const int hi       = __shfl((int)(my_src >> 32), src_lane, 64);
        const int64_t rr   = ((int64_t)hi << 32) | (uint32_t)lo;
        if (live && deg > kNb && c0 + j < deg) acc += *reinterpret_cast<const f32x4*>(x + rr * ldx + f0);
      }
    }
    if (live && deg > kNb) {
      if (a.mean) acc /= (float)deg;
      *reinterpret_cast<f32x4*>(a_tile + r * S + f0) = acc;
    }
  }
}

// ---- transform phase: wave tw of the team computes the 64 x 64 block  A[64 x K] @ W^T[K x 64*tw ..]  + epilogue ------
// K is walked in chunks of 16: k-step j (0..3) of a chunk multiplies chunk rows { 4*kq + j : kq = 0..3 } (any order of the
// K sum is the same product), so that lane (kq, lm) needs A[row][16c + 4kq .. +3] = ONE 16-B LDS read per row tile and
// chunk, and B[16c + 4kq + j][16ct + lm] = 16 dwords per chunk that it loads itself straight from the L2-resident
// weight — no LDS traffic for B at all.  Operands of chunk c+1 are requested before the 64 MFMAs of chunk c are issued
// (two register sets, chunk loop unrolled by two), so neither the L2 nor the LDS latency is ever waited for.
struct chunk_regs {
  f32x4 a[4];     // [row tile] -> A values for k-steps j = 0..3
  float b[4][4];  // [k-step j][col tile]
};

__device__ __forceinline__ void load_chunk(chunk_regs& r, const float* arow, int S, const float* wcol, int64_t ldw, int K,
                                           int k0, int kq)
{
#pragma unroll
  for (int rt = 0; rt < 4; rt++) r.a[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 16 * S + k0);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k  = k0 + 4 * kq + j;
    const int kc = k < K ? k : K - 1;  // rows past K: the A tile holds zeros there; keep the address valid
#pragma unroll
    for (int ct = 0; ct < 4; ct++) r.b[j][ct] = wcol[(int64_t)kc * ldw + ct * 16];
  }
}

__device__ __forceinline__ void mma_chunk(f32x4 (&c)[4][4], const chunk_regs& r)
{
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++)
        c[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[rt][j], r.b[j][ct], c[rt][ct], 0, 0, 0);
}

__device__ __forceinline__ void transform_tile(const layer_args& a, int64_t tile, const float* a_tile, int S, int tw, int lane)
{
  const int K = 2 * a.F;
  f32x4 c[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; rt++)
#pragma unroll
    for (int ct = 0; ct < 4; ct++) c[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kq = lane >> 4, lm = lane & 15;
  const float* arow = a_tile + lm * S + 4 * kq;  // + rt * 16 * S + 16 c
  const float* wcol = a.w_t + tw * 64 + lm;      // + k * ldw + ct * 16
  const int chunks  = (K + 15) / 16;
  float bj[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) bj[ct] = a.bias ? a.bias[tw * 64 + ct * 16 + lm] : 0.f;
  chunk_regs r0, r1;
  load_chunk(r0, arow, S, wcol, a.ldw, K, 0, kq);
  for (int ch = 0; ch < chunks; ch += 2) {
    if (ch + 1 < chunks) load_chunk(r1, arow, S, wcol, a.ldw, K, (ch + 1) * 16, kq);
    mma_chunk(c, r0);
    if (ch + 1 < chunks) {
      if (ch + 2 < chunks) load_chunk(r0, arow, S, wcol, a.ldw, K, (ch + 2) * 16, kq);
      mma_chunk(c, r1);
    }
  }
  // epilogue: bias, activation, store (C/D map: col = lane & 15, row = 4 * (lane >> 4) + reg).  Whole tiles take a
  // branch-free path: a store under a per-lane branch makes the compiler wait vmcnt(0) — for the PREVIOUS store — 64 times
  const int64_t row0 = tile * kTileRows;
#pragma unroll
  for (int ct = 0; ct < 4; ct++)
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        float v = c[rt][ct][rg] + bj[ct];
        c[rt][ct][rg] = a.relu ? fmaxf(v, 0.f) : v;
      }
  float* obase = a.out + (row0 + kq * 4) * a.ldo + tw * 64 + lm;
  if (row0 + kTileRows <= a.n_rows) {
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) obase[(int64_t)(rt * 16 + rg) * a.ldo + ct * 16] = c[rt][ct][rg];
  } else {
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
          if (row0 + rt * 16 + kq * 4 + rg < a.n_rows) obase[(int64_t)(rt * 16 + rg) * a.ldo + ct * 16] = c[rt][ct][rg];
  }
}

// floats per A-tile row: K rounded up to the 16-wide chunk (the pad columns stay zero) + 4 (16-B aligned rows whose
// fragment reads spread over the LDS banks)
__host__ __device__ inline int tile_stride(int F) { return (2 * F + 15) / 16 * 16 + 4; }

// LG = lanes per destination row in the gather phase (power of two >= F/4), WAVES = N / 64 waves per team,
// TEAMS = 2: ping-pong (team t handles the workgroup's tiles t, t+2, ... one phase behind team t-1), TEAMS = 1: plain
template <typename IdT, int LG, int WAVES, int TEAMS, bool OFF32>
__global__ void __launch_bounds__(WAVES * 64 * TEAMS, 2)
sage_layer_fused_kernel(layer_args a)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [TEAMS][kTileRows][S]
  const int S    = tile_stride(a.F);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int team = wave / WAVES, tw = wave % WAVES;
  float* a_tile  = lds + (size_t)team * kTileRows * S;
  // the pad columns K .. S-1 of every tile row are read by the last chunk and never written by the gather: zero once
  for (int i = threadIdx.x; i < TEAMS * kTileRows * (S - 2 * a.F); i += blockDim.x) {
    const int row = i / (S - 2 * a.F), cpad = i % (S - 2 * a.F);
    lds[(size_t)row * S + 2 * a.F + cpad] = 0.f;
  }
  __syncthreads();
  const int64_t n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  // tiles of this workgroup: blockIdx.x + n * gridDim.x, n = 0 .. mine-1; team t owns n = t, t + TEAMS, ...
  const int64_t mine = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  // step s: team t is (s - t) phases into its own sequence gather(n0) transform(n0) gather(n1) ...
  const int64_t steps = 2 * ((mine + TEAMS - 1) / TEAMS) + (TEAMS - 1);
  for (int64_t step = 0; step < steps; step++) {
    const int64_t my = step - team;
    if (my >= 0) {
      const int64_t n    = (my >> 1) * TEAMS + team;
      const int64_t tile = blockIdx.x + n * gridDim.x;
      if (n < mine) {
        if ((my & 1) == 0) gather_tile<IdT, LG, WAVES, OFF32>(a, tile, a_tile, S, tw, lane);
        else transform_tile(a, tile, a_tile, S, tw, lane);
      }
    }
    __syncthreads();
  }
}

template <typename IdT, int LG, int WAVES>
void launch(const layer_args& a, hipStream_t st)
{
  const size_t tile_bytes = sizeof(float) * (size_t)kTileRows * (size_t)tile_stride(a.F);
  const int cus         = stream_cu_count(st);
  const int64_t n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  const size_t b_bytes  = 0;
  // WGAMD_SAGE_NO_PINGPONG=1: single-team workgroups even when two tiles fit (tuning aid, F > 128 only)
  static const bool no_pingpong = getenv("WGAMD_SAGE_NO_PINGPONG") != nullptr;   // read once per process, not per launch
  const bool pingpong   = 2 * tile_bytes + b_bytes <= 160 * 1024 && (LG < 64 || !no_pingpong);
  const size_t lds      = (pingpong ? 2 * tile_bytes : tile_bytes) + b_bytes;
  const int per_cu      = (!pingpong && 2 * lds <= 160 * 1024) ? 2 : 1;
  const int grid        = (int)std::min<int64_t>((n_tiles + (pingpong ? 1 : 0)) / (pingpong ? 2 : 1), (int64_t)cus * per_cu);
  auto go = [&](auto kern, int threads) {
    if (lds > 64 * 1024)
      WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<grid < 1 ? 1 : grid, threads, lds, st>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  };
  const bool off32 = a.x_rows > 0 && (uint64_t)a.x_rows * (uint64_t)a.ldx * 4u < (1ull << 32);
  if (pingpong && off32) go(sage_layer_fused_kernel<IdT, LG, WAVES, 2, true>, WAVES * 128);
  else if (pingpong) go(sage_layer_fused_kernel<IdT, LG, WAVES, 2, false>, WAVES * 128);
  else if constexpr (LG == 64) {  // only F > 128 can need the single-team kernel (two tiles beyond 160 KB of LDS)
    if (off32) go(sage_layer_fused_kernel<IdT, LG, WAVES, 1, true>, WAVES * 64);
    else go(sage_layer_fused_kernel<IdT, LG, WAVES, 1, false>, WAVES * 64);
  } else {
    throw logic_error("two operand tiles must fit LDS for F <= 128");  // unreachable: 2 x 64 x 260 x 4 B = 133 KB
  }
}

template <typename IdT, int LG>
void launch_waves(int N, const layer_args& a, hipStream_t st)
{
  switch (N / 64) {
    case 1: launch<IdT, LG, 1>(a, st); break;
    case 2: launch<IdT, LG, 2>(a, st); break;
    default: launch<IdT, LG, 4>(a, st); break;
  }
}

template <typename IdT>
void launch_groups(int N, const layer_args& a, hipStream_t st)
{
  const int units = a.F / 4;
  if (units <= 8) launch_waves<IdT, 8>(N, a, st);
  else if (units <= 16) launch_waves<IdT, 16>(N, a, st);
  else if (units <= 32) launch_waves<IdT, 32>(N, a, st);
  else launch_waves<IdT, 64>(N, a, st);
}

}  // namespace
}  // namespace wgamd

extern "C" wholememory_error_code_t wgamd_sage_layer_fused_f32(const int* row_ptr, const int* col, int64_t n_rows,
                                                               const float* x, int64_t ldx, int64_t x_rows, int F,
                                                               const void* src_ids,
                                                               wholememory_dtype_t src_ids_dtype, const int64_t* self_rows,
                                                               int mean, const float* w_t, int64_t ldw, int N,
                                                               const float* bias, int relu, float* out, int64_t ldo,
                                                               void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sage_layer_fused_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && F > 0 && N > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && self_rows && w_t && out, "null pointer");
    // shapes the kernel is built for; callers fall back to wgamd_sage_aggregate_f32 + a library GEMM otherwise
    if (F % 4 != 0 || F > 256 || (N != 64 && N != 128 && N != 256) || ldx % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(x) & 15) != 0)
      throw logic_error(fmt("unsupported shape: F=%d (multiple of 4, <= 256), N=%d (64, 128 or 256), 16-B aligned rows", F, N));
    WG_REQUIRE_INPUT(ldw >= N && ldo >= N, "leading dimensions smaller than N");
    layer_args a{row_ptr, col, n_rows, x, ldx, x_rows, F, src_ids, self_rows, mean, w_t, ldw, bias, relu, out, ldo};
    auto st = static_cast<hipStream_t>(stream);
    if (src_ids == nullptr) launch_groups<void>(N, a, st);
    else if (src_ids_dtype == WHOLEMEMORY_DT_INT) launch_groups<int32_t>(N, a, st);
    else if (src_ids_dtype == WHOLEMEMORY_DT_INT64) launch_groups<int64_t>(N, a, st);
    else throw invalid_input("src_ids must be INT or INT64");
  });
}
