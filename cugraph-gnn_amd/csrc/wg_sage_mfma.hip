// One kernel for a whole SAGEConv layer over a sampled hop, HBM-bound by construction:
//     out[i, :] = act( [ mean_{j in N(i)} x[j] | x[self(i)] ] @ [W_l | W_r]^T + b )
// (semantics of torch_geometric.nn.SAGEConv as the reference uses it,
//  python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59).
//
// Why a second one-kernel layer next to wg_sage_fused.hip: that kernel multiplies in exact fp32 on the matrix pipe
// (v_mfma_f32_16x16x4_f32 = the fp32 VECTOR rate, 157 TF/s), so the layer can never run faster than its ~0.36 ms of
// fp32 MFMA issue — and the gather team's VALU work queues behind those MFMAs.  Here the fp32 product is evaluated on the
// bf16 pipe (16x the rate) with a 3-way split of BOTH operands:  a = a_hi + a_mid + a_lo  exactly (an fp32 significand
// is 24 bits = 3 x 8, each piece is a bf16 by truncation, the residuals are exact in fp32), and
//     a * b ~= a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_mid b_mid + a_hi b_lo + a_lo b_hi)
// — the six products of weight >= 2^-16, accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three dropped products
// are <= 2^-23 |a b| in total: the same class as fp32 round-off (checked against the fp64 oracle at 1e-5 x scale like the
// fp32 kernel, tests/test_gpu_aggregate.py).  Six bf16 MFMAs cost 6/16 of one fp32 MFMA, so the matrix work is ~0.14 ms
// of issue for the products layer-1 shape and the kernel is bound by its 2.85 GB of HBM traffic.
//
// Structure (one 512-thread workgroup per CU, persistent over 64-row tiles):
//   * waves CW..CW+3 are PRODUCERS: they fetch CSR bounds / neighbour ids / neighbour rows with branch-free 16-B loads kept
//     in a register ring, sum in CSR order (bit-identical to wgamd_sage_aggregate_f32), split the fp32 sums and the self
//     row into the three bf16 planes and store them to the LDS tile of step s.  Row metadata is software-pipelined ACROSS
//     tiles (bounds two tiles ahead, neighbour ids one tile ahead, the first rows of the next tile are requested before
//     the barrier), so the CU's memory queue never drains at a tile boundary.
//   * waves 0..CW-1 are CONSUMERS (one per SIMD): each owns 64 output columns = 2 x 2 accumulator tiles of 32 x 32 and
//     multiplies the tile of step s-1: A fragments by ds_read_b128 from the planes, B fragments (the pre-split weight,
//     L2-resident, laid out so that a wave-load is 1 KiB contiguous) one k-step ahead straight from global memory.
//   * one s_barrier per step behind an LDS-only wait (s_waitcnt lgkmcnt(0)): neither side's global loads are drained.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

using f32x4  = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4  = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2  = __attribute__((ext_vector_type(2))) uint32_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kProducerWaves = 4;

template <typename IdT>
__device__ __forceinline__ int64_t table_row(const IdT* ids, int64_t local)
{
  if constexpr (std::is_same<IdT, void>::value) return local;
  else return (int64_t)ids[local];
}

struct mfma_args {
  const int* row_ptr;
  const int* col;
  int64_t n_rows;
  const float* x;
  int64_t ldx;
  int F;
  const void* src_ids;
  const int64_t* self_rows;
  int mean;
  const uint32_t* w_planes;  // [3][KS][N][8 dwords]: bf16 plane p, k-step s, column n, 16 consecutive k
  int N;
  int KS;                    // ceil(2F / 16)
  const float* bias;
  int relu;
  float* out;
  int64_t ldo;
  int SD;                    // dwords per LDS tile row and plane (>= F, = 4 * odd: conflict-free ds_read_b128)
};

// a == hi + mid + lo exactly; every piece has <= 8 significant bits, i.e. is a bf16 (the top half of the fp32 word)
__device__ __forceinline__ void split3(float a, uint32_t& h, uint32_t& m, uint32_t& l)
{
  h              = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(h);
  m              = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);
  l              = __float_as_uint(r2);
}
// (lo word's bf16, hi word's bf16) -> one dword: bytes {a.2, a.3, b.2, b.3}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ void store_split(uint32_t* plane0, int plane_stride_dw, int dw, f32x4 v)
{
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; i++) split3(v[i], h[i], m[i], l[i]);
  *reinterpret_cast<u32x2*>(plane0 + dw)                       = u32x2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
  *reinterpret_cast<u32x2*>(plane0 + plane_stride_dw + dw)     = u32x2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
  *reinterpret_cast<u32x2*>(plane0 + 2 * plane_stride_dw + dw) = u32x2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
}

// LDS-only wait + workgroup barrier: in-flight global loads (prefetched rows / weight fragments) and stores stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------------
// producer side
// ---------------------------------------------------------------------------------------------------------------------
template <int IT>
struct bounds_t {
  int s[IT], e[IT];
};
template <int IT>
struct ids_t {
  int lcol[IT];
  int64_t lself[IT];
};
template <int IT, typename off_t>
struct meta_t {
  int d[IT];       // degree; -1 = row past n_rows (no neighbours, zero self row)
  off_t src[IT];   // byte offset of THIS lane's neighbour row (lane `sub` holds neighbour `sub` of the row)
  off_t self[IT];  // byte offset of the self row
};

template <typename IdT, int LG, int TR, bool OFF32>
struct producer {
  using off_t                         = typename std::conditional<OFF32, uint32_t, int64_t>::type;
  static constexpr int kGroupsPerWave = 64 / LG;
  static constexpr int kGroups        = kGroupsPerWave * kProducerWaves;
  static constexpr int IT             = TR / kGroups;       // rows of a tile per lane group
  static constexpr int kNb            = LG < 10 ? LG : 10;  // neighbour rows prefetched per destination row (fan-out 10)
  static constexpr int kDepth         = IT < 2 ? 1 : 2;     // rows in flight per lane group
  static_assert(TR % kGroups == 0, "lane groups must tile the rows evenly");

  const mfma_args& a;
  const int sub, gbase, group, f0, f0c;
  const bool live;

  __device__ producer(const mfma_args& a_, int pw, int lane)
    : a(a_),
      sub(lane & (LG - 1)),
      gbase(lane & ~(LG - 1)),
      group(pw * kGroupsPerWave + lane / LG),
      f0((lane & (LG - 1)) * 4),
      f0c(((lane & (LG - 1)) * 4 < a_.F) ? (lane & (LG - 1)) * 4 : a_.F - 4),
      live((lane & (LG - 1)) * 4 < a_.F)
  {
  }

  __device__ __forceinline__ int64_t row_of(int64_t tile, int it) const { return tile * TR + group + it * kGroups; }

  // stage A: CSR bounds (two tiles ahead of the rows being summed)
  __device__ __forceinline__ void load_bounds(int64_t tile, bounds_t<IT>& b) const
  {
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int64_t row  = row_of(tile, it);
      const int64_t rowc = row < a.n_rows ? row : a.n_rows - 1;
      b.s[it]            = a.row_ptr[rowc];
      b.e[it]            = a.row_ptr[rowc + 1];
    }
  }
  // stage B: this lane's neighbour id of every row + the self row ids (one tile ahead); unconditional loads
  __device__ __forceinline__ void load_ids(int64_t tile, const bounds_t<IT>& b, ids_t<IT>& v) const
  {
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int64_t row = row_of(tile, it);
      const int* pc     = (sub < b.e[it] - b.s[it]) ? a.col + b.s[it] + sub : a.row_ptr;  // row_ptr[0] == 0: a valid row
      v.lcol[it]        = *pc;
      v.lself[it]       = a.self_rows[row < a.n_rows ? row : a.n_rows - 1];
    }
  }
  // stage C: byte offsets (with the id indirection of the fused-fetch variant: one more dependent load)
  __device__ __forceinline__ void finish(int64_t tile, const bounds_t<IT>& b, const ids_t<IT>& v, meta_t<IT, off_t>& m) const
  {
    const IdT* src_ids = static_cast<const IdT*>(a.src_ids);
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int64_t row = row_of(tile, it);
      m.d[it]           = row < a.n_rows ? b.e[it] - b.s[it] : -1;
      m.src[it]         = (off_t)(table_row<IdT>(src_ids, (int64_t)v.lcol[it]) * a.ldx * 4);
      m.self[it]        = (off_t)(table_row<IdT>(src_ids, v.lself[it]) * a.ldx * 4);
    }
  }
  // request the kNb neighbour rows + the self row of row `it` (every load unconditional: slots past the degree read row 0)
  __device__ __forceinline__ void issue(const meta_t<IT, off_t>& m, int it, f32x4* v) const
  {
    const char* xb = reinterpret_cast<const char*>(a.x);
#pragma unroll
    for (int k = 0; k < kNb; k++) {
      const int src_lane = gbase | (k & (LG - 1));
      off_t off;
      if constexpr (OFF32) {
        off = (uint32_t)__shfl((int)m.src[it], src_lane, 64);
      } else {
        const int lo = __shfl((int)(m.src[it] & 0xffffffff), src_lane, 64);
        const int hi = __shfl((int)(m.src[it] >> 32), src_lane, 64);
        off          = ((int64_t)hi << 32) | (uint32_t)lo;
      }
      off  = k < m.d[it] ? off : (off_t)0;
      v[k] = *reinterpret_cast<const f32x4*>(xb + off + (off_t)(f0c * 4));
    }
    v[kNb] = *reinterpret_cast<const f32x4*>(xb + (m.d[it] >= 0 ? m.self[it] : (off_t)0) + (off_t)(f0c * 4));
  }
  // sum row `it` from its ring slot and store the split planes
  __device__ __forceinline__ void reduce_store(const meta_t<IT, off_t>& m, int it, const f32x4* v, uint32_t* tile_lds) const
  {
    const int deg = m.d[it];
    f32x4 acc     = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kNb; k++) acc += k < deg ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};  // select, never multiply by 0
    if (a.mean && deg > 0) acc /= (float)deg;   // rows longer than the window are redone by long_rows()
    if (live) {
      const int r     = group + it * kGroups;
      uint32_t* prow  = tile_lds + r * a.SD;
      const int pl_dw = TR * a.SD;
      store_split(prow, pl_dw, f0 >> 1, acc);
      store_split(prow, pl_dw, (a.F + f0) >> 1, deg >= 0 ? v[kNb] : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
  // rows longer than the prefetched window (rare: deg > 10): the whole sum again, in CSR order, chunk by chunk
  __device__ __forceinline__ void long_rows(int64_t tile, const meta_t<IT, off_t>& m, uint32_t* tile_lds) const
  {
    const IdT* src_ids = static_cast<const IdT*>(a.src_ids);
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int deg = m.d[it];
      if (__ballot(deg > kNb) == 0ull) continue;
      const int64_t row  = row_of(tile, it);
      const int64_t rowc = row < a.n_rows ? row : a.n_rows - 1;
      const int s        = a.row_ptr[rowc];
      f32x4 acc          = {0.f, 0.f, 0.f, 0.f};
      int maxdeg         = deg > kNb ? deg : 0;
#pragma unroll
      for (int dd = 32; dd >= LG; dd >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, dd, 64));
      for (int c0 = 0; c0 < maxdeg; c0 += LG) {
        const int64_t my_src = (deg > kNb && c0 + sub < deg) ? table_row<IdT>(src_ids, (int64_t)a.col[s + c0 + sub]) : 0;
        const int chunk      = min(LG, maxdeg - c0);
        for (int j = 0; j < chunk; j++) {
          const int src_lane = gbase | (j & (LG - 1));
          const int lo       = __shfl((int)(my_src & 0xffffffff), src_lane, 64);
          const int hi       = __shfl((int)(my_src >> 32), src_lane, 64);
          const int64_t rr   = ((int64_t)hi << 32) | (uint32_t)lo;
          if (live && deg > kNb && c0 + j < deg) acc += *reinterpret_cast<const f32x4*>(a.x + rr * a.ldx + f0);
        }
      }
      if (live && deg > kNb) {
        if (a.mean) acc /= (float)deg;
        store_split(tile_lds + (group + it * kGroups) * a.SD, TR * a.SD, f0 >> 1, acc);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// consumer side: wave cw multiplies the [TR x 2F] tile by columns [64 cw, 64 cw + 64) of the weight
// ---------------------------------------------------------------------------------------------------------------------
template <int RT>
struct frag_t {
  u32x4 a[RT][3];  // [row tile][plane]
  u32x4 b[2][3];   // [col tile][plane]
};

template <int RT>
__device__ __forceinline__ void load_frags(frag_t<RT>& f, const uint32_t* a_lane, int plane_dw, int sd, const uint32_t* b_lane,
                                           int64_t b_plane_dw, int n_cols, int ks)
{
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int p = 0; p < 3; p++) f.a[rt][p] = *reinterpret_cast<const u32x4*>(a_lane + p * plane_dw + rt * 32 * sd + ks * 8);
#pragma unroll
  for (int ct = 0; ct < 2; ct++)
#pragma unroll
    for (int p = 0; p < 3; p++)
      f.b[ct][p] = *reinterpret_cast<const u32x4*>(b_lane + p * b_plane_dw + ((int64_t)ks * n_cols + ct * 32) * 8);
}

template <int RT>
__device__ __forceinline__ void mma_frags(f32x16 (&c)[RT][2], const frag_t<RT>& f)
{
  // smallest terms first; per accumulator tile the six products are independent MFMAs on the same accumulator
  constexpr int pa[6] = {2, 0, 1, 1, 0, 0};
  constexpr int pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < 6; t++)
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
        c[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[rt][pa[t]]),
                                                            __builtin_bit_cast(bf16x8, f.b[ct][pb[t]]), c[rt][ct], 0, 0, 0);
}

template <int TR>
__device__ __forceinline__ void consume_tile(const mfma_args& a, int64_t tile, const uint32_t* tile_lds, int cw, int lane)
{
  constexpr int RT = TR / 32;
  f32x16 c[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int i = 0; i < 16; i++) c[rt][ct][i] = 0.f;
  const int lm = lane & 31, lh = lane >> 5;
  const int plane_dw       = TR * a.SD;
  const uint32_t* a_lane   = tile_lds + lm * a.SD + lh * 4;
  const int64_t b_plane_dw = (int64_t)a.KS * a.N * 8;
  const uint32_t* b_lane   = a.w_planes + ((int64_t)(cw * 64 + lm)) * 8 + lh * 4;
  frag_t<RT> f0, f1;
  load_frags<RT>(f0, a_lane, plane_dw, a.SD, b_lane, b_plane_dw, a.N, 0);
  for (int ks = 0; ks < a.KS; ks += 2) {
    if (ks + 1 < a.KS) load_frags<RT>(f1, a_lane, plane_dw, a.SD, b_lane, b_plane_dw, a.N, ks + 1);
    mma_frags<RT>(c, f0);
    if (ks + 1 < a.KS) {
      if (ks + 2 < a.KS) load_frags<RT>(f0, a_lane, plane_dw, a.SD, b_lane, b_plane_dw, a.N, ks + 2);
      mma_frags<RT>(c, f1);
    }
  }
  // epilogue: bias, activation, store.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int64_t row0 = tile * TR;
  float bj[2];
#pragma unroll
  for (int ct = 0; ct < 2; ct++) bj[ct] = a.bias ? a.bias[cw * 64 + ct * 32 + lm] : 0.f;
  float* obase = a.out + (row0 + lh * 4) * a.ldo + cw * 64 + lm;
  if (row0 + TR <= a.n_rows) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int i = 0; i < 16; i++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const float v = c[rt][ct][i] + bj[ct];
          obase[(int64_t)(rt * 32 + (i & 3) + 8 * (i >> 2)) * a.ldo + ct * 32] = a.relu ? fmaxf(v, 0.f) : v;
        }
  } else {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int i = 0; i < 16; i++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const int r   = rt * 32 + (i & 3) + 8 * (i >> 2);
          const float v = c[rt][ct][i] + bj[ct];
          if (row0 + lh * 4 + r < a.n_rows) obase[(int64_t)r * a.ldo + ct * 32] = a.relu ? fmaxf(v, 0.f) : v;
        }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// CW = N / 64 consumer waves + 4 producer waves; TR = rows per tile (64, or 32 when two 64-row tiles exceed the LDS)
// ---------------------------------------------------------------------------------------------------------------------
template <typename IdT, int LG, int TR, int CW, bool OFF32>
__global__ void __launch_bounds__((CW + kProducerWaves) * 64)
sage_layer_mfma_kernel(mfma_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // [2 tiles][3 planes][TR][SD] + 16 dwords of slack
  const int tile_dw = 3 * TR * a.SD;
  // every word the MFMA can touch must be a finite bf16 pair: pad columns, the rows of a tile that is still being
  // written for the first time, and the few dwords the last k-step reads past a row (times the zero rows of the weight)
  for (int i = threadIdx.x; i < 2 * tile_dw + 16; i += blockDim.x) lds[i] = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_tiles = (a.n_rows + TR - 1) / TR;
  const int64_t mine    = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  auto tile_of          = [&](int64_t n) { return (int64_t)blockIdx.x + n * gridDim.x; };

  if (wave >= CW) {
    using P     = producer<IdT, LG, TR, OFF32>;
    using off_t = typename P::off_t;
    constexpr int IT = P::IT, kNb = P::kNb, kDepth = P::kDepth;
    P p(a, wave - CW, lane);
    bounds_t<IT> b_next, b_next2;   // bounds of tile n+1, n+2
    ids_t<IT> i_next;               // neighbour / self ids of tile n+1
    meta_t<IT, off_t> cur;
    f32x4 buf[kDepth][kNb + 1];
    // prologue: tile 0 completely, ids of tile 1, bounds of tile 2, first rows of tile 0 in flight
    {
      bounds_t<IT> b0;
      ids_t<IT> i0;
      p.load_bounds(tile_of(0), b0);
      p.load_bounds(tile_of(1), b_next);
      p.load_ids(tile_of(0), b0, i0);
      p.load_bounds(tile_of(2), b_next2);
      p.finish(tile_of(0), b0, i0, cur);
      p.load_ids(tile_of(1), b_next, i_next);
    }
#pragma unroll
    for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
    for (int64_t n = 0; n <= mine; n++) {
      if (n < mine) {
        uint32_t* tile_lds = lds + (n & 1) * tile_dw;
#pragma unroll
        for (int it = 0; it < IT; it++) {
          if (it + kDepth - 1 < IT) p.issue(cur, it + kDepth - 1, buf[(it + kDepth - 1) % kDepth]);
          p.reduce_store(cur, it, buf[it % kDepth], tile_lds);
        }
        p.long_rows(tile_of(n), cur, tile_lds);
        // roll the metadata pipeline one tile forward and put the first rows of tile n+1 in flight BEFORE the barrier
        meta_t<IT, off_t> nxt;
        p.finish(tile_of(n + 1), b_next, i_next, nxt);
        p.load_ids(tile_of(n + 2), b_next2, i_next);
        b_next = b_next2;
        p.load_bounds(tile_of(n + 3), b_next2);
        cur = nxt;
        if (n + 1 < mine) {
#pragma unroll
          for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
        }
      }
      lds_barrier();
    }
  } else {
    for (int64_t n = 0; n <= mine; n++) {
      if (n >= 1) consume_tile<TR>(a, tile_of(n - 1), lds + ((n - 1) & 1) * tile_dw, wave, lane);
      lds_barrier();
    }
  }
}

// ---- pre-split weight ------------------------------------------------------------------------------------------------
// w_t [K, N] fp32 row-major (ldw)  ->  planes [3][KS][N][16] bf16, rows K .. 16 KS - 1 zero
__global__ void split_weight_kernel(const float* __restrict__ w_t, int64_t ldw, int K, int N, int KS, uint32_t* __restrict__ planes)
{
  const int64_t total = (int64_t)KS * N * 8;  // dwords per plane
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk2 = (int)(i & 7);
    const int n   = (int)((i >> 3) % N);
    const int ks  = (int)((i >> 3) / N);
    const int k0  = ks * 16 + kk2 * 2;
    const float v0 = k0 < K ? w_t[(int64_t)k0 * ldw + n] : 0.f;
    const float v1 = k0 + 1 < K ? w_t[(int64_t)(k0 + 1) * ldw + n] : 0.f;
    uint32_t h0, m0, l0, h1, m1, l1;
    split3(v0, h0, m0, l0);
    split3(v1, h1, m1, l1);
    planes[i]             = pack_hi16(h0, h1);
    planes[total + i]     = pack_hi16(m0, m1);
    planes[2 * total + i] = pack_hi16(l0, l1);
  }
}

__host__ inline int row_stride_dw(int F)
{
  int sd = (F + 3) / 4 * 4;      // F dwords hold 2F bf16
  if ((sd / 4) % 2 == 0) sd += 4;  // sd = 4 * odd
  return sd;
}
constexpr size_t kLdsBudget = 160 * 1024;
__host__ inline size_t lds_bytes(int F, int TR) { return (size_t)(2 * 3 * TR * row_stride_dw(F) + 16) * 4; }

template <typename IdT, int LG, int TR, int CW>
void launch(const mfma_args& a, bool off32, hipStream_t st)
{
  int dev = 0, cus = 256;
  WG_HIP_CHECK(hipGetDevice(&dev));
  WG_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int64_t n_tiles = (a.n_rows + TR - 1) / TR;
  const size_t lds      = lds_bytes(a.F, TR);
  const int per_cu      = std::max<int>(1, (int)std::min<size_t>(kLdsBudget / lds, (size_t)(2048 / ((CW + kProducerWaves) * 64))));
  const int grid        = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)cus * per_cu));
  auto go               = [&](auto kern) {
    if (lds > 64 * 1024)
      WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<grid, (CW + kProducerWaves) * 64, lds, st>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  };
  if (off32) go(sage_layer_mfma_kernel<IdT, LG, TR, CW, true>);
  else go(sage_layer_mfma_kernel<IdT, LG, TR, CW, false>);
}

template <typename IdT, int LG, int TR>
void launch_cw(const mfma_args& a, bool off32, hipStream_t st)
{
  switch (a.N / 64) {
    case 1: launch<IdT, LG, TR, 1>(a, off32, st); break;
    case 2: launch<IdT, LG, TR, 2>(a, off32, st); break;
    default: launch<IdT, LG, TR, 4>(a, off32, st); break;
  }
}

// 64-row tiles whenever two of them fit the LDS (F <= 100), 32-row tiles otherwise; only the (LG, TR) pairs that can
// occur are instantiated: F <= 64 always fits, F > 128 never does
template <typename IdT, int LG>
void launch_tr(const mfma_args& a, bool off32, hipStream_t st)
{
  const bool fits64 = lds_bytes(a.F, 64) <= kLdsBudget;
  if constexpr (LG <= 16) {
    launch_cw<IdT, LG, 64>(a, off32, st);
  } else if constexpr (LG == 32) {
    if (fits64) launch_cw<IdT, LG, 64>(a, off32, st);
    else launch_cw<IdT, LG, 32>(a, off32, st);
  } else {
    launch_cw<IdT, LG, 32>(a, off32, st);
  }
}

template <typename IdT>
void launch_groups(const mfma_args& a, bool off32, hipStream_t st)
{
  const int units = a.F / 4;
  if (units <= 8) launch_tr<IdT, 8>(a, off32, st);
  else if (units <= 16) launch_tr<IdT, 16>(a, off32, st);
  else if (units <= 32) launch_tr<IdT, 32>(a, off32, st);
  else launch_tr<IdT, 64>(a, off32, st);
}

}  // namespace
}  // namespace wgamd

extern "C" size_t wgamd_sage_weight_planes_bytes(int K, int N) { return (size_t)3 * ((K + 15) / 16) * (size_t)N * 32; }

extern "C" int wgamd_sage_layer_bf16x3_supported(int F, int N)
{
  return F > 0 && F % 4 == 0 && (N == 64 || N == 128 || N == 256) && wgamd::lds_bytes(F, 32) <= wgamd::kLdsBudget;
}

extern "C" wholememory_error_code_t wgamd_sage_split_weight_bf16x3(const float* w_t, int64_t ldw, int K, int N, void* planes,
                                                                   void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sage_split_weight_bf16x3", [&] {
    WG_REQUIRE_INPUT(w_t && planes && K > 0 && N > 0 && ldw >= N, "bad weight");
    const int KS        = (K + 15) / 16;
    const int64_t total = (int64_t)KS * N * 8;
    const int grid      = (int)std::min<int64_t>((total + 255) / 256, 2048);
    split_weight_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(w_t, ldw, K, N, KS, static_cast<uint32_t*>(planes));
    WG_HIP_CHECK(hipGetLastError());
  });
}

extern "C" wholememory_error_code_t wgamd_sage_layer_fused_bf16x3(const int* row_ptr, const int* col, int64_t n_rows,
                                                                  const float* x, int64_t ldx, int64_t x_rows, int F,
                                                                  const void* src_ids, wholememory_dtype_t src_ids_dtype,
                                                                  const int64_t* self_rows, int mean, const void* w_planes,
                                                                  int N, const float* bias, int relu, float* out, int64_t ldo,
                                                                  void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sage_layer_fused_bf16x3", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && F > 0 && N > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && self_rows && w_planes && out, "null pointer");
    if (!wgamd_sage_layer_bf16x3_supported(F, N) || ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0)
      throw logic_error(fmt("unsupported shape: F=%d (multiple of 4, two 32-row tiles within 160 KB of LDS), N=%d (64, 128 or "
                            "256), 16-B aligned rows", F, N));
    WG_REQUIRE_INPUT(ldo >= N, "leading dimension smaller than N");
    mfma_args a{row_ptr, col, n_rows, x, ldx, F, src_ids, self_rows, mean, static_cast<const uint32_t*>(w_planes), N,
                (2 * F + 15) / 16, bias, relu, out, ldo, row_stride_dw(F)};
    auto st          = static_cast<hipStream_t>(stream);
    const bool off32 = x_rows > 0 && (uint64_t)x_rows * (uint64_t)ldx * 4u < (1ull << 32);
    if (src_ids == nullptr) launch_groups<void>(a, off32, st);
    else if (src_ids_dtype == WHOLEMEMORY_DT_INT) launch_groups<int32_t>(a, off32, st);
    else if (src_ids_dtype == WHOLEMEMORY_DT_INT64) launch_groups<int64_t>(a, off32, st);
    else throw invalid_input("src_ids must be INT or INT64");
  });
}
