// Shared device pieces of the one-kernel SAGE layer (wg_sage_mfma.hip: producer / consumer waves; wg_sage_ws.hip: the
// weight-stationary variant): argument block, the exact 3-way bf16 split, the fetching side (`producer`), fragment loads,
// the six-product MFMA step and the LDS-transposed epilogue.  See wg_sage_mfma.hip for the design notes.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace sage_mfma {

using f32x4  = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4  = __attribute__((ext_vector_type(4))) uint32_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

#ifndef WG_MFMA_PRODUCTS
#define WG_MFMA_PRODUCTS 6   // (tuning only: fewer products = wrong results, used to price the matrix work)
#endif
#ifndef WG_MFMA_DEPTH
#define WG_MFMA_DEPTH 2   // destination rows in flight per producer lane group
#endif
constexpr int kProducerWaves = 4;
constexpr int kRingDepth     = WG_MFMA_DEPTH;

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
  uint32_t x_bytes;          // extent of x when it is below 2 GB (32-bit offsets, buffer loads), else 0
  int F;
  const void* src_ids;
  const int64_t* self_rows;
  int mean;
  const float* w_tiles;      // [KS][N][16] fp32: k-step s, column n, 16 consecutive k (wgamd_sage_split_weight_bf16x3);
                             // HALF mode (F > 148): pre-split bf16 planes [3][KS][N][8 dwords] behind the same pointer
  int N;
  int KS;                    // ceil(2F / 16)
  const float* bias;
  int relu;
  float* out;
  int64_t ldo;
  int SD;                    // floats per LDS tile row (>= 2F, = 4 * odd: conflict-free ds_read_b128 across rows)
  int debug;                 // tuning harness only, bit mask: 1 no consumers, 2 no producers, 4 no output stores,
                             // 64 roles by SIMD instead of by wave order, 128 no epilogue stagger, 16 / 32 s_setprio 3 for
                             // producers / consumers
  unsigned long long* stamps;  // tuning harness only: s_memtime stamps of workgroup 0, [step][wave][begin, work done]
  int64_t row_scale;         // bytes per unit of a row number: 4 ldx — or 1 when src_ids holds BYTE offsets (a peer-mapped table)
  float* agg_out;            // training (nullable): the finished mean / sum rows [n_rows, F] also go to HBM — the weight-gradient
  int64_t ld_agg;            // kernel (wg_sage_bwd.hip) reads them back instead of fetching every neighbour row a second time
  int full_tiles;            // host side only: w_tiles holds fp32 tiles and the launch takes whole 32-row tiles even where F
                             // would take 64-row half tiles (relu flag WGAMD_SAGE_FULL_TILES: a small launch, see wgamd_ext.h)
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

// LDS-only wait + workgroup barrier: in-flight global loads (prefetched rows / weight fragments) and stores stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__host__ __device__ constexpr int row_stride_dw(int F)
{
  int sd = (2 * F + 3) / 4 * 4;
  return (sd / 4) % 2 == 0 ? sd + 4 : sd;   // 4 * odd
}

// HALF mode (two 64-row tiles of [mean | self] do not fit the LDS: F > 148): the two LDS buffers hold the MEAN halves of two
// consecutive 64-row tiles, F floats per row (the self half is read from global memory by the multiplying waves); a row
// stride of 4 * odd >= F keeps ds_read_b128 conflict-free
__host__ __device__ constexpr int row_stride_half_dw(int F)
{
  int sd = (F + 3) / 4 * 4;
  return (sd / 4) % 2 == 0 ? sd + 4 : sd;
}

// ---------------------------------------------------------------------------------------------------------------------
// producer side
// ---------------------------------------------------------------------------------------------------------------------
template <int IT>
struct bounds_t {
  int s[IT], e[IT];
};
template <int IT>
struct ids_t {
  int deg[IT];     // e - s of the bounds (the bounds die when the ids are requested); -1 = row past n_rows
  int lcol[IT];
  int lself[IT];   // self row (a local row of x / src_ids: < 2^31)
};
template <int IT, typename off_t>
struct meta_t {
  int d[IT];       // degree; -1 = row past n_rows (no neighbours, zero self row)
  off_t src[IT];   // byte offset of THIS lane's neighbour row (lane `sub` holds neighbour `sub` of the row)
  off_t self[IT];  // byte offset of the self row
};

template <typename IdT, int LG, int TR, bool OFF32, bool HALF = false>
struct producer {
  using off_t                         = typename std::conditional<OFF32, uint32_t, int64_t>::type;
  static constexpr int kGroupsPerWave = 64 / LG;
  static constexpr int kGroups        = kGroupsPerWave * kProducerWaves;
  static constexpr int IT             = TR / kGroups;       // rows of a tile per lane group
  static constexpr int kNb            = LG < 10 ? LG : 10;  // neighbour rows prefetched per destination row (fan-out 10)
  static constexpr int kDepth         = IT < kRingDepth ? IT : kRingDepth;  // rows in flight per lane group
  static_assert(TR % kGroups == 0, "lane groups must tile the rows evenly");

  const mfma_args& a;
  const int sub, gbase, group, f0, f0c;
  const bool live;
  __amdgpu_buffer_rsrc_t rsrc;   // x as a raw buffer (OFF32): out-of-range offsets read as zero

  __device__ producer(const mfma_args& a_, int pw, int lane)
    : a(a_),
      sub(lane & (LG - 1)),
      gbase(lane & ~(LG - 1)),
      group(pw * kGroupsPerWave + lane / LG),
      f0((lane & (LG - 1)) * 4),
      f0c(((lane & (LG - 1)) * 4 < a_.F) ? (lane & (LG - 1)) * 4 : a_.F - 4),
      live((lane & (LG - 1)) * 4 < a_.F),
      rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_.x), 0, (int)a_.x_bytes, 0x00020000))
  {
  }

  __device__ __forceinline__ int64_t row_of(int64_t tile, int it) const { return tile * TR + group + it * kGroups; }

  // stage A: CSR bounds of the next tile (requested when a tile starts)
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
  // stage B: this lane's neighbour id of every row + the self row ids (requested half-way through the tile); unconditional
  __device__ __forceinline__ void load_ids(int64_t tile, const bounds_t<IT>& b, ids_t<IT>& v) const
  {
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int64_t row = row_of(tile, it);
      v.deg[it]         = row < a.n_rows ? b.e[it] - b.s[it] : -1;
      const int* pc     = (sub < b.e[it] - b.s[it]) ? a.col + b.s[it] + sub : a.row_ptr;  // row_ptr[0] == 0: a valid row
      v.lcol[it]        = *pc;
      v.lself[it]       = (int)a.self_rows[row < a.n_rows ? row : a.n_rows - 1];
    }
  }
  // stage C: byte offsets (with the id indirection of the fused-fetch variant: one more dependent load)
  __device__ __forceinline__ void finish(const ids_t<IT>& v, meta_t<IT, off_t>& m) const
  {
    const IdT* src_ids = static_cast<const IdT*>(a.src_ids);
#pragma unroll
    for (int it = 0; it < IT; it++) {
      m.d[it]    = v.deg[it];
      m.src[it]  = (off_t)(table_row<IdT>(src_ids, (int64_t)v.lcol[it]) * a.row_scale);
      m.self[it] = (off_t)(table_row<IdT>(src_ids, (int64_t)v.lself[it]) * a.row_scale);
    }
  }
  // request the kNb neighbour rows + the self row of row `it`; every load is unconditional
  __device__ __forceinline__ void issue(const meta_t<IT, off_t>& m, int it, f32x4* v) const
  {
#ifndef WG_ISSUE_READLANE64
#define WG_ISSUE_READLANE64 1
#endif
#ifndef WG_ISSUE_DPP
#define WG_ISSUE_DPP 1
#endif
#ifndef WG_ISSUE_BPERMUTE   // (tuning build: the round-3 form below for every lane-group width)
    // 32-lane groups: a neighbour's offset reaches the group through two v_readlane + a select instead of ds_bpermute —
    // no LDS round trip next to the multiplying waves' fragment reads (bench.py: 3.92 -> 4.02 G edges/s on one box, two runs each)
    if constexpr (LG == 32 && WG_ISSUE_DPP) {
      issue_dpp(m, it, v);
      return;
    }
    if constexpr (LG == 32 || (LG == 64 && WG_ISSUE_READLANE64)) {
      issue_all<0>(m, it, v);
      return;
    }
#endif
    if constexpr (OFF32) {
      // slots past the degree (and the self slot of a row past n_rows) get an out-of-range offset: zeros, no memory access
#pragma unroll
      for (int k = 0; k < kNb; k++) {
        const uint32_t off = (uint32_t)__shfl((int)m.src[it], gbase | (k & (LG - 1)), 64);
        v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, k < m.d[it] ? off + f0c * 4 : a.x_bytes, 0, 0));
      }
      if constexpr (!HALF)
        v[kNb] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, m.d[it] >= 0 ? (uint32_t)m.self[it] + f0c * 4 : a.x_bytes, 0, 0));
    } else {
      const char* xb = reinterpret_cast<const char*>(a.x);
#pragma unroll
      for (int k = 0; k < kNb; k++) {
        const int src_lane = gbase | (k & (LG - 1));
        const int lo       = __shfl((int)(m.src[it] & 0xffffffff), src_lane, 64);
        const int hi       = __shfl((int)((int64_t)m.src[it] >> 32), src_lane, 64);
        int64_t off        = ((int64_t)hi << 32) | (uint32_t)lo;
        off                = k < m.d[it] ? off : (int64_t)0;   // slots past the degree read row 0 (L1-resident), masked below
        v[k]               = *reinterpret_cast<const f32x4*>(xb + off + f0c * 4);
      }
      if constexpr (!HALF)
        v[kNb] = *reinterpret_cast<const f32x4*>(xb + (m.d[it] >= 0 ? (int64_t)m.self[it] : (int64_t)0) + f0c * 4);
    }
  }
  // 32-lane groups, the cheapest form: ONE v_permlane16_swap (gfx950) of the offset register with itself yields the register
  // with its even 16-lane rows duplicated into the odd ones — lane k < 16 of a group is then in BOTH rows of the group and a DPP
  // row_newbcast:k hands it to all 32 lanes in one instruction (two v_readlane + a select before).  The fetching waves are
  // issue-bound next to the multiplying waves of their SIMD: per destination row this removes ~40 of their instructions, and the
  // wide-x path drops its per-load select as well (a slot past the degree reads the offset of a lane that holds row 0's; the sum
  // masks it).  Layer-1 launch of bench.py 1.607 -> 1.561 ms, 4.00 -> 4.02 G edges/s (same box, four runs each).
  // (Also tried on top of it: dead slots pointing at a ZERO row so that the sum needs no per-slot select either — 40 fewer
  //  instructions per row and SLOWER, 1.57 -> 1.615 ms; not kept.)
  template <int k>
  __device__ __forceinline__ void issue_dpp_one(const meta_t<IT, off_t>& m, int it, f32x4* v, int dlo, int dhi) const
  {
    if constexpr (k < kNb) {
      const int lo = __builtin_amdgcn_update_dpp(0, dlo, 0x150 + k, 0xf, 0xf, false);
      if constexpr (OFF32) {
        v[k] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, k < m.d[it] ? (uint32_t)lo + f0c * 4 : a.x_bytes, 0, 0));
      } else {
        const int hi      = __builtin_amdgcn_update_dpp(0, dhi, 0x150 + k, 0xf, 0xf, false);
        const int64_t off = ((int64_t)hi << 32) | (uint32_t)lo;   // (lanes past the degree hold a valid row's offset: masked in reduce_store)
        v[k]              = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x) + off + f0c * 4);
      }
      issue_dpp_one<k + 1>(m, it, v, dlo, dhi);
    }
  }
  __device__ __forceinline__ void issue_dpp(const meta_t<IT, off_t>& m, int it, f32x4* v) const
  {
    static_assert(kNb <= 16, "the first window sits in the even row of the group");
    const int slo = (int)((uint64_t)m.src[it] & 0xffffffffu);
    const int dlo = __builtin_amdgcn_permlane16_swap(slo, slo, false, false)[0];
    int dhi       = 0;
    if constexpr (!OFF32) {
      const int shi = (int)((uint64_t)m.src[it] >> 32);
      dhi           = __builtin_amdgcn_permlane16_swap(shi, shi, false, false)[0];
    }
    issue_dpp_one<0>(m, it, v, dlo, dhi);
    if constexpr (!HALF) issue_one<kNb>(m, it, v);
  }
  // value of lane k of this lane's group.  Two 32-lane groups per wave: two v_readlane + a select — no LDS round trip
  // (ds_bpermute) and no lgkmcnt wait, which an in-order wave with MFMAs queued behind it cannot afford eleven times per row
  template <int k>
  __device__ __forceinline__ int group_lane(int v) const
  {
    if constexpr (LG == 32) {
      const int lo = __builtin_amdgcn_readlane(v, k), hi = __builtin_amdgcn_readlane(v, 32 + k);
      return gbase ? hi : lo;
    } else if constexpr (LG == 64) {
      return __builtin_amdgcn_readlane(v, k);   // one group per wave: the offset is wave-uniform
    } else {
      return __shfl(v, gbase | (k & (LG - 1)), 64);
    }
  }
  template <int k>
  __device__ __forceinline__ void issue_all(const meta_t<IT, off_t>& m, int it, f32x4* v) const
  {
    if constexpr (k < kNb + (HALF ? 0 : 1)) {
      issue_one<k>(m, it, v);
      issue_all<k + 1>(m, it, v);
    }
  }
  // ONE of the kNb + 1 loads of issue(): load k of row `it` (k == kNb: the self row)
  template <int k>
  __device__ __forceinline__ void issue_one(const meta_t<IT, off_t>& m, int it, f32x4* v) const
  {
    static_assert(!HALF || k < kNb, "no self slot in half-tile mode");
    if constexpr (OFF32) {
      if constexpr (k < kNb) {
        const uint32_t off = (uint32_t)group_lane<k>((int)m.src[it]);
        v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, k < m.d[it] ? off + f0c * 4 : a.x_bytes, 0, 0));
      } else {
        v[kNb] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, m.d[it] >= 0 ? (uint32_t)m.self[it] + f0c * 4 : a.x_bytes, 0, 0));
      }
    } else {
      const char* xb = reinterpret_cast<const char*>(a.x);
      if constexpr (k < kNb) {
        const int lo = group_lane<k>((int)(m.src[it] & 0xffffffff));
        const int hi = group_lane<k>((int)((int64_t)m.src[it] >> 32));
        int64_t off  = ((int64_t)hi << 32) | (uint32_t)lo;
        off          = k < m.d[it] ? off : (int64_t)0;
        v[k]         = *reinterpret_cast<const f32x4*>(xb + off + f0c * 4);
      } else {
        v[kNb] = *reinterpret_cast<const f32x4*>(xb + (m.d[it] >= 0 ? (int64_t)m.self[it] : (int64_t)0) + f0c * 4);
      }
    }
  }
  // sum row `it` from its ring slot (CSR order) and store [mean | self] as fp32
  __device__ __forceinline__ void reduce_store(const meta_t<IT, off_t>& m, int it, const f32x4* v, float* tile_lds) const
  {
    const int deg = m.d[it];
    f32x4 acc     = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kNb; k++) {
      if constexpr (OFF32) acc += v[k];                                   // the hardware already zeroed the dead slots
      else acc += k < deg ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};             // select, never multiply by 0
    }
    if (a.mean && deg > 0 && deg <= kNb) acc *= __frcp_rn((float)deg);   // (longer rows: long_rows() continues this sum)
    if (live) {
      float* prow = tile_lds + (group + it * kGroups) * a.SD;
      *reinterpret_cast<f32x4*>(prow + f0) = acc;
      if constexpr (!HALF) {
        f32x4 self = v[kNb];
        if constexpr (!OFF32) self = deg >= 0 ? self : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(prow + a.F + f0) = self;
      }
    }
  }
  // SECOND WINDOW: neighbours kNb .. kNb + kW2 - 1 of the rows that have them.  Their byte offsets are already in registers —
  // load_ids / finish gave lane `sub` of the group the offset of neighbour `sub`, whatever the degree — so a row behind a
  // fan-out of 25 needs no further id loads: kW2 row loads go out together and continue the window's UNSCALED partial sum
  // (left in the tile by reduce_store) in CSR order.  Rows with deg <= kNb + kW2 are finished here.
  static constexpr int kW2 = LG >= 32 ? 16 : (LG > kNb ? LG - kNb : 0);
  __device__ __forceinline__ void second_window(const meta_t<IT, off_t>& m, float* tile_lds) const
  {
    if constexpr (kW2 > 0) {
#pragma unroll
      for (int it = 0; it < IT; it++) {
        const int deg = m.d[it];
        if (__ballot(deg > kNb) == 0ull) continue;
        const bool mine = live && deg > kNb;
        f32x4 v[kW2];
#pragma unroll
        for (int k = 0; k < kW2; k++) {
          const int kk = kNb + k;
          if constexpr (OFF32) {
            const uint32_t off = (uint32_t)__shfl((int)m.src[it], gbase | kk, 64);
            v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kk < deg ? off + f0c * 4 : a.x_bytes, 0, 0));
          } else {
            const int lo = __shfl((int)(m.src[it] & 0xffffffff), gbase | kk, 64);
            const int hi = __shfl((int)((int64_t)m.src[it] >> 32), gbase | kk, 64);
            int64_t off  = ((int64_t)hi << 32) | (uint32_t)lo;
            off          = kk < deg ? off : (int64_t)0;
            v[k]         = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x) + off + f0c * 4);
          }
        }
        float* prow = tile_lds + (group + it * kGroups) * a.SD + f0;
        f32x4 acc   = {0.f, 0.f, 0.f, 0.f};
        if (mine) acc = *reinterpret_cast<const f32x4*>(prow);
#pragma unroll
        for (int k = 0; k < kW2; k++) {
          if constexpr (OFF32) acc += v[k];
          else acc += kNb + k < deg ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (mine) {
          if (a.mean && deg <= kNb + kW2) acc *= __frcp_rn((float)deg);
          *reinterpret_cast<f32x4*>(prow) = acc;
        }
      }
    }
  }
  // rows longer than BOTH windows (deg > 26 at F >= 100): the tile holds the UNSCALED sum of their first kNb + kW2 neighbours;
  // the rest is added to it in CSR order, kLongUnroll row
  // loads in flight at a time (one at a time — a dependent round trip per neighbour — made the 47-class head of the products
  // model, whose hop has fan-out 25, three times slower than aggregate + GEMM)
  static constexpr int kLongUnroll = 8;
  __device__ __forceinline__ void long_rows(int64_t tile, const meta_t<IT, off_t>& m, float* tile_lds) const
  {
    long_rows_from<kNb + kW2, kLongUnroll>(tile, m, tile_lds);
  }
  // (kStart = kNb without second_window(): every row past the first window, kUnroll row loads in flight — the
  //  weight-stationary kernel has no registers for sixteen)
  template <int kStart, int kUnroll>
  __device__ __forceinline__ void long_rows_from(int64_t tile, const meta_t<IT, off_t>& m, float* tile_lds) const
  {
    const IdT* src_ids = static_cast<const IdT*>(a.src_ids);
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int deg = m.d[it];
      if (__ballot(deg > kStart) == 0ull) continue;
      const bool mine    = live && deg > kStart;
      const int64_t row  = row_of(tile, it);
      const int64_t rowc = row < a.n_rows ? row : a.n_rows - 1;
      const int s        = a.row_ptr[rowc];
      float* prow        = tile_lds + (group + it * kGroups) * a.SD + f0;
      f32x4 acc          = {0.f, 0.f, 0.f, 0.f};
      if (mine) acc = *reinterpret_cast<const f32x4*>(prow);
      int maxdeg = deg > kStart ? deg : 0;
#pragma unroll
      for (int dd = 32; dd >= LG; dd >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, dd, 64));
      for (int c0 = kStart; c0 < maxdeg; c0 += LG) {
        const int64_t my_src = (deg > kStart && c0 + sub < deg) ? table_row<IdT>(src_ids, (int64_t)a.col[s + c0 + sub]) : 0;
        const int chunk      = min(LG, maxdeg - c0);
        for (int j0 = 0; j0 < chunk; j0 += kUnroll) {
          f32x4 v[kUnroll];
#pragma unroll
          for (int u = 0; u < kUnroll; u++) {
            const int j        = j0 + u;
            const int src_lane = gbase | (j & (LG - 1));
            const int lo       = __shfl((int)(my_src & 0xffffffff), src_lane, 64);
            const int hi       = __shfl((int)(my_src >> 32), src_lane, 64);
            const int64_t rr   = (mine && j < chunk && c0 + j < deg) ? (((int64_t)hi << 32) | (uint32_t)lo) : (int64_t)0;
            // (dead slots read row 0, masked below)
            v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x) + rr * a.row_scale + f0c * 4);
          }
#pragma unroll
          for (int u = 0; u < kUnroll; u++) {
            const int j = j0 + u;
            acc += (mine && j < chunk && c0 + j < deg) ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      if (mine) {
        if (a.mean) acc *= __frcp_rn((float)deg);
        *reinterpret_cast<f32x4*>(prow) = acc;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// consumer side: wave cw multiplies the [TR x 2F] tile by columns [64 cw, 64 cw + 64) of the weight
// ---------------------------------------------------------------------------------------------------------------------
template <int RT>
struct araw_t {
  f32x4 v[RT][2];  // [row tile][k 0-3 | k 4-7 of this lane's half k-step]
};
template <int RT>
struct afrag_t {
  u32x4 v[RT][3];  // [row tile][plane]
};
struct bfrag_t {
  u32x4 v[2][3];  // [col tile][plane]
};
struct braw_t {
  f32x4 v[2][2];  // [col tile][k 0-3 | k 4-7 of this lane's half k-step]: the fp32 weight as it travels
};

template <int RT>
__device__ __forceinline__ void load_a_raw(araw_t<RT>& f, const float* a_lane, int sd, int ks)
{
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    f.v[rt][0] = *reinterpret_cast<const f32x4*>(a_lane + rt * 32 * sd + ks * 16);
    f.v[rt][1] = *reinterpret_cast<const f32x4*>(a_lane + rt * 32 * sd + ks * 16 + 4);
  }
}
// fp32 fragment -> the three bf16 planes (VALU work that issues under the wave's own MFMAs)
template <int RT>
__device__ __forceinline__ void split_a(const araw_t<RT>& r, afrag_t<RT>& f)
{
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
#ifdef WG_ABL_NO_SPLIT   // tuning build: no split work (wrong results) — prices the VALU side of the multiplying waves
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f.v[rt][0][j] = __float_as_uint(r.v[rt][0][j]);
      f.v[rt][1][j] = __float_as_uint(r.v[rt][1][j]);
      f.v[rt][2][j] = __float_as_uint(r.v[rt][0][j]);
    }
    continue;
#endif
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      split3(r.v[rt][0][i], h[i], m[i], l[i]);
      split3(r.v[rt][1][i], h[4 + i], m[4 + i], l[4 + i]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f.v[rt][0][j] = pack_hi16(h[2 * j], h[2 * j + 1]);
      f.v[rt][1][j] = pack_hi16(m[2 * j], m[2 * j + 1]);
      f.v[rt][2][j] = pack_hi16(l[2 * j], l[2 * j + 1]);
    }
  }
}
// The weight travels as fp32 (4 B per element) and is split into its three bf16 planes by the multiplying wave, in the issue
// slots under its own MFMAs — the pre-split planes of round 2 were 6 B per element, and the weight stream (once per 64-row
// tile per CU, through the same vector-memory pipeline as the row fetches) is what the layer's time is most sensitive to:
// each third of it costs 0.045 ms of the 0.55 ms layer-1 launch (compile-time ablations).  Same products,
// bit-identical results.
__device__ __forceinline__ void load_b(braw_t& f, const float* b_lane, int n_cols, int ks)
{
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    const float* p = b_lane + ((int64_t)ks * n_cols + ct * 32) * 16;
    f.v[ct][0]     = *reinterpret_cast<const f32x4*>(p);
    f.v[ct][1]     = *reinterpret_cast<const f32x4*>(p + 4);
  }
}
// HALF mode (F > 148, K = 512: the multiplying waves are the busier side there and the split of the weight in registers
// costs more than the bytes it saves: 1.74 vs 1.69 ms at 256 -> 256 in round 2; with the round-3 loop there are no
// registers left for fp32 fragments two k-steps ahead AND split planes) keeps the pre-split planes [3][KS][N][8 dwords]
__device__ __forceinline__ void load_b_planes(bfrag_t& f, const uint32_t* b_lane, int64_t b_plane_dw, int n_cols, int ks)
{
#ifdef WG_ABL_NO_B
  return;
#endif
#pragma unroll
  for (int ct = 0; ct < 2; ct++)
#pragma unroll
    for (int p = 0; p < 3; p++)
      f.v[ct][p] = *reinterpret_cast<const u32x4*>(b_lane + p * b_plane_dw + ((int64_t)ks * n_cols + ct * 32) * 8);
}
__device__ __forceinline__ void split_b(const braw_t& r, bfrag_t& f)
{
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      split3(r.v[ct][0][i], h[i], m[i], l[i]);
      split3(r.v[ct][1][i], h[4 + i], m[4 + i], l[4 + i]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f.v[ct][0][j] = pack_hi16(h[2 * j], h[2 * j + 1]);
      f.v[ct][1][j] = pack_hi16(m[2 * j], m[2 * j + 1]);
      f.v[ct][2][j] = pack_hi16(l[2 * j], l[2 * j + 1]);
    }
  }
}

template <int RT>
__device__ __forceinline__ void mma_frags(f32x16 (&c)[RT][2], const afrag_t<RT>& fa, const bfrag_t& fb)
{
  // smallest terms first; per accumulator tile the six products are independent MFMAs on the same accumulator
  constexpr int pa[6] = {2, 0, 1, 1, 0, 0};
  constexpr int pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < WG_MFMA_PRODUCTS; t++)
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
        c[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa.v[rt][pa[t]]),
                                                            __builtin_bit_cast(bf16x8, fb.v[ct][pb[t]]), c[rt][ct], 0, 0, 0);
}

// ---- epilogue: bias, activation, 16-B stores ---------------------------------------------------------------------------
// C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane owns one column, so a
// direct store is 4 B per lane (two 128-B segments per instruction, 64 instructions per wave and tile, each one a slot in
// the CU's memory pipeline that the producers' row fetches queue behind).  The four registers 4g .. 4g+3 of all lanes are
// the 8 consecutive rows 8g .. 8g+7 of a row tile: they go through this wave's 2 KiB LDS scratch [8][64] and leave as
// 16 B per lane, 256 B per row and wave — 16 store instructions of 1 KiB per wave and tile.
constexpr int kScratchDw = 8 * 64;   // per consumer wave

template <int RT>
__device__ __forceinline__ void epilogue(const mfma_args& a, f32x16 (&c)[RT][2], int64_t row0, int cw, int lane, float* scratch)
{
  const int lm = lane & 31, lh = lane >> 5;
  float bj[2];
#pragma unroll
  for (int ct = 0; ct < 2; ct++) bj[ct] = a.bias ? a.bias[cw * 64 + ct * 32 + lm] : 0.f;
  if (a.debug & 4) {
    if (c[0][0][0] == 12345.678f) a.out[0] = c[0][1][3] + c[RT - 1][1][5];
    return;
  }
  const bool full = row0 + RT * 32 <= a.n_rows;
  const int rl = lane >> 4, cl = (lane & 15) * 4;
  float* obase = a.out + (row0 + rl) * a.ldo + cw * 64 + cl;
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const float v = c[rt][ct][4 * g + jj] + bj[ct];
          scratch[(jj + 4 * lh) * 64 + ct * 32 + lm] = a.relu ? fmaxf(v, 0.f) : v;
        }
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (rl + 4 * pass) * 64 + cl);
        const int r   = rt * 32 + 8 * g + 4 * pass;   // + rl
        // (non-temporal stores here: 0.518 -> 0.514 ms, inside the noise — not taken)
        if (full || row0 + r + rl < a.n_rows) *reinterpret_cast<f32x4*>(obase + (int64_t)r * a.ldo) = v;
      }
    }
}

// Training: the multiplying waves copy the aggregate half of the tile they are about to multiply (complete behind the barrier,
// scaled) to a.agg_out — wave cw of CW the rows [cw TR / CW, (cw + 1) TR / CW), 16 B per lane.  Nothing else changes: the
// forward result is bit for bit the one of the inference launch.
template <int TR, int CW>
__device__ __forceinline__ void store_agg(const mfma_args& a, const float* tile_lds, int64_t row0, int cw, int lane)
{
  if (a.agg_out == nullptr) return;
  constexpr int kRows = TR / CW;
  const int q_row = a.F >> 2, total = kRows * q_row;
  for (int i = lane; i < total; i += 64) {
    const int r = i / q_row, q = i - r * q_row;
    const int64_t row = row0 + cw * kRows + r;
    if (row < a.n_rows)
      *reinterpret_cast<f32x4*>(a.agg_out + row * a.ld_agg + q * 4) =
        *reinterpret_cast<const f32x4*>(tile_lds + (cw * kRows + r) * a.SD + q * 4);
  }
}

// The same epilogue in eight pieces — piece (rt, g) = the 8 rows 32 rt + 8 g .. + 7 — for a caller that spreads the stores
// between other work (wg_sage_ws.hip)
struct epilogue_lane_t {
  float bj[2];
  float* obase;
  int64_t row0;
  bool full;
};
template <int RT>
__device__ __forceinline__ void epilogue_begin(const mfma_args& a, epilogue_lane_t& e, int64_t row0, int cw, int lane)
{
#pragma unroll
  for (int ct = 0; ct < 2; ct++) e.bj[ct] = a.bias ? a.bias[cw * 64 + ct * 32 + (lane & 31)] : 0.f;
  e.row0  = row0;
  e.full  = row0 + RT * 32 <= a.n_rows;
  e.obase = a.out + (row0 + (lane >> 4)) * a.ldo + cw * 64 + (lane & 15) * 4;
}
template <int RT, int rt, int g>
__device__ __forceinline__ void epilogue_piece(const mfma_args& a, const epilogue_lane_t& e, f32x16 (&c)[RT][2], int lane,
                                               float* scratch)
{
  const int lm = lane & 31, lh = lane >> 5, rl = lane >> 4, cl = (lane & 15) * 4;
#pragma unroll
  for (int ct = 0; ct < 2; ct++)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const float v = c[rt][ct][4 * g + jj] + e.bj[ct];
      scratch[(jj + 4 * lh) * 64 + ct * 32 + lm] = a.relu ? fmaxf(v, 0.f) : v;
    }
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (rl + 4 * pass) * 64 + cl);
    const int r   = rt * 32 + 8 * g + 4 * pass;   // + rl
    if (e.full || e.row0 + r + rl < a.n_rows) *reinterpret_cast<f32x4*>(e.obase + (int64_t)r * a.ldo) = v;
  }
}

// weight-stationary variant (wg_sage_ws.hip); id_kind: 0 = no indirection, 1 = int32 src_ids, 2 = int64 src_ids
bool sage_ws_supported(int F, int N);
void sage_ws_launch(const mfma_args& a, int id_kind, hipStream_t st);

}  // namespace sage_mfma
}  // namespace wgamd
