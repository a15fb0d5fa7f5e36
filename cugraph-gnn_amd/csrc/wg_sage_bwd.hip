// Weight gradient of the one-kernel SAGEConv layer (wg_sage_mfma.hip) over a sampled hop:
//     dW_t [2F, N] = [ agg | X[self] ]^T (n_rows x 2F)  @  dZ (n_rows x N),       dZ = grad_out masked by the layer's ReLU
// — the training half of what the reference's models do with torch_geometric.nn.SAGEConv
// (python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59,119-125: forward, loss.backward(), optimizer step).
//
// Why its own kernel: the product is a "TN" GEMM whose reduction runs over the ROWS of the hop (1.8 M per call group of the
// products workload) with a small [2F, N] = [200, 256] result — a library fp32 GEMM of that shape is bound by the fp32 matrix
// rate (180 GFLOP at ~110 TF/s: 1.6 ms) and needs the [n_rows, 2F] operand materialised first.  Here:
//   * the aggregate half comes from the forward launch (wgamd_sage_layer_fused_bf16x3_train keeps it: n_rows F 4 bytes written
//     once, instead of E F 4 bytes of neighbour rows fetched a second time), the self half is gathered through self_rows /
//     src_ids like the forward does;
//   * split-K: every workgroup owns a contiguous range of rows and keeps its whole [2F, N-block] partial sum in MFMA
//     accumulators (8 waves x up to 8 tiles of 32 x 32) for its whole life; it is written once, and a second launch adds the
//     partial sums in workgroup order (no atomics: same bits from run to run);
//   * the product is the exact 3-way bf16 split of the forward (six v_mfma_f32_32x32x16_bf16 per fp32 product, fp32
//     accumulate).  Both MFMA operands are K-major here (K = hop rows): the A side ([agg | self] rows) is split ONCE per
//     element by the thread that loaded it and stored TRANSPOSED into LDS as three bf16 planes [feature][row] (a fragment is
//     one ds_read_b128); the B side (dZ) never touches the LDS — a wave owns 32 output columns, so its fragment is 8 rows of
//     one column per lane: eight 4-byte loads whose wave-wide footprint is two 128-B segments each, split in registers by the
//     only wave that needs them;
//   * double-buffered over 32-row (F <= 128) or 16-row tiles, one barrier per tile: the loads of tile t+1 are in flight
//     during the MFMAs of tile t, their split + LDS store follows, the row ids of the self half run two tiles ahead.
#include "wg_sage_mfma_parts.hpp"

namespace wgamd {
namespace {
using namespace sage_mfma;

struct wgrad_args {
  const float* agg;
  int64_t ld_agg;
  const float* x;
  int64_t ldx;                  // floats per unit of self_global: the row stride — or 1 when the rows were given as byte offsets
  int F;
  const int64_t* self_global;   // row of x holding destination i itself (self_rows with the src_ids indirection applied)
  int64_t n_rows;
  const float* g;
  int64_t ldg;
  const float* act;             // nullable: the layer's ReLU output — dZ = g where act > 0
  int64_t ld_act;
  int N;
  float* part;                  // [gridDim.y][gridDim.x][KL + 1][NB]: partial sums, row KL = the bias gradient
  int64_t rows_per_block;       // multiple of TR
  int KL;                       // feature rows of the LDS planes and of a partial sum: 2F rounded up to 32
};

// kind: 1 int32 row numbers, 2 int64 row numbers, 3 int64 BYTE offsets (-> float offsets: rows are 16-B aligned)
__global__ void compose_self_kernel(const int64_t* __restrict__ self_rows, const void* __restrict__ src_ids, int kind,
                                    int64_t n, int64_t* __restrict__ out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = self_rows[i];
    const int64_t v = kind == 1 ? (int64_t) static_cast<const int32_t*>(src_ids)[s] : static_cast<const int64_t*>(src_ids)[s];
    out[i]          = kind == 3 ? v >> 2 : v;
  }
}

// TR rows per tile; wave w owns columns [32 (w % WN), +32) of the workgroup's N-block and the (up to) MT feature tiles
// [MT (w / WN), +MT) of 32 that exist (a.KL / 32 of them)
// MASK: a.act is given (a runtime test inside the unrolled load loops made the compiler peel the mask loads into a rolled loop
// over a stack array, every load waited for on the spot: 14 us per 32-row tile instead of 3)
template <int TR, int MT, int WN, bool MASK>
__global__ void __launch_bounds__(512) sage_wgrad_kernel(wgrad_args a)
{
  constexpr int TRP = TR + 8, KS = TR / 16, NB = WN * 32;
  constexpr int RG = TR / 4, QG = 16 / RG;   // staging: 4-row groups per tile, quads (4 features) per 16-lane group
  const int kPlane = a.KL * TRP, kBuf = 3 * kPlane;   // bf16 elements
  extern __shared__ __attribute__((aligned(16))) uint16_t planes[];   // [2][3][KL][TRP]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave % WN, wm = wave / WN, lm = lane & 31, lh = lane >> 5;
  for (int i = t; i < kBuf; i += 512) reinterpret_cast<uint32_t*>(planes)[i] = 0u;   // 2 kBuf bf16 = kBuf dwords
  __syncthreads();

  const int mt_live = min(MT, a.KL / 32 - wm * MT);   // (<= 0: this wave only helps staging)
  const int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_block;
  const int64_t r_end   = min(a.n_rows, r_begin + a.rows_per_block);
  const int64_t n_tiles = r_begin < r_end ? (r_end - r_begin + TR - 1) / TR : 0;

  // ---- staging task of this thread: quad q (features 4q .. 4q+3 of [agg | self]) x rows 4 rg .. 4 rg + 3 of the tile ----
  const int l16 = t & 15, rg = l16 & (RG - 1), q = (t >> 4) * QG + l16 / RG;
  const int FQ = a.F >> 2;
  const bool stage = q < 2 * FQ, self_half = q >= FQ;
  f32x4 st[4];
  int64_t idx_next[4];   // self half: rows of x for the tile after the one being staged
  auto load_idx = [&](int64_t tile) {
    if (!(stage && self_half)) return;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t row = r_begin + tile * TR + rg * 4 + j;
      idx_next[j]       = a.self_global[row < r_end ? row : r_begin];
    }
  };
  // Tile rows are read as RAW BUFFER loads: the descriptor is the tile (wave-uniform base, rows that exist as its extent),
  // the lane part of the address is one 32-bit register fixed for the whole launch — one VALU add per load, and a row past
  // the end of the matrix reads as zero without a select (64-bit per-load addresses made the compiler keep 32 pointers live
  // and spill them).
  auto tile_rsrc = [&](const float* base, int64_t ld, int64_t row0) {
    const int64_t rows = min((int64_t)TR, a.n_rows - row0);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + row0 * ld), 0, (int)(rows * ld * 4), 0x00020000);
  };
  const uint32_t agg_lane = ((uint32_t)(rg * 4) * (uint32_t)a.ld_agg + (uint32_t)q * 4u) * 4u;   // bytes
  auto load_stage = [&](int64_t tile) {
    if (!stage) return;
#ifdef WG_ABL_NO_STAGE   // tuning build (wrong results): no [agg | self] row loads
    return;
#endif
    if (self_half) {
#pragma unroll
      for (int j = 0; j < 4; j++) st[j] = *reinterpret_cast<const f32x4*>(a.x + idx_next[j] * a.ldx + (q - FQ) * 4);
    } else {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(a.agg, a.ld_agg, r_begin + tile * TR);
      uint32_t al = agg_lane;
      asm volatile("" : "+v"(al));
#pragma unroll
      for (int j = 0; j < 4; j++)
        st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, al + (uint32_t)(j * 4) * (uint32_t)a.ld_agg, 0, 0));
    }
  };
  auto store_stage = [&](int64_t tile, uint16_t* buf) {
    if (!stage) return;
#ifdef WG_ABL_NO_STORE   // tuning build (wrong results): no split + transposed LDS store of the A side
    return;
#endif
    uint32_t h[4][4], m[4][4], l[4][4];   // [row j][feature i]
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool ok = r_begin + tile * TR + rg * 4 + j < r_end;
#pragma unroll
      for (int i = 0; i < 4; i++) split3(ok ? st[j][i] : 0.f, h[j][i], m[j][i], l[j][i]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint16_t* dst = buf + (q * 4 + i) * TRP + rg * 4;
      using u32x2   = __attribute__((ext_vector_type(2))) uint32_t;
      *reinterpret_cast<u32x2*>(dst)              = u32x2{pack_hi16(h[0][i], h[1][i]), pack_hi16(h[2][i], h[3][i])};
      *reinterpret_cast<u32x2*>(dst + kPlane)     = u32x2{pack_hi16(m[0][i], m[1][i]), pack_hi16(m[2][i], m[3][i])};
      *reinterpret_cast<u32x2*>(dst + 2 * kPlane) = u32x2{pack_hi16(l[0][i], l[1][i]), pack_hi16(l[2][i], l[3][i])};
    }
  };

  // ---- dZ fragments of this wave: column `col`, rows 16 ks + 8 lh + j ----------------------------------------------------
  const int col      = blockIdx.y * NB + wn * 32 + lm;
  const bool col_ok  = col < a.N;
  const int colc     = col_ok ? col : 0;
  const bool z_wave  = mt_live > 0;   // (a wave without live feature tiles needs no dZ; wm == 0 always has some)
  float zr[KS][8], zm[KS][8];
  u32x4 zf[KS][3];
  float bsum = 0.f;
  const uint32_t g_lane = ((uint32_t)(lh * 8) * (uint32_t)a.ldg + (uint32_t)colc) * 4u;      // bytes
  const uint32_t m_lane = ((uint32_t)(lh * 8) * (uint32_t)a.ld_act + (uint32_t)colc) * 4u;
  auto load_z = [&](int64_t tile) {
    if (!z_wave) return;
#ifdef WG_ABL_NO_Z   // tuning build (wrong results): no dZ / mask loads — prices the B side's memory instructions
    return;
#endif
    const __amdgpu_buffer_rsrc_t rg_ = tile_rsrc(a.g, a.ldg, r_begin + tile * TR);
    // (opaque to the optimiser: the 16 + 16 per-load offsets are then one add each per tile, not 32 registers kept live
    //  across the whole loop next to 112 accumulators)
    uint32_t gl = g_lane, ml = m_lane;
    asm volatile("" : "+v"(gl), "+v"(ml));
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int j = 0; j < 8; j++)
        zr[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg_, gl + (uint32_t)((ks * 16 + j) * 4) * (uint32_t)a.ldg, 0, 0));
    if constexpr (MASK) {
      const __amdgpu_buffer_rsrc_t rm_ = tile_rsrc(a.act, a.ld_act, r_begin + tile * TR);
#pragma unroll
      for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++)
          zm[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm_, ml + (uint32_t)((ks * 16 + j) * 4) * (uint32_t)a.ld_act, 0, 0));
    }
  };
  auto split_z = [&](int64_t tile) {
    if (!z_wave) return;
    const int lim = (int)min((int64_t)TR, r_end - (r_begin + tile * TR)) - lh * 8;   // rows of this lane's half inside the block
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      uint32_t h[8], m[8], l[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool ok = col_ok && ks * 16 + j < lim && (!MASK || zm[ks][j] > 0.f);
        const float v = ok ? zr[ks][j] : 0.f;
        if (wm == 0) bsum += v;
        split3(v, h[j], m[j], l[j]);
      }
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        zf[ks][0][jj] = pack_hi16(h[2 * jj], h[2 * jj + 1]);
        zf[ks][1][jj] = pack_hi16(m[2 * jj], m[2 * jj + 1]);
        zf[ks][2][jj] = pack_hi16(l[2 * jj], l[2 * jj + 1]);
      }
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[m][i] = 0.f;

  if (n_tiles > 0) {
    // prologue: tile 0 staged and stored, tile 1's row ids known
    load_idx(0);
    load_stage(0);
    load_z(0);
    load_idx(1);
    store_stage(0, planes);
    split_z(0);
    lds_barrier();
    // Where a tile's time goes (compile-time ablations WG_ABL_*, products layer-1 hop, 1.38 ms): no dZ loads -0.47 ms, no
    // [agg | self] loads -0.35, no MFMAs -0.44 — the parts ADD; without MFMAs the loads alone run at 4.9 TB/s.  Loading and
    // multiplying on the same SIMD time-slice on this chip instead of overlapping (wg_sage_mfma.hip measured the same between
    // its fetching and multiplying waves).  What was tried against it, in this order: the two waves of a SIMD out of phase
    // (one requests then multiplies, the other the reverse): +3 % at F = 100, -13 % at F = 256; the dZ loads of tile t + 1
    // dealt BETWEEN the MFMA blocks of tile t with a VALU-computed offset: +14 % (the register allocator reused destinations
    // of loads in flight for the offsets and put vmcnt(0) before every group); the same with the row part of the address in
    // the instruction's scalar offset (no VALU temporary): -1 % / -15 %.  Kept: the last one.
    constexpr int pa_[6] = {2, 0, 1, 1, 0, 0};   // smallest terms first (as the forward)
    constexpr int pb_[6] = {0, 2, 1, 0, 1, 0};
    auto frags = [&](const uint16_t* ap, u32x4 (&fa)[3]) {
#pragma unroll
      for (int p = 0; p < 3; p++) fa[p] = *reinterpret_cast<const u32x4*>(ap + p * kPlane);
    };
    constexpr int kBlocks = KS * ((MT + 1) / 2), kEntries = KS * 8;       // MFMA blocks per tile; (k-step, row) pairs of dZ
    constexpr int kPerBlock = (kEntries + kBlocks - 1) / kBlocks;
    // all MT tiles of this wave exist (every launch shape of the BASELINE models): straight-line code, feature tiles in PAIRS so
    // that consecutive MFMAs go to different accumulators
    auto multiply_full = [&](const uint16_t* pb, bool more, int64_t tile) {
      __amdgpu_buffer_rsrc_t rg_ = tile_rsrc(a.g, a.ldg, r_begin + (tile + 1) * TR), rm_ = rg_;
      if constexpr (MASK) rm_ = tile_rsrc(a.act, a.ld_act, r_begin + (tile + 1) * TR);
      uint32_t gl = g_lane, ml = m_lane;
      asm volatile("" : "+v"(gl), "+v"(ml));
      // (the row part of the address rides in the instruction's SCALAR offset: no VALU temporary per load — with one, the
      //  register allocator reused the destinations of loads still in flight for it and put a vmcnt(0) before every group.
      //  The scalar offset is outside the descriptor's range check, so the one tile that crosses the end of the matrix takes
      //  load_z() up front instead.)
      const bool inside = r_begin + (tile + 2) * TR <= a.n_rows;
      if (more && !inside) load_z(tile + 1);
      more = more && inside;
      auto zload = [&](int e) {
        const int ks = e / 8, j = e % 8;
#ifndef WG_ABL_NO_Z
        zr[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg_, gl, (ks * 16 + j) * 4 * (int)a.ldg, 0));
        if constexpr (MASK)
          zm[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm_, ml, (ks * 16 + j) * 4 * (int)a.ld_act, 0));
#endif
      };
      int blk = 0;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
#pragma unroll
        for (int m = 0; m < MT; m += 2) {
          u32x4 fa[3], fb[3];
          frags(pb + m * 32 * TRP + ks * 16, fa);
          if (m + 1 < MT) frags(pb + (m + 1) * 32 * TRP + ks * 16, fb);
#ifdef WG_ABL_NO_MFMA   // tuning build (wrong results): fragments are read, nothing is multiplied
          acc[m][0] += __uint_as_float(fa[0][0] ^ fa[1][1] ^ fa[2][2]);
#else
#pragma unroll
          for (int k6 = 0; k6 < 6; k6++) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[pa_[k6]]),
                                                             __builtin_bit_cast(bf16x8, zf[ks][pb_[k6]]), acc[m], 0, 0, 0);
            if (m + 1 < MT)
              acc[m + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[pa_[k6]]),
                                                                   __builtin_bit_cast(bf16x8, zf[ks][pb_[k6]]), acc[m + 1], 0, 0, 0);
          }
#endif
          if (more) {
#pragma unroll
            for (int e = 0; e < kPerBlock; e++)
              if (blk * kPerBlock + e < kEntries) zload(blk * kPerBlock + e);
          }
          __builtin_amdgcn_sched_barrier(0);
          blk++;
        }
      }
    };
    auto multiply_some = [&](const uint16_t* pb) {
#pragma unroll
      for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int m = 0; m < MT; m++)
          if (m < mt_live) {
            u32x4 fa[3];
            frags(pb + m * 32 * TRP + ks * 16, fa);
#pragma unroll
            for (int k6 = 0; k6 < 6; k6++)
              acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[pa_[k6]]),
                                                               __builtin_bit_cast(bf16x8, zf[ks][pb_[k6]]), acc[m], 0, 0, 0);
          }
    };
    for (int64_t tile = 0; tile < n_tiles; tile++) {
      const bool more = tile + 1 < n_tiles;
      if (more) {
        load_stage(tile + 1);   // (uses the ids requested one iteration ago)
        load_idx(tile + 2);
      }
      const uint16_t* pb = planes + (tile & 1) * kBuf + ((wm * MT) * 32 + lm) * TRP + lh * 8;
      if (mt_live == MT) multiply_full(pb, more, tile);
      else {
        if (more) load_z(tile + 1);
        multiply_some(pb);
      }
      if (more) {
        store_stage(tile + 1, planes + ((tile + 1) & 1) * kBuf);
        split_z(tile + 1);
      }
      lds_barrier();   // (LDS-only wait: the row ids requested for tile + 2 stay in flight)
    }
  }

  // ---- partial sums: [KM + 1][NB] of this workgroup; C/D map: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* mine = a.part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (int64_t)(a.KL + 1) * NB;
#pragma unroll
  for (int m = 0; m < MT; m++)
    if (m < mt_live) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int f = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        mine[(int64_t)f * NB + wn * 32 + lm] = acc[m][r];
      }
    }
  if (wm == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (lh == 0) mine[(int64_t)a.KL * NB + wn * 32 + lm] = bsum;
  }
}

// grad_w_l [N, F], grad_w_r [N, F], grad_bias [N]  (+)=  sum over the workgroups, in workgroup order
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int grid_x, int KL, int NB, int F, int N,
                                    float* __restrict__ gwl, float* __restrict__ gwr, float* __restrict__ gb, int accumulate)
{
  const int64_t total = (int64_t)(2 * F + 1) * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % N), f = (int)(i / N);   // f == 2F: the bias row
    const int y = n / NB, nl = n - y * NB;
    const float* p = part + ((int64_t)y * grid_x * (KL + 1) + (f == 2 * F ? KL : f)) * NB + nl;
    // eight loads in flight, eight running sums combined in a fixed tree: the order is a function of grid_x alone (run-to-run
    // deterministic); one serial sum over 115 partials was 19 us of a mini-batch's 330-us training step
    const int64_t step = (int64_t)(KL + 1) * NB;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= grid_x; b += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = p[(int64_t)(b + k) * step];
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] += v[k];
    }
    for (int k = 0; b < grid_x; b++, k++) acc[k] += p[(int64_t)b * step];
    const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    float* dst = f == 2 * F ? (gb ? gb + n : nullptr) : (f < F ? gwl + (int64_t)n * F + f : gwr + (int64_t)n * F + (f - F));
    if (dst) *dst = accumulate ? *dst + s : s;
  }
}

struct wgrad_plan {
  int TR, MT, WN, KL, NB, grid_y;
};
__host__ inline wgrad_plan plan_for(int F, int N)
{
  const int n_ct = (N + 31) / 32, mtiles = (2 * F + 31) / 32;
  int WN = 2;   // (N <= 32 runs with one dead column wave: no instances for a single one)
  while (WN < n_ct && WN < 8) WN *= 2;
  while (WN > 2 && (mtiles + (8 / WN) - 1) / (8 / WN) > 8) WN /= 2;   // at most 8 accumulator tiles per wave
  // (rows per tile: the LDS holds two tiles of three bf16 planes of 2F x (TR + 8); narrow shapes — the dense tails of the GAT layers,
  //  F = 64 — take 64-row tiles: a tile is so few MFMAs that the per-tile staging and barriers were the launch's time, 1.5 TB/s)
  static const bool wide_tiles = [] { const char* e = getenv("WGAMD_WGRAD_TR64"); return !(e && e[0] == '0'); }();
  const int WM = 8 / WN, mt = (mtiles + WM - 1) / WM, TR = (F <= 64 && wide_tiles) ? 64 : (F <= 128 ? 32 : 16);
  // (accumulator tiles per wave are a template parameter: 4, 8, and 7 for the products layer-1 shape, whose 16 registers
  //  fewer keep that instance clear of spills)
  const int MT = mt <= 4 ? 4 : (mt == 7 && TR == 32 && WN == 8 ? 7 : 8);
  return {TR, MT, WN, mtiles * 32, WN * 32, (n_ct + WN - 1) / WN};
}
__host__ inline size_t part_bytes(const wgrad_plan& p, int grid_x) { return (size_t)p.grid_y * grid_x * (p.KL + 1) * p.NB * 4; }
// workgroups along the rows: one per CU of the DEVICE at most (the workspace query does not know the stream's CU mask)
__host__ inline int max_grid_x(const wgrad_plan& p) { return std::max(1, stream_cu_count(nullptr) / p.grid_y); }

template <int TR, int MT>
void launch_wn(const wgrad_plan& p, const wgrad_args& a, dim3 grid, hipStream_t st)
{
  const size_t lds = (size_t)2 * 3 * p.KL * (TR + 8) * 2;
  auto go = [&](auto kern) {
    WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<grid, 512, lds, st>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  };
  const bool mask = a.act != nullptr;
  switch (p.WN) {
    case 2: mask ? go(sage_wgrad_kernel<TR, MT, 2, true>) : go(sage_wgrad_kernel<TR, MT, 2, false>); break;
    case 4: mask ? go(sage_wgrad_kernel<TR, MT, 4, true>) : go(sage_wgrad_kernel<TR, MT, 4, false>); break;
    default: mask ? go(sage_wgrad_kernel<TR, MT, 8, true>) : go(sage_wgrad_kernel<TR, MT, 8, false>); break;
  }
}
template <int TR>
void launch_mt(const wgrad_plan& p, const wgrad_args& a, dim3 grid, hipStream_t st)
{
  if (p.MT == 4) return launch_wn<TR, 4>(p, a, grid, st);
  if constexpr (TR == 32) {
    if (p.MT == 7) {
      const size_t lds = (size_t)2 * 3 * p.KL * (TR + 8) * 2;
      auto go          = [&](auto kern) {
        WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 512, lds, st>>>(a);
        WG_HIP_CHECK(hipGetLastError());
      };
      a.act != nullptr ? go(sage_wgrad_kernel<32, 7, 8, true>) : go(sage_wgrad_kernel<32, 7, 8, false>);
      return;
    }
  }
  launch_wn<TR, 8>(p, a, grid, st);
}

}  // namespace
}  // namespace wgamd

extern "C" size_t wgamd_sage_wgrad_workspace_bytes(int64_t n_rows, int F, int N)
{
  using namespace wgamd;
  if (F <= 0 || N <= 0 || F > 256 || N > 256) return 0;
  const wgrad_plan p = plan_for(F, N);
  return ((part_bytes(p, max_grid_x(p)) + 255) & ~(size_t)255) + (size_t)std::max<int64_t>(n_rows, 0) * 8 + 256;
}

extern "C" wholememory_error_code_t wgamd_sage_wgrad_bf16x3(const float* agg, int64_t ld_agg, const float* x, int64_t ldx, int F,
                                                            const void* src_ids, wholememory_dtype_t src_ids_dtype,
                                                            const int64_t* self_rows, int64_t n_rows, const float* grad_out,
                                                            int64_t ldg, const float* act_out, int64_t ld_act, int N,
                                                            float* grad_w_l, float* grad_w_r, float* grad_bias, int accumulate,
                                                            void* workspace, size_t workspace_bytes, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sage_wgrad_bf16x3", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && F > 0 && N > 0, "bad sizes");
    if (F % 4 != 0 || F > 256 || N > 256) throw logic_error(fmt("unsupported shape: F=%d (multiple of 4, <= 256), N=%d (<= 256)", F, N));
    WG_REQUIRE_INPUT(grad_w_l && grad_w_r && workspace, "null pointer");
    WG_REQUIRE_INPUT(workspace_bytes >= wgamd_sage_wgrad_workspace_bytes(n_rows, F, N), "workspace too small");
    auto st = static_cast<hipStream_t>(stream);
    if (n_rows == 0) {
      if (!accumulate) {
        WG_HIP_CHECK(hipMemsetAsync(grad_w_l, 0, (size_t)N * F * 4, st));
        WG_HIP_CHECK(hipMemsetAsync(grad_w_r, 0, (size_t)N * F * 4, st));
        if (grad_bias) WG_HIP_CHECK(hipMemsetAsync(grad_bias, 0, (size_t)N * 4, st));
      }
      return;
    }
    WG_REQUIRE_INPUT(agg && x && self_rows && grad_out, "null pointer");
    WG_REQUIRE_INPUT(ld_agg >= F && ldx >= F && ldg >= N && (act_out == nullptr || ld_act >= N), "leading dimension too small");
    if (ld_agg % 4 != 0 || ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(agg) & 15) != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0)
      throw logic_error("agg / x rows must be 16-B aligned");
    const bool byte_offsets = src_ids != nullptr && src_ids_dtype == WGAMD_IDS_BYTE_OFFSETS;
    if (src_ids != nullptr && src_ids_dtype != WHOLEMEMORY_DT_INT && src_ids_dtype != WHOLEMEMORY_DT_INT64 && !byte_offsets)
      throw invalid_input("src_ids must be INT, INT64 or WGAMD_IDS_BYTE_OFFSETS");
    const wgrad_plan p = plan_for(F, N);
    const int cus      = stream_cu_count(st);
    const int64_t tiles = (n_rows + p.TR - 1) / p.TR;
    // every workgroup leaves a [2F + 1, N] partial sum that wgrad_reduce_kernel adds up: with few row tiles (one mini-batch:
    // 338 tiles) a workgroup takes at least three of them.  Measured per launch at F = 100, N = 256, 10 k rows: 256 workgroups
    // 16 us + 31 us of reduction; 42 workgroups 47 + 8; the model a + b tiles / grid + c grid puts the optimum near 110.
    const int64_t tiles_x = std::max<int64_t>(1, tiles / 3);
    const int grid_x   = (int)std::max<int64_t>(1, std::min<int64_t>({tiles_x, (int64_t)std::max(1, cus / p.grid_y), (int64_t)max_grid_x(p)}));
    char* ws           = static_cast<char*>(workspace);
    ws                 = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    float* part        = reinterpret_cast<float*>(ws);
    const int64_t* self_global = self_rows;
    if (src_ids != nullptr) {
      int64_t* composed = reinterpret_cast<int64_t*>(ws + ((part_bytes(p, max_grid_x(p)) + 255) & ~(size_t)255));
      compose_self_kernel<<<(int)std::min<int64_t>((n_rows + 255) / 256, 4096), 256, 0, st>>>(
        self_rows, src_ids, byte_offsets ? 3 : (src_ids_dtype == WHOLEMEMORY_DT_INT ? 1 : 2), n_rows, composed);
      WG_HIP_CHECK(hipGetLastError());
      self_global = composed;
    }
    wgrad_args a{agg, ld_agg, x, byte_offsets ? (int64_t)1 : ldx, F, self_global, n_rows, grad_out, ldg, act_out, ld_act, N, part,
                 (tiles + grid_x - 1) / grid_x * p.TR, p.KL};
    const dim3 grid(grid_x, p.grid_y);
    if (p.TR == 64) launch_mt<64>(p, a, grid, st);
    else if (p.TR == 32) launch_mt<32>(p, a, grid, st);
    else launch_mt<16>(p, a, grid, st);
    const int64_t total = (int64_t)(2 * F + 1) * N;
    wgrad_reduce_kernel<<<(int)((total + 255) / 256), 256, 0, st>>>(part, grid_x, p.KL, p.NB, F, N, grad_w_l, grad_w_r, grad_bias,
                                                                     accumulate);
    WG_HIP_CHECK(hipGetLastError());
  });
}
