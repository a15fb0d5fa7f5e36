// One-kernel SAGEConv layer, WEIGHT-STATIONARY variant for the BASELINE layer-1 shapes (F = 100 -> 256; semantics of
// torch_geometric.nn.SAGEConv as python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59 uses it).
//
// Why: the round-3 ablations of wg_sage_mfma.hip priced that kernel's multiplying waves at exactly their
// MEMORY instructions — per 64-row tile every CU re-streams the whole [2F x 256] weight (205 KB of L2 hits) through the same
// in-order vector-memory pipeline as its 262 KB of HBM row fetches: no weight loads 0.422 ms against 0.516 as built.  A wave
// of a 512-thread workgroup has 256 registers, not enough to keep its 64-column slice of the weight (2F x 64 fp32 = 200+
// registers per lane) next to 64 accumulators and the fragment pipeline.  Here the workgroup is FOUR waves, one per SIMD,
// each with the SIMD's whole 512-entry register file (amdgpu_waves_per_eu(1, 1)): the slice is loaded ONCE per launch and
// stays in registers (the unified VGPR / AGPR file: the compiler parks it in the AGPR half), so the weight stream is gone
// from the memory pipeline.  With one wave per SIMD there are no separate fetching and multiplying waves: every wave does
// both, interleaved in ONE instruction stream — the row loads are asynchronous, their latency passes under the wave's own
// MFMAs and splits:
//     tile n:  [ fetch + sum 8 destination rows per lane group into LDS buffer n & 1 ]  interleaved, row by row, with
//              [ the 13 k-steps of tile n - 1 from buffer (n - 1) & 1 ],  then the output stores of tile n - 1, one barrier.
// The fetching side is wg_sage_mfma.hip's `producer` unchanged (branch-free 16-B buffer loads in a register ring, metadata
// pipelined across tiles), the multiplying side its six-product bf16x3 step with the same split — results are bit-identical
// to that kernel's (same products, same accumulation order per output element).
#include "wg_sage_mfma_parts.hpp"

namespace wgamd {
namespace {
using namespace sage_mfma;

#ifndef WS_SPREAD
#define WS_SPREAD 12   // MFMA pairs a destination row's kNb + 1 loads are spread over (12 = one k-step)
#endif
template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// split_b with the bf16 mask in an OPAQUE scalar: the weight is loop-invariant, and with the literal mask LICM hoists its
// three planes out of the tile loop (312 more live registers).  Same instructions, same values.
__device__ __forceinline__ void split_b_masked(const braw_t& r, bfrag_t& f, uint32_t mask)
{
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float a  = r.v[ct][i >> 2][i & 3];
      h[i]           = __float_as_uint(a) & mask;
      const float r1 = a - __uint_as_float(h[i]);
      m[i]           = __float_as_uint(r1) & mask;
      l[i]           = __float_as_uint(r1 - __uint_as_float(m[i]));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f.v[ct][0][j] = pack_hi16(h[2 * j], h[2 * j + 1]);
      f.v[ct][1][j] = pack_hi16(m[2 * j], m[2 * j + 1]);
      f.v[ct][2][j] = pack_hi16(l[2 * j], l[2 * j + 1]);
    }
  }
}

// Where the wave keeps k-step ks of its 64-column weight slice (F = 100: 13 k-steps of 16 registers):
//   * k-steps 1, 2, 3, 5, 6, 7, 9: REGISTERS, loaded once per launch;
//   * the last three: the LDS (12 KB per wave — what the two operand tiles leave free), read back with ds_read_b128;
//   * k-steps 0, 4, 8: STREAMED through ONE 16-register slot, requested at the end of destination row 6 / 0 / 2 and
//     multiplied two rows later.  A wave's loads return in issue order, so a load may only be waited for where everything
//     older has been waited for anyway: the slot's load follows the request of row it + 2 and is consumed after row it + 2 has
//     been summed — it costs no wait of its own (a register SPILL, in contrast, is reloaded right behind freshly requested
//     rows and drains the whole queue: the first version of this kernel, 26 spilled registers, was compute-bound at 2.25 ms).
//     48 KB of L2 hits per tile and CU instead of the 205 KB of the producer / consumer kernel.
template <int FC>
struct ws_consumer {
  static constexpr int RT = 2, KSC = (2 * FC + 15) / 16, SD = row_stride_dw(FC);
  static constexpr int KL = 3, KR = 7;
  static_assert(KSC == 13, "k-step placement is laid out for 2F = 200");
  __host__ __device__ static constexpr bool streamed(int ks) { return ks < KSC - KL && ks % 4 == 0; }   // 0, 4, 8
  __host__ __device__ static constexpr bool in_lds(int ks) { return ks >= KSC - KL; }
  __host__ __device__ static constexpr int reg_index(int ks) { return ks - 1 - (ks > 4) - (ks > 8); }
  braw_t w[KR];
  braw_t ws;           // the streamed slot
  const float* w_lds;  // [k-step - (KSC - KL)][col tile][half] x 64 lanes x 16 B, lane-linear
  const char* w_glb;   // this lane's fragment of k-step 0 in the tiled weight
  int w_step;          // bytes between k-steps
  f32x16 c[RT][2];
  araw_t<RT> raw;
  afrag_t<RT> fa[2];
  const float* a_lane;
  uint32_t mask;

  __device__ __forceinline__ void load_frag(braw_t& f, int ks) const
  {
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const char* p = w_glb + (size_t)ks * w_step + ct * (32 * 64);
      f.v[ct][0]    = *reinterpret_cast<const f32x4*>(p);
      f.v[ct][1]    = *reinterpret_cast<const f32x4*>(p + 16);
    }
  }
  __device__ __forceinline__ void load_weights(const mfma_args& a, int cw, int lane, float* w_lds_wave)
  {
    w_glb  = reinterpret_cast<const char*>(a.w_tiles) + ((cw * 64 + (lane & 31)) * 16 + (lane >> 5) * 8) * 4;
    w_step = a.N * 64;
    w_lds  = w_lds_wave + lane * 4;
#pragma unroll
    for (int ks = 0; ks < KSC; ks++) {
      if (streamed(ks)) continue;
      braw_t f;
      load_frag(f, ks);
      if (!in_lds(ks)) {
        w[reg_index(ks)] = f;
      } else {
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
          for (int h = 0; h < 2; h++)
            *reinterpret_cast<f32x4*>(const_cast<float*>(w_lds) + (((ks - (KSC - KL)) * 2 + ct) * 2 + h) * 256) = f.v[ct][h];
      }
    }
  }
  template <int KS>
  __device__ __forceinline__ void load_stream()
  {
    static_assert(streamed(KS), "not a streamed k-step");
    load_frag(ws, KS);
  }
  __device__ __forceinline__ void begin(const float* tile_lds, int lane)
  {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int i = 0; i < 16; i++) c[rt][ct][i] = 0.f;
    a_lane = tile_lds + (lane & 31) * SD + (lane >> 5) * 8;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(mask));   // (once per tile: see split_b_masked)
    load_a_raw<RT>(raw, a_lane, SD, 0);
    split_a<RT>(raw, fa[0]);
  }
  // one k-step; vm(slot) is called after every second MFMA (12 slots): the caller's row loads go out one at a time between
  // the MFMAs — a burst of eleven into a saturated memory pipeline blocks the (only, in-order) wave of the SIMD at the load
  // instructions for as long as the pipeline takes to accept them, and nothing multiplies meanwhile
  template <int KS, typename VM>
  __device__ __forceinline__ void step(VM&& vm)
  {
    if constexpr (KS + 1 < KSC) load_a_raw<RT>(raw, a_lane, SD, KS + 1);
    bfrag_t fb;
    if constexpr (streamed(KS)) {
      split_b_masked(ws, fb, mask);
    } else if constexpr (!in_lds(KS)) {
      split_b_masked(w[reg_index(KS)], fb, mask);
    } else {
      braw_t wl;
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int h = 0; h < 2; h++)
          wl.v[ct][h] = *reinterpret_cast<const f32x4*>(w_lds + (((KS - (KSC - KL)) * 2 + ct) * 2 + h) * 256);
      split_b_masked(wl, fb, mask);
    }
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0};   // (mma_frags' order: smallest terms first)
    constexpr int pb[6] = {0, 2, 1, 0, 1, 0};
    static_for<0, 6 * RT>([&](auto S) {
      constexpr int t = decltype(S)::value / RT, rt = decltype(S)::value % RT;
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
        c[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[KS & 1].v[rt][pa[t]]),
                                                            __builtin_bit_cast(bf16x8, fb.v[ct][pb[t]]), c[rt][ct], 0, 0, 0);
      vm(S);
    });
    if constexpr (KS + 1 < KSC) split_a<RT>(raw, fa[(KS + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);   // keep the fragment pipeline as written (hoisted reads / splits cost registers)
  }
  template <int KS>
  __device__ __forceinline__ void step()
  {
    step<KS>([](auto) {});
  }
};

// k-steps of tile n - 1 multiplied after destination row `it` of tile n has been summed (KSC k-steps over IT rows)
// — two per row for rows 0 .. 4, three for row 5; rows 6 and 7 carry the output stores of the finished tile, so that the
// stores, too, leave between row loads instead of as one burst of sixteen at the end
__host__ __device__ constexpr int ks_cut(int it, int IT, int KSC) { return it >= IT - 2 ? KSC : (2 * it < KSC ? 2 * it : KSC); }

template <typename IdT, bool OFF32, int FC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
sage_layer_ws_kernel(mfma_args a)
{
  constexpr int TR = 64, LG = 32;
  using C = ws_consumer<FC>;
  constexpr int SD = C::SD, KSC = C::KSC, tile_dw = TR * SD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 2 * tile_dw + 16; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t n_tiles = (a.n_rows + TR - 1) / TR;
  const int64_t mine    = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  if (mine == 0) return;
  auto tile_of = [&](int64_t n) { return (int64_t)blockIdx.x + n * gridDim.x; };

  using P     = producer<IdT, LG, TR, OFF32>;
  using off_t = typename P::off_t;
  constexpr int IT = P::IT, kNb = P::kNb, kDepth = P::kDepth, kHalf = IT / 2;
  P p(a, wave, lane);
  C cons;
  float* scratch = lds + 2 * tile_dw + 16 + wave * kScratchDw;
  cons.load_weights(a, wave, lane, lds + 2 * tile_dw + 16 + 4 * kScratchDw + wave * (C::KL * 16 * 64));

  bounds_t<IT> b_next;
  ids_t<IT> i_next;
  meta_t<IT, off_t> cur, nxt;
  f32x4 buf[2][kNb + 1];
  static_assert(kDepth == 2 && IT >= 6, "two destination rows in flight per lane group");
  p.load_bounds(tile_of(0), b_next);
  p.load_ids(tile_of(0), b_next, i_next);
  p.finish(i_next, cur);
  p.issue(cur, 0, buf[0]);
  p.issue(cur, 1, buf[1]);

  // One wave is in order: a row's loads must be REQUESTED as early as possible and WAITED FOR as late as possible.  Row `it`
  // is summed, the ring slot it frees takes row it + 2 at once (the last two rows of a tile: rows 0 and 1 of the next tile,
  // whose metadata is therefore complete three rows before the tile ends), and only then the wave turns to its k-steps — two
  // rows per lane group stay in flight under them (with "request it + 1, sum it, multiply" the steady state is one row in
  // flight and half of every multiply is exposed: 2.20 ms against 1.94 for the producer / consumer kernel).
  auto fetch_row = [&](auto I, int64_t n, float* tile_lds) {
    constexpr int it = decltype(I)::value;
    if constexpr (it == 0) p.load_bounds(tile_of(n + 1), b_next);
    if constexpr (it == 2) p.load_ids(tile_of(n + 1), b_next, i_next);
    if constexpr (it == IT - 3) p.finish(i_next, nxt);
    p.reduce_store(cur, it, buf[it & 1], tile_lds);
    if constexpr (it + 2 < IT) p.issue(cur, it + 2, buf[it & 1]);
    else p.issue(nxt, it + 2 - IT, buf[it & 1]);   // (past the last tile: clamped / zero-length rows, never summed)
    __builtin_amdgcn_sched_barrier(0);
  };

  // the streamed k-step requested after row `it` (multiplied two rows later)
  auto stream_after = [&](auto I) {
    constexpr int it = decltype(I)::value;
    if constexpr (it == 6 || it == 0 || it == 2) {
      cons.template load_stream<it == 6 ? 0 : (it == 0 ? 4 : 8)>();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  static_assert(IT == 8 && ks_cut(0, IT, KSC) == 0 && ks_cut(2, IT, KSC) == 4 && ks_cut(4, IT, KSC) == 8 && ks_cut(6, IT, KSC) == KSC,
                "a streamed k-step is multiplied two destination rows after its request");

  // tile 0: nothing to multiply yet
  static_for<0, IT>([&](auto I) {
    fetch_row(I, 0, lds);
    if constexpr (decltype(I)::value == 6) stream_after(I);
  });
  p.template long_rows_from<kNb, 4>(tile_of(0), cur, lds);
  cur = nxt;
  lds_barrier();
  // steady state: rows of tile n come back under the k-steps of tile n - 1
  for (int64_t n = 1; n < mine; n++) {
    float* tile_lds = lds + (n & 1) * tile_dw;
    cons.begin(lds + ((n - 1) & 1) * tile_dw, lane);
    epilogue_lane_t ep;
    static_for<0, IT>([&](auto I) {
      constexpr int it = decltype(I)::value;
      constexpr int k0 = ks_cut(it, IT, KSC), k1 = ks_cut(it + 1, IT, KSC), slots = 12 * (k1 - k0);
      if constexpr (it == 0) p.load_bounds(tile_of(n + 1), b_next);
      if constexpr (it == 2) p.load_ids(tile_of(n + 1), b_next, i_next);
      if constexpr (it == IT - 3) p.finish(i_next, nxt);
      p.reduce_store(cur, it, buf[it & 1], tile_lds);
      __builtin_amdgcn_sched_barrier(0);
      // row it + 2 takes the freed slot, its kNb + 1 loads one at a time between the work of this row
      auto load_j = [&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (it + 2 < IT) p.template issue_one<j>(cur, it + 2, buf[it & 1]);
        else p.template issue_one<j>(nxt, it + 2 - IT, buf[it & 1]);   // (past the last tile: clamped / zero-length rows)
      };
      if constexpr (it < IT - 2) {
        static_assert(slots >= kNb + 1, "a destination row's loads fit between the MFMAs of its k-steps");
        static_for<k0, k1>([&](auto K) {   // load j after MFMA pair j * slots / 11
          constexpr int ks = decltype(K)::value;
          cons.template step<ks>([&](auto S) {
            constexpr int g = (ks - k0) * 12 + decltype(S)::value;
            static_for<0, kNb + 1>([&](auto J) {
              if constexpr (decltype(J)::value * WS_SPREAD / (kNb + 1) == g) load_j(J);
            });
          });
        });
      } else {
        // four of the eight output pieces of tile n - 1, three row loads behind each
        if constexpr (it == IT - 2) epilogue_begin<C::RT>(a, ep, tile_of(n - 1) * TR, wave, lane);
        static_for<0, 4>([&](auto Q) {
          constexpr int q = decltype(Q)::value, piece = (it - (IT - 2)) * 4 + q;
          epilogue_piece<C::RT, piece / 4, piece % 4>(a, ep, cons.c, lane, scratch);
          static_for<3 * q, (3 * q + 3 < kNb + 1 ? 3 * q + 3 : kNb + 1)>(load_j);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      stream_after(I);
    });
    p.template long_rows_from<kNb, 4>(tile_of(n), cur, tile_lds);
    cur = nxt;
    lds_barrier();
  }
  // the last tile: nothing left to fetch
  cons.begin(lds + ((mine - 1) & 1) * tile_dw, lane);
  static_for<0, KSC>([&](auto K) {
    constexpr int ks = decltype(K)::value;
    if constexpr (ks > 0 && C::streamed(ks)) cons.template load_stream<ks>();   // (k-step 0 was requested by the last tile's row 6)
    cons.template step<ks>();
  });
  epilogue<C::RT>(a, cons.c, tile_of(mine - 1) * TR, wave, lane, scratch);
}

template <typename IdT, int FC>
void launch_ws(const mfma_args& a, hipStream_t st)
{
  const int cus         = stream_cu_count(st);
  const int64_t n_tiles = (a.n_rows + 63) / 64;
  const size_t lds      = (size_t)(2 * 64 * row_stride_dw(FC) + 16 + 4 * kScratchDw + 4 * ws_consumer<FC>::KL * 16 * 64) * 4;
  const int grid        = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)cus));   // ONE workgroup per CU
  auto go               = [&](auto kern) {
    WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<grid, 256, lds, st>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  };
  if (a.x_bytes != 0) go(sage_layer_ws_kernel<IdT, true, FC>);
#ifndef WS_ONE_INST
  else go(sage_layer_ws_kernel<IdT, false, FC>);
#endif
}

}  // namespace

namespace sage_mfma {

// WGAMD_SAGE_WS=1 selects this kernel for its shape (measured in round 4: bit-identical to the producer / consumer kernel and
// 2-3 % SLOWER — DESIGN.md §3.5 — so it is opt-in)
bool sage_ws_supported(int F, int N)
{
  static const bool on = [] { const char* e = getenv("WGAMD_SAGE_WS"); return e && e[0] == '1'; }();
  return on && F == 100 && N == 256;
}

// id_kind: 0 = no indirection, 1 = int32 src_ids, 2 = int64 src_ids
void sage_ws_launch(const mfma_args& a, int id_kind, hipStream_t st)
{
  if (id_kind == 0) launch_ws<void, 100>(a, st);
#ifndef WS_ONE_INST
  else if (id_kind == 1) launch_ws<int32_t, 100>(a, st);
  else launch_ws<int64_t, 100>(a, st);
#endif
}

}  // namespace sage_mfma
}  // namespace wgamd
