// The dense tail of aggregate-first GATConv on the matrix pipe at fp32 accuracy:
//     out[r(i), h C + c] = act( sum_k agg[i, h F + k] W[k, h C + c]  (+ acc_in[i, h C + c])  (+ bias[h C + c]) ),   C = 64
// — the H small GEMMs after wgamd_gat_aggregate_heads_f32, HeteroConv's sum over the relations of a destination type
// (`acc_in`), and the layer's bias + ReLU + row placement (`out_rows`) in ONE pass over the aggregate.  (Reference semantics:
// torch_geometric's HeteroConv{GATConv} as examples/mag_lp_mnmg.py:141 and python/pylibwholegraph/.../torch/gnn_model.py:45-59
// build it; the aggregate-first identity is DESIGN.md §3.5.)
//
// Why not the library: a strided batched fp32 GEMM runs the fp32 MFMA (157 TF/s) and took 2.4 ms + 0.57 ms (bias / ReLU pass)
// per call group of the ogbn-mag-like workload for 157 GFLOP.  Here the product is the exact 3-way bf16 split of
// wg_sage_mfma.hip (six v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate: the class of fp32 round-off) and the
// kernel is bound by its own traffic: the aggregate is read once (2 KB per row at 4 x 128), the output written once.
//
// Structure: no LDS for the operands — every element of `agg` is used by exactly ONE wave (the wave of its head), so a lane
// loads its MFMA A fragment (8 consecutive k of one row = 32 B) straight from global memory, three k-steps ahead, continuing
// across tiles.  A wave owns a head: its [F x 64] weight slice stays in REGISTERS as fp32 fragments for the whole launch when
// F <= 128 (8 k-steps x 16 registers) and is split into its planes next to the A fragment (the literal-mask split would be
// hoisted out of the tile loop and cost 192 registers: the mask lives in an opaque scalar); wider inputs (layer 2: F = 256)
// stream the fragments from L2 beside the A fragments.  32-row tiles (one 32 x 32 accumulator pair), 4 waves per workgroup,
// two workgroups per CU.  Outputs leave through the 2 KiB per-wave LDS transposition of wg_sage_mfma.hip as 16-byte accesses,
// which is also where `acc_in`, bias and ReLU are applied.
#include "wg_sage_mfma_parts.hpp"

namespace wgamd {
namespace {
using namespace sage_mfma;

struct gt_args {
  const float* agg;
  int64_t ld_agg;
  int64_t n_rows;
  int H;
  const float* w_tiles;   // [KS][N = H * 64][16] fp32 (wgamd_gat_transform_weight_tiles)
  const float* acc_in;    // nullable
  int64_t ld_acc;
  const float* bias;      // nullable, [H * 64]
  int relu;
  const int64_t* out_rows;   // nullable
  float* out;
  int64_t ldo;
};

// k-steps of A (and streamed weight) fragments in flight; divides every KS (the ring position of a k-step must not depend on
// the tile: the stream continues across tiles).  The stationary-weight instantiation has 128 registers of weight: with 4 it
// spills, and a spilled 64-bit pointer reloaded inside the loop returned STALE fragments from the second tile of a wave on
// (results wrong from tile gridDim.x on; `s_waitcnt vmcnt(0)` at the loop top or no spill both cure it) — 2 there.
template <bool STAT>
constexpr int ahead() { return STAT ? 2 : 4; }

__device__ __forceinline__ void split_b_opaque(const braw_t& r, bfrag_t& f, uint32_t mask)
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

template <int KS, bool STAT>
__global__ void __launch_bounds__(256, 2) gat_transform_kernel(gt_args a)
{
  constexpr int kAhead = ahead<STAT>();
  static_assert(KS % kAhead == 0, "ring position of a k-step is the same in every tile");
  __shared__ __attribute__((aligned(16))) float scratch_all[4 * kScratchDw];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lm = lane & 31, lh = lane >> 5, rl = lane >> 4, cl = (lane & 15) * 4;
  float* scratch          = scratch_all + wave * kScratchDw;
  constexpr int F         = KS * 16;
  const int N             = a.H * 64;
  const int64_t n_tiles   = (a.n_rows + 31) / 32;
  // work units (tile, head).  STAT: the wave keeps ONE head (H divides 4): wave w of block b walks tiles
  // b (4 / H) + w / H + n gridDim (4 / H); otherwise units are dealt round-robin, head fastest
  int head;
  int64_t first, stride, count;
  if constexpr (STAT) {
    const int per = 4 / a.H;
    head          = wave % a.H;
    first         = (int64_t)blockIdx.x * per + wave / a.H;
    stride        = (int64_t)gridDim.x * per;
    count         = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
  } else {
    head   = 0;
    first  = (int64_t)blockIdx.x * 4 + wave;
    stride = (int64_t)gridDim.x * 4;
    const int64_t units = n_tiles * a.H;
    count  = first < units ? (units - first + stride - 1) / stride : 0;
  }
  if (count == 0) return;
  auto unit = [&](int64_t n, int64_t& tile, int& hd) {
    const int64_t u = first + n * stride;
    if constexpr (STAT) { tile = u; hd = head; }
    else { tile = u / a.H; hd = (int)(u % a.H); }
  };
  auto a_ptr = [&](int64_t tile, int hd) {
    const int64_t row  = tile * 32 + lm;
    const int64_t rowc = row < a.n_rows ? row : a.n_rows - 1;
    return a.agg + rowc * a.ld_agg + hd * F + lh * 4;
  };
  auto w_ptr = [&](int hd) { return a.w_tiles + ((int64_t)(hd * 64 + lm)) * 16 + lh * 8; };
  auto load_a = [&](araw_t<1>& f, const float* ap, int ks) {
    // k-order inside a k-step: lane half lh holds k = 4 lh .. 4 lh + 3 and 8 + 4 lh .. 8 + 4 lh + 3 (the weight tiles are laid
    // out to match), so that one load instruction reads 32 CONTIGUOUS bytes per row instead of two 16-byte pieces
    f.v[0][0] = *reinterpret_cast<const f32x4*>(ap + ks * 16);
    f.v[0][1] = *reinterpret_cast<const f32x4*>(ap + ks * 16 + 8);
  };
  auto load_w = [&](braw_t& f, const float* wp, int ks) {
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const float* p = wp + ((int64_t)ks * N + ct * 32) * 16;
      f.v[ct][0]     = *reinterpret_cast<const f32x4*>(p);
      f.v[ct][1]     = *reinterpret_cast<const f32x4*>(p + 4);
    }
  };

  braw_t wst[STAT ? KS : 1];
  if constexpr (STAT) {
    const float* wp = w_ptr(head);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) load_w(wst[ks], wp, ks);
  }
  araw_t<1> ra[kAhead];
  braw_t rb[STAT ? 1 : kAhead];
  int64_t tile;
  int hd;
  unit(0, tile, hd);
  const float* ap = a_ptr(tile, hd);
  const float* wp = w_ptr(hd);
#pragma unroll
  for (int j = 0; j < kAhead; j++) {
    load_a(ra[j], ap, j);
    if constexpr (!STAT) load_w(rb[j], wp, j);
  }
  for (int64_t n = 0; n < count; n++) {
    int64_t tile_n;
    int hd_n;
    unit(n + 1 < count ? n + 1 : n, tile_n, hd_n);   // (the last unit prefetches itself again: harmless)
    const float* ap_n = a_ptr(tile_n, hd_n);
    const float* wp_n = w_ptr(hd_n);
    uint32_t mask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(mask));
    f32x16 c[1][2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int i = 0; i < 16; i++) c[0][ct][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      afrag_t<1> fa;
      bfrag_t fb;
      split_a<1>(ra[ks % kAhead], fa);
      if constexpr (STAT) split_b_opaque(wst[ks], fb, mask);
      else split_b_opaque(rb[ks % kAhead], fb, mask);
      // the slot just read takes k-step ks + kAhead — of this tile, or of the next one
      if (ks + kAhead < KS) {
        load_a(ra[ks % kAhead], ap, ks + kAhead);
        if constexpr (!STAT) load_w(rb[ks % kAhead], wp, ks + kAhead);
      } else {
        load_a(ra[ks % kAhead], ap_n, ks + kAhead - KS);
        if constexpr (!STAT) load_w(rb[ks % kAhead], wp_n, ks + kAhead - KS);
      }
      mma_frags<1>(c, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: 8 rows at a time through the wave's LDS scratch; (+ acc_in) (+ bias) (ReLU) on the 16-byte side ----
    const int64_t row0 = tile * 32;
    const int colb     = hd * 64 + cl;
    f32x4 b4           = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + colb);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      f32x4 prev[2];
      int64_t orow[2];
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        const int64_t row  = row0 + 8 * g + 4 * pass + rl;
        const int64_t rowc = row < a.n_rows ? row : a.n_rows - 1;
        prev[pass]         = a.acc_in ? *reinterpret_cast<const f32x4*>(a.acc_in + rowc * a.ld_acc + colb) : f32x4{0.f, 0.f, 0.f, 0.f};
        orow[pass]         = a.out_rows ? a.out_rows[rowc] : rowc;
      }
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) scratch[(jj + 4 * lh) * 64 + ct * 32 + lm] = c[0][ct][4 * g + jj];
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (rl + 4 * pass) * 64 + cl);
        v       = v + prev[pass] + b4;
        if (a.relu) {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.f);
        }
        if (row0 + 8 * g + 4 * pass + rl < a.n_rows) *reinterpret_cast<f32x4*>(a.out + orow[pass] * a.ldo + colb) = v;
      }
    }
    tile = tile_n, hd = hd_n, ap = ap_n, wp = wp_n;
  }
}

__global__ void gt_tile_weight_kernel(const float* __restrict__ w, int64_t ldw, int K, int N, int KS, float* __restrict__ tiles)
{
  const int64_t total = (int64_t)KS * N * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 15);   // position in the tile: lane half lh = kk >> 3 reads positions 8 lh .. 8 lh + 7
    const int n  = (int)((i >> 4) % N);
    const int ks = (int)((i >> 4) / N);
    const int j  = kk & 7, lh = kk >> 3;
    const int k  = ks * 16 + (j < 4 ? 4 * lh + j : 8 + 4 * lh + (j - 4));   // (the A fragment's k-order, see load_a)
    tiles[i]     = k < K ? w[(int64_t)k * ldw + n] : 0.f;
  }
}

template <int KS>
void launch_gt(const gt_args& a, hipStream_t st)
{
  const int cus         = stream_cu_count(st);
  const int64_t n_tiles = (a.n_rows + 31) / 32;
  static const bool no_stat = [] { const char* e = getenv("WGAMD_GT_STAT"); return e && e[0] == '0'; }();   // (tuning)
  const bool stat       = !no_stat && KS <= 8 && (a.H == 1 || a.H == 2 || a.H == 4);
  const int64_t blocks  = stat ? (n_tiles + (4 / a.H) - 1) / (4 / a.H) : (n_tiles * a.H + 3) / 4;
  const int grid        = (int)std::max<int64_t>(1, std::min<int64_t>(blocks, 2 * (int64_t)cus));
  if constexpr (KS <= 8) {
    if (stat) {
      gat_transform_kernel<KS, true><<<grid, 256, 0, st>>>(a);
      WG_HIP_CHECK(hipGetLastError());
      return;
    }
  }
  gat_transform_kernel<KS, false><<<grid, 256, 0, st>>>(a);
  WG_HIP_CHECK(hipGetLastError());
}

}  // namespace
}  // namespace wgamd

extern "C" int wgamd_gat_transform_heads_supported(int F, int H, int C)
{
  return C == 64 && H >= 1 && (F == 64 || F == 128 || F == 256);
}

extern "C" size_t wgamd_gat_transform_weight_bytes(int F, int H, int C) { return (size_t)((F + 15) / 16) * (size_t)(H * C) * 64; }

extern "C" wholememory_error_code_t wgamd_gat_transform_weight_tiles(const float* w, int64_t ldw, int F, int H, int C, void* tiles,
                                                                     void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gat_transform_weight_tiles", [&] {
    WG_REQUIRE_INPUT(w && tiles && F > 0 && H > 0 && C > 0 && ldw >= (int64_t)H * C, "bad weight");
    const int KS = (F + 15) / 16, N = H * C;
    const int64_t total = (int64_t)KS * N * 16;
    gt_tile_weight_kernel<<<(int)std::min<int64_t>((total + 255) / 256, 2048), 256, 0, static_cast<hipStream_t>(stream)>>>(
      w, ldw, F, N, KS, static_cast<float*>(tiles));
    WG_HIP_CHECK(hipGetLastError());
  });
}

extern "C" wholememory_error_code_t wgamd_gat_transform_heads_bf16x3(const float* agg, int64_t ld_agg, int64_t n_rows, int F, int H,
                                                                     int C, const void* w_tiles, const float* acc_in,
                                                                     int64_t ld_acc, const float* bias, int relu,
                                                                     const int64_t* out_rows, float* out, int64_t ldo, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gat_transform_heads_bf16x3", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && F > 0 && H > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(agg && w_tiles && out, "null pointer");
    if (!wgamd_gat_transform_heads_supported(F, H, C)) throw logic_error(fmt("unsupported shape: F=%d (64, 128 or 256), C=%d (64)", F, C));
    WG_REQUIRE_INPUT(ld_agg >= (int64_t)H * F && ldo >= (int64_t)H * C && (!acc_in || ld_acc >= (int64_t)H * C), "leading dimension");
    if (ld_agg % 4 != 0 || ldo % 4 != 0 || (acc_in && ld_acc % 4 != 0) || (reinterpret_cast<uintptr_t>(agg) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(out) & 15) != 0 || (reinterpret_cast<uintptr_t>(acc_in) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(bias) & 15) != 0)
      throw logic_error("rows (and the bias) must be 16-B aligned");
    gt_args a{agg, ld_agg, n_rows, H, static_cast<const float*>(w_tiles), acc_in, ld_acc, bias, relu, out_rows, out, ldo};
    auto st = static_cast<hipStream_t>(stream);
    if (F == 64) launch_gt<4>(a, st);
    else if (F == 128) launch_gt<8>(a, st);
    else launch_gt<16>(a, st);
  });
}
