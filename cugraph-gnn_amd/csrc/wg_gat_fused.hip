// One kernel for a relation of an aggregate-first GATConv layer over a sampled hop (H = 4 heads, F = 128 -> C = 64 per head):
//     agg[i, h, :] = sum_{e in row i} alpha_e^h x[col[e], :],   alpha^h = softmax_e( leaky_relu(a_src[col[e], h] + a_dst[r(i), h]) )
//     out[o(i), h C + c] = act( agg[i, h, :] @ W[:, h C + c]  (+ acc_in[i, h C + c])  (+ bias[h C + c]) )
// = wgamd_gat_aggregate_heads_f32 followed by wgamd_gat_transform_heads_bf16x3 WITHOUT the [n_rows, H F] aggregate (2 KB per
// row) ever leaving the CU: 4.9 GB written and 4.9 GB read back per call group of the ogbn-mag-like workload, a third of the
// two kernels' traffic.  (Semantics: torch_geometric's HeteroConv{GATConv} as the reference's examples build it,
// examples/mag_lp_mnmg.py:141, python/pylibwholegraph/.../torch/gnn_model.py:45-59; aggregate-first identity: DESIGN.md §3.5.)
//
// Structure = the one-kernel SAGE layer's (wg_sage_mfma.hip): one workgroup per CU, persistent over 32-row tiles, two fp32
// [32 x 4 x 128] operand tiles in LDS (2 x 66 KB), one barrier per tile.
//   * 4 FETCHING waves, two 32-lane groups each, a destination row per group and step: the row's <= 10 neighbour rows (fan-out
//     10: the deep hop of a [25, 10] walk) sit in a two-row register ring, 16 B per lane; the per-row metadata runs ahead as a
//     ROW-granular software pipeline over the group's row sequence q — CSR bounds for q + 8, neighbour / a_dst ids for q + 6,
//     attention terms and byte offsets for q + 4, the row loads for q + 2, the softmax-weighted sum of row q — every value is
//     consumed two steps after its load was issued, i.e. where everything older has been waited for anyway (a wave's loads
//     return in issue order).  The softmax is the plain two-pass one over the <= 10 register-resident neighbours (scores
//     broadcast with v_readlane, no LDS round trip); rows longer than the window continue ONLINE, one neighbour at a time
//     (correct, slow: the caller routes hops with a larger fan-out to the two-kernel path).
//   * 4 MULTIPLYING waves, one per head: wave h multiplies columns [128 h, 128 h + 128) of the tile by its [128 x 64] weight
//     slice, held in registers for the whole launch (wg_gat_transform.hip's stationary form, the same exact 3-way bf16 split),
//     and stores through the same epilogue (running HeteroConv sum, bias, ReLU, row placement).
#include "wg_sage_mfma_parts.hpp"

namespace wgamd {
namespace {
using namespace sage_mfma;

struct gf_args {
  const int* row_ptr;
  const int* col;
  int64_t n_rows;
  const float* x;
  int64_t ldx;
  const int64_t* src_ids;    // nullable: neighbour j's row of x is src_ids[j] (fetch in the layer: x = the feature table,
                             // src_ids = the call group's node list; a_src stays indexed by j unless terms_by_id & 1)
  const int64_t* dst_ids;    // node list of the destination type (terms_by_id & 2)
  int terms_by_id;           // bit 0: a_src holds the terms of the TABLE's rows, row src_ids[j]; bit 1: a_dst likewise, row
                             // dst_ids[dst_rows ? dst_rows[i] : i] — the per-list terms are then never made
  const float* a_src;        // [n_src, 4]
  const float* a_dst;        // [*, 4], row dst_rows ? dst_rows[i] : i
  const int64_t* dst_rows;   // nullable
  float slope;
  const float* w_tiles;      // wgamd_gat_transform_weight_tiles(W [128, 256])
  const float* acc_in;       // nullable
  int64_t ld_acc;
  const float* bias;         // nullable
  int relu;
  const int64_t* out_rows;   // nullable
  float* out;
  int64_t ldo;
};

constexpr int kNbG   = 10;    // neighbour rows of a destination row held in registers
constexpr int kF     = 128, kH = 4, kKS = kF / 16;
constexpr int kSDA   = kH * kF + 4;   // floats per LDS tile row: 4 * odd -> conflict-free ds_read_b128 across rows
constexpr int kTileDw = 32 * kSDA;

template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct st_a { int s, e, valid; };
struct st_b { int deg, s, colk; int64_t dst; };
struct st_c { int deg, s; f32x4 asrc, adst; int64_t off, dsti; };   // off: the neighbour's row of x (scaled to bytes at issue)

__device__ __forceinline__ void split_b_opaque2(const braw_t& r, bfrag_t& f, uint32_t mask)
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

template <bool IDS>
__global__ void __launch_bounds__(512) gat_layer_fused_kernel(gf_args a)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][32][kSDA] + [4][kScratchDw]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_tiles = (a.n_rows + 31) / 32;
  const int64_t mine    = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  if (mine == 0) return;
  auto tile_of = [&](int64_t n) { return (int64_t)blockIdx.x + n * gridDim.x; };

  if (wave >= 4) {
    // ------------------------------------------------------------------ fetching waves ----------------------------------
    const int sub = lane & 31, gbase = lane & 32, group = (wave - 4) * 2 + (lane >> 5), f0 = sub * 4;
    const char* xb = reinterpret_cast<const char*>(a.x);
    st_a ra[2];
    st_b rb[2];
    st_c rc[4];
    f32x4 v[2][kNbG];
    // lane k of this lane's 16-lane row (DPP row_newbcast: one instruction, folded into its consumer where the ISA allows)
#define WG_ROW_BCAST(val, k) __builtin_amdgcn_update_dpp(0, (val), 0x150 + (k), 0xf, 0xf, false)
    auto bcast = [&](int val, int k) __attribute__((always_inline)) {
      const int lo = __builtin_amdgcn_readlane(val, k), hi = __builtin_amdgcn_readlane(val, 32 + k);
      return gbase ? hi : lo;
    };
    // (n, it) + d rows ahead in the group's sequence -> global row, or -1 past the block's last tile / the last row
    auto row_at = [&](int64_t n, int it, int d) __attribute__((always_inline)) -> int64_t {
      const int itd    = it + d;
      const int64_t nn = n + (itd >> 2);
      if (nn < 0 || nn >= mine) return -1;
      const int64_t row = tile_of(nn) * 32 + group + (itd & 3) * 8;
      return row < a.n_rows ? row : -1;
    };
    auto stage_a = [&](st_a& o, int64_t row) __attribute__((always_inline)) {
      const int64_t rc_ = row >= 0 ? row : 0;
      o.s     = a.row_ptr[rc_];
      o.e     = a.row_ptr[rc_ + 1];
      o.valid = row >= 0;
    };
    auto stage_b = [&](st_b& o, const st_a& i, int64_t row) __attribute__((always_inline)) {
      const int64_t rc_ = row >= 0 ? row : 0;
      o.deg  = i.valid ? i.e - i.s : -1;
      o.s    = i.s;
      o.colk = a.col[(sub & 15) < o.deg ? i.s + (sub & 15) : 0];   // neighbour k in lane k of BOTH 16-lane rows of the group
      o.dst  = a.dst_rows ? a.dst_rows[rc_] : rc_;
    };
    auto stage_c = [&](st_c& o, const st_b& i) __attribute__((always_inline)) {
      o.deg  = i.deg;
      o.s    = i.s;
      if constexpr (IDS) {
        o.off = a.src_ids[i.colk];   // (one more load of the stage: consumed at `issue`, two steps on)
        if (!(a.terms_by_id & 1)) o.asrc = *reinterpret_cast<const f32x4*>(a.a_src + (int64_t)i.colk * 4);
        if (a.terms_by_id & 2) o.dsti = a.dst_ids[i.dst];
        else o.adst = *reinterpret_cast<const f32x4*>(a.a_dst + i.dst * 4);
      } else {
        o.off  = (int64_t)i.colk;
        o.asrc = *reinterpret_cast<const f32x4*>(a.a_src + (int64_t)i.colk * 4);
        o.adst = *reinterpret_cast<const f32x4*>(a.a_dst + i.dst * 4);
      }
    };
    auto issue = [&](st_c& m, f32x4* vv) __attribute__((always_inline)) {
      if constexpr (IDS) {   // terms of the table's rows: the ids arrived with stage c, the terms are consumed at `reduce`, two steps on
        if (a.terms_by_id & 1) m.asrc = *reinterpret_cast<const f32x4*>(a.a_src + m.off * 4);
        if (a.terms_by_id & 2) m.adst = *reinterpret_cast<const f32x4*>(a.a_dst + m.dsti * 4);
      }
      const int64_t boff = m.off * a.ldx * 4;
      static_for<0, kNbG>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const int lo = WG_ROW_BCAST((int)(boff & 0xffffffff), k), hi = WG_ROW_BCAST((int)(boff >> 32), k);
        int64_t off  = ((int64_t)hi << 32) | (uint32_t)lo;
        off          = k < m.deg ? off : (int64_t)0;   // slots past the degree read row 0 (cache-resident), weight 0 below
        vv[k]        = *reinterpret_cast<const f32x4*>(xb + off + f0 * 4);
      });
    };
    auto reduce = [&](const st_c& m, const f32x4* vv, float* row_lds) __attribute__((always_inline)) {
      const int deg = m.deg;
      float mx[kH], den[kH];
      f32x4 acc[kH];
      // lane k < kNbG of the group owns neighbour k: its four head scores, the row maximum and the sum of the exponentials by a
      // DPP butterfly over the 16-lane row (lanes past the window / the degree carry -inf / 0), then the ten weights of a head
      // are broadcast (v_readlane) for the weighted sum — 40 exponentials per row instead of 40 per LANE
      float sl[kH], pl[kH];
      const bool mine_k = (sub & 15) < kNbG && (sub & 15) < deg;
#pragma unroll
      for (int h = 0; h < kH; h++) {
        float t = m.asrc[h] + m.adst[h];
        t       = t > 0.f ? t : t * a.slope;
        sl[h]   = mine_k ? t : -INFINITY;
      }
      auto row_max = [](float v_) __attribute__((always_inline)) {
        v_ = fmaxf(v_, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0xB1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
        v_ = fmaxf(v_, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0x4E, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
        v_ = fmaxf(v_, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0x141, 0xf, 0xf, false)));   // row_half_mirror
        v_ = fmaxf(v_, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0x140, 0xf, 0xf, false)));   // row_mirror
        return v_;
      };
      auto row_sum = [](float v_) __attribute__((always_inline)) {
        v_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0xB1, 0xf, 0xf, false));
        v_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0x4E, 0xf, 0xf, false));
        v_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0x141, 0xf, 0xf, false));
        v_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), 0x140, 0xf, 0xf, false));
        return v_;
      };
#pragma unroll
      for (int h = 0; h < kH; h++) {
        const float mh = row_max(sl[h]);                       // (lanes 0 .. 15 of the group hold it; lane 0 is read below)
        pl[h]          = mine_k ? __expf(sl[h] - mh) : 0.f;
        const float dh = row_sum(pl[h]);
        mx[h]          = mh;      // (every lane of the row holds the row's value)
        den[h]         = dh;
        f32x4 ah       = {0.f, 0.f, 0.f, 0.f};
        static_for<0, kNbG>([&](auto K) {
          constexpr int k = decltype(K)::value;
          ah += __int_as_float(WG_ROW_BCAST(__float_as_int(pl[h]), k)) * vv[k];
        });
        acc[h] = ah;
      }
      if (__ballot(deg > kNbG) != 0ull) {
        // rows past the register window: the same softmax continued online, one neighbour at a time
        int maxdeg = deg;
#pragma unroll
        for (int dd = 16; dd >= 1; dd >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, dd, 64));
        maxdeg = max(maxdeg, __shfl_xor(maxdeg, 32, 64));
        for (int k = kNbG; k < maxdeg; k++) {
          const bool on   = k < deg;
          const int idx   = a.col[on ? m.s + k : 0];
          int64_t xrow    = idx;
          if constexpr (IDS) xrow = a.src_ids[idx];
          const f32x4 a4  = *reinterpret_cast<const f32x4*>(a.a_src + ((IDS && (a.terms_by_id & 1)) ? xrow : (int64_t)idx) * 4);
          const f32x4 xv  = *reinterpret_cast<const f32x4*>(xb + xrow * a.ldx * 4 + f0 * 4);
#pragma unroll
          for (int h = 0; h < kH; h++) {
            float t        = a4[h] + m.adst[h];
            t              = t > 0.f ? t : t * a.slope;
            const float s_ = on ? t : -INFINITY;
            const float mn = fmaxf(mx[h], s_);
            const float r  = on ? __expf(mx[h] - mn) : 1.f, p = on ? __expf(s_ - mn) : 0.f;   // (off: the row's state stays)
            acc[h]         = acc[h] * r + p * xv;
            den[h]         = den[h] * r + p;
            mx[h]          = mn;
          }
        }
      }
#pragma unroll
      for (int h = 0; h < kH; h++) {
        const float inv = deg > 0 ? 1.0f / den[h] : 0.f;
        *reinterpret_cast<f32x4*>(row_lds + h * kF + f0) = acc[h] * inv;
      }
    };
    // one step of the row pipeline: (n, it) is row q of the group's sequence
    auto step = [&](auto IT_, auto DO_B, auto DO_C, auto DO_ISSUE, auto DO_REDUCE, int64_t n) __attribute__((always_inline)) {
      constexpr int it = decltype(IT_)::value;
      if constexpr (decltype(DO_REDUCE)::value) reduce(rc[it & 3], v[it & 1], lds + (n & 1) * kTileDw + (group + it * 8) * kSDA);
      if constexpr (decltype(DO_ISSUE)::value) issue(rc[(it + 2) & 3], v[it & 1]);
      if constexpr (decltype(DO_C)::value) stage_c(rc[it & 3], rb[it & 1]);
      if constexpr (decltype(DO_B)::value) stage_b(rb[it & 1], ra[it & 1], row_at(n, it, 6));
      stage_a(ra[it & 1], row_at(n, it, 8));
      __builtin_amdgcn_sched_barrier(0);
    };
    using T = std::true_type;
    using N_ = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // fill: virtual rows q = -8 .. -1
    step(I0{}, N_{}, N_{}, N_{}, N_{}, -2);
    step(I1{}, N_{}, N_{}, N_{}, N_{}, -2);
    step(I2{}, T{}, N_{}, N_{}, N_{}, -2);
    step(I3{}, T{}, N_{}, N_{}, N_{}, -2);
    step(I0{}, T{}, T{}, N_{}, N_{}, -1);
    step(I1{}, T{}, T{}, N_{}, N_{}, -1);
    step(I2{}, T{}, T{}, T{}, N_{}, -1);
    step(I3{}, T{}, T{}, T{}, N_{}, -1);
    for (int64_t n = 0; n <= mine; n++) {
      if (n < mine) {
        step(I0{}, T{}, T{}, T{}, T{}, n);
        step(I1{}, T{}, T{}, T{}, T{}, n);
        step(I2{}, T{}, T{}, T{}, T{}, n);
        step(I3{}, T{}, T{}, T{}, T{}, n);
      }
      lds_barrier();
    }
  } else {
    // ------------------------------------------------------------------ multiplying waves: wave = head --------------------
    const int lm = lane & 31, lh = lane >> 5, rl = lane >> 4, cl = (lane & 15) * 4;
    const int hd = wave;
    float* scratch = lds + 2 * kTileDw + wave * kScratchDw;
    const int N    = kH * 64;
    braw_t wst[kKS];
    {
      const float* wp = a.w_tiles + ((int64_t)(hd * 64 + lm)) * 16 + lh * 8;
#pragma unroll
      for (int ks = 0; ks < kKS; ks++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const float* p  = wp + ((int64_t)ks * N + ct * 32) * 16;
          wst[ks].v[ct][0] = *reinterpret_cast<const f32x4*>(p);
          wst[ks].v[ct][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
    }
    const int colb = hd * 64 + cl;
    f32x4 b4       = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + colb);
    for (int64_t n = 0; n <= mine; n++) {
      if (n >= 1) {
        const float* a_lane = lds + ((n - 1) & 1) * kTileDw + lm * kSDA + hd * kF + lh * 4;   // (k-order of the weight tiles)
        uint32_t mask;
        asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(mask));
        f32x16 c[1][2];
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
          for (int i = 0; i < 16; i++) c[0][ct][i] = 0.f;
        araw_t<1> raw[2];
        raw[0].v[0][0] = *reinterpret_cast<const f32x4*>(a_lane);
        raw[0].v[0][1] = *reinterpret_cast<const f32x4*>(a_lane + 8);
#pragma unroll
        for (int ks = 0; ks < kKS; ks++) {
          if (ks + 1 < kKS) {
            raw[(ks + 1) & 1].v[0][0] = *reinterpret_cast<const f32x4*>(a_lane + (ks + 1) * 16);
            raw[(ks + 1) & 1].v[0][1] = *reinterpret_cast<const f32x4*>(a_lane + (ks + 1) * 16 + 8);
          }
          afrag_t<1> fa;
          bfrag_t fb;
          split_a<1>(raw[ks & 1], fa);
          split_b_opaque2(wst[ks], fb, mask);
          mma_frags<1>(c, fa, fb);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int64_t row0 = tile_of(n - 1) * 32;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          f32x4 prev[2];
          int64_t orow[2];
#pragma unroll
          for (int pass = 0; pass < 2; pass++) {
            const int64_t row  = row0 + 8 * g + 4 * pass + rl;
            const int64_t rowc = row < a.n_rows ? row : a.n_rows - 1;
            prev[pass] = a.acc_in ? *reinterpret_cast<const f32x4*>(a.acc_in + rowc * a.ld_acc + colb) : f32x4{0.f, 0.f, 0.f, 0.f};
            orow[pass] = a.out_rows ? a.out_rows[rowc] : rowc;
          }
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) scratch[(jj + 4 * lh) * 64 + ct * 32 + lm] = c[0][ct][4 * g + jj];
#pragma unroll
          for (int pass = 0; pass < 2; pass++) {
            f32x4 vv = *reinterpret_cast<const f32x4*>(scratch + (rl + 4 * pass) * 64 + cl);
            vv       = vv + prev[pass] + b4;
            if (a.relu) {
#pragma unroll
              for (int i = 0; i < 4; i++) vv[i] = fmaxf(vv[i], 0.f);
            }
            if (row0 + 8 * g + 4 * pass + rl < a.n_rows) *reinterpret_cast<f32x4*>(a.out + orow[pass] * a.ldo + colb) = vv;
          }
        }
      }
      lds_barrier();
    }
  }
}

}  // namespace
}  // namespace wgamd

extern "C" int wgamd_gat_layer_fused_supported(int F, int H, int C) { return F == 128 && H == 4 && C == 64; }

extern "C" wholememory_error_code_t wgamd_gat_layer_fused_ids_bf16x3(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                                 int64_t ldx, const int64_t* src_ids, const int64_t* dst_ids, int terms_by_id,
                                                                 int F, const float* a_src, const float* a_dst, int H,
                                                                 int C, float negative_slope, const int64_t* dst_rows,
                                                                 const void* w_tiles, const float* acc_in, int64_t ld_acc,
                                                                 const float* bias, int relu, const int64_t* out_rows, float* out,
                                                                 int64_t ldo, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gat_layer_fused_ids_bf16x3", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && a_src && a_dst && w_tiles && out, "null pointer");
    if (!wgamd_gat_layer_fused_supported(F, H, C)) throw logic_error(fmt("unsupported shape: F=%d H=%d C=%d (128, 4, 64)", F, H, C));
    WG_REQUIRE_INPUT(ldx >= F && ldo >= (int64_t)H * C && (!acc_in || ld_acc >= (int64_t)H * C), "leading dimension");
    if (ldx % 4 != 0 || ldo % 4 != 0 || (acc_in && ld_acc % 4 != 0) || (reinterpret_cast<uintptr_t>(x) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(out) & 15) != 0 || (reinterpret_cast<uintptr_t>(acc_in) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(bias) & 15) != 0 || (reinterpret_cast<uintptr_t>(a_src) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(a_dst) & 15) != 0)
      throw logic_error("rows, attention terms and the bias must be 16-B aligned");
    WG_REQUIRE_INPUT(terms_by_id >= 0 && terms_by_id <= 3 && (terms_by_id == 0 || src_ids) && (!(terms_by_id & 2) || dst_ids),
                     "terms_by_id needs src_ids (and dst_ids for bit 1)");
    gf_args a{row_ptr, col, n_rows, x, ldx, src_ids, dst_ids, terms_by_id, a_src, a_dst, dst_rows, negative_slope, static_cast<const float*>(w_tiles),
              acc_in, ld_acc, bias, relu, out_rows, out, ldo};
    auto st               = static_cast<hipStream_t>(stream);
    const int cus         = stream_cu_count(st);
    const int64_t n_tiles = (n_rows + 31) / 32;
    const size_t lds      = (size_t)(2 * kTileDw + 4 * kScratchDw) * 4;
    const int grid        = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)cus));
    auto kernel = src_ids ? gat_layer_fused_kernel<true> : gat_layer_fused_kernel<false>;
    WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kernel<<<grid, 512, lds, st>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  });
}

extern "C" wholememory_error_code_t wgamd_gat_layer_fused_bf16x3(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                                 int64_t ldx, int F, const float* a_src, const float* a_dst, int H,
                                                                 int C, float negative_slope, const int64_t* dst_rows,
                                                                 const void* w_tiles, const float* acc_in, int64_t ld_acc,
                                                                 const float* bias, int relu, const int64_t* out_rows, float* out,
                                                                 int64_t ldo, void* stream)
{
  return wgamd_gat_layer_fused_ids_bf16x3(row_ptr, col, n_rows, x, ldx, nullptr, nullptr, 0, F, a_src, a_dst, H, C, negative_slope, dst_rows, w_tiles,
                                          acc_in, ld_acc, bias, relu, out_rows, out, ldo, stream);
}
