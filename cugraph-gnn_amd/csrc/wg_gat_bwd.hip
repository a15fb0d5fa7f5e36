// Backward of the fused GAT aggregation (wgamd_gat_csr_f32):  out_i = sum_j alpha_ij x_j,  alpha = softmax_j(e_ij),
// e_ij = LeakyReLU(a_src[j] + a_dst[i]) per head.  Given g = dL/dout:
//   dalpha_ij = <g_i, x_j>_head,  ds_ij = alpha_ij (dalpha_ij - sum_k alpha_ik dalpha_ik),  de_ij = ds_ij * LeakyReLU'(.)
//   ga_dst[i] = sum_j de_ij,   ga_src[j] = sum_i de_ij,   gx_j = sum_i alpha_ij g_i
// Two kernels, no atomics, no [E, H, C] temporaries (the torch-op version needs three of them):
//   1. destination-major (the forward's CSR): one lane group per destination row; the per-head dot products are reduced
//      with xor-shuffles inside the head's lanes; writes de[E, H] and ga_dst;
//   2. source-major (the transposed CSR: edge ids sorted by source, stable): gathers g rows weighted by alpha, sums de.
// Semantics of torch_geometric.nn.GATConv's message/aggregate as used by the reference's model zoo
// (python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:45-59); the formulas are the standard softmax backward.
#include <algorithm>

#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

// lanes = lane group per row (power of two, >= H*C/4); LH = C/4 lanes per head (power of two)
__global__ void __launch_bounds__(256)
gat_bwd_dst_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col, int64_t n_rows, const float* __restrict__ x,
                   int64_t ldx, const float* __restrict__ a_src, const float* __restrict__ a_dst, int H, int C, float slope,
                   const float* __restrict__ alpha, const float* __restrict__ g, int64_t ldg, float* __restrict__ de,
                   float* __restrict__ ga_dst, int log2_lanes)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const int HC = H * C, LH = C / 4;
  const int f0    = sub * 4;
  const bool live = f0 < HC;
  const int h     = live ? f0 / C : 0;
  const int t     = sub & (LH - 1);  // my index among the lanes of my head
  const int64_t iters = (n_rows + ngroups - 1) / ngroups;
  for (int64_t it = 0; it < iters; it++) {  // all lanes of a wave iterate together: the shuffles below need them
    const int64_t row = group + it * ngroups;
    const bool valid  = row < n_rows && live;
    int s = 0, e = 0;
    if (row < n_rows) {
      s = row_ptr[row];
      e = row_ptr[row + 1];
    }
    int maxdeg = e - s;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, d, 64));
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) g4 = *reinterpret_cast<const float4*>(g + row * ldg + f0);
    // pass 1: dalpha per edge and head (parked in de), dot = sum alpha * dalpha
    float dot = 0.f;
    for (int j = 0; j < maxdeg; j++) {
      float p = 0.f;
      const bool on = valid && s + j < e;
      if (on) {
        const float4 x4 = *reinterpret_cast<const float4*>(x + (int64_t)col[s + j] * ldx + f0);
        p = g4.x * x4.x + g4.y * x4.y + g4.z * x4.z + g4.w * x4.w;
      }
      for (int d = LH >> 1; d >= 1; d >>= 1) p += __shfl_xor(p, d, 64);
      if (on) {
        dot += alpha[(int64_t)(s + j) * H + h] * p;
        if (t == 0) de[(int64_t)(s + j) * H + h] = p;
      }
    }
    // pass 2: the LH lanes of a head split the row's edges
    float gad = 0.f;
    if (valid) {
      const float ad = a_dst[row * H + h];
      for (int j = s + t; j < e; j += LH) {
        const int64_t eh = (int64_t)j * H + h;
        const float sc   = a_src[(int64_t)col[j] * H + h] + ad;
        float ds         = alpha[eh] * (de[eh] - dot);
        ds               = sc > 0.f ? ds : ds * slope;
        de[eh]           = ds;
        gad += ds;
      }
    }
    for (int d = LH >> 1; d >= 1; d >>= 1) gad += __shfl_xor(gad, d, 64);
    if (valid && t == 0) ga_dst[row * H + h] = gad;
  }
}

// The source-major pass walks the TRANSPOSED hop, which is power-law (a hub source is a neighbour of thousands of sampled
// rows): with one lane group per source and one dependent load chain per entry a 3,800-entry row alone took milliseconds.
// Rows are therefore summed in PIECES of kSegEntries entries (as in wgamd_spmm_csr_segmented_f32): row r < n_main sums its
// first piece, every further piece of a long row is an extra "row" n_main + x handed out by gat_plan_kernel whose sums go
// to partial buffers, and gat_addup_kernel adds the pieces of a row up in order — deterministic.  Four entries are in
// flight per lane group (edge id -> alpha / destination -> gradient row is a chain of dependent loads).
constexpr int kSegEntries = 64;
struct gat_segments {
  int seg;                  // 0 = plain rows
  int64_t n_main;
  const int* extra_start;
  const int* extra_end;
  const int* n_extra_dev;
  float* partial_gx;        // [extras, ldp]
  int64_t ldp;
  float* partial_gas;       // [extras, H]
};
struct gat_long_row {
  int row, base, pieces;
};

// counters: [0] extra slots handed out, [1] long rows, [2] OVERFLOW flag — a long row whose pieces did not fit the `cap` slots
// of the workspace gets none (its gradient then misses the entries past its first piece) and raises the flag instead of
// writing past the buffers; with a workspace sized by wgamd_gat_csr_bwd_workspace_bytes(row_ptr_t[n_src], H, C) it cannot
// happen (at most E / 64 further pieces exist).
__global__ void __launch_bounds__(256) gat_plan_kernel(const int* __restrict__ row_ptr_t, int64_t n_src, int seg, int cap,
                                                       int* __restrict__ counters, int* __restrict__ extra_start,
                                                       int* __restrict__ extra_end, gat_long_row* __restrict__ long_rows)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_src) return;
  const int s = row_ptr_t[r], e = row_ptr_t[r + 1];
  if (e - s <= seg) return;
  const int pieces = (e - s + seg - 1) / seg - 1;
  const int base   = atomicAdd(counters, pieces);
  if (base + pieces > cap) {
    atomicSub(counters, pieces);   // hand the slots back: later rows may still fit, and the count stays <= cap
    counters[2] = 1;
    return;
  }
  long_rows[atomicAdd(counters + 1, 1)] = gat_long_row{(int)r, base, pieces};
  for (int j = 0; j < pieces; j++) {
    extra_start[base + j] = s + (j + 1) * seg;
    extra_end[base + j]   = min(e, s + (j + 2) * seg);
  }
}

template <bool SEG>
__global__ void __launch_bounds__(256)
gat_bwd_src_kernel(const int* __restrict__ row_ptr_t, const int* __restrict__ edge_perm, const int* __restrict__ edge_dst,
                   int64_t n_rows, int H, int C, const float* __restrict__ alpha, const float* __restrict__ de,
                   const float* __restrict__ g, int64_t ldg, float* __restrict__ gx, int64_t ldgx,
                   float* __restrict__ ga_src, int log2_lanes, gat_segments sg)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const int HC = H * C;
  const int f0 = sub * 4;
  if (f0 >= HC) return;
  const int h = f0 / C;
  for (int64_t row = group; row < n_rows; row += ngroups) {
    int s, e;
    float* gx_row;
    float* gas_row;
    if (!SEG || row < sg.n_main) {
      s = row_ptr_t[row];
      e = row_ptr_t[row + 1];
      if constexpr (SEG) e = min(e, s + sg.seg);
      gx_row  = gx + row * ldgx;
      gas_row = ga_src + row * H;
    } else {
      const int64_t xs = row - sg.n_main;
      if (xs >= (int64_t)*sg.n_extra_dev) break;   // extras are handed out from 0: nothing further for this lane group
      s       = sg.extra_start[xs];
      e       = sg.extra_end[xs];
      gx_row  = sg.partial_gx + xs * sg.ldp;
      gas_row = sg.partial_gas + xs * H;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float gas  = 0.f;
    for (int j = s; j < e; j += 4) {
      int eid[4];
      float al[4], dv[4];
      float4 t4[4];
#pragma unroll
      for (int k = 0; k < 4; k++) eid[k] = edge_perm[min(j + k, e - 1)];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        al[k] = alpha[(int64_t)eid[k] * H + h];
        dv[k] = de[(int64_t)eid[k] * H + h];
        t4[k] = *reinterpret_cast<const float4*>(g + (int64_t)edge_dst[eid[k]] * ldg + f0);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (j + k < e) {
          acc.x += al[k] * t4[k].x;
          acc.y += al[k] * t4[k].y;
          acc.z += al[k] * t4[k].z;
          acc.w += al[k] * t4[k].w;
          gas += dv[k];
        }
      }
    }
    *reinterpret_cast<float4*>(gx_row + f0) = acc;
    if ((f0 % C) == 0) gas_row[h] = gas;
  }
}

// one wave per long row: grad_x[row] += its pieces, grad_a_src[row] += its pieces, in piece order
__global__ void __launch_bounds__(256)
gat_addup_kernel(const gat_long_row* __restrict__ long_rows, const int* __restrict__ counters, const float* __restrict__ partial_gx,
                 int64_t ldp, const float* __restrict__ partial_gas, int H, int HC, float* __restrict__ gx, int64_t ldgx,
                 float* __restrict__ ga_src)
{
  const int lane = threadIdx.x & 63;
  for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; k < counters[1]; k += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const gat_long_row lr = long_rows[k];
    for (int f = lane; f < HC; f += 64) {
      float acc = gx[(int64_t)lr.row * ldgx + f];
      for (int j = 0; j < lr.pieces; j++) acc += partial_gx[(int64_t)(lr.base + j) * ldp + f];
      gx[(int64_t)lr.row * ldgx + f] = acc;
    }
    if (lane < H) {
      float acc = ga_src[(int64_t)lr.row * H + lane];
      for (int j = 0; j < lr.pieces; j++) acc += partial_gas[(int64_t)(lr.base + j) * H + lane];
      ga_src[(int64_t)lr.row * H + lane] = acc;
    }
  }
}

inline int log2_ceil(int v)
{
  int l = 0;
  while ((1 << l) < v) l++;
  return l;
}

}  // namespace
}  // namespace wgamd

extern "C" size_t wgamd_gat_csr_bwd_workspace_bytes(int64_t n_entries, int H, int C)
{
  if (n_entries < 0 || H <= 0 || C <= 0) return 0;
  // one slot per piece of a long source row beyond its first: at most n_entries / 64 of them
  const size_t cap       = (size_t)(n_entries / wgamd::kSegEntries) + 1;
  const size_t per_extra = 2 * sizeof(int) + sizeof(wgamd::gat_long_row) + (size_t)H * C * sizeof(float) + (size_t)H * sizeof(float);
  return 1024 + cap * (per_extra + 16);
}

namespace wgamd {
namespace {
// piece slots a workspace of `bytes` holds (the inverse of wgamd_gat_csr_bwd_workspace_bytes)
size_t gat_bwd_slots(size_t bytes, int H, int C)
{
  const size_t per_extra = 2 * sizeof(int) + sizeof(gat_long_row) + (size_t)H * C * sizeof(float) + (size_t)H * sizeof(float) + 16;
  return bytes > 1024 + per_extra ? (bytes - 1024) / per_extra : 0;
}

// n_entries < 0: the capacity comes from the workspace size alone (the entry point without n_entries)
wholememory_error_code_t gat_csr_bwd(const char* op, const int* row_ptr, const int* col, int64_t n_rows, const float* x, int64_t ldx,
                                     const float* a_src, const float* a_dst, int H, int C, float negative_slope, const float* alpha,
                                     const float* grad_out, int64_t ldg, const int* row_ptr_t, const int* edge_perm,
                                     const int* edge_dst, int64_t n_src, float* de, float* grad_x, int64_t ldgx, float* grad_a_src,
                                     float* grad_a_dst, int64_t n_entries, void* workspace, size_t workspace_bytes, void* stream)
{
  return guarded(op, [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && n_src >= 0 && H > 0 && C > 0, "bad sizes");
    const int HC = H * C, LH = C / 4;
    if (C % 4 != 0 || (LH & (LH - 1)) != 0 || HC > 256 || ldx % 4 != 0 || ldg % 4 != 0 || ldgx % 4 != 0)
      throw logic_error(fmt("unsupported shape: H=%d C=%d (C/4 a power of two, H*C <= 256, 16-B aligned rows)", H, C));
    WG_REQUIRE_INPUT(row_ptr && col && x && a_src && a_dst && alpha && grad_out && row_ptr_t && edge_perm && edge_dst && de &&
                       grad_x && grad_a_src && grad_a_dst,
                     "null pointer");
    auto st      = static_cast<hipStream_t>(stream);
    const int l2 = log2_ceil(HC / 4);
    const int gpb = 256 >> l2;  // lane groups per workgroup
    if (n_rows > 0) {
      const int grid = (int)std::min<int64_t>((n_rows + gpb - 1) / gpb, 256 * 16);
      gat_bwd_dst_kernel<<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope, alpha,
                                               grad_out, ldg, de, grad_a_dst, l2);
    }
    if (n_src > 0 && workspace == nullptr) {   // plain rows (a caller without scratch)
      const int grid = (int)std::min<int64_t>((n_src + gpb - 1) / gpb, 256 * 16);
      gat_bwd_src_kernel<false><<<grid, 256, 0, st>>>(row_ptr_t, edge_perm, edge_dst, n_src, H, C, alpha, de, grad_out, ldg,
                                                      grad_x, ldgx, grad_a_src, l2, gat_segments{});
    } else if (n_src > 0) {
      // every further piece of a long row needs a slot: at most n_entries / kSegEntries of them, which is what the
      // workspace must hold — the plan kernel hands slots out with atomics and has no other bound
      if (n_entries >= 0)
        WG_REQUIRE_INPUT(workspace_bytes >= wgamd_gat_csr_bwd_workspace_bytes(n_entries, H, C),
                         "workspace smaller than wgamd_gat_csr_bwd_workspace_bytes(n_entries, H, C)");
      const size_t cap = n_entries >= 0 ? (size_t)(n_entries / kSegEntries) + 1 : gat_bwd_slots(workspace_bytes, H, C);
      WG_REQUIRE_INPUT(cap >= 1 && cap < ((size_t)1 << 31), "workspace too small for a single piece slot");
      char* ws         = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
      auto carve       = [&](size_t bytes) {
        char* at = ws;
        ws += (bytes + 15) / 16 * 16;
        return at;
      };
      int* counters    = reinterpret_cast<int*>(carve(256));
      int* extra_start = reinterpret_cast<int*>(carve(cap * sizeof(int)));
      int* extra_end   = reinterpret_cast<int*>(carve(cap * sizeof(int)));
      auto* long_rows  = reinterpret_cast<gat_long_row*>(carve(cap * sizeof(gat_long_row)));
      float* pgx       = reinterpret_cast<float*>(carve(cap * (size_t)HC * sizeof(float)));
      float* pgas      = reinterpret_cast<float*>(carve(cap * (size_t)H * sizeof(float)));
      WG_HIP_CHECK(hipMemsetAsync(counters, 0, 4 * sizeof(int), st));
      gat_plan_kernel<<<(int)((n_src + 255) / 256), 256, 0, st>>>(row_ptr_t, n_src, kSegEntries, (int)cap, counters, extra_start,
                                                                   extra_end, long_rows);
      const int64_t n_all = n_src + (int64_t)cap;
      const int grid      = (int)std::min<int64_t>((n_all + gpb - 1) / gpb, 256 * 16);
      gat_segments sg{kSegEntries, n_src, extra_start, extra_end, counters, pgx, HC, pgas};
      gat_bwd_src_kernel<true><<<grid, 256, 0, st>>>(row_ptr_t, edge_perm, edge_dst, n_all, H, C, alpha, de, grad_out, ldg,
                                                     grad_x, ldgx, grad_a_src, l2, sg);
      gat_addup_kernel<<<(int)std::min<size_t>((cap + 3) / 4, 4096), 256, 0, st>>>(long_rows, counters, pgx, HC, pgas, H, HC,
                                                                                    grad_x, ldgx, grad_a_src);
    }
    WG_HIP_CHECK(hipGetLastError());
  });
}
// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the AGGREGATE-FIRST GAT aggregation (wgamd_gat_aggregate_heads[_ids]_f32):
//     agg[i, h, :] = sum_e alpha_e^h x[r(e), :],   alpha^h = softmax_e(LeakyReLU(a_src[t(e), h] + a_dst[d(i), h]))
// (x untransformed, F floats shared by the H heads; r / t / d = the row of x, of a_src, of a_dst an edge / a destination reads:
// plain, or through the id lists of the fetch-in-the-layer forward).  Given g = dL/dagg [n_rows, H F]:
//     p_e^h  = <g[i, h, :], x[r(e), :]>,   ds_e^h = alpha_e^h (p_e^h - sum_k alpha_k^h p_k^h),   de_e^h = ds_e^h LeakyReLU'(.)
//     ga_dst[d(i), h] += sum_e de_e^h,   ga_src[t(e), h] += de_e^h,   gx[r(e), :] += sum_h alpha_e^h g[i, h, :]   (gx optional)
// One lane group (F / 4 lanes) per destination row — the forward's mapping: the row's H gradient slices sit in registers, every
// neighbour row is read once (16 B per lane), the H dot products are reduced with xor-shuffles inside the group, p is parked in
// `de` between the two passes.  ga_src / ga_dst / gx are ACCUMULATED with float atomics into caller-zeroed buffers: with the
// terms of the table's rows one table row collects from every mini-batch that sampled it, and a source row from every
// destination that drew it (a sampled hop has at most `fan-out` edges per row: no long-row path).  The sums' order is not
// fixed: results are reproducible to fp32 rounding, not bit for bit.  Semantics: torch_geometric.nn.GATConv's message /
// aggregate (pylibwholegraph/torch/gnn_model.py:45-59) in the aggregate-first form of DESIGN.md section 3.5.
// Sum / maximum over a lane group of 2^LG lanes, the result in every lane.  Inside a 16-lane row the butterfly runs on DPP
// moves folded into VALU instructions (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: after the first
// two steps a quad is uniform, so the mirrors act as xor 4 and xor 8); the steps across rows (groups of 32 / 64 lanes) are
// permlane swaps — nothing goes through the LDS crossbar.  The generic `__shfl_xor` loop is ds_bpermute + wait + add per step: 5 LDS round trips per reduction
// in a 32-lane group, and this kernel makes H of them per edge — most of its issue time (round 6).  LG < 0: runtime width.
template <int LG, bool MAX>
__device__ __forceinline__ float group_reduce(float v, int lanes_rt)
{
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
#define WG_DPP(ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, false))
  if constexpr (LG < 0) {
    for (int d = lanes_rt >> 1; d >= 1; d >>= 1) v = op(v, __shfl_xor(v, d, 64));
  } else {
    if constexpr (LG >= 1) v = op(v, WG_DPP(0xB1));
    if constexpr (LG >= 2) v = op(v, WG_DPP(0x4E));
    if constexpr (LG >= 3) v = op(v, WG_DPP(0x141));
    if constexpr (LG >= 4) v = op(v, WG_DPP(0x140));
    // across rows: gfx950's v_permlane16_swap / v_permlane32_swap (VALU; swap the odd rows / the upper half of the first
    // operand with the even rows / the lower half of the second): with both operands = v, a lane finds its partner's value in
    // the first result when it sits in an odd row (upper half), in the second otherwise
    if constexpr (LG >= 5) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v            = op(v, __uint_as_float((__lane_id() & 16) ? r[0] : r[1]));
    }
    if constexpr (LG >= 6) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v            = op(v, __uint_as_float((__lane_id() & 32) ? r[0] : r[1]));
    }
  }
#undef WG_DPP
  return v;
}

template <int H, int LG>
__global__ void __launch_bounds__(256)
gat_aggregate_heads_bwd_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col, int64_t n_rows,
                               const float* __restrict__ x, int64_t ldx, int F, const float* __restrict__ a_src,
                               const float* __restrict__ a_dst, float slope, const int64_t* __restrict__ dst_rows,
                               const int64_t* __restrict__ src_ids, const int64_t* __restrict__ dst_ids, int terms_by_id,
                               const float* __restrict__ g, int64_t ldg, float* __restrict__ de, float* __restrict__ ga_src,
                               float* __restrict__ ga_dst, float* __restrict__ gx, int64_t ldgx, int log2_lanes,
                               float* __restrict__ stats)
{
  // stats (nullable): [2][n_rows][H] — the row's softmax maximum and denominator per head, for the source-major gx kernel
  // Lane k of the group OWNS edge c0 + k of the current chunk of `lanes` edges: its column, term row, scores and attention
  // weights are loaded / computed once, by that lane (coalesced), and broadcast where the whole group needs them; the
  // neighbour rows of a chunk are requested four at a time.
  constexpr int EIF     = 4;
  if constexpr (LG >= 0) log2_lanes = LG;
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int gbase       = (int)(threadIdx.x & 63) & ~(lanes - 1);
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const bool live       = sub * 4 < F;
  const int f0          = live ? sub * 4 : 0;
  auto group_sum = [&](float v) { return group_reduce<LG, false>(v, lanes); };
  auto group_max = [&](float v) { return group_reduce<LG, true>(v, lanes); };
  // A lane group works on ONE destination row at a time, and a row starts with a chain of dependent loads — row_ptr -> col ->
  // id list -> rows of x / of the terms (and dst_rows -> id list -> a_dst): four round trips before the first useful byte, which
  // with 24 rows in flight per CU was the kernel's time (2.6 ms for the 1.8 M rows of a products hop).  The chain is software-
  // pipelined over the group's rows: stage A (row_ptr, dst_rows) runs three rows ahead, B (col, the destination's id) two, C (the
  // source's id, a_dst, the row's gradient slices) one — every load a row's arithmetic waits for was issued a row earlier.
  auto stage_a = [&](int64_t r, int& s_, int& e_, int64_t& ar_) {
    const bool in = r < n_rows;
    s_  = in ? row_ptr[r] : 0;
    e_  = in ? row_ptr[r + 1] : 0;
    ar_ = in ? (dst_rows ? dst_rows[r] : r) : 0;
  };
  auto stage_b = [&](int s_, int e_, int64_t& ar_, int& c_) {
    c_ = s_ + sub < e_ ? col[s_ + sub] : 0;
    if (terms_by_id & 2) ar_ = dst_ids[ar_];
  };
  auto stage_c = [&](int64_t r, int c_, int64_t ar_, int64_t& xr_, float (&ad_)[H], float4 (&g_)[H]) {
    xr_ = src_ids ? src_ids[c_] : (int64_t)c_;
#pragma unroll
    for (int h = 0; h < H; h++) {
      ad_[h] = a_dst[ar_ * H + h];
      g_[h]  = (live && r < n_rows) ? *reinterpret_cast<const float4*>(g + r * ldg + (int64_t)h * F + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int s1, e1, c1, s2, e2, c2, s3, e3;
  int64_t ar1, ar2, ar3, xr1;
  float ad1[H];
  float4 g1[H];
  stage_a(group, s1, e1, ar1);
  stage_b(s1, e1, ar1, c1);
  stage_c(group, c1, ar1, xr1, ad1, g1);
  stage_a(group + ngroups, s2, e2, ar2);
  stage_b(s2, e2, ar2, c2);
  stage_a(group + 2 * ngroups, s3, e3, ar3);
  for (int64_t row = group; row < n_rows; row += ngroups) {
    const int s = s1, e = e1, c_own = c1;
    const int64_t arow = ar1, xr_own = xr1;
    float ad[H], m[H], den[H], dot[H], gad[H];
    float4 g4[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
      ad[h]  = ad1[h];
      g4[h]  = g1[h];
      m[h]   = -INFINITY;
      den[h] = 0.f;
      dot[h] = 0.f;
      gad[h] = 0.f;
    }
    s1 = s2, e1 = e2, ar1 = ar2, c1 = c2;
    stage_c(row + ngroups, c1, ar1, xr1, ad1, g1);
    s2 = s3, e2 = e3, ar2 = ar3;
    stage_b(s2, e2, ar2, c2);
    stage_a(row + 3 * ngroups, s3, e3, ar3);
    if (s == e) continue;   // (no edge: nothing flows back; uniform over the lane group)
    // what the owner of edge j needs: the row of x, the row of the terms, the raw scores
    auto own = [&](int j, int64_t& xr, int64_t& tr, float (&raw)[H]) {
      const int c = col[j];
      xr          = src_ids ? src_ids[c] : (int64_t)c;
      tr          = (terms_by_id & 1) ? xr : (int64_t)c;
#pragma unroll
      for (int h = 0; h < H; h++) raw[h] = a_src[tr * H + h] + ad[h];
    };
    auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
    if (e - s <= lanes) {
      // ---- the usual case (a sampled hop: at most `fan-out` edges per row): ONE chunk, the owner of edge s + sub keeps its
      // column, rows and scores in registers through all phases — one dependent col -> id -> term chain per row instead of four
      const bool on = s + sub < e;
      const int cnt = e - s;
      float raw[H], al[H], pown[H];
      const int64_t xr = xr_own, tr = (terms_by_id & 1) ? xr_own : (int64_t)c_own;   // (stages B and C of this row, a row ago)
      // the neighbour rows of the first four edges are requested BEFORE the scores are waited for (they do not depend on them)
      float4 x4[EIF];
      int64_t xu[EIF];
      auto request = [&](int k, float4 (&xx)[EIF], int64_t (&uu)[EIF]) {
#pragma unroll
        for (int u = 0; u < EIF; u++) {
          const int from = gbase | min(k + u, cnt - 1);
          uu[u] = ((int64_t)__shfl((int)(xr >> 32), from, 64) << 32) | (uint32_t)__shfl((int)xr, from, 64);
          xx[u] = live ? *reinterpret_cast<const float4*>(x + uu[u] * ldx + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      request(0, x4, xu);
#pragma unroll
      for (int h = 0; h < H; h++) raw[h] = on ? a_src[tr * H + h] + ad[h] : 0.f;
#pragma unroll
      for (int h = 0; h < H; h++) {
        const float sc = on ? leaky(raw[h]) : -INFINITY;
        m[h]           = group_max(sc);
        const float ex = on ? expf(sc - m[h]) : 0.f;
        den[h]         = group_sum(ex);
        al[h]          = ex / den[h];
        pown[h]        = 0.f;
      }
      for (int k = 0; k < cnt; k += EIF) {
        float4 xn[EIF];
        int64_t un[EIF];
        if (k + EIF < cnt) request(k + EIF, xn, un);   // (uniform over the group) the next four, under this batch's arithmetic
#pragma unroll
        for (int u = 0; u < EIF; u++) {
          if (k + u >= cnt) break;   // (uniform over the group)
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int h = 0; h < H; h++) {
            const float p = group_sum(g4[h].x * x4[u].x + g4[h].y * x4[u].y + g4[h].z * x4[u].z + g4[h].w * x4[u].w);
            if (sub == k + u) pown[h] = p;
            if (gx != nullptr) {
              const float a_ = __shfl(al[h], gbase | (k + u), 64);
              acc.x += a_ * g4[h].x; acc.y += a_ * g4[h].y; acc.z += a_ * g4[h].z; acc.w += a_ * g4[h].w;
            }
          }
          if (gx != nullptr && live) {
            float* q = gx + xu[u] * ldgx + f0;
            unsafeAtomicAdd(q, acc.x); unsafeAtomicAdd(q + 1, acc.y); unsafeAtomicAdd(q + 2, acc.z); unsafeAtomicAdd(q + 3, acc.w);
          }
        }
        if (k + EIF < cnt) {
#pragma unroll
          for (int u = 0; u < EIF; u++) {
            x4[u] = xn[u];
            xu[u] = un[u];
          }
        }
      }
#pragma unroll
      for (int h = 0; h < H; h++) {
        dot[h]   = group_sum(al[h] * pown[h]);
        float ds = al[h] * (pown[h] - dot[h]);
        ds       = raw[h] > 0.f ? ds : ds * slope;
        if (on) {
          de[(int64_t)(s + sub) * H + h] = ds;
#ifndef WG_ABL_NO_GASRC
          unsafeAtomicAdd(ga_src + tr * H + h, ds);
#endif
        }
        gad[h] = on ? ds : 0.f;
      }
    } else {
      // the row's softmax statistics
      for (int c0 = s; c0 < e; c0 += lanes)
        if (c0 + sub < e) {
          int64_t xr, tr;
          float raw[H];
          own(c0 + sub, xr, tr, raw);
#pragma unroll
          for (int h = 0; h < H; h++) m[h] = fmaxf(m[h], leaky(raw[h]));
        }
#pragma unroll
      for (int h = 0; h < H; h++) m[h] = group_max(m[h]);
      for (int c0 = s; c0 < e; c0 += lanes)
        if (c0 + sub < e) {
          int64_t xr, tr;
          float raw[H];
          own(c0 + sub, xr, tr, raw);
#pragma unroll
          for (int h = 0; h < H; h++) den[h] += expf(leaky(raw[h]) - m[h]);
        }
#pragma unroll
      for (int h = 0; h < H; h++) den[h] = group_sum(den[h]);
      // pass 1: p per edge and head (kept by the owner, parked in de), dot = sum alpha p, gx
      for (int c0 = s; c0 < e; c0 += lanes) {
        const bool on = c0 + sub < e;
        const int cnt = min(lanes, e - c0);
        int64_t xr = 0, tr = 0;
        float al[H], pown[H];
        {
          float raw[H];
#pragma unroll
          for (int h = 0; h < H; h++) raw[h] = 0.f;
          if (on) own(c0 + sub, xr, tr, raw);
#pragma unroll
          for (int h = 0; h < H; h++) {
            al[h]   = on ? expf(leaky(raw[h]) - m[h]) / den[h] : 0.f;
            pown[h] = 0.f;
          }
        }
        for (int k = 0; k < cnt; k += EIF) {
          float4 x4[EIF];
          int64_t xu[EIF];
#pragma unroll
          for (int u = 0; u < EIF; u++) {
            const int from = gbase | min(k + u, cnt - 1);
            xu[u] = ((int64_t)__shfl((int)(xr >> 32), from, 64) << 32) | (uint32_t)__shfl((int)xr, from, 64);
            x4[u] = live ? *reinterpret_cast<const float4*>(x + xu[u] * ldx + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < EIF; u++) {
            if (k + u >= cnt) break;   // (uniform over the group)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < H; h++) {
  #ifdef WG_ABL_NO_PSUM
            const float p = (g4[h].x * x4[u].x + g4[h].y * x4[u].y + g4[h].z * x4[u].z + g4[h].w * x4[u].w);
#else
            const float p = group_sum(g4[h].x * x4[u].x + g4[h].y * x4[u].y + g4[h].z * x4[u].z + g4[h].w * x4[u].w);
#endif
              if (sub == k + u) pown[h] = p;
              if (gx != nullptr) {
                const float a_ = __shfl(al[h], gbase | (k + u), 64);
                acc.x += a_ * g4[h].x; acc.y += a_ * g4[h].y; acc.z += a_ * g4[h].z; acc.w += a_ * g4[h].w;
              }
            }
            if (gx != nullptr && live) {
              float* q = gx + xu[u] * ldgx + f0;
              unsafeAtomicAdd(q, acc.x); unsafeAtomicAdd(q + 1, acc.y); unsafeAtomicAdd(q + 2, acc.z); unsafeAtomicAdd(q + 3, acc.w);
            }
          }
        }
        if (on) {
#pragma unroll
          for (int h = 0; h < H; h++) {
            dot[h] += al[h] * pown[h];
            de[(int64_t)(c0 + sub) * H + h] = pown[h];
          }
        }
      }
#pragma unroll
      for (int h = 0; h < H; h++) dot[h] = group_sum(dot[h]);
      // pass 2: every owner finishes its edges (it reads back what it parked itself)
      for (int c0 = s; c0 < e; c0 += lanes)
        if (c0 + sub < e) {
          const int j = c0 + sub;
          int64_t xr, tr;
          float raw[H];
          own(j, xr, tr, raw);
#pragma unroll
          for (int h = 0; h < H; h++) {
            float ds = expf(leaky(raw[h]) - m[h]) / den[h] * (de[(int64_t)j * H + h] - dot[h]);
            ds       = raw[h] > 0.f ? ds : ds * slope;
            de[(int64_t)j * H + h] = ds;
            gad[h] += ds;
  #ifndef WG_ABL_NO_GASRC
          unsafeAtomicAdd(ga_src + tr * H + h, ds);
#endif
          }
        }
    }
#pragma unroll
    for (int h = 0; h < H; h++) {
      const float v = group_sum(gad[h]);
      if (sub == 0) {
        unsafeAtomicAdd(ga_dst + arow * H + h, v);
        if (stats != nullptr) {
          stats[row * H + h]            = m[h];
          stats[(n_rows + row) * H + h] = den[h];
        }
      }
    }
  }
}

// gx[j, :] = sum over the edges (i, j) of sum_h alpha_{ij}^h g[i, h, :] — the gradient of the SOURCE rows of the aggregate-first
// aggregation, source-major over the transposed hop (row_ptr_t / col_t = destination row of every entry, wgamd_csr_transpose_i32):
// one lane group per source row, every row of gx written exactly once, no atomics (the destination-major kernel's alternative is
// F atomics per edge: 1.25 G of them for the seeds' layer of a products call group).  alpha is recomputed from the terms and the
// destination rows' softmax statistics the destination-major kernel left in `stats`.  Plain addressing only (a hidden-state
// input: a_src per source row, a_dst at dst_rows[i]).
template <int H>
__global__ void __launch_bounds__(256)
gat_aggregate_heads_bwd_gx_kernel(const int* __restrict__ row_ptr_t, const int* __restrict__ col_t, int64_t n_src, int64_t n_rows,
                                  int F, const float* __restrict__ a_src, const float* __restrict__ a_dst, float slope,
                                  const int64_t* __restrict__ dst_rows, const float* __restrict__ stats,
                                  const float* __restrict__ g, int64_t ldg, float* __restrict__ gx, int64_t ldgx, int log2_lanes)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int gbase       = (int)(threadIdx.x & 63) & ~(lanes - 1);
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const bool live       = sub * 4 < F;
  const int f0          = live ? sub * 4 : 0;
  for (int64_t j = group; j < n_src; j += ngroups) {
    const int s = row_ptr_t[j], e = row_ptr_t[j + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float as_[H];
#pragma unroll
    for (int h = 0; h < H; h++) as_[h] = a_src[j * H + h];
    for (int c0 = s; c0 < e; c0 += lanes) {
      const bool on = c0 + sub < e;
      const int cnt = min(lanes, e - c0);
      const int i   = on ? col_t[c0 + sub] : 0;
      float w[H];
      {
        const int64_t arow = dst_rows ? dst_rows[i] : (int64_t)i;
#pragma unroll
        for (int h = 0; h < H; h++) {
          float sc = as_[h] + a_dst[arow * H + h];
          sc       = sc > 0.f ? sc : sc * slope;
          w[h]     = on ? expf(sc - stats[(int64_t)i * H + h]) / stats[(n_rows + i) * H + h] : 0.f;
        }
      }
      for (int k = 0; k < cnt; k++) {
        const int ik = __shfl(i, gbase | k, 64);
#pragma unroll
        for (int h = 0; h < H; h++) {
          const float wk = __shfl(w[h], gbase | k, 64);
          if (live) {
            const float4 g4 = *reinterpret_cast<const float4*>(g + (int64_t)ik * ldg + (int64_t)h * F + f0);
            acc.x += wk * g4.x; acc.y += wk * g4.y; acc.z += wk * g4.z; acc.w += wk * g4.w;
          }
        }
      }
    }
    if (live) *reinterpret_cast<float4*>(gx + j * ldgx + f0) = acc;
  }
}

}  // namespace
}  // namespace wgamd

extern "C" wholememory_error_code_t wgamd_gat_aggregate_heads_bwd_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                                      int64_t ldx, const int64_t* src_ids, const int64_t* dst_ids,
                                                                      int terms_by_id, int F, const float* a_src, const float* a_dst,
                                                                      int H, float negative_slope, const int64_t* dst_rows,
                                                                      const float* grad_agg, int64_t ldg, float* de, float* grad_a_src,
                                                                      float* grad_a_dst, float* grad_x, int64_t ldgx, float* stats,
                                                                      void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gat_aggregate_heads_bwd_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && H > 0 && F > 0, "bad sizes");
    WG_REQUIRE_INPUT(terms_by_id >= 0 && terms_by_id <= 3 && (terms_by_id == 0 || src_ids) && (!(terms_by_id & 2) || dst_ids),
                     "terms_by_id needs src_ids (and dst_ids for bit 1)");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && a_src && a_dst && grad_agg && de && grad_a_src && grad_a_dst, "null pointer");
    if (F % 4 != 0 || F > 256 || !(H == 1 || H == 2 || H == 4 || H == 8) || (ldx & 3) != 0 || (ldg & 3) != 0 ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_agg)) & 15) != 0 ||
        (grad_x && ((ldgx & 3) != 0 || (reinterpret_cast<uintptr_t>(grad_x) & 15) != 0)))
      throw logic_error(fmt("unsupported shape: F=%d (multiple of 4, <= 256), H=%d (1, 2, 4 or 8), 16-B aligned rows", F, H));
    WG_REQUIRE_INPUT(ldx >= F && ldg >= (int64_t)H * F && (!grad_x || ldgx >= F), "leading dimension");
    auto st      = static_cast<hipStream_t>(stream);
    int l2       = 0;
    while ((1 << l2) < F / 4 && l2 < 6) l2++;
    const int64_t groups_per_block = 256 >> l2;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_rows + groups_per_block - 1) / groups_per_block, 256 * 16));
#define WG_GAT_AGG_BWD_(HH, LL)                                                                                                       \
  gat_aggregate_heads_bwd_kernel<HH, LL><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, F, a_src, a_dst, negative_slope,          \
                                                               dst_rows, src_ids, dst_ids, terms_by_id, grad_agg, ldg, de, grad_a_src, \
                                                               grad_a_dst, grad_x, ldgx, l2, stats)
    // (the group width is a compile-time constant for rows of 32 floats and more: the reductions then run on DPP moves)
#define WG_GAT_AGG_BWD(HH)                                                                                                         \
  do {                                                                                                                             \
    if (l2 == 3) WG_GAT_AGG_BWD_(HH, 3); else if (l2 == 4) WG_GAT_AGG_BWD_(HH, 4); else if (l2 == 5) WG_GAT_AGG_BWD_(HH, 5);       \
    else if (l2 == 6) WG_GAT_AGG_BWD_(HH, 6); else WG_GAT_AGG_BWD_(HH, -1);                                                        \
  } while (0)
    switch (H) {
      case 1: WG_GAT_AGG_BWD(1); break;
      case 2: WG_GAT_AGG_BWD(2); break;
      case 4: WG_GAT_AGG_BWD(4); break;
      default: WG_GAT_AGG_BWD(8); break;
    }
#undef WG_GAT_AGG_BWD
#undef WG_GAT_AGG_BWD_
    WG_HIP_CHECK(hipGetLastError());
  });
}

extern "C" wholememory_error_code_t wgamd_gat_aggregate_heads_bwd_gx_f32(const int* row_ptr_t, const int* col_t, int64_t n_src,
                                                                         int64_t n_rows, int F, const float* a_src, const float* a_dst,
                                                                         int H, float negative_slope, const int64_t* dst_rows,
                                                                         const float* stats, const float* grad_agg, int64_t ldg,
                                                                         float* grad_x, int64_t ldgx, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gat_aggregate_heads_bwd_gx_f32", [&] {
    WG_REQUIRE_INPUT(n_src >= 0 && n_rows >= 0 && H > 0 && F > 0, "bad sizes");
    if (n_src == 0) return;
    WG_REQUIRE_INPUT(row_ptr_t && a_src && a_dst && stats && grad_agg && grad_x, "null pointer");
    if (F % 4 != 0 || F > 256 || !(H == 1 || H == 2 || H == 4 || H == 8) || (ldg & 3) != 0 || (ldgx & 3) != 0 ||
        ((reinterpret_cast<uintptr_t>(grad_agg) | reinterpret_cast<uintptr_t>(grad_x)) & 15) != 0)
      throw logic_error(fmt("unsupported shape: F=%d (multiple of 4, <= 256), H=%d (1, 2, 4 or 8), 16-B aligned rows", F, H));
    WG_REQUIRE_INPUT(ldg >= (int64_t)H * F && ldgx >= F, "leading dimension");
    auto st = static_cast<hipStream_t>(stream);
    int l2  = 0;
    while ((1 << l2) < F / 4 && l2 < 6) l2++;
    const int64_t groups_per_block = 256 >> l2;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_src + groups_per_block - 1) / groups_per_block, 256 * 16));
#define WG_GAT_GX(HH)                                                                                                            \
  gat_aggregate_heads_bwd_gx_kernel<HH><<<grid, 256, 0, st>>>(row_ptr_t, col_t, n_src, n_rows, F, a_src, a_dst, negative_slope,     \
                                                              dst_rows, stats, grad_agg, ldg, grad_x, ldgx, l2)
    switch (H) {
      case 1: WG_GAT_GX(1); break;
      case 2: WG_GAT_GX(2); break;
      case 4: WG_GAT_GX(4); break;
      default: WG_GAT_GX(8); break;
    }
#undef WG_GAT_GX
    WG_HIP_CHECK(hipGetLastError());
  });
}

/* the entry point of rounds 1-2 (no n_entries): the piece capacity is whatever the workspace holds */
extern "C" wholememory_error_code_t wgamd_gat_csr_bwd_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                          int64_t ldx, const float* a_src, const float* a_dst, int H, int C,
                                                          float negative_slope, const float* alpha, const float* grad_out,
                                                          int64_t ldg, const int* row_ptr_t, const int* edge_perm,
                                                          const int* edge_dst, int64_t n_src, float* de, float* grad_x,
                                                          int64_t ldgx, float* grad_a_src, float* grad_a_dst, void* workspace,
                                                          size_t workspace_bytes, void* stream)
{
  return wgamd::gat_csr_bwd("wgamd_gat_csr_bwd_f32", row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope, alpha,
                            grad_out, ldg, row_ptr_t, edge_perm, edge_dst, n_src, de, grad_x, ldgx, grad_a_src, grad_a_dst,
                            /*n_entries=*/-1, workspace, workspace_bytes, stream);
}

extern "C" wholememory_error_code_t wgamd_gat_csr_bwd_f32_v2(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                             int64_t ldx, const float* a_src, const float* a_dst, int H, int C,
                                                             float negative_slope, const float* alpha, const float* grad_out,
                                                             int64_t ldg, const int* row_ptr_t, const int* edge_perm,
                                                             const int* edge_dst, int64_t n_src, float* de, float* grad_x,
                                                             int64_t ldgx, float* grad_a_src, float* grad_a_dst,
                                                             int64_t n_entries, void* workspace, size_t workspace_bytes,
                                                             void* stream)
{
  if (n_entries < 0) return WHOLEMEMORY_INVALID_INPUT;
  return wgamd::gat_csr_bwd("wgamd_gat_csr_bwd_f32_v2", row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope, alpha,
                            grad_out, ldg, row_ptr_t, edge_perm, edge_dst, n_src, de, grad_x, ldgx, grad_a_src, grad_a_dst,
                            n_entries, workspace, workspace_bytes, stream);
}
