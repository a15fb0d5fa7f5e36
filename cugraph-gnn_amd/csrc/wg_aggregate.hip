// Mini-batch aggregation over the sampler's per-hop CSR: segmented sum/mean SpMM (GraphSAGE) and
// edge-softmax SDDMM + weighted SpMM (GAT), hand-written for gfx950.
//
// The reference has no such kernel (it calls torch_geometric.nn.SAGEConv/GATConv; call sites
// /root/reference/python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59,178-199).
// Semantics are PyG's public formulas, fp32; see include/wgamd_ext.h.
//
// Roofline: HBM-bound gather-reduce.  Algorithmic bytes per layer:
//   SpMM  E*(4F+4) + N_dst*(4F+8);   flops E*F  (0.25 flop/byte -> no MFMA here; the dense
//   lin_l/lin_r tail is a hipBLASLt GEMM in the host layer).
// Layout: one power-of-two lane group per destination row, 16 B (float4) per lane along the
// feature axis (F=100: 25 of 32 lanes, two rows per wave64; F=128: 32; F=256: the whole wave).
// The row's neighbour ids are fetched with ONE coalesced load per group and broadcast with
// bpermute, so the feature-row loads of different neighbours are independent (no
// col -> x dependent-load chain); 4 neighbour rows are kept in flight per lane.
// Sums run in CSR order => bit-identical to a sequential fp32 loop.
#include "wg_common.hpp"

namespace wgamd {
namespace {

template <typename IdT>
__device__ __forceinline__ int64_t src_row(const IdT* src_ids, int c)
{
  return (int64_t)src_ids[c];
}
template <>
__device__ __forceinline__ int64_t src_row<void>(const void*, int c)
{
  return (int64_t)c;
}

// neighbour rows a lane group has in flight before it adds them up: a row is one 16-B load per lane and ~2 us away
constexpr int kSpmmRowsInFlight = 4;   // 8 costs the F = 100 launch 6 % (registers), and buys the F = 256 one nothing

// Segmented mode (the backward pass gathers over the TRANSPOSED hop, which is power-law: a hub source is a neighbour of
// thousands of sampled rows, and a row is walked by one lane group — the launch would last as long as its longest row).
// Row r < n_main sums only its first `seg` entries; every further piece of a long row is an EXTRA row n_main + x with its
// own [start, end) and its sum goes to partial[x]; segment_addup_kernel then adds the pieces of a long row to its first one
// in order — deterministic whatever slots the plan kernel's atomics hand out.
struct spmm_segments {
  int seg;                  // entries per piece; 0 = plain CSR
  int64_t n_main;
  const int* extra_start;
  const int* extra_end;
  const int* n_extra_dev;   // extras in use
  float* partial;
  int64_t ldp;
};
struct long_row {
  int row, base, pieces;
};

__global__ void __launch_bounds__(256) segment_plan_kernel(const int* __restrict__ row_ptr, int64_t n_rows, int seg,
                                                           int* __restrict__ counters /*[0] extras, [1] long rows*/,
                                                           int* __restrict__ extra_start, int* __restrict__ extra_end,
                                                           long_row* __restrict__ long_rows)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int s = row_ptr[r], e = row_ptr[r + 1];
  if (e - s <= seg) return;
  const int pieces = (e - s + seg - 1) / seg - 1;   // beyond the first
  const int base   = atomicAdd(counters, pieces);
  long_rows[atomicAdd(counters + 1, 1)] = long_row{(int)r, base, pieces};
  for (int j = 0; j < pieces; j++) {
    extra_start[base + j] = s + (j + 1) * seg;
    extra_end[base + j]   = min(e, s + (j + 2) * seg);
  }
}

// one wave per long row: out[row, :] += partial[base, :] + partial[base + 1, :] + ... (in this order)
__global__ void __launch_bounds__(256) segment_addup_kernel(const long_row* __restrict__ long_rows, const int* __restrict__ counters,
                                                            const float* __restrict__ partial, int64_t ldp, int F,
                                                            float* __restrict__ out, int64_t ldo)
{
  const int lane = threadIdx.x & 63;
  for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; k < counters[1]; k += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const long_row lr = long_rows[k];
    for (int f = lane; f < F; f += 64) {
      float acc = out[(int64_t)lr.row * ldo + f];
      for (int j = 0; j < lr.pieces; j++) acc += partial[(int64_t)(lr.base + j) * ldp + f];
      out[(int64_t)lr.row * ldo + f] = acc;
    }
  }
}

// VEC = 4 (float4 path: F % 4 == 0, 16 B aligned rows) or 1.
template <int VEC, typename IdT, bool SEG = false>
__global__ void __launch_bounds__(256) spmm_csr_kernel(const int* __restrict__ row_ptr,
                                                       const int* __restrict__ col,
                                                       int64_t n_rows,
                                                       const float* __restrict__ x,
                                                       int64_t ldx,
                                                       int F,
                                                       const IdT* __restrict__ src_ids,
                                                       int mean,
                                                       float* __restrict__ out,
                                                       int64_t ldo,
                                                       int log2_lanes,
                                                       const int64_t* __restrict__ self_rows,
                                                       spmm_segments sg)
{
  const int lanes       = 1 << log2_lanes;
  const int lane        = threadIdx.x & 63;
  const int sub         = lane & (lanes - 1);
  const int gbase       = lane & ~(lanes - 1);  // first lane of my group inside the wave
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;

  // grid-stride over rows; all lanes of a wave iterate together (rows past the end idle) so the
  // bpermute broadcasts below are executed by the full wave.
  const int64_t rows_per_iter = ngroups;
  const int64_t iters         = (n_rows + rows_per_iter - 1) / rows_per_iter;
  for (int64_t it = 0; it < iters; it++) {
    const int64_t row = group + it * rows_per_iter;
    int s = 0, e = 0;
    bool writes = row < n_rows;
    if (row < n_rows) {
      if (!SEG || row < sg.n_main) {
        s = row_ptr[row];
        e = row_ptr[row + 1];
        if constexpr (SEG) e = min(e, s + sg.seg);   // segmented: the first piece of the row; the rest are extra "rows"
      } else {
        const int64_t xs = row - sg.n_main;
        writes           = xs < (int64_t)*sg.n_extra_dev;
        if (writes) {
          s = sg.extra_start[xs];
          e = sg.extra_end[xs];
        }
      }
    }
    const int deg = e - s;
    for (int f0 = sub * VEC; f0 < ((F + lanes * VEC - 1) / (lanes * VEC)) * (lanes * VEC); f0 += lanes * VEC) {
      const bool live = f0 < F;
      float acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) acc[v] = 0.f;
      // longest row in this wave decides the trip count of the broadcast loop
      int maxdeg = deg;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, d, 64));
      for (int c0 = 0; c0 < maxdeg; c0 += lanes) {
        // one coalesced fetch of up to `lanes` neighbour ids of my row
        int my_c = (c0 + sub < deg) ? col[s + c0 + sub] : 0;
        int64_t my_src = (c0 + sub < deg) ? src_row<IdT>(src_ids, my_c) : 0;
        const int chunk = min(lanes, maxdeg - c0);
        for (int j0 = 0; j0 < chunk; j0 += kSpmmRowsInFlight) {
          float vals[kSpmmRowsInFlight][VEC];
          bool ok[kSpmmRowsInFlight];
#pragma unroll
          for (int k = 0; k < kSpmmRowsInFlight; k++) {
            // broadcast neighbour (j0+k) of my group's row
            int src_lane  = gbase | ((j0 + k) & (lanes - 1));
            int lo        = __shfl((int)(my_src & 0xffffffff), src_lane, 64);
            int hi        = __shfl((int)(my_src >> 32), src_lane, 64);
            int64_t r     = ((int64_t)hi << 32) | (uint32_t)lo;
            ok[k]         = live && (j0 + k < chunk) && (c0 + j0 + k < deg);
            if (ok[k]) {
              const float* p = x + r * ldx + f0;
              if constexpr (VEC == 4) {
                float4 t   = *reinterpret_cast<const float4*>(p);
                vals[k][0] = t.x; vals[k][1] = t.y; vals[k][2] = t.z; vals[k][3] = t.w;
              } else {
                vals[k][0] = *p;
              }
            }
          }
#pragma unroll
          for (int k = 0; k < kSpmmRowsInFlight; k++) {
            if (ok[k]) {
#pragma unroll
              for (int v = 0; v < VEC; v++) acc[v] += vals[k][v];
            }
          }
        }
      }
      if (live && writes) {
        const float denom = (mean && deg > 0) ? (float)deg : 1.0f;
        float* q          = (SEG && row >= sg.n_main) ? sg.partial + (row - sg.n_main) * sg.ldp + f0 : out + row * ldo + f0;
        if constexpr (VEC == 4) {
          *reinterpret_cast<float4*>(q) = make_float4(acc[0] / denom, acc[1] / denom, acc[2] / denom, acc[3] / denom);
        } else {
          q[0] = acc[0] / denom;
        }
        if (self_rows != nullptr) {
          // SAGEConv root term: out[row, F:2F] = x[self_rows[row], :]  ->  one GEMM over [mean | self]
          const float* p = x + src_row<IdT>(src_ids, (int)self_rows[row]) * ldx + f0;
          if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(q + F) = *reinterpret_cast<const float4*>(p);
          } else {
            q[F] = p[0];
          }
        }
      }
    }
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) spmm_csr_bwd_kernel(const int* __restrict__ row_ptr,
                                                           const int* __restrict__ col,
                                                           int64_t n_rows,
                                                           const float* __restrict__ g,
                                                           int64_t ldg,
                                                           int F,
                                                           int mean,
                                                           float* __restrict__ gx,
                                                           int64_t ldx,
                                                           int log2_lanes)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  for (int64_t row = group; row < n_rows; row += ngroups) {
    const int s = row_ptr[row], e = row_ptr[row + 1];
    if (e <= s) continue;
    const float scale = mean ? 1.0f / (float)(e - s) : 1.0f;
    for (int f0 = sub * VEC; f0 < F; f0 += lanes * VEC) {
      float gv[VEC];
      if constexpr (VEC == 4) {
        float4 t = *reinterpret_cast<const float4*>(g + row * ldg + f0);
        gv[0] = t.x * scale; gv[1] = t.y * scale; gv[2] = t.z * scale; gv[3] = t.w * scale;
      } else {
        gv[0] = g[row * ldg + f0] * scale;
      }
      for (int j = s; j < e; j++) {
        float* q = gx + (int64_t)col[j] * ldx + f0;
#pragma unroll
        for (int v = 0; v < VEC; v++) atomicAdd(q + v, gv[v]);
      }
    }
  }
}

// GAT: one lane group per destination row, lane -> VEC consecutive channels of one head;
// single pass over the neighbour rows with an online (running max / running sum) softmax.
template <int VEC>
__global__ void __launch_bounds__(256) gat_csr_kernel(const int* __restrict__ row_ptr,
                                                      const int* __restrict__ col,
                                                      int64_t n_rows,
                                                      const float* __restrict__ x,
                                                      int64_t ldx,
                                                      const float* __restrict__ a_src,
                                                      const float* __restrict__ a_dst,
                                                      int H,
                                                      int C,
                                                      float slope,
                                                      float* __restrict__ alpha_out,
                                                      float* __restrict__ out,
                                                      int64_t ldo,
                                                      int log2_lanes)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const int HC          = H * C;
  for (int64_t row = group; row < n_rows; row += ngroups) {
    const int s = row_ptr[row], e = row_ptr[row + 1];
    for (int f0 = sub * VEC; f0 < HC; f0 += lanes * VEC) {
      const int h    = f0 / C;  // VEC divides C on the VEC=4 path, so all VEC channels share a head
      const float ad = a_dst[row * H + h];
      float m = -INFINITY, d = 0.f;
      float acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) acc[v] = 0.f;
      for (int j = s; j < e; j++) {
        const int c   = col[j];
        float sc      = a_src[(int64_t)c * H + h] + ad;
        sc            = sc > 0.f ? sc : sc * slope;
        const float mn = fmaxf(m, sc);
        const float rescale = expf(m - mn);  // exp(-inf) = 0 on the first edge
        const float p  = expf(sc - mn);
        d              = d * rescale + p;
        const float* xp = x + (int64_t)c * ldx + f0;
        if constexpr (VEC == 4) {
          float4 t = *reinterpret_cast<const float4*>(xp);
          acc[0] = acc[0] * rescale + p * t.x;
          acc[1] = acc[1] * rescale + p * t.y;
          acc[2] = acc[2] * rescale + p * t.z;
          acc[3] = acc[3] * rescale + p * t.w;
        } else {
          acc[0] = acc[0] * rescale + p * xp[0];
        }
        m = mn;
      }
      const float inv = e > s ? 1.0f / d : 0.f;
      float* q        = out + row * ldo + f0;
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(q) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
      } else {
        q[0] = acc[0] * inv;
      }
      // the first lane of every head also writes the attention coefficients
      if (alpha_out != nullptr && (f0 % C) == 0) {
        for (int j = s; j < e; j++) {
          float sc = a_src[(int64_t)col[j] * H + h] + ad;
          sc       = sc > 0.f ? sc : sc * slope;
          alpha_out[(int64_t)j * H + h] = expf(sc - m) * inv;
        }
      }
    }
  }
}

// GAT, HBM-bound version (H*C % 4 == 0, H*C <= 256): one lane group per destination row, one float4 of the H*C row per
// lane.  The row's neighbour ids are fetched with ONE coalesced load per group of edges and broadcast by shuffle (no
// col -> a_src / x dependent chain per edge), and FOUR neighbour rows + their head scores are in flight per lane group;
// the online softmax takes the four scores in one update (one rescale per four edges).
// ROWS: the rows of this launch are a subset of a larger destination list (a heterogeneous hop: the frontier entries of one
// hop and edge type) — `dst_rows[i]` is row i's place in a_dst / out; `accumulate`: out += (HeteroConv sums the relations
// that end in one node type; launches are stream-ordered, so no two of them touch a row at the same time).
template <bool ROWS>
__global__ void __launch_bounds__(256)
gat_csr_v4_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col, int64_t n_rows, const float* __restrict__ x,
                  int64_t ldx, const float* __restrict__ a_src, const float* __restrict__ a_dst, int H, int C, float slope,
                  const int64_t* __restrict__ dst_rows, int accumulate, float* __restrict__ alpha_out,
                  float* __restrict__ out, int64_t ldo, int log2_lanes)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int gbase       = (int)(threadIdx.x & 63) & ~(lanes - 1);
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const int HC          = H * C;
  const bool live       = sub * 4 < HC;
  const int f0          = live ? sub * 4 : 0;
  const int h           = f0 / C;
  for (int64_t row = group; row < n_rows; row += ngroups) {
    const int s = row_ptr[row], e = row_ptr[row + 1];
    const int64_t orow = ROWS ? dst_rows[row] : row;
    const float ad     = a_dst[orow * H + h];
    float m = -INFINITY, d = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = s; c0 < e; c0 += lanes) {
      const int mine = c0 + sub < e ? col[c0 + sub] : 0;
      const int cnt  = min(lanes, e - c0);
      for (int k = 0; k < cnt; k += 4) {
        int idx[4];
        float4 t[4];
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) idx[u] = __shfl(mine, gbase | min(k + u, cnt - 1), 64);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          t[u]  = *reinterpret_cast<const float4*>(x + (int64_t)idx[u] * ldx + f0);
          sc[u] = a_src[(int64_t)idx[u] * H + h];
        }
        float mn = m;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          float v = sc[u] + ad;
          v       = v > 0.f ? v : v * slope;
          sc[u]   = k + u < cnt ? v : -INFINITY;   // a slot past the chunk repeats its last edge: weight exp(-inf) = 0
          mn      = fmaxf(mn, sc[u]);
        }
        const float rescale = expf(m - mn);      // exp(-inf) = 0 on the first edges
        float psum = 0.f;
        acc.x *= rescale; acc.y *= rescale; acc.z *= rescale; acc.w *= rescale;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float pu = expf(sc[u] - mn);
          psum += pu;
          acc.x += pu * t[u].x; acc.y += pu * t[u].y; acc.z += pu * t[u].z; acc.w += pu * t[u].w;
        }
        d = d * rescale + psum;
        m = mn;
      }
    }
    const float inv = e > s ? 1.0f / d : 0.f;
    if (live) {
      float4* q = reinterpret_cast<float4*>(out + orow * ldo + f0);
      float4 r  = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
      if (accumulate) {
        const float4 o = *q;
        r = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
      }
      *q = r;
      // the first lane of every head also writes the attention coefficients
      if (alpha_out != nullptr && (f0 % C) == 0) {
        for (int j = s; j < e; j++) {
          float v = a_src[(int64_t)col[j] * H + h] + ad;
          v       = v > 0.f ? v : v * slope;
          alpha_out[(int64_t)j * H + h] = expf(v - m) * inv;
        }
      }
    }
  }
}

// GAT, AGGREGATE-FIRST (sampled hops: far fewer destination rows than source rows).  The attention-weighted sum is linear,
//   out[i, h, :] = sum_e alpha_e^h (W_h x_src(e)) = W_h (sum_e alpha_e^h x_src(e)),
// so the kernel aggregates the UNTRANSFORMED source rows per head — agg[i, h, :] = sum_e alpha_e^h x[col[e], :], F floats per
// head — and the dense transform runs afterwards over the destination rows only (H small [n_rows, F] x [F, C] GEMMs).  A
// mini-batch hop has 10-20x fewer destinations than sources, so the lin GEMM over every source row — the dominant cost of the
// transform-first formulation — disappears, and an edge moves F floats instead of H*C.
// One lane group (F/4 lanes, one float4 of the row per lane) per destination row; neighbour ids by one coalesced load per
// chunk + shuffle; EIF neighbour rows and their H scores in flight; one online softmax per head.
template <int H, int EIF>
__global__ void __launch_bounds__(256)
gat_aggregate_heads_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col, int64_t n_rows,
                           const float* __restrict__ x, int64_t ldx, int F, const float* __restrict__ a_src,
                           const float* __restrict__ a_dst, float slope, const int64_t* __restrict__ dst_rows,
                           float* __restrict__ out, int64_t ldo, int log2_lanes, const int64_t* __restrict__ src_ids,
                           const int64_t* __restrict__ dst_ids, int terms_by_id)
{
  // terms_by_id bit 0: a_src holds the terms of the TABLE's rows (row src_ids[j]); bit 1: a_dst likewise (row dst_ids[dst])
  // src_ids (nullable): neighbour j's row of x is src_ids[j] — x is then the feature table itself and src_ids the node list
  // of the call group (fetch in the layer); the attention terms stay indexed by j
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int gbase       = (int)(threadIdx.x & 63) & ~(lanes - 1);
  const int64_t group   = tid >> log2_lanes;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const bool live       = sub * 4 < F;
  const int f0          = live ? sub * 4 : 0;
  for (int64_t row = group; row < n_rows; row += ngroups) {
    const int s = row_ptr[row], e = row_ptr[row + 1];
    int64_t arow = dst_rows ? dst_rows[row] : row;
    if (terms_by_id & 2) arow = dst_ids[arow];
    float ad[H], m[H], d[H];
    float4 acc[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
      ad[h]  = a_dst[arow * H + h];
      m[h]   = -INFINITY;
      d[h]   = 0.f;
      acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int c0 = s; c0 < e; c0 += lanes) {
      const int mine = c0 + sub < e ? col[c0 + sub] : 0;
      const int64_t mine_row = src_ids ? src_ids[mine] : (int64_t)mine;   // (one coalesced-by-chunk load per lane, shuffled below)
      const int cnt  = min(lanes, e - c0);
      for (int k = 0; k < cnt; k += EIF) {
        int idx[EIF];
        int64_t xr[EIF];
        float4 t[EIF];
        float sc[EIF][H];
#pragma unroll
        for (int u = 0; u < EIF; u++) {
          const int from = gbase | min(k + u, cnt - 1);
          idx[u]         = __shfl(mine, from, 64);
          xr[u]          = src_ids ? (((int64_t)__shfl((int)(mine_row >> 32), from, 64) << 32) | (uint32_t)__shfl((int)mine_row, from, 64))
                                   : (int64_t)idx[u];
        }
#pragma unroll
        for (int u = 0; u < EIF; u++) {
          t[u] = *reinterpret_cast<const float4*>(x + xr[u] * ldx + f0);
          const int64_t trow = (terms_by_id & 1) ? xr[u] : (int64_t)idx[u];
          if constexpr (H == 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(a_src + trow * 4);
            sc[u][0] = a4.x; sc[u][1] = a4.y; sc[u][2] = a4.z; sc[u][3] = a4.w;
          } else {
#pragma unroll
            for (int h = 0; h < H; h++) sc[u][h] = a_src[trow * H + h];
          }
        }
#pragma unroll
        for (int h = 0; h < H; h++) {
          float mn = m[h];
#pragma unroll
          for (int u = 0; u < EIF; u++) {
            float v  = sc[u][h] + ad[h];
            v        = v > 0.f ? v : v * slope;
            sc[u][h] = k + u < cnt ? v : -INFINITY;   // a slot past the chunk repeats its last edge with weight exp(-inf) = 0
            mn       = fmaxf(mn, sc[u][h]);
          }
          const float rescale = expf(m[h] - mn);
          float psum = 0.f;
          float4 a   = acc[h];
          a.x *= rescale; a.y *= rescale; a.z *= rescale; a.w *= rescale;
#pragma unroll
          for (int u = 0; u < EIF; u++) {
            const float pu = expf(sc[u][h] - mn);
            psum += pu;
            a.x += pu * t[u].x; a.y += pu * t[u].y; a.z += pu * t[u].z; a.w += pu * t[u].w;
          }
          acc[h] = a;
          d[h]   = d[h] * rescale + psum;
          m[h]   = mn;
        }
      }
    }
    if (live) {
#pragma unroll
      for (int h = 0; h < H; h++) {
        const float inv = e > s ? 1.0f / d[h] : 0.f;
        *reinterpret_cast<float4*>(out + row * ldo + (int64_t)h * F + f0) =
          make_float4(acc[h].x * inv, acc[h].y * inv, acc[h].z * inv, acc[h].w * inv);
      }
    }
  }
}

// out[dst_rows ? dst_rows[i] : i, :] = act(in[i, :] + bias): the tail of a HeteroConv layer (sum over relations done, bias,
// ReLU, rows placed in the destination type's compact list) in ONE pass instead of three torch passes + an index_copy.
__global__ void __launch_bounds__(256)
bias_act_rows_kernel(const float* __restrict__ in, int64_t ldi, int64_t n_rows, int C, const float* __restrict__ bias, int relu,
                     const int64_t* __restrict__ dst_rows, float* __restrict__ out, int64_t ldo)
{
  const int c4n = C / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * c4n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c     = (int)(i % c4n) * 4;
    float4 v        = *reinterpret_cast<const float4*>(in + r * ldi + c);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + c);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    const int64_t o = dst_rows ? dst_rows[r] : r;
    *reinterpret_cast<float4*>(out + o * ldo + c) = v;
  }
}

inline int lanes_log2_for(int units)
{
  int l = 0;
  while ((1 << l) < units && l < 6) l++;
  return l;
}

inline int grid_rows(int64_t n_rows, int log2_lanes)
{
  int64_t groups_per_block = 256 >> log2_lanes;
  int64_t blocks           = (n_rows + groups_per_block - 1) / groups_per_block;
  static const int per_cu = getenv("WGAMD_SPMM_WG_PER_CU") ? atoi(getenv("WGAMD_SPMM_WG_PER_CU")) : 16;
  if (blocks > 256 * (int64_t)per_cu) blocks = 256 * (int64_t)per_cu;
  return (int)(blocks < 1 ? 1 : blocks);
}

inline bool vec4_ok(const void* a, int64_t lda, const void* b, int64_t ldb, int F)
{
  return F % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(b) & 15) == 0;
}

}  // namespace
}  // namespace wgamd

extern "C" {

static wholememory_error_code_t spmm_entry(const char* name, const int* row_ptr, const int* col, int64_t n_rows,
                                           const float* x, int64_t ldx, int F, const void* src_ids,
                                           wholememory_dtype_t src_ids_dtype, int mean, float* out, int64_t ldo,
                                           const int64_t* self_rows, void* stream)
{
  using namespace wgamd;
  return guarded(name, [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && F > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && out, "null pointer");
    WG_REQUIRE_INPUT(src_ids == nullptr || src_ids_dtype == WHOLEMEMORY_DT_INT || src_ids_dtype == WHOLEMEMORY_DT_INT64,
                     "src_ids dtype must be INT|INT64");
    auto st        = static_cast<hipStream_t>(stream);
    const bool v4  = vec4_ok(x, ldx, out, ldo, F);
    const int l2   = lanes_log2_for(v4 ? F / 4 : F);
    const int grid = grid_rows(n_rows, l2);
#define WG_SPMM(VEC, IDT, IDP) \
  spmm_csr_kernel<VEC, IDT><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, F, IDP, mean, out, ldo, l2, self_rows, \
                                                  spmm_segments{})
    if (src_ids == nullptr) {
      if (v4) WG_SPMM(4, void, (const void*)nullptr); else WG_SPMM(1, void, (const void*)nullptr);
    } else if (src_ids_dtype == WHOLEMEMORY_DT_INT) {
      if (v4) WG_SPMM(4, int32_t, static_cast<const int32_t*>(src_ids)); else WG_SPMM(1, int32_t, static_cast<const int32_t*>(src_ids));
    } else {
      if (v4) WG_SPMM(4, int64_t, static_cast<const int64_t*>(src_ids)); else WG_SPMM(1, int64_t, static_cast<const int64_t*>(src_ids));
    }
#undef WG_SPMM
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_spmm_csr_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                            int64_t ldx, int F, const void* src_ids,
                                            wholememory_dtype_t src_ids_dtype, int mean, float* out, int64_t ldo,
                                            void* stream)
{
  return spmm_entry("wgamd_spmm_csr_f32", row_ptr, col, n_rows, x, ldx, F, src_ids, src_ids_dtype, mean, out, ldo,
                    nullptr, stream);
}

wholememory_error_code_t wgamd_sage_aggregate_fetch_f32(const int* row_ptr, const int* col, int64_t n_rows,
                                                        const float* table, int64_t ldt, int F, const void* src_ids,
                                                        wholememory_dtype_t src_ids_dtype, const int64_t* self_rows,
                                                        int mean, float* out, int64_t ldo, void* stream)
{
  if (self_rows == nullptr || src_ids == nullptr || ldo < 2 * (int64_t)F) {
    fprintf(stderr, "[wholegraph_amd] wgamd_sage_aggregate_fetch_f32: self_rows / src_ids is NULL or ldo < 2F\n");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  return spmm_entry("wgamd_sage_aggregate_fetch_f32", row_ptr, col, n_rows, table, ldt, F, src_ids, src_ids_dtype, mean,
                    out, ldo, self_rows, stream);
}

wholememory_error_code_t wgamd_sage_aggregate_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                  int64_t ldx, int F, const int64_t* self_rows, int mean, float* out,
                                                  int64_t ldo, void* stream)
{
  if (self_rows == nullptr || ldo < 2 * (int64_t)F) {
    fprintf(stderr, "[wholegraph_amd] wgamd_sage_aggregate_f32: self_rows is NULL or ldo < 2F\n");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  return spmm_entry("wgamd_sage_aggregate_f32", row_ptr, col, n_rows, x, ldx, F, nullptr, WHOLEMEMORY_DT_UNKNOWN,
                    mean, out, ldo, self_rows, stream);
}

constexpr int kSegmentEntries = 64;   // entries per piece of a long row

size_t wgamd_spmm_csr_segmented_workspace_bytes(int64_t n_entries, int F)
{
  if (n_entries < 0 || F <= 0) return 0;
  const size_t cap = (size_t)(n_entries / kSegmentEntries) + 1;   // pieces beyond the first, over all rows
  const size_t ldp = ((size_t)F + 3) / 4 * 4;
  return 256 + 2 * ((cap * sizeof(int) + 255) / 256 * 256) + (cap * sizeof(wgamd::long_row) + 255) / 256 * 256 +
         cap * ldp * sizeof(float) + 256;
}

wholememory_error_code_t wgamd_spmm_csr_segmented_f32(const int* row_ptr, const int* col, int64_t n_rows, int64_t n_entries,
                                                      const float* x, int64_t ldx, int F, float* out, int64_t ldo,
                                                      void* workspace, size_t workspace_bytes, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_spmm_csr_segmented_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && n_entries >= 0 && F > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && (col || n_entries == 0) && x && out && workspace, "null pointer");
    WG_REQUIRE_INPUT(workspace_bytes >= wgamd_spmm_csr_segmented_workspace_bytes(n_entries, F), "workspace too small");
    auto st           = static_cast<hipStream_t>(stream);
    const size_t cap  = (size_t)(n_entries / kSegmentEntries) + 1;
    const size_t ib   = (cap * sizeof(int) + 255) / 256 * 256;
    const int64_t ldp = ((int64_t)F + 3) / 4 * 4;
    char* ws          = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    int* counters     = reinterpret_cast<int*>(ws);
    int* extra_start  = reinterpret_cast<int*>(ws + 256);
    int* extra_end    = reinterpret_cast<int*>(ws + 256 + ib);
    auto* long_rows   = reinterpret_cast<long_row*>(ws + 256 + 2 * ib);
    float* partial    = reinterpret_cast<float*>(ws + 256 + 2 * ib + (cap * sizeof(long_row) + 255) / 256 * 256);
    WG_HIP_CHECK(hipMemsetAsync(counters, 0, 2 * sizeof(int), st));
    segment_plan_kernel<<<ceil_div(n_rows, 256), 256, 0, st>>>(row_ptr, n_rows, kSegmentEntries, counters, extra_start,
                                                               extra_end, long_rows);
    const bool v4  = vec4_ok(x, ldx, out, ldo, F);
    const int l2   = lanes_log2_for(v4 ? F / 4 : F);
    const int64_t n_all = n_rows + (int64_t)cap;
    const int grid = grid_rows(n_all, l2);
    spmm_segments sg{kSegmentEntries, n_rows, extra_start, extra_end, counters, partial, ldp};
    if (v4)
      spmm_csr_kernel<4, void, true><<<grid, 256, 0, st>>>(row_ptr, col, n_all, x, ldx, F, nullptr, 0, out, ldo, l2, nullptr, sg);
    else
      spmm_csr_kernel<1, void, true><<<grid, 256, 0, st>>>(row_ptr, col, n_all, x, ldx, F, nullptr, 0, out, ldo, l2, nullptr, sg);
    segment_addup_kernel<<<(int)std::min<size_t>((cap + 3) / 4, 4096), 256, 0, st>>>(long_rows, counters, partial, ldp, F, out, ldo);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_spmm_csr_bwd_f32(const int* row_ptr, const int* col, int64_t n_rows,
                                                const float* grad_out, int64_t ldg, int F, int mean, float* grad_x,
                                                int64_t ldx, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_spmm_csr_bwd_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && F > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && grad_out && grad_x, "null pointer");
    auto st        = static_cast<hipStream_t>(stream);
    const bool v4  = vec4_ok(grad_out, ldg, grad_x, ldx, F);
    const int l2   = lanes_log2_for(v4 ? F / 4 : F);
    const int grid = grid_rows(n_rows, l2);
    if (v4)
      spmm_csr_bwd_kernel<4><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, grad_out, ldg, F, mean, grad_x, ldx, l2);
    else
      spmm_csr_bwd_kernel<1><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, grad_out, ldg, F, mean, grad_x, ldx, l2);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_bias_act_rows_f32(const float* in, int64_t ldi, int64_t n_rows, int C, const float* bias, int relu,
                                                 const int64_t* dst_rows, float* out, int64_t ldo, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_bias_act_rows_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && C > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(in && out && ldi >= C && ldo >= C, "null pointer / leading dimension smaller than C");
    if (!vec4_ok(in, ldi, out, ldo, C) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15) != 0))
      throw logic_error("rows must be 16-byte aligned and C a multiple of 4");
    const int64_t work = n_rows * (C / 4);
    const int grid     = (int)std::min<int64_t>((work + 255) / 256, 256 * 32);
    bias_act_rows_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(in, ldi, n_rows, C, bias, relu, dst_rows, out, ldo);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_gat_aggregate_heads_ids_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                           int64_t ldx, const int64_t* src_ids, const int64_t* dst_ids,
                                                           int terms_by_id, int F, const float* a_src,
                                                           const float* a_dst, int H, float negative_slope,
                                                           const int64_t* dst_rows, float* out, int64_t ldo, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gat_aggregate_heads_ids_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && H > 0 && F > 0, "bad sizes");
    WG_REQUIRE_INPUT(terms_by_id >= 0 && terms_by_id <= 3 && (terms_by_id == 0 || src_ids) && (!(terms_by_id & 2) || dst_ids),
                     "terms_by_id needs src_ids (and dst_ids for bit 1)");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && a_src && a_dst && out, "null pointer");
    if (F % 4 != 0 || F > 256 || !(H == 1 || H == 2 || H == 4 || H == 8) || !vec4_ok(x, ldx, out, ldo, F) ||
        (H == 4 && (reinterpret_cast<uintptr_t>(a_src) & 15) != 0))
      throw logic_error(fmt("unsupported shape: F=%d (multiple of 4, <= 256), H=%d (1, 2, 4 or 8), 16-B aligned rows", F, H));
    WG_REQUIRE_INPUT(ldo >= (int64_t)H * F, "output rows hold H * F floats");
    auto st        = static_cast<hipStream_t>(stream);
    const int l2   = lanes_log2_for(F / 4);
    const int grid = grid_rows(n_rows, l2);
#define WG_GAT_AGG(HH, EE)                                                                                                 \
  gat_aggregate_heads_kernel<HH, EE><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, F, a_src, a_dst, negative_slope,     \
                                                          dst_rows, out, ldo, l2, src_ids, dst_ids, terms_by_id)
    switch (H) {
      case 1: WG_GAT_AGG(1, 4); break;
      case 2: WG_GAT_AGG(2, 4); break;
      case 4: WG_GAT_AGG(4, 4); break;
      default: WG_GAT_AGG(8, 2); break;
    }
#undef WG_GAT_AGG
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_gat_aggregate_heads_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                       int64_t ldx, int F, const float* a_src, const float* a_dst, int H,
                                                       float negative_slope, const int64_t* dst_rows, float* out, int64_t ldo,
                                                       void* stream)
{
  return wgamd_gat_aggregate_heads_ids_f32(row_ptr, col, n_rows, x, ldx, nullptr, nullptr, 0, F, a_src, a_dst, H, negative_slope, dst_rows, out,
                                           ldo, stream);
}

wholememory_error_code_t wgamd_gat_csr_rows_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                                int64_t ldx, const float* a_src, const float* a_dst, int H, int C,
                                                float negative_slope, const int64_t* dst_rows, int accumulate,
                                                float* alpha_out, float* out, int64_t ldo, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gat_csr_rows_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && H > 0 && C > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && a_src && a_dst && out, "null pointer");
    auto st        = static_cast<hipStream_t>(stream);
    const int HC   = H * C;
    const bool v4  = (C % 4 == 0) && vec4_ok(x, ldx, out, ldo, HC);
    const int l2   = lanes_log2_for(v4 ? HC / 4 : HC);
    const int grid = grid_rows(n_rows, l2);
    if (v4 && HC <= 256) {
      if (dst_rows)
        gat_csr_v4_kernel<true><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope, dst_rows,
                                                      accumulate, alpha_out, out, ldo, l2);
      else
        gat_csr_v4_kernel<false><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope, nullptr,
                                                       accumulate, alpha_out, out, ldo, l2);
    } else {
      if (dst_rows != nullptr || accumulate)
        throw logic_error("row indirection / accumulation need H*C % 4 == 0, H*C <= 256 and 16-B aligned rows");
      if (v4)
        gat_csr_kernel<4><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope,
                                                alpha_out, out, ldo, l2);
      else
        gat_csr_kernel<1><<<grid, 256, 0, st>>>(row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope,
                                                alpha_out, out, ldo, l2);
    }
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_gat_csr_f32(const int* row_ptr, const int* col, int64_t n_rows, const float* x,
                                           int64_t ldx, const float* a_src, const float* a_dst, int H, int C,
                                           float negative_slope, float* alpha_out, float* out, int64_t ldo,
                                           void* stream)
{
  return wgamd_gat_csr_rows_f32(row_ptr, col, n_rows, x, ldx, a_src, a_dst, H, C, negative_slope, nullptr, 0, alpha_out, out,
                                ldo, stream);
}

}  // extern "C"
