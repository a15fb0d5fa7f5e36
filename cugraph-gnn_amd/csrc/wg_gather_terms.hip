// Feature gather with a narrow product folded in:
//     out_x[i, :]     = table[ids[i], :]                      (the row gather of wholememory_gather, fp32)
//     out_terms[i, :] = table[ids[i], :] @ V                  (V [F, T], T <= 32)
// in ONE pass over the gathered rows.  What it is for: the attention logits of GATConv,
//     alpha_src[j, h] = ((x_j W).view(H, C) * att_src).sum(-1) = x_j . fold(W, att_src)[:, h]
// (torch_geometric.nn.GATConv as the reference's models use it, /root/reference/python/pylibwholegraph/pylibwholegraph/torch/
// gnn_model.py:45-59), need one [F, H] product per relation end and node type over EVERY gathered row: for the ogbn-mag
// call group (BASELINE configs[4], bench_mag.py) that was a second pass over 4 GB of x — a library GEMM with N = 8..20,
// 1.9 ms next to the 1.4 ms gather.  Here the rows pass through registers once.
//
// The product runs on the matrix pipe in exact fp32 (v_mfma_f32_16x16x4_f32), computed TRANSPOSED so that the row gather
// itself produces the operand layout: out^T [T x rows] = V^T [T x F] . x^T [F x rows].  A wave takes 16 rows at a time, four
// lanes per row; lane (r, q) loads the 16-byte chunks q, q + 4, q + 8, ... of row r (64 contiguous bytes of a row per
// load instruction) and stores them to out_x as they are.  The B operand of a 16x16x4 MFMA wants B[k = lane / 16][col = lane % 16]:
// with col = the row r and the k-order of step (m, i) chosen as k = 16 m + 4 q + i, that is exactly float i of the lane's
// chunk m — no shuffle, no LDS.  The A operand (V^T in the same k-order) sits in registers for the whole launch.  The
// accumulator tile D[t = 4 (lane / 16) + v][r = lane % 16] leaves as one 16-byte store per lane: out_terms[r, 4 q' .. 4 q' + 3].
#include <cstdlib>

#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// KM = F / 16 chunks per lane, TT = 16-term tiles (T <= 16 TT)
template <typename IdT, int KM, int TT>
__global__ void __launch_bounds__(256)
gather_terms_kernel(const float* __restrict__ table, int64_t ldt, const IdT* __restrict__ ids, int64_t n, const float* __restrict__ v,
                    int T, float* __restrict__ out_x, int64_t ldx, float* __restrict__ out_terms, int64_t ldo, int group)
{
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  // A operand: a[tt][m][i] = V[16 m + 4 q + i][16 tt + r]  (zero past T)
  float a[TT][KM][4];
#pragma unroll
  for (int tt = 0; tt < TT; tt++) {
    const int t = tt * 16 + r;
#pragma unroll
    for (int m = 0; m < KM; m++)
#pragma unroll
      for (int i = 0; i < 4; i++) a[tt][m][i] = t < T ? v[(int64_t)(16 * m + 4 * q + i) * T + t] : 0.f;
  }
  const int64_t n_tiles  = (n + 15) / 16;
  const int64_t wave     = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves  = (int64_t)gridDim.x * (blockDim.x >> 6);
  // the id of the NEXT tile's row is requested before this tile's rows are waited for: one dependent round trip per tile, not two
  int64_t id_next = (wave < n_tiles && wave * 16 + r < n) ? (int64_t)ids[wave * 16 + r] : -1;
  for (int64_t tile = wave; tile < n_tiles; tile += n_waves) {
    const int64_t row  = tile * 16 + r;
    const bool in      = row < n;
    const int64_t id   = id_next;
    {
      const int64_t nrow = (tile + n_waves) * 16 + r;
      id_next            = (tile + n_waves < n_tiles && nrow < n) ? (int64_t)ids[nrow] : -1;
    }
    const bool live    = id >= 0;                         // a negative id: the row is skipped, its terms are zero
    const float* src   = table + (live ? id : 0) * ldt + 4 * q;
    f32x4 x[KM];
#pragma unroll
    for (int m = 0; m < KM; m++) x[m] = *reinterpret_cast<const f32x4*>(src + 16 * m);
    if (live) {
      float* dst = out_x + row * ldx + 4 * q;
#pragma unroll
      for (int m = 0; m < KM; m++) *reinterpret_cast<f32x4*>(dst + 16 * m) = x[m];
    }
    f32x4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; tt++) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < KM; m++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float b = live ? x[m][i] : 0.f;
#pragma unroll
        for (int tt = 0; tt < TT; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][m][i], b, acc[tt], 0, 0, 0);
      }
    // D[t = 16 tt + 4 q + v][row r]: four consecutive terms of one row per lane
    if (in) {
#pragma unroll
      for (int tt = 0; tt < TT; tt++) {
        const int t0 = tt * 16 + 4 * q;
        if (group == 4) {   // slabs [T / 4][n][4]: the four terms of one (row, relation end) are one 16-byte store
          if (t0 < T) *reinterpret_cast<f32x4*>(out_terms + ((int64_t)(t0 >> 2) * n + row) * 4) = acc[tt];
          continue;
        }
        float* o = out_terms + row * ldo + t0;
        if (t0 + 4 <= T && (ldo & 3) == 0) {
          *reinterpret_cast<f32x4*>(o) = acc[tt];
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (t0 + i < T) o[i] = acc[tt][i];
        }
      }
    }
  }
}

// The same tile step, software-pipelined: the A operand (V^T) lives in LDS (one ds_read_b128 per (term tile, chunk) — 64
// registers back for F = 128, T > 16), which pays for a SECOND set of row registers: the loads of tile t + 1 are in flight
// while tile t is stored and multiplied (112-115 registers, four waves per SIMD with two tiles each).  Same MFMA order,
// bit-identical results; the default (WGAMD_GATHER_TERMS_PIPELINED=0 selects the kernel above).  Same box, ogbn-mag call
// group (tools/bench_gather_terms.py): paper rows (10.0 M, T = 24) 2.35 -> 2.12 ms, author (3.85 M, T = 12) 0.85 -> 0.79,
// field_of_study (2.25 M, T = 8) 0.46 -> 0.43.  Measured without effect on top: non-temporal stores of the rows, 4 / 6 / 16
// workgroups per CU instead of 8.
template <typename IdT, int KM, int TT>
__global__ void __launch_bounds__(256)
gather_terms_pipelined_kernel(const float* __restrict__ table, int64_t ldt, const IdT* __restrict__ ids, int64_t n,
                              const float* __restrict__ v, int T, float* __restrict__ out_x, int64_t ldx,
                              float* __restrict__ out_terms, int64_t ldo, int group)
{
  __shared__ f32x4 a_lds[TT * KM * 64];
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  if (threadIdx.x < 64) {   // a_lds[(tt KM + m) 64 + lane] = V[16 m + 4 q + 0..3][16 tt + r]  (zero past T)
#pragma unroll
    for (int tt = 0; tt < TT; tt++) {
      const int t = tt * 16 + r;
#pragma unroll
      for (int m = 0; m < KM; m++) {
        f32x4 a;
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = t < T ? v[(int64_t)(16 * m + 4 * q + i) * T + t] : 0.f;
        a_lds[(tt * KM + m) * 64 + lane] = a;
      }
    }
  }
  __syncthreads();
  const int64_t n_tiles = (n + 15) / 16;
  const int64_t wave    = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  auto id_of = [&](int64_t tile) -> int64_t {
    const int64_t row = tile * 16 + r;
    return (tile < n_tiles && row < n) ? (ids != nullptr ? (int64_t)ids[row] : row) : -1;   // no id list: the rows as they lie
  };
  auto fetch = [&](int64_t id, f32x4 (&x)[KM]) {
    const float* src = table + (id >= 0 ? id : 0) * ldt + 4 * q;
#pragma unroll
    for (int m = 0; m < KM; m++) x[m] = *reinterpret_cast<const f32x4*>(src + 16 * m);
  };
  auto finish = [&](int64_t tile, int64_t id, const f32x4 (&x)[KM]) {
    const int64_t row = tile * 16 + r;
    const bool live   = id >= 0;
    if (live && out_x != nullptr) {   // (no out_x: only the terms are wanted)
      float* dst = out_x + row * ldx + 4 * q;
#pragma unroll
      for (int m = 0; m < KM; m++) *reinterpret_cast<f32x4*>(dst + 16 * m) = x[m];
    }
    f32x4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; tt++) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int slot = lane;
    asm volatile("" : "+v"(slot));   // opaque per tile: the operand is re-read from LDS, not hoisted back into 64 registers
#pragma unroll
    for (int m = 0; m < KM; m++) {
      f32x4 a[TT];
#pragma unroll
      for (int tt = 0; tt < TT; tt++) a[tt] = a_lds[(tt * KM + m) * 64 + slot];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float b = live ? x[m][i] : 0.f;
#pragma unroll
        for (int tt = 0; tt < TT; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][i], b, acc[tt], 0, 0, 0);
      }
    }
    if (row < n) {
#pragma unroll
      for (int tt = 0; tt < TT; tt++) {
        const int t0 = tt * 16 + 4 * q;
        if (group == 4) {
          if (t0 < T) *reinterpret_cast<f32x4*>(out_terms + ((int64_t)(t0 >> 2) * n + row) * 4) = acc[tt];
          continue;
        }
        float* o = out_terms + row * ldo + t0;
        if (t0 + 4 <= T && (ldo & 3) == 0) {
          *reinterpret_cast<f32x4*>(o) = acc[tt];
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (t0 + i < T) o[i] = acc[tt][i];
        }
      }
    }
  };
  // tiles wave, wave + n_waves, ...: rows of tile k + 1 are requested before tile k is stored and multiplied; the ids run
  // one more tile ahead of the rows
  f32x4 x0[KM], x1[KM];
  int64_t tile = wave;
  int64_t id0 = id_of(tile), id1 = id_of(tile + n_waves);
  if (tile < n_tiles) fetch(id0, x0);
  while (tile < n_tiles) {
    int64_t id2 = id_of(tile + 2 * n_waves);
    if (tile + n_waves < n_tiles) fetch(id1, x1);
    finish(tile, id0, x0);
    tile += n_waves;
    if (tile >= n_tiles) break;
    id0 = id_of(tile + 2 * n_waves);
    if (tile + n_waves < n_tiles) fetch(id2, x0);
    finish(tile, id1, x1);
    tile += n_waves;
    id1 = id0;
    id0 = id2;
  }
}

template <typename IdT, int KM>
void launch_tt(int TT, int grid, hipStream_t st, const float* table, int64_t ldt, const IdT* ids, int64_t n, const float* v, int T,
               float* out_x, int64_t ldx, float* out_terms, int64_t ldo, int group)
{
  static const bool pipelined = [] {
    const char* e = getenv("WGAMD_GATHER_TERMS_PIPELINED");
    return e == nullptr || e[0] != '0';
  }();
  if (pipelined || ids == nullptr || out_x == nullptr) {
    if (TT == 1) gather_terms_pipelined_kernel<IdT, KM, 1><<<grid, 256, 0, st>>>(table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group);
    else gather_terms_pipelined_kernel<IdT, KM, 2><<<grid, 256, 0, st>>>(table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group);
    return;
  }
  if (TT == 1) gather_terms_kernel<IdT, KM, 1><<<grid, 256, 0, st>>>(table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group);
  else gather_terms_kernel<IdT, KM, 2><<<grid, 256, 0, st>>>(table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group);
}

template <typename IdT>
void launch_km(int KM, int TT, int grid, hipStream_t st, const float* table, int64_t ldt, const IdT* ids, int64_t n, const float* v,
               int T, float* out_x, int64_t ldx, float* out_terms, int64_t ldo, int group)
{
  switch (KM) {
    case 2: launch_tt<IdT, 2>(TT, grid, st, table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group); break;
    case 4: launch_tt<IdT, 4>(TT, grid, st, table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group); break;
    case 8: launch_tt<IdT, 8>(TT, grid, st, table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group); break;
    default: launch_tt<IdT, 16>(TT, grid, st, table, ldt, ids, n, v, T, out_x, ldx, out_terms, ldo, group); break;
  }
}

// out[k][i][0..3] = in[k][ids[i]][0..3] for the K slabs of a node type's attention terms: the terms of a call group's rows
// taken from the terms of the TABLE's rows (a call group lists a table row once per mini-batch that sampled it: when the
// table is shorter than the group's node list, x @ v over the table + this 16-byte-row gather replaces x[ids] @ v over the
// list).  One lane per listed row: the id is read once, the K loads are independent, stores are coalesced per slab.
template <typename IdT>
__global__ void __launch_bounds__(256)
gather_term_slabs_kernel(const f32x4* __restrict__ in, int64_t n_in, int K, const IdT* __restrict__ ids, int64_t n,
                         f32x4* __restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = (int64_t)ids[i];
    const bool live  = id >= 0 && id < n_in;          // (a skipped row: zero terms, as wgamd_gather_terms_f32 writes)
    for (int k = 0; k < K; k++) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live) v = in[(int64_t)k * n_in + id];
      __builtin_nontemporal_store(v, out + (int64_t)k * n + i);
    }
  }
}

// dV [F, T] += X^T [F, n] . dT [n, T]: the weight gradient of `terms = X @ V` for a narrow V (T <= 32) and a LONG X (a whole
// feature table: K = n is the reduction).  A library GEMM sees an output of 128 x 12 elements and runs it at 0.5 TB/s of
// X; here every block streams its stretch of rows once — thread (row lane, feature f) reads X[row, f] (a coalesced row per
// lane group), the rows' dT values come from an LDS tile as broadcast float4 reads — and adds its [T] partial sums to dV with
// float atomics (F x T per block).  TT = ceil(T / 4) float4 accumulators per thread.
constexpr int kXtRows = 128;   // dT rows per LDS tile
template <int TT>
__global__ void __launch_bounds__(256)
rows_terms_bwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int F, int fp_log2, const float* __restrict__ dt, int T,
                      float* __restrict__ dv, int64_t rows_per_block)
{
  __shared__ f32x4 tile[kXtRows][TT];
  const int t  = threadIdx.x;
  const int fp = 1 << fp_log2, f = t & (fp - 1), rl = t >> fp_log2, n_rl = 256 >> fp_log2;
  const bool live = f < F;
  f32x4 acc[TT];
#pragma unroll
  for (int q = 0; q < TT; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int64_t base = r0; base < r1; base += kXtRows) {
    __syncthreads();   // the previous tile's readers are done
    for (int i = t; i < kXtRows * TT; i += 256) {
      const int64_t row = base + i / TT;
      const int q       = i % TT;
      f32x4 v           = {0.f, 0.f, 0.f, 0.f};
      if (row < r1) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (4 * q + k < T) v[k] = dt[row * T + 4 * q + k];
      }
      tile[i / TT][q] = v;
    }
    __syncthreads();
    const int rows = (int)(r1 - base < kXtRows ? r1 - base : kXtRows);
    if (live) {
      int r = rl;
      for (; r + 3 * n_rl < rows; r += 4 * n_rl) {   // four rows' loads in flight per lane
        float xv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = x[(base + r + u * n_rl) * ldx + f];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int q = 0; q < TT; q++) acc[q] += xv[u] * tile[r + u * n_rl][q];
      }
      for (; r < rows; r += n_rl) {
        const float xv = x[(base + r) * ldx + f];
#pragma unroll
        for (int q = 0; q < TT; q++) acc[q] += xv * tile[r][q];
      }
    }
  }
  if (live) {
#pragma unroll
    for (int q = 0; q < TT; q++)
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (4 * q + k < T) unsafeAtomicAdd(dv + (int64_t)f * T + 4 * q + k, acc[q][k]);
  }
}

}  // namespace
}  // namespace wgamd

extern "C" {

int wgamd_gather_terms_supported(int F, int T) { return (F == 32 || F == 64 || F == 128 || F == 256) && T > 0 && T <= 32; }

wholememory_error_code_t wgamd_gather_terms_f32(const float* table, int64_t ldt, const void* ids, wholememory_dtype_t id_dtype,
                                                int64_t n, int F, const float* v, int T, float* out_x, int64_t ldx,
                                                float* out_terms, int64_t ldo, int term_group, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gather_terms_f32", [&] {
    WG_REQUIRE_INPUT(id_dtype == WHOLEMEMORY_DT_INT || id_dtype == WHOLEMEMORY_DT_INT64, "id dtype must be INT|INT64");
    WG_REQUIRE_INPUT(n >= 0 && n < ((int64_t)1 << 40), "bad row count");
    if (!wgamd_gather_terms_supported(F, T)) throw logic_error("gather_terms: F must be 32 | 64 | 128 | 256 and 0 < T <= 32");
    if (n == 0) return;
    WG_REQUIRE_INPUT(table && v && out_terms, "null pointer");   // ids NULL: rows 0 .. n-1; out_x NULL: terms only
    WG_REQUIRE_INPUT(term_group == 0 || (term_group == 4 && T % 4 == 0), "term_group must be 0 (rows [n, T]) or 4 (slabs [T/4][n][4])");
    WG_REQUIRE_INPUT(ldt >= F && (out_x == nullptr || ldx >= F) && (term_group != 0 || ldo >= T), "leading dimension smaller than the row");
    if (out_x == nullptr) ldx = 4;
    if (((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out_x)) & 15) != 0 || (ldt & 3) != 0 || (ldx & 3) != 0 ||
        (reinterpret_cast<uintptr_t>(out_terms) & 15) != 0)
      throw logic_error("gather_terms: rows must be 16-byte aligned");
    hipStream_t st        = static_cast<hipStream_t>(stream);
    const int64_t n_tiles = (n + 15) / 16;
    const int cus         = stream_cu_count(st);
    const int grid        = (int)std::min<int64_t>((n_tiles + 3) / 4, (int64_t)cus * 8);
    const int KM = F / 16, TT = (T + 15) / 16;
    if (id_dtype == WHOLEMEMORY_DT_INT)
      launch_km<int32_t>(KM, TT, grid, st, table, ldt, static_cast<const int32_t*>(ids), n, v, T, out_x, ldx, out_terms, ldo, term_group);
    else
      launch_km<int64_t>(KM, TT, grid, st, table, ldt, static_cast<const int64_t*>(ids), n, v, T, out_x, ldx, out_terms, ldo, term_group);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_gather_term_slabs_f32(const float* slabs_in, int64_t n_in, int n_slabs, const void* ids,
                                                     wholememory_dtype_t id_dtype, int64_t n, float* slabs_out, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_gather_term_slabs_f32", [&] {
    WG_REQUIRE_INPUT(id_dtype == WHOLEMEMORY_DT_INT || id_dtype == WHOLEMEMORY_DT_INT64, "id dtype must be INT|INT64");
    WG_REQUIRE_INPUT(n >= 0 && n_in >= 0 && n_slabs >= 0 && n_slabs <= 64, "bad sizes");
    if (n == 0 || n_slabs == 0) return;
    WG_REQUIRE_INPUT(slabs_in && ids && slabs_out, "null pointer");
    if (((reinterpret_cast<uintptr_t>(slabs_in) | reinterpret_cast<uintptr_t>(slabs_out)) & 15) != 0)
      throw logic_error("gather_term_slabs: slabs must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)stream_cu_count(st) * 16);
    if (id_dtype == WHOLEMEMORY_DT_INT)
      gather_term_slabs_kernel<int32_t><<<grid, 256, 0, st>>>(reinterpret_cast<const f32x4*>(slabs_in), n_in, n_slabs,
                                                              static_cast<const int32_t*>(ids), n, reinterpret_cast<f32x4*>(slabs_out));
    else
      gather_term_slabs_kernel<int64_t><<<grid, 256, 0, st>>>(reinterpret_cast<const f32x4*>(slabs_in), n_in, n_slabs,
                                                              static_cast<const int64_t*>(ids), n, reinterpret_cast<f32x4*>(slabs_out));
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_rows_terms_bwd_f32(const float* x, int64_t ldx, int64_t n, int F, const float* dterms, int T,
                                                  float* dv, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_rows_terms_bwd_f32", [&] {
    WG_REQUIRE_INPUT(n >= 0 && F > 0 && F <= 256 && T > 0 && T <= 32 && ldx >= F, "bad sizes (F <= 256, T <= 32)");
    if (n == 0) return;
    WG_REQUIRE_INPUT(x && dterms && dv, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int fp_log2    = 0;
    while ((1 << fp_log2) < F) fp_log2++;
    const int64_t blocks_wanted = (int64_t)stream_cu_count(st) * 8;
    int64_t rows_per_block      = std::max<int64_t>(kXtRows, (n + blocks_wanted - 1) / blocks_wanted);
    rows_per_block              = (rows_per_block + kXtRows - 1) / kXtRows * kXtRows;
    const int grid              = (int)((n + rows_per_block - 1) / rows_per_block);
    const int TT                = (T + 3) / 4;
#define WG_XT(N_) rows_terms_bwd_kernel<N_><<<grid, 256, 0, st>>>(x, ldx, n, F, fp_log2, dterms, T, dv, rows_per_block)
    switch (TT) {
      case 1: WG_XT(1); break;
      case 2: WG_XT(2); break;
      case 3: WG_XT(3); break;
      case 4: WG_XT(4); break;
      case 5: WG_XT(5); break;
      case 6: WG_XT(6); break;
      case 7: WG_XT(7); break;
      default: WG_XT(8); break;
    }
#undef WG_XT
    WG_HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"
