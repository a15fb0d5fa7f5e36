// One kernel for a whole SAGEConv layer over a sampled hop, built to be HBM-bound:
//     out[i, :] = act( [ mean_{j in N(i)} x[j] | x[self(i)] ] @ [W_l | W_r]^T + b )
// (semantics of torch_geometric.nn.SAGEConv as the reference uses it,
//  python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59).
//
// Why a second one-kernel layer next to wg_sage_fused.hip: that kernel multiplies in exact fp32 on the matrix pipe
// (v_mfma_f32_16x16x4_f32 = the fp32 VECTOR rate, 157 TF/s), so the layer can never run faster than its ~0.36 ms of
// fp32 MFMA issue.  Here the fp32 product is evaluated on the bf16 pipe (16x the rate) with a 3-way split of BOTH
// operands:  a = a_hi + a_mid + a_lo  EXACTLY (an fp32 significand is 24 bits = 3 x 8; each piece is a bf16 by truncation,
// the residuals are exact in fp32), and
//     a * b ~= a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_mid b_mid + a_hi b_lo + a_lo b_hi)
// — the six products of weight >= 2^-16, accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three dropped products
// are <= 2^-23 |a b| in total: the class of fp32 round-off (checked against the fp64 oracle at 1e-5 x scale like the
// fp32 kernel, tests/test_gpu_aggregate.py).  Six bf16 MFMAs cost 6/16 of one fp32 MFMA.
//
// Structure (one workgroup per CU, persistent over 64-row tiles, two fp32 operand tiles in LDS):
//   * 4 PRODUCER waves fetch CSR bounds / neighbour ids / neighbour rows with branch-free 16-B loads kept in a register
//     ring, sum in CSR order and store the fp32 [mean | self] rows to the LDS tile of step s.  Neighbour slots past a row's
//     degree are BUFFER loads with an out-of-range offset: the hardware returns zeros without touching memory, so the
//     sum needs no per-element select.  Row metadata is software-pipelined ACROSS tiles (bounds at the start of the
//     previous tile, ids half-way, offsets at its end; the first rows of the next tile are requested before the barrier).
//   * CW = N/64 CONSUMER waves multiply the tile of step s-1: each owns 64 output columns = 2 x 2 accumulator tiles of
//     32 x 32.  They read fp32 A fragments (2 x ds_read_b128 per row tile and k-step) and do the 3-way split themselves,
//     in the issue slots under their own MFMAs (an MFMA holds the matrix pipe for 32 cycles but the issue port for ~4);
//     the weight travels as fp32 tiles laid out so that a wave-load is 2 x 1 KiB contiguous (split into its bf16 planes in
//     registers too: 4 B per element instead of the 6 of pre-split planes), and its fragment stream runs two k-steps ahead,
//     continuing across tiles.
//   * A producer and a consumer wave share each SIMD.  Measured on gfx950 (tools/tune/sage_mfma_harness.cpp, timeline of
//     s_memtime stamps): they time-slice rather than overlap — while a consumer streams MFMAs the co-resident producer runs
//     at ~25 % of its stand-alone speed — and the CU's vector-memory pipeline returns data in issue order across waves, so
//     the consumers' weight fragments (L2 hits) and output stores queue behind the producers' HBM row fetches.  Putting the
//     roles on different SIMDs (wave rank by HW_REG_HW_ID, `kRolesBySimd`) was tried: the two consumers of a SIMD then
//     saturate its matrix pipe and the step becomes consumer-bound; it is kept as a switch, off.  What is left on the
//     table: stand-alone the producers take 0.40 ms (5.6 TB/s of HBM traffic) and the consumers 0.27 ms for the products
//     layer-1 call group; together 0.63-0.75 ms depending on the box (fp32-MFMA kernel: 0.84 ms).
//   * Round 3, compile-time ablations inside bench.py: the multiplying waves' MATRIX work is free (half the
//     MFMAs: 0.572 -> 0.564 ms); what they cost is their MEMORY instructions, which share the CU's vector-memory pipeline
//     with the row fetches — no weight loads: 0.422, no output stores: 0.470, neither: 0.338 = the fetching waves alone
//     (0.332).  Hence the fp32 weight tiles (two thirds of the bytes of pre-split planes): 0.570 -> 0.516 ms.
//   * F > 148 (two [mean | self] tiles do not fit the LDS; the hidden-256 layers): only the MEAN half goes through the LDS,
//     the multiplying waves read the SELF half of A from global memory, their loop is software-pipelined (split of k-step
//     ks+1 dealt between the MFMAs of k-step ks by sched_group_barrier) and the weight planes run two k-steps ahead:
//     256 -> 256 layer 1.69 -> 1.51 ms (no weight loads at all: 1.34; the multiplying waves alone: 0.90, of which 0.43 is
//     the matrix pipe; the fetching waves alone: 0.73 = the HBM floor of the shape).
//   * one s_barrier per step behind an LDS-only wait (s_waitcnt lgkmcnt(0)): neither side's global loads are drained.
#include "wg_sage_mfma_parts.hpp"

namespace wgamd {
namespace {
using namespace sage_mfma;

// ---- runtime shape: weight fragments one k-step ahead ------------------------------------------------------------------
template <int TR>
__device__ __forceinline__ void consume_tile(const mfma_args& a, int64_t tile, const float* tile_lds, int cw, int lane,
                                             float* scratch)
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
  const float* a_lane      = tile_lds + lm * a.SD + lh * 8;
  const float* b_lane = a.w_tiles + ((int64_t)(cw * 64 + lm)) * 16 + lh * 8;
  braw_t b0, b1;
  bfrag_t fb;
  araw_t<RT> raw;
  afrag_t<RT> fa;
  load_b(b0, b_lane, a.N, 0);
  for (int ks = 0; ks < a.KS; ks += 2) {
    load_a_raw<RT>(raw, a_lane, a.SD, ks);
    if (ks + 1 < a.KS) load_b(b1, b_lane, a.N, ks + 1);
    split_a<RT>(raw, fa);
    split_b(b0, fb);
    mma_frags<RT>(c, fa, fb);
    if (ks + 1 < a.KS) {
      load_a_raw<RT>(raw, a_lane, a.SD, ks + 1);
      if (ks + 2 < a.KS) load_b(b0, b_lane, a.N, ks + 2);
      split_a<RT>(raw, fa);
      split_b(b1, fb);
      mma_frags<RT>(c, fa, fb);
    }
  }
  epilogue<RT>(a, c, tile * TR, cw, lane, scratch);
}

// Issue-order hint for one software-pipelined k-step of the runtime-shape multiplying loops: the 24 MFMAs of k-step ks with
// the VALU work of the NEXT k-step's fragment split dealt between them.  A wave issues in order and an MFMA blocks issue
// while the matrix pipe is busy with the one before (32 cycles each here); written as "24 MFMAs, then the split" the pipe
// idles for the whole split (~90 VALU instructions), as "MFMA, 5 VALU, MFMA, ..." the split is free.
template <int HEAD, int BODY, int VALU_PER>
__device__ __forceinline__ void mfma_valu_interleave()
{
#pragma unroll
  for (int i = 0; i < HEAD; i++) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
  for (int i = 0; i < BODY; i++) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);
  }
}

// ---- HALF mode, one range of k-steps with the weight fragments TWO k-steps ahead (ring of three) ---------------------------
// load_a(dst, ks) requests the fp32 A fragment of k-step ks (LDS: AD = 1 k-step ahead; global self rows: AD = 2).  The
// weight stream is one sequence over the whole kernel — k-step (ks + 2) mod KS is requested in step ks — and the ring is
// rotated back to position 0 when a range ends, so every range starts with b[0] = its first k-step, b[1] = its second.
template <int RT, int AD, typename LoadA>
__device__ __forceinline__ void consume_range(const mfma_args& a, f32x16 (&c)[RT][2], bfrag_t (&b)[3], int ks0, int ks1, int cw,
                                              int lane, LoadA load_a)
{
  const int lm = lane & 31, lh = lane >> 5;
  const int64_t b_plane_dw = (int64_t)a.KS * a.N * 8;
  const uint32_t* b_lane   = reinterpret_cast<const uint32_t*>(a.w_tiles) + ((int64_t)(cw * 64 + lm)) * 8 + lh * 4;
  araw_t<RT> raw[AD];
  afrag_t<RT> fa[2];
  const int kl = ks1 - 1;
  load_a(raw[0], ks0);
  if constexpr (AD == 2) load_a(raw[1], ks0 + 1 < kl ? ks0 + 1 : kl);
  split_a<RT>(raw[0], fa[0]);
  auto step = [&](auto I, int ks) {
    constexpr int i = decltype(I)::value;
    load_a(raw[i % AD], ks + AD < kl ? ks + AD : kl);
    int kb = ks + 2;
    kb     = kb >= a.KS ? kb - a.KS : kb;
    load_b_planes(b[(i + 2) % 3], b_lane, b_plane_dw, a.N, kb);
    mma_frags<RT>(c, fa[i & 1], b[i % 3]);
    split_a<RT>(raw[(i + 1) % AD], fa[(i + 1) & 1]);
    mfma_valu_interleave<4, 6 * RT * 2 - 4, 5>();
    __builtin_amdgcn_sched_barrier(0);
  };
  int ks = ks0;
  for (; ks + 5 < ks1; ks += 6) {
    step(std::integral_constant<int, 0>{}, ks);
    step(std::integral_constant<int, 1>{}, ks + 1);
    step(std::integral_constant<int, 2>{}, ks + 2);
    step(std::integral_constant<int, 3>{}, ks + 3);
    step(std::integral_constant<int, 4>{}, ks + 4);
    step(std::integral_constant<int, 5>{}, ks + 5);
  }
  const int r = ks1 - ks;
  if (r > 0) step(std::integral_constant<int, 0>{}, ks);
  if (r > 1) step(std::integral_constant<int, 1>{}, ks + 1);
  if (r > 2) step(std::integral_constant<int, 2>{}, ks + 2);
  if (r > 3) step(std::integral_constant<int, 3>{}, ks + 3);
  if (r > 4) step(std::integral_constant<int, 4>{}, ks + 4);
  const int rot = (ks1 - ks0) % 3;
  if (rot == 1) {
    const bfrag_t t = b[0];
    b[0] = b[1], b[1] = b[2], b[2] = t;
  } else if (rot == 2) {
    const bfrag_t t = b[0];
    b[0] = b[2], b[2] = b[1], b[1] = t;
  }
}

// ---- compile-time feature width (the BASELINE shapes): fully unrolled, weight fragments kPD k-steps ahead ---------------
// The weight is the same for every tile, so its fragment stream simply continues across tiles: the last kPD k-steps of a
// tile request fragments 0 .. kPD-1 of the NEXT tile into dedicated "head" registers, i.e. before this tile's output stores
// are issued — waiting for them later never waits for a store (gfx950 retires loads and stores of a wave in order), and the
// next tile starts multiplying the moment the barrier opens.  Under load a weight fragment takes > 1 us to come back (the
// CU's memory pipeline is full of the producers' row fetches); two k-steps = 48 MFMAs = ~1500 cycles of cover.
// With F a constant every LDS address is `lane base + immediate`.  The fp32 fragment of k-step ks+1 is read at the start of
// k-step ks and split at its end: both overlap the 24 MFMAs of k-step ks.
constexpr int kPD = 2;
__host__ __device__ constexpr int b_slot(int ks) { return ks < kPD ? ks : kPD + (ks % kPD); }
// Round 4: the LAST k-steps of the weight wait in the LDS the two operand tiles leave free (F = 100: 48 KB = 3 of 13 k-steps,
// F = 128: 1 of 16), written once per launch by the multiplying waves in the order they read it (lane-linear 16-B chunks) —
// that share of the weight stream leaves the CU's vector-memory pipeline, where each third of it was priced at 0.045 ms of the
// 0.55 ms layer-1 launch; the fragments still arrive kPD k-steps ahead, from ds_read_b128 instead of L2.
__host__ __device__ constexpr int w_lds_ksteps(int FC)
{
#ifdef WG_NO_WLDS   // (tuning build: the round-3 kernel)
  return 0;
#endif
  const int tiles = (2 * 64 * row_stride_dw(FC) + 16 + 4 * kScratchDw + 16) * 4;
  const int free_ = 160 * 1024 - tiles;
  const int ks    = free_ / (256 * 64);
  return ks < 0 ? 0 : (ks > 4 ? 4 : ks);
}

template <int TR, int FC>
struct static_consumer {
  static constexpr int RT = TR / 32, KSC = (2 * FC + 15) / 16, SD = row_stride_dw(FC);
  static constexpr int KL = TR == 64 ? w_lds_ksteps(FC) : 0;   // k-steps [KSC - KL, KSC) are read from the LDS
  braw_t bb[2 * kPD];    // fp32 fragments: [0, kPD): heads = k-steps 0 .. kPD-1 of a tile; [kPD, 2 kPD): ring for the rest
  uint32_t b_lane_off;   // bytes
  const float* w_lds;    // this lane's 16-B chunk of [k-step - (KSC - KL)][col tile][half] x 64 lanes in the wave's LDS slice
  f32x16 c[RT][2];       // accumulators of the tile being multiplied / waiting to be stored

  __device__ __forceinline__ void load_b_static(const mfma_args& a, braw_t& f, int ks) const
  {
#ifdef WG_ABL_NO_B   // tuning build: no weight loads (stale registers, wrong results) — prices the weight stream
    return;
#endif
#ifdef WG_ABL_HALF_B   // tuning build: every other k-step's weight fragments only — prices HALF the weight stream
    if (ks & 1) return;
#endif
    if (ks >= KSC - KL) {   // (ks is a compile-time constant at every call site)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int h = 0; h < 2; h++)
          f.v[ct][h] = *reinterpret_cast<const f32x4*>(w_lds + (((ks - (KSC - KL)) * 2 + ct) * 2 + h) * 256);
      return;
    }
    const char* wb = reinterpret_cast<const char*>(a.w_tiles);   // uniform: stays in SGPRs
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const size_t at = ((size_t)ks * a.N + ct * 32) * 64 + b_lane_off;
      f.v[ct][0]      = *reinterpret_cast<const f32x4*>(wb + at);
      f.v[ct][1]      = *reinterpret_cast<const f32x4*>(wb + at + 16);
    }
  }

  __device__ __forceinline__ void prime(const mfma_args& a, int cw, int lane, float* w_lds_wave)
  {
    b_lane_off = (uint32_t)(((cw * 64 + (lane & 31)) * 16 + (lane >> 5) * 8) * 4);
    w_lds      = w_lds_wave + lane * 4;
    if constexpr (KL > 0) {
      const char* wb = reinterpret_cast<const char*>(a.w_tiles);
#pragma unroll
      for (int ks = KSC - KL; ks < KSC; ks++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
          for (int h = 0; h < 2; h++)
            *reinterpret_cast<f32x4*>(const_cast<float*>(w_lds) + (((ks - (KSC - KL)) * 2 + ct) * 2 + h) * 256) =
              *reinterpret_cast<const f32x4*>(wb + ((size_t)ks * a.N + ct * 32) * 64 + b_lane_off + h * 16);
      // (read back by this wave only: the LDS executes a wave's operations in order)
    }
#pragma unroll
    for (int j = 0; j < kPD; j++) load_b_static(a, bb[j], j);
  }

  __device__ __forceinline__ void multiply(const mfma_args& a, const float* tile_lds, int lane)
  {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int i = 0; i < 16; i++) c[rt][ct][i] = 0.f;
    const float* a_lane = tile_lds + (lane & 31) * SD + (lane >> 5) * 8;
    araw_t<RT> raw;
    afrag_t<RT> fa[2];
    load_a_raw<RT>(raw, a_lane, SD, 0);
    split_a<RT>(raw, fa[0]);
#pragma unroll
    for (int ks = 0; ks < KSC; ks++) {
      if (ks + 1 < KSC) load_a_raw<RT>(raw, a_lane, SD, ks + 1);
      bfrag_t fb;
      split_b(bb[b_slot(ks)], fb);
      mma_frags<RT>(c, fa[ks & 1], fb);
      const int nk = ks + kPD;   // the slot just multiplied from (ring) or long since consumed (head) is free again
      if (nk < KSC) load_b_static(a, bb[b_slot(nk)], nk);
      else load_b_static(a, bb[nk - KSC], nk - KSC);
      if (ks + 1 < KSC) split_a<RT>(raw, fa[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distances as written (hoisted loads cost registers)
    }
  }
  __device__ __forceinline__ void store(const mfma_args& a, int64_t tile, int cw, int lane, float* scratch)
  {
    epilogue<RT>(a, c, tile * TR, cw, lane, scratch);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// CW = N / 64 consumer waves + 4 producer waves; TR = rows per tile; FC = compile-time feature width (0 = runtime a.F)
// ---------------------------------------------------------------------------------------------------------------------
template <typename IdT, int LG, int TR, int CW, bool OFF32, int FC, bool HALF = false>
__global__ void __launch_bounds__((CW + kProducerWaves) * 64)
sage_layer_mfma_kernel(mfma_args a)
{
  static_assert(!HALF || FC == 0, "the half-tile mode runs the runtime-shape consumer");
  // [2 tiles][TR][SD] fp32 + 16 floats of slack + [CW][8][64] epilogue scratch + role keys
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tile_dw = TR * a.SD;
  // every word the MFMA can touch must be finite: pad columns, the rows of a tile that is still being written for the
  // first time, and the few floats the last k-step reads past a row (times the zero rows of the weight)
  for (int i = threadIdx.x; i < 2 * tile_dw + 16; i += blockDim.x) lds[i] = 0.f;   // (scratch and role keys need no init)
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // Optional ROLES BY SIMD (see the header; off): every wave reads its SIMD id (HW_REG_HW_ID bits 5:4), the workgroup ranks
  // its waves by (SIMD class, wave) and the CW lowest ranks multiply.  Any placement gives exactly CW consumers and 4
  // producers; the usual 2-waves-per-SIMD placement makes SIMDs 0 and 2 multiply and SIMDs 1 and 3 fetch and sum.
  int wave = threadIdx.x >> 6;
  if (a.debug & 64) {
    uint32_t* keys = reinterpret_cast<uint32_t*>(lds + 2 * tile_dw + 16 + CW * kScratchDw);   // [CW + kProducerWaves]
    const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);                // HW_ID[5:4]
    if (lane == 0) keys[wave] = (uint32_t)(((simd & 1) * 2 + (simd >> 1)) * 16 + wave);
    __syncthreads();
    const uint32_t mine_key = keys[wave];
    int rank = 0;
#pragma unroll
    for (int w = 0; w < CW + kProducerWaves; w++) rank += keys[w] < mine_key ? 1 : 0;
    wave = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();
  }
  const int64_t n_tiles = (a.n_rows + TR - 1) / TR;
  const int64_t mine    = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  auto tile_of          = [&](int64_t n) { return (int64_t)blockIdx.x + n * gridDim.x; };
  auto stamp            = [&](int64_t n, int which) {
    if (a.stamps && blockIdx.x == 0 && lane == 0 && n < 64) a.stamps[(n * 8 + wave) * 2 + which] = __builtin_readcyclecounter();
  };

  if constexpr (HALF) {
    // The LDS holds only the MEAN halves, double-buffered (buffer n & 1 = tile n, F floats per row); the SELF half of the A
    // operand never passes through it: the multiplying waves read their self rows x[self(i)] straight from global memory
    // (a lane's half k-step of its row is 32 contiguous bytes).  One step per tile: the fetching waves sum the mean rows of
    // tile n into buffer n & 1 while the multiplying waves run tile n - 1 — the MEAN k-steps [0, KS/2) from the other
    // buffer, then the SELF k-steps [KS/2, KS) from global memory — and store it.  (Round 2 kept [mean | self] of ONE tile
    // in the two buffers and took two steps per tile, the second with the fetching waves only copying self rows while half
    // the K was multiplied: 1.69 ms at 256 -> 256 against 1.51 ms this way.  Self k-steps first, before the barrier:
    // 1.516 vs 1.505 — their first row loads then have nothing in front of them to hide behind.)
    const int KSh = a.KS / 2;   // F % 16 == 0: k-steps [0, KSh) = mean half (W_l rows), [KSh, KS) = self half (W_r rows)
    if (wave >= CW) {
      // the 64-row tile is produced as two 32-row SUB-TILES (sub-tile 2 t + h = rows 32 (2 t + h) ...): eight rows of
      // metadata per lane group in registers instead of sixteen; the metadata pipeline simply runs across the sub-tiles
      using P     = producer<IdT, LG, 32, OFF32, true>;
      using off_t = typename P::off_t;
      constexpr int IT = P::IT, kNb = P::kNb, kDepth = P::kDepth;
      static_assert(TR == 64, "two 32-row sub-tiles per tile");
      P p(a, wave - CW, lane);
      bounds_t<IT> b_next;
      ids_t<IT> i_next;
      meta_t<IT, off_t> cur;
      f32x4 buf[kDepth][kNb + 1];
      auto sub_of = [&](int64_t j) { return 2 * tile_of(j >> 1) + (j & 1); };
      p.load_bounds(sub_of(0), b_next);
      p.load_ids(sub_of(0), b_next, i_next);
      p.finish(i_next, cur);
#pragma unroll
      for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
      constexpr int kHalf = IT / 2;
      for (int64_t n = 0; n <= mine; n++) {
        if (n < mine && !(a.debug & 2)) {
          float* tile_lds = lds + (n & 1) * tile_dw;
#pragma unroll
          for (int sub = 0; sub < 2; sub++) {
            const int64_t j = 2 * n + sub;
            float* rows_lds = tile_lds + sub * 32 * a.SD;
            p.load_bounds(sub_of(j + 1), b_next);
#pragma unroll
            for (int it = 0; it < IT; it++) {
              if (it == kHalf) p.load_ids(sub_of(j + 1), b_next, i_next);
              if (it + kDepth - 1 < IT) p.issue(cur, it + kDepth - 1, buf[(it + kDepth - 1) % kDepth]);
              p.reduce_store(cur, it, buf[it % kDepth], rows_lds);
            }
            p.second_window(cur, rows_lds);
            p.long_rows(sub_of(j), cur, rows_lds);
            p.finish(i_next, cur);
            if (j + 1 < 2 * mine) {
#pragma unroll
              for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
            }
          }
        }
        lds_barrier();
      }
    } else {
      constexpr int RT = TR / 32;
      float* scratch = lds + 2 * tile_dw + 16 + wave * kScratchDw;
      f32x16 c[RT][2];
      bfrag_t b[3];   // the weight stream, two k-steps ahead (consume_range)
      {
        const uint32_t* b_lane =
          reinterpret_cast<const uint32_t*>(a.w_tiles) + ((int64_t)(wave * 64 + (lane & 31))) * 8 + (lane >> 5) * 4;
        load_b_planes(b[0], b_lane, (int64_t)a.KS * a.N * 8, a.N, 0);
        load_b_planes(b[1], b_lane, (int64_t)a.KS * a.N * 8, a.N, 1);
      }
      const IdT* src_ids = static_cast<const IdT*>(a.src_ids);
      for (int64_t n = 0; n <= mine; n++) {
        if (n >= 1 && !(a.debug & 1)) {
          const int64_t row0 = tile_of(n - 1) * TR;
          // this lane's self row of every row tile (rows past the end: any row, never stored); the two dependent loads
          // are requested here and waited for after the mean half
          const float* self_ptr[RT];
#pragma unroll
          for (int rt = 0; rt < RT; rt++) {
            const int64_t row  = row0 + rt * 32 + (lane & 31);
            const int64_t srow = a.self_rows[row < a.n_rows ? row : a.n_rows - 1];
            self_ptr[rt]       = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + table_row<IdT>(src_ids, srow) * a.row_scale) +
                           (lane >> 5) * 8;
          }
#pragma unroll
          for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
              for (int i = 0; i < 16; i++) c[rt][ct][i] = 0.f;
          store_agg<TR, CW>(a, lds + ((n - 1) & 1) * tile_dw, row0, wave, lane);
          const float* a_lane = lds + ((n - 1) & 1) * tile_dw + (lane & 31) * a.SD + (lane >> 5) * 8;
          consume_range<RT, 1>(a, c, b, 0, KSh, wave, lane, [&](araw_t<RT>& f, int ks) { load_a_raw<RT>(f, a_lane, a.SD, ks); });
          consume_range<RT, 2>(a, c, b, KSh, a.KS, wave, lane, [&](araw_t<RT>& f, int ks) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
              const float* p = self_ptr[rt] + (ks - KSh) * 16;
              f.v[rt][0]     = *reinterpret_cast<const f32x4*>(p);
              f.v[rt][1]     = *reinterpret_cast<const f32x4*>(p + 4);
            }
          });
          epilogue<RT>(a, c, row0, wave, lane, scratch);
        }
        lds_barrier();
      }
    }
  } else if (wave >= CW) {
    if (a.debug & 16) __builtin_amdgcn_s_setprio(3);
    using P     = producer<IdT, LG, TR, OFF32>;
    using off_t = typename P::off_t;
    constexpr int IT = P::IT, kNb = P::kNb, kDepth = P::kDepth;
    P p(a, wave - CW, lane);
    // metadata of tile n+1 is fetched WHILE tile n is summed, in three dependent stages spread over the tile so that no
    // stage ever waits: CSR bounds at the start of the tile, neighbour / self ids (they need the bounds) half-way, byte
    // offsets (they need the ids) at the end.  Bounds and ids are never live together.
    bounds_t<IT> b_next;
    ids_t<IT> i_next;
    meta_t<IT, off_t> cur;
    f32x4 buf[kDepth][kNb + 1];
    {
      p.load_bounds(tile_of(0), b_next);
      p.load_ids(tile_of(0), b_next, i_next);
      p.finish(i_next, cur);
    }
#pragma unroll
    for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
    constexpr int kHalf = IT / 2;
    for (int64_t n = 0; n <= mine; n++) {
      stamp(n, 0);
      if (n < mine && !(a.debug & 2)) {
        float* tile_lds = lds + (n & 1) * tile_dw;
        p.load_bounds(tile_of(n + 1), b_next);
#pragma unroll
        for (int it = 0; it < IT; it++) {
          if (it == kHalf) p.load_ids(tile_of(n + 1), b_next, i_next);
          if (it + kDepth - 1 < IT) p.issue(cur, it + kDepth - 1, buf[(it + kDepth - 1) % kDepth]);
          p.reduce_store(cur, it, buf[it % kDepth], tile_lds);
        }
        p.second_window(cur, tile_lds);
        p.long_rows(tile_of(n), cur, tile_lds);
        // offsets of tile n+1, and its first rows in flight BEFORE the barrier
        p.finish(i_next, cur);
        if (n + 1 < mine) {
#pragma unroll
          for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
        }
      }
      stamp(n, 1);
      lds_barrier();
    }
  } else {
    if (a.debug & 32) __builtin_amdgcn_s_setprio(3);
    float* scratch = lds + 2 * tile_dw + 16 + wave * kScratchDw;
    if constexpr (FC > 0) {
      static_consumer<TR, FC> cons;
      cons.prime(a, wave, lane, lds + 2 * tile_dw + 16 + CW * kScratchDw + 16 + wave * (decltype(cons)::KL * 16 * 64));
      // Two consumer waves share a SIMD (ranks 2k, 2k+1) and one matrix pipe.  They are kept out of phase: the even one
      // multiplies a tile and stores it in the same step, the odd one stores the PREVIOUS tile first (its accumulators stay
      // live across the barrier) and multiplies afterwards — so one wave's epilogue (LDS transpose, stores, waits) runs
      // under the other wave's MFMAs instead of both idling the pipe together.
      const bool late = (wave & 1) && !(a.debug & 128);
      int64_t pending = -1;
      for (int64_t n = 0; n <= mine; n++) {
        stamp(n, 0);
        if (n >= 1 && !(a.debug & 1)) {
          const float* tile_lds = lds + ((n - 1) & 1) * tile_dw;
          store_agg<TR, CW>(a, tile_lds, tile_of(n - 1) * TR, wave, lane);
          if (late) {
            if (pending >= 0) cons.store(a, pending, wave, lane, scratch);
            cons.multiply(a, tile_lds, lane);
            pending = tile_of(n - 1);
          } else {
            cons.multiply(a, tile_lds, lane);
            cons.store(a, tile_of(n - 1), wave, lane, scratch);
          }
        }
        stamp(n, 1);
        lds_barrier();
      }
      if (late && pending >= 0) cons.store(a, pending, wave, lane, scratch);
    } else {
      for (int64_t n = 0; n <= mine; n++) {
        if (n >= 1 && !(a.debug & 1)) {
          store_agg<TR, CW>(a, lds + ((n - 1) & 1) * tile_dw, tile_of(n - 1) * TR, wave, lane);
          consume_tile<TR>(a, tile_of(n - 1), lds + ((n - 1) & 1) * tile_dw, wave, lane, scratch);
        }
        lds_barrier();
      }
    }
  }
}

// ---- weight in the order the multiplying waves read it -----------------------------------------------------------------
// HALF mode: w_t [K, N] fp32 row-major (ldw)  ->  planes [3][KS][N][16] bf16, rows K .. 16 KS - 1 zero
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

// w_t [K, N] fp32 row-major (ldw)  ->  tiles [KS][N][16] fp32 (k-step, column, 16 consecutive k), rows K .. 16 KS - 1 zero:
// a lane's half k-step of one column is 32 contiguous bytes, a wave's load 1 KiB + 1 KiB contiguous
__global__ void tile_weight_kernel(const float* __restrict__ w_t, int64_t ldw, int K, int N, int KS, float* __restrict__ tiles)
{
  const int64_t total = (int64_t)KS * N * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 15);
    const int n  = (int)((i >> 4) % N);
    const int ks = (int)((i >> 4) / N);
    const int k  = ks * 16 + kk;
    tiles[i]     = k < K ? w_t[(int64_t)k * ldw + n] : 0.f;
  }
}

// The same two orders straight from a layer's parameters: the [K = 2F, N] operand is [W_l | W_r]^T (torch.nn.Linear keeps
// [N, F] row-major), its columns and the bias zero-padded from N to Np — ONE launch instead of cat / transpose / pad / split
// (ten launches per layer pair in a captured per-mini-batch training step, where the weights change on every replay).
__global__ void layer_weight_kernel(const float* __restrict__ w_l, int64_t ldl, const float* __restrict__ w_r, int64_t ldr,
                                    const float* __restrict__ bias, int F, int N, int Np, int KS, int half,
                                    void* __restrict__ planes_, float* __restrict__ bias_out)
{
  auto w_at = [&](int k, int n) -> float {
    if (n >= N || k >= 2 * F) return 0.f;
    return k < F ? w_l[(int64_t)n * ldl + k] : w_r[(int64_t)n * ldr + (k - F)];
  };
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  if (bias_out)
    for (int64_t n = tid; n < Np; n += stride) bias_out[n] = (bias && n < N) ? bias[n] : 0.f;
  if (half) {
    uint32_t* planes    = static_cast<uint32_t*>(planes_);
    const int64_t total = (int64_t)KS * Np * 8;
    for (int64_t i = tid; i < total; i += stride) {
      const int kk2 = (int)(i & 7), n = (int)((i >> 3) % Np), ks = (int)((i >> 3) / Np);
      const int k0  = ks * 16 + kk2 * 2;
      uint32_t h0, m0, l0, h1, m1, l1;
      split3(w_at(k0, n), h0, m0, l0);
      split3(w_at(k0 + 1, n), h1, m1, l1);
      planes[i]             = pack_hi16(h0, h1);
      planes[total + i]     = pack_hi16(m0, m1);
      planes[2 * total + i] = pack_hi16(l0, l1);
    }
  } else {
    float* tiles        = static_cast<float*>(planes_);
    const int64_t total = (int64_t)KS * Np * 16;
    for (int64_t i = tid; i < total; i += stride) {
      const int kk = (int)(i & 15), n = (int)((i >> 4) % Np), ks = (int)((i >> 4) / Np);
      tiles[i]     = w_at(ks * 16 + kk, n);
    }
  }
}

constexpr size_t kLdsBudget = 160 * 1024;
__host__ inline size_t lds_bytes(int F, int TR) { return (size_t)(2 * TR * row_stride_dw(F) + 16 + 4 * kScratchDw + 16) * 4; }
__host__ inline size_t lds_bytes_half(int F) { return (size_t)(2 * 64 * row_stride_half_dw(F) + 16 + 4 * kScratchDw + 16) * 4; }
// 64-row tiles in two halves: when two whole 64-row tiles do not fit but the halves of one do, and the halves are whole
// k-steps (WGAMD_SAGE_HALF_TILES=0 keeps the 32-row tiles)
__host__ inline bool use_half_tiles(int F)
{
  static const bool off = [] { const char* e = getenv("WGAMD_SAGE_HALF_TILES"); return e && e[0] == '0'; }();
  return !off && F % 16 == 0 && lds_bytes(F, 64) > kLdsBudget && lds_bytes_half(F) <= kLdsBudget;
}

template <typename IdT, int LG, int CW>
void launch_half(mfma_args a, hipStream_t st)
{
  const int cus         = stream_cu_count(st);
  a.SD                  = row_stride_half_dw(a.F);
  const int64_t n_tiles = (a.n_rows + 63) / 64;
  const size_t lds      = lds_bytes_half(a.F);
  const int grid        = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)cus));
  auto go               = [&](auto kern) {
    WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<grid, (CW + kProducerWaves) * 64, lds, st>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  };
  if (a.x_bytes != 0) go(sage_layer_mfma_kernel<IdT, LG, 64, CW, true, 0, true>);
  else go(sage_layer_mfma_kernel<IdT, LG, 64, CW, false, 0, true>);
}

template <typename IdT, int LG, int TR, int CW, int FC = 0>
void launch(const mfma_args& a, hipStream_t st)
{
  const int cus         = stream_cu_count(st);
  const int64_t n_tiles = (a.n_rows + TR - 1) / TR;
  const size_t lds      = lds_bytes(a.F, TR) + (FC > 0 && TR == 64 ? (size_t)w_lds_ksteps(FC) * CW * 64 * 64 : 0);
  // ONE workgroup per CU: the SIMD role split needs the CU to itself (two waves per SIMD)
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)cus));
  auto go        = [&](auto kern) {
    if (lds > 64 * 1024)
      WG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<grid, (CW + kProducerWaves) * 64, lds, st>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  };
  if (a.x_bytes != 0) go(sage_layer_mfma_kernel<IdT, LG, TR, CW, true, FC>);
  else go(sage_layer_mfma_kernel<IdT, LG, TR, CW, false, FC>);
}

// the feature width is a compile-time constant for the BASELINE layer shapes (F = 100: products, F = 128: papers100M / mag)
// with N = 256; every other shape takes the runtime-shape consumer
template <typename IdT, int LG, int TR>
void launch_cw(const mfma_args& a, hipStream_t st)
{
  switch (a.N / 64) {
    case 1: launch<IdT, LG, TR, 1>(a, st); break;
    case 2: launch<IdT, LG, TR, 2>(a, st); break;
    default:
      if constexpr (LG == 32 && TR == 64) {
        if (a.F == 100) return launch<IdT, LG, TR, 4, 100>(a, st);
        if (a.F == 128) return launch<IdT, LG, TR, 4, 128>(a, st);
      }
      // (F = 256, RMAT-26's layers: a compile-time consumer was tried — 32 unrolled k-steps spill 63 VGPRs and lose 13 %;
      //  that shape is bound by its 6 x 2 N_dst 2F N bf16 flops on time-sliced SIMDs, not by the weight prefetch)
      launch<IdT, LG, TR, 4>(a, st);
      break;
  }
}

// F <= 128: two whole 64-row tiles; wider rows: 64-row tiles in two halves when the halves are whole k-steps and fit, 32-row
// tiles otherwise; only the (LG, TR) pairs that can occur are instantiated
template <typename IdT, int LG>
void launch_tr(const mfma_args& a, hipStream_t st)
{
  if constexpr (LG <= 32) {
    launch_cw<IdT, LG, 64>(a, st);
  } else {
    // (128 < F <= 148: two whole 64-row tiles would fit, but a 64-lane group then carries sixteen rows of metadata per tile
    //  and the kernel spills ~100 VGPRs; 32-row tiles do not)
    if (use_half_tiles(a.F) && !a.full_tiles) {
      switch (a.N / 64) {
        case 1: launch_half<IdT, LG, 1>(a, st); break;
        case 2: launch_half<IdT, LG, 2>(a, st); break;
        default: launch_half<IdT, LG, 4>(a, st); break;
      }
    } else launch_cw<IdT, LG, 32>(a, st);
  }
}

template <typename IdT>
void launch_groups(const mfma_args& a, hipStream_t st)
{
  const int units = a.F / 4;
  if (units <= 8) launch_tr<IdT, 8>(a, st);
  else if (units <= 16) launch_tr<IdT, 16>(a, st);
  else if (units <= 32) launch_tr<IdT, 32>(a, st);
  else launch_tr<IdT, 64>(a, st);
}

}  // namespace
}  // namespace wgamd

#ifndef WG_MFMA_TUNE_HARNESS
// (sized for the larger of the two formats: bf16 planes are 96 B per (k-step, column), fp32 tiles 64 B)
extern "C" size_t wgamd_sage_weight_planes_bytes(int K, int N) { return (size_t)3 * ((K + 15) / 16) * (size_t)N * 32; }

extern "C" int wgamd_sage_layer_bf16x3_supported(int F, int N)
{
  return F > 0 && F % 4 == 0 && F <= 256 && (N == 64 || N == 128 || N == 256) && wgamd::lds_bytes(F, 32) <= wgamd::kLdsBudget;
}

extern "C" wholememory_error_code_t wgamd_sage_split_weight_bf16x3(const float* w_t, int64_t ldw, int K, int N, void* planes,
                                                                   void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sage_split_weight_bf16x3", [&] {
    WG_REQUIRE_INPUT(w_t && planes && K > 0 && N > 0 && ldw >= N, "bad weight");
    const int KS        = (K + 15) / 16;
    const int64_t total = (int64_t)KS * N * 16;
    const int grid      = (int)std::min<int64_t>((total + 255) / 256, 2048);
    // K = 2F: the layer kernel of this F decides the format (half-tile mode reads pre-split planes, the others fp32 tiles)
    if (K % 2 == 0 && use_half_tiles(K / 2))
      split_weight_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(w_t, ldw, K, N, KS, static_cast<uint32_t*>(planes));
    else
      tile_weight_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(w_t, ldw, K, N, KS, static_cast<float*>(planes));
    WG_HIP_CHECK(hipGetLastError());
  });
}

extern "C" int wgamd_sage_layer_uses_half_tiles(int F) { return wgamd::use_half_tiles(F) ? 1 : 0; }

extern "C" wholememory_error_code_t wgamd_sage_layer_weight_planes(const float* w_l, int64_t ldl, const float* w_r, int64_t ldr,
                                                                  const float* bias, int F, int N, int Np, void* planes,
                                                                  float* bias_out, int full_tiles, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sage_layer_weight_planes", [&] {
    WG_REQUIRE_INPUT(w_l && w_r && planes && F > 0 && N > 0 && Np >= N && ldl >= F && ldr >= F, "bad weights");
    const int K = 2 * F, KS = (K + 15) / 16;
    const int half      = (use_half_tiles(F) && !full_tiles) ? 1 : 0;
    const int64_t total = (int64_t)KS * Np * 16;
    const int grid      = (int)std::min<int64_t>((total + 255) / 256, 2048);
    layer_weight_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(w_l, ldl, w_r, ldr, bias, F, N, Np, KS, half, planes, bias_out);
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
  return wgamd_sage_layer_fused_bf16x3_train(row_ptr, col, n_rows, x, ldx, x_rows, F, src_ids, src_ids_dtype, self_rows, mean,
                                             w_planes, N, bias, relu, out, ldo, nullptr, 0, stream);
}

extern "C" wholememory_error_code_t wgamd_sage_layer_fused_bf16x3_train(const int* row_ptr, const int* col, int64_t n_rows,
                                                                        const float* x, int64_t ldx, int64_t x_rows, int F,
                                                                        const void* src_ids, wholememory_dtype_t src_ids_dtype,
                                                                        const int64_t* self_rows, int mean, const void* w_planes,
                                                                        int N, const float* bias, int relu, float* out,
                                                                        int64_t ldo, float* agg_out, int64_t ld_agg, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sage_layer_fused_bf16x3", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && F > 0 && N > 0, "bad sizes");
    if (n_rows == 0) return;
    WG_REQUIRE_INPUT(row_ptr && col && x && self_rows && w_planes && out, "null pointer");
    if (!wgamd_sage_layer_bf16x3_supported(F, N) || ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0)
      throw logic_error(fmt("unsupported shape: F=%d (multiple of 4, <= 256), N=%d (64, 128 or 256), 16-B aligned rows", F, N));
    WG_REQUIRE_INPUT(ldo >= N, "leading dimension smaller than N");
    if (ldo % 4 != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0) throw logic_error("output rows must be 16-B aligned");
    if (agg_out != nullptr && (ld_agg < F || ld_agg % 4 != 0 || (reinterpret_cast<uintptr_t>(agg_out) & 15) != 0))
      throw logic_error("agg_out rows must hold F floats and be 16-B aligned");
    // x below 2 GB (extent known): 32-bit row offsets and buffer loads whose out-of-range slots read as zero
    const uint64_t xb = x_rows > 0 ? (uint64_t)x_rows * (uint64_t)ldx * 4u : 0;
    mfma_args a{row_ptr, col, n_rows, x, ldx, (uint32_t)(xb > 0 && xb < (1ull << 31) ? xb : 0), F, src_ids, self_rows, mean,
                static_cast<const float*>(w_planes), N, (2 * F + 15) / 16, bias, relu & 1, out, ldo, row_stride_dw(F), 0, nullptr,
                ldx * 4, agg_out, ld_agg, (relu & WGAMD_SAGE_FULL_TILES) != 0};
    const bool byte_offsets = src_ids != nullptr && src_ids_dtype == WGAMD_IDS_BYTE_OFFSETS;
    if (byte_offsets) {      // rows addressed by byte offsets from x (a peer-mapped table): 64-bit addressing, no extent
      a.row_scale = 1;
      a.x_bytes   = 0;
    }
    // WGAMD_SAGE_DEBUG=<bits> (tuning only; results are WRONG with 4 / 8): the ablation switches of mfma_args::debug
    static const int dbg = [] { const char* e = getenv("WGAMD_SAGE_DEBUG"); return e ? atoi(e) : 0; }();
    a.debug = dbg;
    auto st = static_cast<hipStream_t>(stream);
    if (agg_out == nullptr && !byte_offsets && !a.full_tiles && sage_ws_supported(F, N)) {
      if (src_ids != nullptr && src_ids_dtype != WHOLEMEMORY_DT_INT && src_ids_dtype != WHOLEMEMORY_DT_INT64)
        throw invalid_input("src_ids must be INT or INT64");
      return sage_ws_launch(a, src_ids == nullptr ? 0 : (src_ids_dtype == WHOLEMEMORY_DT_INT ? 1 : 2), st);
    }
    if (src_ids == nullptr) launch_groups<void>(a, st);
    else if (src_ids_dtype == WHOLEMEMORY_DT_INT) launch_groups<int32_t>(a, st);
    else if (src_ids_dtype == WHOLEMEMORY_DT_INT64 || byte_offsets) launch_groups<int64_t>(a, st);
    else throw invalid_input("src_ids must be INT, INT64 or WGAMD_IDS_BYTE_OFFSETS");
  });
}
#endif  // WG_MFMA_TUNE_HARNESS
