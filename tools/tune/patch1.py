import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sage_mfma.hip')
s = open(p).read()
i = s.index('// ---- compile-time k-step count (the BASELINE shapes)')
j = s.index('// ---------------------------------------------------------------------------------------------------------------------\n// CW = N / 64 consumer waves')
new = '''// ---- compile-time feature width (the BASELINE shapes): fully unrolled, weight fragments kPD k-steps ahead ---------------
// The weight is the same for every tile, so its fragment stream simply continues across tiles: the last kPD k-steps of a
// tile request fragments 0 .. kPD-1 of the NEXT tile into dedicated "head" registers, i.e. before this tile's output stores
// are issued — waiting for them later never waits for a store (gfx950 retires loads and stores of a wave in order), and the
// next tile starts multiplying the moment the barrier opens.  Under load a weight fragment takes > 1 us to come back (the
// CU's memory pipeline is full of the producers' row fetches); two k-steps = 48 MFMAs = ~1500 cycles of cover.
// With F a constant every LDS address is `lane base + immediate` and every weight address `uniform base (SGPR) + lane
// offset`: no address registers per fragment (the runtime-shape version kept 70+ VGPRs of precomputed addresses).
constexpr int kPD = 2;
__host__ __device__ constexpr int b_slot(int ks) { return ks < kPD ? ks : kPD + (ks % kPD); }
__host__ __device__ constexpr int row_stride_dw(int F)
{
  int sd = (F + 3) / 4 * 4;        // F dwords hold 2F bf16
  return (sd / 4) % 2 == 0 ? sd + 4 : sd;   // sd = 4 * odd
}

template <int TR, int FC>
struct static_consumer {
  static constexpr int RT = TR / 32, KSC = (2 * FC + 15) / 16, SD = row_stride_dw(FC);
  bfrag_t bb[2 * kPD];   // [0, kPD): heads = k-steps 0 .. kPD-1 of a tile; [kPD, 2 kPD): ring for the rest
  uint32_t b_lane_off;   // bytes

  __device__ __forceinline__ void load_b_static(const mfma_args& a, bfrag_t& f, int ks) const
  {
    const char* wb       = reinterpret_cast<const char*>(a.w_planes);   // uniform: stays in SGPRs
    const size_t plane_b = (size_t)KSC * a.N * 32;
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int p = 0; p < 3; p++)
        f.v[ct][p] = *reinterpret_cast<const u32x4*>(wb + (p * plane_b + ((size_t)ks * a.N + ct * 32) * 32) + b_lane_off);
  }

  __device__ __forceinline__ void prime(const mfma_args& a, int cw, int lane)
  {
    b_lane_off = (uint32_t)(((cw * 64 + (lane & 31)) * 8 + (lane >> 5) * 4) * 4);
#pragma unroll
    for (int j = 0; j < kPD; j++) load_b_static(a, bb[j], j);
  }

  __device__ __forceinline__ void tile(const mfma_args& a, int64_t tile, const uint32_t* tile_lds, int cw, int lane, float* scratch)
  {
    f32x16 c[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int i = 0; i < 16; i++) c[rt][ct][i] = 0.f;
    const uint32_t* a_lane = tile_lds + (lane & 31) * SD + (lane >> 5) * 4;
    afrag_t<RT> aa[2];
    load_a<RT>(aa[0], a_lane, TR * SD, SD, 0);
#pragma unroll
    for (int ks = 0; ks < KSC; ks++) {
      if (ks + 1 < KSC) load_a<RT>(aa[(ks + 1) & 1], a_lane, TR * SD, SD, ks + 1);
      mma_frags<RT>(c, aa[ks & 1], bb[b_slot(ks)]);
      const int nk = ks + kPD;   // the slot just multiplied from (ring) or long since consumed (head) is free again
      if (nk < KSC) load_b_static(a, bb[b_slot(nk)], nk);
      else load_b_static(a, bb[nk - KSC], nk - KSC);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distances as written (hoisted loads cost registers)
    }
    epilogue<RT>(a, c, tile * TR, cw, lane, scratch);
  }
};

'''
s = s[:i] + new + s[j:]
s = s.replace("// KSC = compile-time k-step count (0 = runtime a.KS)\ntemplate <typename IdT, int LG, int TR, int CW, bool OFF32, int KSC>",
              "// FC = compile-time feature width (0 = runtime a.F)\ntemplate <typename IdT, int LG, int TR, int CW, bool OFF32, int FC>")
s = s.replace("    if constexpr (KSC > 0) {\n      static_consumer<TR, KSC> cons;", "    if constexpr (FC > 0) {\n      static_consumer<TR, FC> cons;")
s = s.replace("template <typename IdT, int LG, int TR, int CW, int KSC = 0>\nvoid launch(", "template <typename IdT, int LG, int TR, int CW, int FC = 0>\nvoid launch(")
s = s.replace("sage_layer_mfma_kernel<IdT, LG, TR, CW, true, KSC>", "sage_layer_mfma_kernel<IdT, LG, TR, CW, true, FC>")
s = s.replace("sage_layer_mfma_kernel<IdT, LG, TR, CW, false, KSC>", "sage_layer_mfma_kernel<IdT, LG, TR, CW, false, FC>")
s = s.replace("        if (a.KS == 13) return launch<IdT, LG, TR, 4, 13>(a, off32, st);", "        if (a.F == 100) return launch<IdT, LG, TR, 4, 100>(a, off32, st);")
s = s.replace("        if (a.KS == 16) return launch<IdT, LG, TR, 4, 16>(a, off32, st);", "        if (a.F == 128) return launch<IdT, LG, TR, 4, 128>(a, off32, st);")
s = s.replace("// the k-step count is a compile-time constant for the BASELINE layer shapes (F = 100 -> 13 k-steps, F = 128 -> 16) with\n// N = 256; every other shape takes the runtime-count consumer",
              "// the feature width is a compile-time constant for the BASELINE layer shapes (F = 100: products, F = 128: papers100M / mag)\n// with N = 256; every other shape takes the runtime-shape consumer")
old = '''__host__ inline int row_stride_dw(int F)
{
  int sd = (F + 3) / 4 * 4;      // F dwords hold 2F bf16
  if ((sd / 4) % 2 == 0) sd += 4;  // sd = 4 * odd
  return sd;
}
'''
assert old in s
s = s.replace(old, '')
open(p, 'w').write(s)
p = os.path.join(ROOT, 'tools/tune/sage_mfma_harness.cpp')
s = open(p).read()
s = s.replace("#ifndef WG_HARNESS_KSC\n#define WG_HARNESS_KSC 13\n#endif", "#ifndef WG_HARNESS_KSC\n#define WG_HARNESS_KSC 100\n#endif")
open(p, 'w').write(s)
p = os.path.join(ROOT, 'tools/tune/build.sh')
s = open(p).read()
s = s.replace('for v in "4 2 13" "4 2 0" "8 2 13"; do', 'for v in "4 2 100" "4 2 0"; do')
open(p, 'w').write(s)
print("patched")
