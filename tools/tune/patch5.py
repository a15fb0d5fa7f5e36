import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sage_mfma.hip')
s = open(p).read()

old = s[s.index("  __device__ __forceinline__ void tile(const mfma_args& a, int64_t tile, const uint32_t* tile_lds, int cw, int lane, float* scratch)\n  {\n    f32x16 c[RT][2];"):s.index("// ---------------------------------------------------------------------------------------------------------------------\n// CW = N / 64 consumer waves")]
new = '''  f32x16 c[RT][2];   // accumulators of the tile being multiplied / waiting to be stored

  __device__ __forceinline__ void multiply(const mfma_args& a, const uint32_t* tile_lds, int lane)
  {
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
  }
  __device__ __forceinline__ void store(const mfma_args& a, int64_t tile, int cw, int lane, float* scratch)
  {
    epilogue<RT>(a, c, tile * TR, cw, lane, scratch);
  }
};

'''
s = s.replace(old, new)

old = '''      for (int64_t n = 0; n <= mine; n++) {
        if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 0] = __builtin_readcyclecounter();
        if (n >= 1 && !(a.debug & 1)) cons.tile(a, tile_of(n - 1), lds + ((n - 1) & 1) * tile_dw, wave, lane, scratch);
        if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 1] = __builtin_readcyclecounter();
        lds_barrier();
        if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 2] = __builtin_readcyclecounter();
      }'''
new = '''      // Two consumer waves share a SIMD (ranks 2k, 2k+1) and one matrix pipe.  They are kept out of phase: the even one
      // multiplies a tile and stores it in the same step, the odd one stores the PREVIOUS tile first (its accumulators stay
      // live across the barrier) and multiplies afterwards — so one wave's epilogue (LDS transpose, stores, waits) always
      // runs under the other wave's MFMAs instead of both idling the pipe together.
      const bool late = (wave & 1) && !(a.debug & 128);
      int64_t pending = -1;
      for (int64_t n = 0; n <= mine; n++) {
        if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 0] = __builtin_readcyclecounter();
        if (n >= 1 && !(a.debug & 1)) {
          const uint32_t* tile_lds = lds + ((n - 1) & 1) * tile_dw;
          if (late) {
            if (pending >= 0) cons.store(a, pending, wave, lane, scratch);
            cons.multiply(a, tile_lds, lane);
            pending = tile_of(n - 1);
          } else {
            cons.multiply(a, tile_lds, lane);
            cons.store(a, tile_of(n - 1), wave, lane, scratch);
          }
        }
        if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 1] = __builtin_readcyclecounter();
        lds_barrier();
        if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 2] = __builtin_readcyclecounter();
      }
      if (late && pending >= 0) cons.store(a, pending, wave, lane, scratch);'''
assert old in s
s = s.replace(old, new)
open(p, 'w').write(s)
print("ok")
