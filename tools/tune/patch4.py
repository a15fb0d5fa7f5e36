import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sage_mfma.hip')
s = open(p).read()

old = '''  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_tiles = (a.n_rows + TR - 1) / TR;'''
new = '''  const int lane = threadIdx.x & 63;
  // ROLES BY SIMD.  Measured on gfx950: a wave issuing back-to-back MFMAs and a VALU / memory-heavy wave on the SAME SIMD
  // do not overlap, they time-slice (producer at ~25 % of its speed while the co-resident consumer multiplies, consumer at
  // ~70 %: step = producer time + consumer time).  Waves on DIFFERENT SIMDs of the CU do run concurrently.  So the consumers
  // are the waves that landed on two of the four SIMDs and the producers the waves on the other two: every wave reads its
  // SIMD id (HW_REG_HW_ID bits 5:4), the workgroup ranks its waves by (SIMD class, wave) and the CW lowest ranks multiply.
  // Any placement gives exactly CW consumers and 4 producers; the usual 2-waves-per-SIMD placement gives a clean split
  // (SIMDs 0 and 2 multiply — half the chip's matrix pipes, still ~1.2 PFLOP/s of bf16 — SIMDs 1 and 3 fetch and sum).
  int wave = threadIdx.x >> 6;
  if (!(a.debug & 64)) {
    uint32_t* keys = lds + 2 * tile_dw + 16 + CW * kScratchDw;   // [CW + kProducerWaves]
    const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_ID[5:4]
    if (lane == 0) keys[wave] = (uint32_t)(((simd & 1) * 2 + (simd >> 1)) * 16 + wave);
    __syncthreads();
    const uint32_t mine_key = keys[wave];
    int rank = 0;
#pragma unroll
    for (int w = 0; w < CW + kProducerWaves; w++) rank += keys[w] < mine_key ? 1 : 0;
    wave = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();
  }
  const int64_t n_tiles = (a.n_rows + TR - 1) / TR;'''
assert old in s
s = s.replace(old, new)
s = s.replace("  for (int i = threadIdx.x; i < 2 * tile_dw + 16; i += blockDim.x) lds[i] = 0u;",
              "  for (int i = threadIdx.x; i < 2 * tile_dw + 16; i += blockDim.x) lds[i] = 0u;   // (scratch and role keys need no init)")
s = s.replace("__host__ inline size_t lds_bytes(int F, int TR) { return (size_t)(2 * 3 * TR * row_stride_dw(F) + 16 + 4 * kScratchDw) * 4; }",
              "__host__ inline size_t lds_bytes(int F, int TR) { return (size_t)(2 * 3 * TR * row_stride_dw(F) + 16 + 4 * kScratchDw + 16) * 4; }")
# stamps use the physical thread of the role: replace threadIdx-based conditions with role-based ones
s = s.replace("blockIdx.x == 0 && threadIdx.x == CW * 64 && n < 64", "blockIdx.x == 0 && wave == CW && lane == 0 && n < 64")
s = s.replace("blockIdx.x == 0 && threadIdx.x == 0 && n < 64", "blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64")
# mean: one exactly-rounded reciprocal per row instead of four IEEE divisions per lane
s = s.replace("    if (a.mean && deg > 0) acc /= (float)deg;   // rows longer than the window are redone by long_rows()",
              "    if (a.mean && deg > 0) acc *= __frcp_rn((float)deg);   // (rows longer than the window are redone by long_rows())")
s = s.replace("        if (a.mean) acc /= (float)deg;\n        store_split(tile_lds + (group + it * kGroups) * a.SD",
              "        if (a.mean) acc *= __frcp_rn((float)deg);\n        store_split(tile_lds + (group + it * kGroups) * a.SD")
open(p, 'w').write(s)
print("ok")
