// Matrix-pipe rate of the consumer's MFMA pattern in isolation (tuning tool, not part of the library):
//   hipcc -O3 --offload-arch=gfx950 tools/tune/mfma_rate.cpp -o tools/tune/bin/mfma_rate && tools/tune/bin/mfma_rate
// 24 x v_mfma_f32_32x32x16_bf16 per "k-step" on NACC independent accumulators, operands in registers, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4  = __attribute__((ext_vector_type(4))) uint32_t;

template <int NACC, int FILL>
__global__ void __launch_bounds__(512) rate_kernel(float* out, int iters, uint32_t seed)
{
  f32x16 c[NACC];
  for (int i = 0; i < NACC; i++)
    for (int j = 0; j < 16; j++) c[i][j] = 0.f;
  u32x4 a[3], b[2][3];
  for (int p = 0; p < 3; p++) {
    a[p] = u32x4{seed + threadIdx.x, seed * 3 + p, seed ^ 0x3f803f80u, 0x3f803f80u};
    for (int ct = 0; ct < 2; ct++) b[ct][p] = u32x4{0x3f803f80u, seed + ct, seed + p, 0x3c003c00u};
  }
  uint32_t fz[16];
  float ff[16];
  for (int i = 0; i < 16; i++) {
    fz[i] = seed * (i + 1) + threadIdx.x;
    ff[i] = (float)(seed + i);
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 24; t++) {
      const int acc = t % NACC;
      c[acc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t % 3]),
                                                       __builtin_bit_cast(bf16x8, b[t & 1][(t / 2) % 3]), c[acc], 0, 0, 0);
      // FILL independent VALU instructions behind every MFMA (the consumer's split: and / sub / perm on other registers)
#pragma unroll
      for (int f = 0; f < FILL; f++) {
        const int r = (t * FILL + f) % 16;
        if (f % 3 == 0) fz[r] = fz[r] & 0xffff0000u;
        else if (f % 3 == 1) ff[r] = ff[r] - __uint_as_float(fz[(r + 5) % 16]);
        else fz[r] = __builtin_amdgcn_perm(fz[(r + 3) % 16], fz[r], 0x07060302u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; i++)
    for (int j = 0; j < 16; j++) s += c[i][j];
  for (int i = 0; i < 16; i++) s += ff[i] + (float)fz[i];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC, int FILL>
void run(int waves_per_cu_simd, int iters)
{
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int threads = 256 * waves_per_cu_simd;
  rate_kernel<NACC, FILL><<<256, threads>>>(out, 10, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate_kernel<NACC, FILL><<<256, threads>>>(out, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * (threads / 64) * (double)iters * 24 * 32 * 32 * 16 * 2;
  printf("NACC=%d FILL=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, FILL, waves_per_cu_simd, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 24 * waves_per_cu_simd));
  hipFree(out);
}

int main()
{
  for (int w = 1; w <= 2; w++) {
    run<4, 0>(w, 4000);
    run<4, 2>(w, 4000);
    run<4, 4>(w, 4000);
    run<4, 5>(w, 4000);
    run<4, 6>(w, 4000);
    run<4, 8>(w, 4000);
  }
  return 0;
}
