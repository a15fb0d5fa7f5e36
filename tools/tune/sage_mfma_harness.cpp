// Tuning harness of the one-kernel SAGE layer (csrc/wg_sage_mfma.hip): compiled here once per variant
// (-DWG_MFMA_PW=.. -DWG_MFMA_DEPTH=..), run on the GPU box without Python.  Products layer-1 shape of one call group.
//   sage_mfma_harness [n_dst] [debug mask, see mfma_args::debug]
#define WG_MFMA_TUNE_HARNESS 1
#ifndef WG_HARNESS_KSC
#define WG_HARNESS_KSC 100
#endif
#include "../../cugraph-gnn_amd/csrc/wg_sage_mfma.hip"

#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv)
{
  using namespace wgamd;
  const int64_t n_dst = argc > 1 ? atoll(argv[1]) : 550000;
  const int F = 100, N = 256;
  const int64_t n_src = n_dst * 6 + 1000;
  std::mt19937_64 rng(1);
  std::vector<int> rp(n_dst + 1, 0);
  for (int64_t i = 0; i < n_dst; i++) rp[i + 1] = rp[i] + 9 + (int)(rng() % 2) - (rng() % 8 == 0 ? 4 : 0);  // mean ~9.3
  const int64_t E = rp[n_dst];
  std::vector<int> col(E);
  for (auto& c : col) c = (int)(rng() % n_src);
  std::vector<int64_t> self(n_dst);
  for (auto& r : self) r = (int64_t)(rng() % n_src);
  std::vector<float> x((size_t)n_src * F), w((size_t)2 * F * N), bias(N, 0.1f);
  for (auto& v : x) v = (float)((rng() >> 40) * (1.0 / (1 << 24))) - 0.5f;
  for (auto& v : w) v = (float)((rng() >> 40) * (1.0 / (1 << 24))) - 0.5f;
  int *d_rp, *d_col; int64_t* d_self; float *d_x, *d_w, *d_bias, *d_out; float* d_planes;
  const int KS = (2 * F + 15) / 16;
  const size_t planes_bytes = (size_t)KS * N * 64;
  hipMalloc(&d_rp, rp.size() * 4); hipMalloc(&d_col, col.size() * 4); hipMalloc(&d_self, self.size() * 8);
  hipMalloc(&d_x, x.size() * 4); hipMalloc(&d_w, w.size() * 4); hipMalloc(&d_bias, N * 4);
  hipMalloc(&d_out, (size_t)n_dst * N * 4); hipMalloc(&d_planes, planes_bytes);
  hipMemcpy(d_rp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_col, col.data(), col.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_self, self.data(), self.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_bias, bias.data(), N * 4, hipMemcpyHostToDevice);
  tile_weight_kernel<<<1024, 256>>>(d_w, N, 2 * F, N, KS, d_planes);
  mfma_args a{d_rp, d_col, n_dst, d_x, F, (uint32_t)(x.size() * 4), F, nullptr, d_self, 1, d_planes, N, KS, d_bias, 1, d_out, N, row_stride_dw(F), 0, nullptr, (int64_t)F * 4, nullptr, 0};
  unsigned long long* d_stamps; hipMalloc(&d_stamps, 64 * 16 * 8); hipMemset(d_stamps, 0, 64 * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)E * (4 * F + 4) + (double)n_dst * (4 * F + 16) + (double)n_dst * 4 * N;
  const int modes[] = {192, 192 | 16, 192 | 32, 192 | 4, 192 | 2, 192 | 1};
  std::vector<int> run_modes(modes, modes + sizeof(modes) / sizeof(modes[0]));
  if (argc > 2) run_modes.assign(1, atoi(argv[2]));
  for (int mode : run_modes) {
    a.debug = mode;
    for (int i = 0; i < 3; i++) launch<void, 32, 64, 4, WG_HARNESS_KSC>(a, 0);
    hipDeviceSynchronize();
    const int iters = 10;
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) launch<void, 32, 64, 4, WG_HARNESS_KSC>(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    printf("FC=%d DEPTH=%d mode=%d n_dst=%lld E=%lld: %.3f ms  %.2f TB/s (%.3f of 8 TB/s)\n", WG_HARNESS_KSC, WG_MFMA_DEPTH, mode,
           (long long)n_dst, (long long)E, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
  }
  if (argc > 3) {   // timeline of workgroup 0 (shader clock ticks)
    a.debug = atoi(argv[2]); a.stamps = d_stamps;
    launch<void, 32, 64, 4, WG_HARNESS_KSC>(a, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(64 * 16);
    hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    for (int n = 8; n < 11; n++) {
      const unsigned long long t0 = st[(n * 8) * 2];
      printf("step %2d (begin -> work done, ticks since the step began on wave 0):", n);
      for (int w = 0; w < 8; w++) printf("  w%d %lld->%lld", w, (long long)(st[(n * 8 + w) * 2] - t0), (long long)(st[(n * 8 + w) * 2 + 1] - t0));
      printf("   | next step begins %lld\n", (long long)(st[((n + 1) * 8) * 2] - t0));
    }
  }
  return 0;
}
