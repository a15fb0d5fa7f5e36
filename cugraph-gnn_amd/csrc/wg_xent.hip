// Softmax cross-entropy of a mini-batch's seed rows (include/wgamd_ext.h: wgamd_softmax_xent_*): the loss of every training loop
// of the reference (pylibwholegraph/torch/gnn_model.py:119-125, cugraph_pyg/examples/gcn_dist_mnmg.py: F.cross_entropy on the
// seeds' logits).  torch spends seven launches on it (log-softmax, nll, their backward, fills) — 35 us of a per-mini-batch
// training step whose kernels all sit on their launch floors; here the forward is ONE launch that keeps log-sum-exp per row,
// the backward ONE launch that multiplies by the upstream gradient on the device.
//   loss = sum_i w_i (lse_i - x[i, t_i]) / sum_i w_i   over rows with t_i >= 0 (torch's ignore_index for negative targets),
//   d x[i, c] = g w_i (exp(x[i, c] - lse_i) - [c == t_i]) / sum_i w_i.
// One wave per row; per-workgroup partial sums added in workgroup order by the last workgroup to finish (ticket): the loss
// is run-to-run deterministic.
#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

constexpr int kXentThreads = 256;

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// state: [0] loss, [1] sum of weights, then per workgroup (loss, weight) pairs, then the ticket (as int)
__global__ void __launch_bounds__(kXentThreads) xent_fwd_kernel(const float* __restrict__ x, int64_t ld, int64_t n_rows, int C,
                                                                const int64_t* __restrict__ target, const float* __restrict__ w,
                                                                float* __restrict__ lse, float* __restrict__ state,
                                                                float* __restrict__ loss_out)
{
  __shared__ float s_loss[kXentThreads / 64], s_w[kXentThreads / 64];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = kXentThreads / 64;
  float loss = 0.f, wsum = 0.f;   // (lane 0 of every wave carries its wave's sums)
  for (int64_t r = (int64_t)blockIdx.x * waves + wave; r < n_rows; r += (int64_t)gridDim.x * waves) {
    const float* row = x + r * ld;
    float m          = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
    m       = wave_max(m);
    float z = 0.f;
    for (int c = lane; c < C; c += 64) z += __expf(row[c] - m);
    z             = wave_sum(z);
    const float l = m + __logf(z);
    const int64_t t = target[r];
    if (lane == 0) {
      lse[r] = l;
      if (t >= 0 && t < C) {
        const float wi = w ? w[r] : 1.f;
        loss += wi * (l - row[t]);
        wsum += wi;
      }
    }
  }
  if (lane == 0) {
    s_loss[wave] = loss;
    s_w[wave]    = wsum;
  }
  __syncthreads();
  float* part = state + 2;
  int* ticket = reinterpret_cast<int*>(state + 2 + 2 * (int64_t)gridDim.x);
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < waves; k++) a += s_loss[k], b += s_w[k];
    part[2 * blockIdx.x]     = a;
    part[2 * blockIdx.x + 1] = b;
    __threadfence();
    s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {   // the whole last workgroup adds the partial sums up: thread k takes workgroups k, k + 256, ... in order, then a fixed tree
    __shared__ float t_loss[kXentThreads], t_w[kXentThreads];
    __threadfence();
    float a = 0.f, b = 0.f;
    for (unsigned k = threadIdx.x; k < gridDim.x; k += kXentThreads) {
      a += __hip_atomic_load(&part[2 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b += __hip_atomic_load(&part[2 * k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t_loss[threadIdx.x] = a;
    t_w[threadIdx.x]    = b;
    __syncthreads();
    for (int half = kXentThreads / 2; half >= 1; half >>= 1) {
      if ((int)threadIdx.x < half) {
        t_loss[threadIdx.x] += t_loss[threadIdx.x + half];
        t_w[threadIdx.x] += t_w[threadIdx.x + half];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      state[0]  = t_loss[0] / t_w[0];
      state[1]  = t_w[0];
      *loss_out = t_loss[0] / t_w[0];
      *ticket   = 0;   // (the state is reusable without a memset)
    }
  }
}

__global__ void __launch_bounds__(kXentThreads) xent_bwd_kernel(const float* __restrict__ x, int64_t ld, int64_t n_rows, int C,
                                                                const int64_t* __restrict__ target, const float* __restrict__ w,
                                                                const float* __restrict__ lse, const float* __restrict__ state,
                                                                const float* __restrict__ grad_loss, float* __restrict__ dx, int64_t ldd)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = kXentThreads / 64;
  const float g  = (grad_loss ? *grad_loss : 1.f) / state[1];
  for (int64_t r = (int64_t)blockIdx.x * waves + wave; r < n_rows; r += (int64_t)gridDim.x * waves) {
    const int64_t t = target[r];
    const bool live = t >= 0 && t < C;
    const float gi  = live ? g * (w ? w[r] : 1.f) : 0.f;
    const float l   = lse[r];
    const float* row = x + r * ld;
    float* out       = dx + r * ldd;
    for (int c = lane; c < C; c += 64) out[c] = live ? gi * (__expf(row[c] - l) - (c == t ? 1.f : 0.f)) : 0.f;
  }
}

inline int xent_grid(int64_t n_rows)
{
  return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_rows, kXentThreads / 64), 1024));
}

}  // namespace
}  // namespace wgamd

extern "C" {

size_t wgamd_softmax_xent_state_bytes(int64_t n_rows) { return sizeof(float) * (size_t)(2 + 2 * wgamd::xent_grid(n_rows) + 1) + 16; }

wholememory_error_code_t wgamd_softmax_xent_forward_f32(const float* logits, int64_t ld, int64_t n_rows, int n_classes,
                                                        const int64_t* target, const float* row_weight, float* lse, void* state,
                                                        int state_is_zeroed, float* loss_out, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_softmax_xent_forward_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && n_classes >= 1 && ld >= n_classes, "bad shape");
    WG_REQUIRE_INPUT(state && loss_out && (n_rows == 0 || (logits && target && lse)), "null pointer");
    auto st = static_cast<hipStream_t>(stream);
    if (!state_is_zeroed) WG_HIP_CHECK(hipMemsetAsync(state, 0, wgamd_softmax_xent_state_bytes(n_rows), st));
    xent_fwd_kernel<<<xent_grid(n_rows), kXentThreads, 0, st>>>(logits, ld, n_rows, n_classes, target, row_weight, lse,
                                                                 static_cast<float*>(state), loss_out);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_softmax_xent_backward_f32(const float* logits, int64_t ld, int64_t n_rows, int n_classes,
                                                         const int64_t* target, const float* row_weight, const float* lse,
                                                         const void* state, const float* grad_loss, float* grad_logits, int64_t ldg,
                                                         void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_softmax_xent_backward_f32", [&] {
    WG_REQUIRE_INPUT(n_rows >= 0 && n_classes >= 1 && ld >= n_classes && ldg >= n_classes, "bad shape");
    WG_REQUIRE_INPUT(state && (n_rows == 0 || (logits && target && lse && grad_logits)), "null pointer");
    if (n_rows == 0) return;
    xent_bwd_kernel<<<xent_grid(n_rows), kXentThreads, 0, static_cast<hipStream_t>(stream)>>>(
      logits, ld, n_rows, n_classes, target, row_weight, lse, static_cast<const float*>(state), grad_loss, grad_logits, ldg);
    WG_HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"
