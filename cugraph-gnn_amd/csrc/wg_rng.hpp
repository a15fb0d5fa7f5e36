// PCG32 XSH-RR generator with the stream/offset conventions of the reference's sampling ops.
//
// The reference draws from raft::random::detail::PCGenerator (rapidsai/raft 26.10 — a third-party
// dependency that is NOT vendored under /root/reference; call sites:
// cpp/src/wholegraph_ops/unweighted_sample_without_replacement_func.cuh:137,183-187,357-358,
// weighted_sample_without_replacement_func.cuh:33-51,255, raft_random_gen.cu:32-50,69-95).
// Its published algorithm: state=0; inc=(subsequence<<1)|1; step; state+=seed; step; then the
// DeviceState constructor used by every call site skips ahead by `subsequence` draws
// (assumption A1 in DESIGN.md — the raw stream is "parity unpinned" against raft, pinned here to
// the PCG32 reference vectors in tests/test_oracle_rng.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace wgamd {

struct Pcg32 {
  uint64_t state;
  uint64_t inc;

  static constexpr uint64_t kMult = 6364136223846793005ULL;

  __host__ __device__ __forceinline__ uint32_t next_u32()
  {
    uint64_t old = state;
    state        = old * kMult + inc;
    uint32_t x   = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (x >> rot) | (x << ((0u - rot) & 31u));
  }

  // LCG jump by `delta` draws in O(log delta)
  __host__ __device__ __forceinline__ void skipahead(uint64_t delta)
  {
    uint64_t acc_mult = 1u, acc_plus = 0u, cur_mult = kMult, cur_plus = inc;
    while (delta) {
      if (delta & 1u) {
        acc_mult *= cur_mult;
        acc_plus = acc_plus * cur_mult + cur_plus;
      }
      cur_plus = (cur_mult + 1u) * cur_plus;
      cur_mult *= cur_mult;
      delta >>= 1u;
    }
    state = acc_mult * state + acc_plus;
  }

  // generator of the op stream `subsequence` for `seed`
  __host__ __device__ __forceinline__ Pcg32(uint64_t seed, uint64_t subsequence)
  {
    state = 0u;
    inc   = (subsequence << 1u) | 1u;
    (void)next_u32();
    state += seed;
    (void)next_u32();
    skipahead(subsequence);
  }

  __host__ __device__ __forceinline__ int32_t next_i31() { return (int32_t)(next_u32() & 0x7fffffffu); }
  __host__ __device__ __forceinline__ uint64_t next_u64()
  {
    uint64_t lo = next_u32();
    uint64_t hi = next_u32();
    return lo | (hi << 32u);
  }
  __host__ __device__ __forceinline__ int64_t next_i63() { return (int64_t)(next_u64() & 0x7fffffffffffffffULL); }
  __host__ __device__ __forceinline__ float next_f32() { return (float)(next_u32() >> 8u) / 16777216.0f; }
};

}  // namespace wgamd
