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

// ---- jump tables -----------------------------------------------------------------------------
// Jumping an LCG  s -> A*s + inc  ahead by n draws is the affine map  s -> A^n * s + inc * S(n),
// S(n) = 1 + A + ... + A^(n-1)  (mod 2^64), and  (A^m, S(m)) o (A^n, S(n)) = (A^m A^n, S(m) A^n + S(n)).
// Neither A^n nor S(n) depends on the stream (inc), so they can be tabulated once: with n written
// in base 256, n = c0 + c1*256 + c2*256^2 + c3*256^3, four lookups and three compositions replace
// the ~4*log2(n) dependent 64-bit multiplies of the generic jump loop — that loop was >50 % of the
// M <= 32 sampling kernel's issue cycles (64-bit multiplies are quarter-rate on CDNA).
struct PcgJump {
  uint64_t a;  // A^n
  uint64_t s;  // S(n)
};

struct PcgJumpTables {
  PcgJump t[4][256];
  constexpr PcgJumpTables() : t{}
  {
    uint64_t base_a = 6364136223846793005ULL, base_s = 1;  // jump by 256^k, k = 0
    for (int k = 0; k < 4; k++) {
      t[k][0].a = 1;
      t[k][0].s = 0;
      for (int c = 1; c < 256; c++) {
        t[k][c].a = t[k][c - 1].a * base_a;
        t[k][c].s = t[k][c - 1].s * base_a + base_s;
      }
      // (base)^256 by eight doublings:  (a, s) o (a, s) = (a*a, s*a + s)
      for (int d = 0; d < 8; d++) {
        uint64_t na = base_a * base_a, ns = base_s * base_a + base_s;
        base_a = na;
        base_s = ns;
      }
    }
  }
};

__device__ __constant__ const PcgJumpTables g_pcg_jump{};

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

  // jump ahead by n < 2^32 draws with the tables (at most 4 lookups, 3 compositions)
  __device__ __forceinline__ void jump_table(uint32_t n)
  {
    PcgJump j = g_pcg_jump.t[0][n & 255u];
#pragma unroll
    for (int k = 1; k < 4; k++) {
      const uint32_t c = (n >> (8 * k)) & 255u;
      if (c) {
        const PcgJump m = g_pcg_jump.t[k][c];
        j.s             = j.s * m.a + m.s;
        j.a             = j.a * m.a;
      }
    }
    state = j.a * state + inc * j.s;
  }

  // Same generator as Pcg32(seed, subsequence) for 0 <= subsequence < 2^31, via the jump tables; `extra_draws`
  // (< 2^31) positions it that many draws further along its stream in the same jump.
  struct table_tag {};
  __device__ __forceinline__ Pcg32(uint64_t seed, uint32_t subsequence, table_tag, uint32_t extra_draws = 0)
  {
    inc   = ((uint64_t)subsequence << 1u) | 1u;
    // two plain steps from state 0 with `seed` added in between:  ((0*A + inc) + seed)*A + inc
    state = (inc + seed) * kMult + inc;
    jump_table(subsequence + extra_draws);
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
