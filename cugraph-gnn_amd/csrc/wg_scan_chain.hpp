// Single-pass ("chained", decoupled look-back) exclusive scan building blocks, shared by the kernels that scan and do
// something else in the same launch (wg_scan.hip: the plain scan; wg_sample.hip: sample counts + offsets of a hop).
//
// A scan over m tiles runs as ONE launch: a workgroup takes the next tile from a ticket counter, reduces it, publishes the
// tile's aggregate in a 64-bit state word, looks back over the words of the tiles before it until it meets one that already
// knows its inclusive prefix, and publishes its own.  Tickets make every tile below mine belong to a workgroup that is
// already running (forward progress: an aggregate is published before anything is waited for).
//
// The state lives in a LIBRARY-OWNED buffer per (device, stream) — never in caller scratch — and is never cleared: a word is
// [epoch:30 | flag:2 | value:32], every launch on the stream gets the next epoch (scan_chain_acquire), and a word of another
// epoch reads as "not there yet".  The two counters (ticket, workgroups done) are reset by the last workgroup to finish, so
// the next launch on the stream finds zeros.  Launches sharing one state buffer are stream-ordered by construction.
#pragma once
#include "wg_common.hpp"

namespace wgamd {

constexpr unsigned kChainAggregate = 1u, kChainPrefix = 2u;
constexpr int kChainThreads = 256;
constexpr int kChainItems   = kScanTile / kChainThreads;   // 8 consecutive items per thread (blocked arrangement)

__device__ __forceinline__ unsigned long long chain_pack(unsigned epoch, unsigned flag, int value)
{
  return ((unsigned long long)epoch << 34) | ((unsigned long long)flag << 32) | (unsigned long long)(unsigned)value;
}

__device__ __forceinline__ int chain_wave_inclusive_scan(int v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int up = __shfl_up(v, d, 64);
    if (lane >= d) v += up;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread workgroup; returns the exclusive prefix, the total in *total
__device__ __forceinline__ int chain_block_exclusive_scan(int v, int* total)
{
  __shared__ int wave_sums[kChainThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = chain_wave_inclusive_scan(v);
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  int wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kChainThreads / 64; w++) {
    int s = wave_sums[w];
    if (w < wave) wave_off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wave_off + inc - v;
}

// the next tile of this launch (every thread of the workgroup gets the same ticket)
__device__ __forceinline__ unsigned chain_next_tile(const scan_chain& c)
{
  __shared__ unsigned s_ticket;
  if (threadIdx.x == 0) s_ticket = atomicAdd(c.counters, 1u);
  __syncthreads();
  const unsigned t = s_ticket;
  __syncthreads();
  return t;
}

// Exclusive prefix (sum of the aggregates of all tiles below `tile`), called by every thread of the workgroup with the
// tile's aggregate; wave 0 publishes and looks back, 64 predecessor words per step.
__device__ __forceinline__ int chain_exclusive_prefix(const scan_chain& c, unsigned tile, int aggregate)
{
  __shared__ int s_prefix;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int exclusive  = 0;
    if (tile == 0) {
      if (lane == 0)
        __hip_atomic_store(c.tiles, chain_pack(c.epoch, kChainPrefix, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0)
        __hip_atomic_store(c.tiles + tile, chain_pack(c.epoch, kChainAggregate, aggregate), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      int64_t t = (int64_t)tile - 1;
      while (true) {
        const int64_t idx    = t - lane;
        unsigned long long w = chain_pack(c.epoch, kChainPrefix, 0);   // "before tile 0": a prefix of 0
        bool ok              = idx < 0;
        while (true) {
          if (!ok) {
            w  = __hip_atomic_load(c.tiles + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(w >> 34) == c.epoch && ((unsigned)(w >> 32) & 3u) != 0u;
          }
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(2);
        }
        const bool is_prefix          = ((unsigned)(w >> 32) & 3u) == kChainPrefix;
        const unsigned long long mask = __ballot(is_prefix);
        const int value               = (int)(unsigned)(w & 0xffffffffull);
        if (mask != 0ull) {   // lanes 0 .. first (the nearest prefix) close the sum
          const int first = __ffsll((long long)mask) - 1;
          int part        = lane <= first ? value : 0;
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
          exclusive += part;
          break;
        }
        int part = value;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        exclusive += part;
        t -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(c.tiles + tile, chain_pack(c.epoch, kChainPrefix, exclusive + aggregate), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = exclusive;
  }
  __syncthreads();
  const int p = s_prefix;
  __syncthreads();
  return p;
}

// once per workgroup, after its last tile: the last workgroup to get here zeroes the counters for the stream's next launch
__device__ __forceinline__ void chain_finish(const scan_chain& c)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(c.counters + 1, 1u);
    if (done == gridDim.x - 1) {
      atomicExch(c.counters, 0u);
      atomicExch(c.counters + 1, 0u);
    }
  }
}

}  // namespace wgamd
