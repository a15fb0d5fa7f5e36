// Test infrastructure: prints draws of the OFFICIAL PCG C++ library's pcg32 (setseq_xsh_rr_64_32, M. E. O'Neill,
// pcg-random.org), as vendored by Apache Arrow in this image (pyarrow/include/arrow/vendored/pcg/pcg_random.hpp — a
// third-party header that is part of the image, not of /root/reference and not of this repo).  The oracle's PCG32 core
// (oracle/wg_oracle.c: wgo_pcg_init / wgo_pcg_u32 / skip-ahead) is pinned against it in tests/test_oracle_rng.py.
// usage: pcg_official seed stream advance n  ->  n draws of pcg32(seed, stream) after advance(advance), one hex word per line
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "arrow/vendored/pcg/pcg_random.hpp"

int main(int argc, char** argv)
{
  if (argc != 5) return 2;
  const uint64_t seed = strtoull(argv[1], nullptr, 0), stream = strtoull(argv[2], nullptr, 0);
  const uint64_t adv = strtoull(argv[3], nullptr, 0);
  const int n = atoi(argv[4]);
  arrow_vendored::pcg32 rng(seed, stream);
  rng.advance(adv);
  for (int i = 0; i < n; i++) printf("%08x\n", (unsigned)rng());
  return 0;
}
