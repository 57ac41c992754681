// rng.hpp — counter-based uniform generator shared by host and device.
// The reference draws from a global boost::lagged_fibonacci607 (graehl/shared/random.hpp:139-181), a sequential
// generator whose stream no reference test pins; a GPU sampler needs a stateless one.  u(seed, sweep, block, step)
// is a splitmix64 finaliser over the four counters, mapped to [0, 1) with 53 bits.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define CARMEL_HD __host__ __device__
#else
#define CARMEL_HD
#endif

CARMEL_HD inline uint64_t gibbs_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
CARMEL_HD inline double gibbs_uniform(uint64_t seed, uint32_t iter, uint32_t block, uint32_t step) {
  uint64_t h = gibbs_mix64(seed ^ 0xD1B54A32D192ED03ull);
  h = gibbs_mix64(h ^ ((uint64_t)iter << 32 | block));
  h = gibbs_mix64(h ^ (uint64_t)step);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
