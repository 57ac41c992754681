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

// Annealing (gibbs_opts.hpp:206-211, gibbs.hpp:838-839): sweep t of n chooses with probabilities raised to 1/T(t),
// T = graehl::clamped_time_series(high, low, n, curvature = linear = -1e8) (time_series.hpp:90-141): an exponential
// between the two temperatures measured from an origin 1e8 * low below zero, i.e. a straight line to ~1e-8.
// the standard normal's cdf and quantile for the prior-scale proposals (gibbs.hip; Wichura's AS 241)
double gibbs_norm_cdf(double z);
double gibbs_norm_quantile(double p);
inline double gibbs_anneal_power(double high, double low, uint32_t n_sweeps, uint32_t sweep) {
  if (high == 0) high = 1;  // unset fields of a zeroed options struct
  if (low == 0) low = 1;
  double temp;
  if (n_sweeps == 0 || high == low)
    temp = high;
  else {
    const double x_origin = low * -100000000.0, x0 = high - x_origin, k = (low - x_origin) / x0;
    const double t = (double)sweep, t_max = (double)n_sweeps;
    temp = t <= 0 ? x0 + x_origin : t >= t_max ? x0 * k + x_origin : x0 * __builtin_pow(k, t / t_max) + x_origin;
  }
  return temp > 0 ? 1. / temp : 1.;
}
