// sweep_math.hpp — the log-semiring arithmetic the sweep kernels share (kernels.hip, tile_sweep.hip): f64 throughout.
// graehl/shared/weight.h:737-801 adds terms one at a time with log1p(exp(-|d|)); here a state's sum is one streaming
// log-sum-exp (running maximum + scaled sum), see kernels.hip's header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carmel_hip {

#define NEG_INF (-__builtin_huge_val())

// streaming logsumexp accumulator: value = m + log(acc).  A state with a single arc (the common case in sparse
// lattices) costs no exp and no log: acc stays exactly 1.
// timing experiment only (tools/lane_bench_real.hip -DCARMEL_FAKE_MATH): what the sweep costs without its f64
// transcendentals
#ifdef CARMEL_FAKE_MATH
#define K_EXP(x) ((x) * 0.5 + 1.0)
#define K_LOG(x) ((x) - 1.0)
#else
#ifdef CARMEL_LIBM_EXP
#define K_EXP(x) exp(x)
#else
#define K_EXP(x) exp_le0(x)
#endif
#ifdef CARMEL_LIBM_LOG
#define K_LOG(x) log(x)
#else
#define K_LOG(x) log_ge1(x)
#endif
#endif
// ln(a) for a finite a >= 1 -- all a streaming log-sum-exp ever asks for: its scaled sum holds exp(0) = 1 for the largest term and
// at most the in-degree.  The library's log spends most of its ~80 instructions on arguments that cannot occur here (denormals,
// zero, negatives, infinities); this one is a frexp, one division and an odd series in s = (m - 1) / (m + 1), m in
// [sqrt(1/2), sqrt(2)): ln m = 2 s (1 + z/3 + z^2/5 + ... + z^10/21), z = s^2 <= 0.0295 (the first dropped term is below 2^-60
// of the sum).  A third of the instructions; within 2 ulp of the library's (tests/test_gpu_parity.py compares every sweep with the
// oracle's libm arithmetic).  On the ambiguous workloads the log per state was what the lane sweeps' arithmetic was made of.
// e^x for the arguments the sweeps have: differences to a running maximum and log-posteriors (x <= 0 up to rounding), -inf for
// dead arcs and padding.  Cody-Waite reduction by ln 2, Taylor to r^13 on |r| <= ln2 / 2 (remainder below 2^-57), ldexp -- which
// also carries arguments below -745 through the denormals to 0; no branches for overflow or NaN inputs, which cannot occur.
__device__ __forceinline__ double exp_le0(double x) {
  x = fmax(x, -1100.0);  // (-inf included; 2^-1587 is 0 through ldexp)
  const double n = rint(x * 1.44269504088896340736);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}
__device__ __forceinline__ double log_ge1(double a) {
  int e;
  double m = frexp(a, &e);  // [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double p = 1.0 / 21.0;
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  const double s2 = s + s;
  const double r = fma(s2 * z, p, s2);  // ln m
  const double de = (double)e;
  return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, r));  // e * ln2 (hi + lo) + ln m
}
struct Lse {
  double m, acc;
  __device__ __forceinline__ void init() {
    m = NEG_INF;
    acc = 0.0;
  }
  __device__ __forceinline__ void add(double x) {
    if (x == NEG_INF) return;
    if (x <= m) {
      acc += K_EXP(x - m);
    } else {
      acc = (m == NEG_INF) ? 1.0 : acc * K_EXP(m - x) + 1.0;
      m = x;
    }
  }
  __device__ __forceinline__ double value() const { return acc == 1.0 ? m : (acc > 0.0 ? m + K_LOG(acc) : NEG_INF); }
};

// ---------------- blocked transposition (TransBucket, lattice.hpp) ----------------
// Workgroup b runs on XCD b % 8 (each XCD has its own L2).  Neighbouring tiles / buckets read neighbouring runs of X --
// the 128-byte lines at the run boundaries are shared -- so neighbours are given to the same XCD, back to back:
// work item = (b % 8) * ceil(n / 8) + b / 8.
__device__ __forceinline__ uint32_t xcd_chunked(uint32_t b, uint32_t n) {
  const uint32_t per = (n + 7) / 8;
  return (b & 7u) * per + (b >> 3);
}

}  // namespace carmel_hip
