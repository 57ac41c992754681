// unrolled.hip: forward-backward over lattices that are never stored (unrolled.hpp): one wavefront per training
// pair, one lane per transducer state, positions walked in a loop.  gfx950 only.
//
// Arithmetic: the lattice of one pair is a chain of sparse matrix-vector products, so the sweep runs in the linear
// domain with one rescaling per position (the classic scaled forward-backward): alpha_hat[o] sums to 1, the scale
// c[o] goes into ln p.  A posterior is alpha_hat[o][src] * W * beta_hat[o+1][dst] / (c[o+1] * alpha_hat[L][final]);
// nothing underflows however long the string is, and no exp/log is spent per lattice arc (two per position).  The
// result equals the log-domain sweeps' to rounding (tests: 1e-9 on ln p, 1e-7 on counts).
//
// Counts: every arc adds its posterior to up to UNROLLED_MAX_CHAIN accumulator slots (the arc itself, or the unlocked
// parameters of its cascade chain) held in LDS per workgroup (ds_add_f64), written out as one partial vector per
// workgroup and summed in a fixed order by unrolled_reduce_kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "unrolled_args.hpp"

namespace carmel_hip {

#define U_MAX_WAVES 8
#define U_BATCH 9
#define U_NEG_INF (-__builtin_huge_val())

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// linear weights in table order: Wf / Wb / We follow f_arc / b_arc / e_arc
__global__ void unrolled_weights_kernel(const uint32_t* __restrict__ arcs, const double* __restrict__ logw,
                                        double* __restrict__ out, uint32_t stride, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t a = arcs[k];
  out[(size_t)k * stride] = a == 0xffffffffu ? 0.0 : exp(logw[a]);  // stride 2: the weight field of a URec
}

__global__ __launch_bounds__(64 * U_MAX_WAVES) void unrolled_sweep_kernel(UnrolledArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const uint32_t S = A.S;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n_waves = blockDim.x >> 6;
  double* acc = lds;                                             // n_slots accumulators, shared by the workgroup
  double* cs = lds + A.n_slots + (size_t)wave * (A.max_len + 2);  // this wave's scales c[o]
  // alpha_hat and beta_hat live in registers (lane = state); a value of another state comes by cross-lane read
  // (ds_bpermute), so LDS holds only the accumulators and many waves fit a CU -- they are what hides the L2 latency
  // of the table reads.  alpha_hat[o][lane] is parked in a scratch row (one coalesced store / load per position).
  double* rows = A.alpha_scratch + ((size_t)blockIdx.x * n_waves + wave) * (size_t)(A.max_len + 1) * 64 + lane;
  for (uint32_t k = threadIdx.x; k < A.n_slots; k += blockDim.x) acc[k] = 0.0;
  __syncthreads();
  const bool on = lane < S;
  const uint32_t ln = on ? lane : 0u;  // idle lanes shadow lane 0 and are masked out: the wave stays converged
  const uint32_t total_waves = gridDim.x * n_waves;
  for (uint64_t q = (uint64_t)blockIdx.x * n_waves + wave; q < A.n_pairs; q += total_waves) {
    const uint64_t s0 = A.seq_off[q];
    const uint32_t L = (uint32_t)(A.seq_off[q + 1] - s0);
    const uint16_t* xs = A.seq_sym + s0;
    // ---------- forward ----------
    double a = (lane == A.start) ? 1.0 : 0.0;
    for (uint32_t e = 0; e < A.n_eps; ++e) {  // *e*:*e* arcs in topological order of their sources
      const double v = __shfl(a, A.e_src[e], 64) * A.We[e];
      if (lane == A.e_dst[e]) a += v;
    }
    double lnz = 0.0;
    bool dead = false;
    {
      const double c0 = wave_sum(a);
      a /= c0;
      lnz = log(c0);
      rows[0] = a;
      if (lane == 0) cs[0] = c0;
    }
    for (uint32_t o = 0; o < L; ++o) {
      const uint32_t x = xs[o];
      const uint32_t off = A.f_off[x], deg = (A.f_off[x + 1] - off) / S;
      double v = 0.0;
      for (uint32_t it0 = 0; it0 < deg; it0 += U_BATCH) {  // loads of a batch issue together
        URec r[U_BATCH];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) r[j] = A.f_rec[off + (it0 + j < deg ? it0 + j : it0) * S + ln];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) {
          const double as = __shfl(a, (int)(r[j].other_slot2 & 0xffu), 64);
          v += (it0 + j < deg) ? as * r[j].w : 0.0;
        }
      }
      if (!on) v = 0.0;
      for (uint32_t e = 0; e < A.n_eps; ++e) {
        const double u = __shfl(v, A.e_src[e], 64) * A.We[e];
        if (lane == A.e_dst[e]) v += u;
      }
      const double c = wave_sum(v);
      if (!(c > 0.0)) {
        dead = true;
        break;
      }
      a = v / c;
      lnz += log(c);
      rows[(size_t)(o + 1) * 64] = a;
      if (lane == 0) cs[o + 1] = c;
    }
    const double pfin = dead ? 0.0 : __shfl(a, (int)A.fin, 64);
    const double lp = (dead || !(pfin > 0.0)) ? U_NEG_INF : lnz + log(pfin);
    if (lane == 0) A.pair_logprob[A.pair_id[q]] = lp;
    if (lp == U_NEG_INF) continue;
    // ---------- backward + posteriors ----------
    const double g = A.pair_weight[q] / pfin;  // "* weight / prob" (derivations.h:445)
    double b = (lane == A.fin) ? 1.0 : 0.0;
    // *e*:*e* arcs of the last position, in reverse order (a = alpha_hat[L] is still in the registers)
    for (uint32_t e = A.n_eps; e-- > 0;) {
      const uint32_t es = A.e_src[e], ed = A.e_dst[e];
      const double u = __shfl(b, ed, 64) * A.We[e];
      const double p = __shfl(a, es, 64) * u * g;
      if (lane == es) b += u;
      if (lane == 0 && p > 0.0)
        for (int j = 0; j < (int)UNROLLED_MAX_CHAIN; ++j) {
          const uint32_t sl = A.e_slot[e * UNROLLED_MAX_CHAIN + j];
          if (sl != UNROLLED_NO_SLOT) atomicAdd(acc + sl, p);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // this wave's scratch rows before it reads them back
    double a_o = L ? rows[(size_t)(L - 1) * 64] : 0.0;
    for (uint32_t o = L; o-- > 0;) {
      const uint32_t x = xs[o];
      const double a_next = o ? rows[(size_t)(o - 1) * 64] : 0.0;  // prefetch the next row
      const double co = cs[o + 1];
      const double ag = on ? a_o * (g / co) : 0.0;
      // one loop over the out-arcs of every source (lane = source) gives both beta_hat[o] and the posteriors: the
      // term W * beta_hat[o+1][dst] is the arc's share of beta, times alpha_hat[o][src] it is the arc's posterior
      const uint32_t off = A.b_off[x], deg = (A.b_off[x + 1] - off) / S;
      double v = 0.0;
      for (uint32_t it0 = 0; it0 < deg; it0 += U_BATCH) {
        URec r[U_BATCH];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) r[j] = A.b_rec[off + (it0 + j < deg ? it0 + j : it0) * S + ln];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) {
          const double bd = __shfl(b, (int)(r[j].other_slot2 & 0xffu), 64);
          const double term = (it0 + j < deg) ? bd * r[j].w : 0.0;
          v += term;
          const double p = ag * term;
          if (p > 0.0 && !A.debug_no_acc) {
            const uint32_t t0 = r[j].slot01 & 0xffffu, t1 = r[j].slot01 >> 16, t2 = r[j].other_slot2 >> 16;
            if (t0 != UNROLLED_NO_SLOT) atomicAdd(acc + t0, p);
            if (t1 != UNROLLED_NO_SLOT) atomicAdd(acc + t1, p);
            if (t2 != UNROLLED_NO_SLOT) atomicAdd(acc + t2, p);
          }
        }
      }
      b = on ? v / co : 0.0;
      for (uint32_t e = A.n_eps; e-- > 0;) {
        const uint32_t es = A.e_src[e], ed = A.e_dst[e];
        const double u = __shfl(b, ed, 64) * A.We[e];
        const double p = __shfl(a_o, es, 64) * u * g;
        if (lane == es) b += u;
        if (lane == 0 && p > 0.0)
          for (int j = 0; j < (int)UNROLLED_MAX_CHAIN; ++j) {
            const uint32_t sl = A.e_slot[e * UNROLLED_MAX_CHAIN + j];
            if (sl != UNROLLED_NO_SLOT) atomicAdd(acc + sl, p);
          }
      }
      a_o = a_next;
    }
  }
  __syncthreads();
  double* out = A.partial + (size_t)blockIdx.x * A.n_slots;
  for (uint32_t k = threadIdx.x; k < A.n_slots; k += blockDim.x) out[k] = acc[k];
}

// counts[slot] = sum over workgroups, in a fixed order
__global__ void unrolled_reduce_kernel(const double* __restrict__ partial, uint32_t n_wg, uint32_t n_slots,
                                       double* __restrict__ counts) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_slots) return;
  double v = 0.0;
  for (uint32_t g = 0; g < n_wg; ++g) v += partial[(size_t)g * n_slots + k];
  counts[k] = v;
}

// cascade: parameter counts = accumulated posteriors + the -f prior once per composed arc that uses the parameter
// (what chain_scatter_kernel produces from composed-arc counts); locked parameters get none
__global__ void unrolled_param_counts_kernel(double* __restrict__ out, const double* __restrict__ counts,
                                             const double* __restrict__ uses, double floor_count,
                                             const uint32_t* __restrict__ group, uint32_t n) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  out[p] = group[p] != 0u ? counts[p] + floor_count * uses[p] : 0.0;
}
hipError_t launch_unrolled_param_counts(double* out, const double* counts, const double* uses, double floor_count,
                                        const uint32_t* group, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(unrolled_param_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, counts, uses, floor_count, group, n);
  return hipGetLastError();
}

size_t unrolled_lds_bytes(const UnrolledArgs& A, uint32_t n_waves) {
  return (A.n_slots + (size_t)n_waves * (A.max_len + 2)) * sizeof(double);
}
// waves per workgroup (0: the accumulators alone do not fit)
uint32_t unrolled_waves(uint32_t n_slots, uint32_t max_len, uint32_t S) {
  (void)S;
  for (uint32_t w = U_MAX_WAVES; w >= 1; --w)
    if ((n_slots + (size_t)w * (max_len + 2)) * sizeof(double) <= 64 * 1024) return w;
  return 0;
}
// doubles of scratch for the alpha rows of n_wg workgroups
size_t unrolled_scratch_doubles(uint32_t n_wg, uint32_t n_waves, uint32_t max_len) {
  return (size_t)n_wg * n_waves * (max_len + 1) * 64;
}

hipError_t launch_unrolled_weights(const uint32_t* arcs, const double* logw, double* out, uint32_t stride_doubles, uint32_t n,
                                   hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(unrolled_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, s, arcs, logw, out, stride_doubles, n);
  return hipGetLastError();
}

hipError_t launch_unrolled_sweep(const UnrolledArgs& A, uint32_t n_wg, double* counts, hipStream_t s) {
  const uint32_t n_waves = unrolled_waves(A.n_slots, A.max_len, A.S);
  if (!n_waves) return hipErrorInvalidValue;
  const size_t lds = unrolled_lds_bytes(A, n_waves);
  hipLaunchKernelGGL(unrolled_sweep_kernel, dim3(n_wg), dim3(64 * n_waves), lds, s, A);
  hipLaunchKernelGGL(unrolled_reduce_kernel, dim3((A.n_slots + 255) / 256), dim3(256), 0, s, A.partial, n_wg, A.n_slots, counts);
  return hipGetLastError();
}

}  // namespace carmel_hip
