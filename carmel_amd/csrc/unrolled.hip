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
#ifndef U_BATCH
#define U_BATCH 9
#endif
#ifndef U_WAVES_PER_EU
#define U_WAVES_PER_EU 4
#endif
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

// WD = lanes per training pair (16, 32 or 64: the smallest that holds the S states): a wavefront sweeps 64 / WD pairs
// side by side, each in its own group of WD lanes, so that a 28-state transducer keeps 56 of the 64 lanes busy in
// every table read, cross-lane read and LDS add instead of 28.  The pairs of a wavefront are neighbours in a list
// sorted by length; the forward passes start together, the backward passes start together (each pair walks its own
// positions L-1 .. 0), and whatever differs between the pairs (symbol, table offset, scale) is a per-lane value.
template <int WD>
__device__ __forceinline__ double sub_sum(double v) {
#pragma unroll
  for (int o = WD / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WD>
__device__ __forceinline__ uint32_t across_max(uint32_t v) {  // max over the pairs of the wavefront
#pragma unroll
  for (int o = 32; o >= WD; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
  return v;
}

template <int WD>
__global__ __launch_bounds__(64 * U_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(U_WAVES_PER_EU)))
void unrolled_sweep_kernel(UnrolledArgs A) {
  constexpr uint32_t P = 64 / WD;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const uint32_t S = A.S;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n_waves = blockDim.x >> 6;
  const uint32_t sub = lane / WD, sl = lane % WD, base = sub * WD;
  double* acc = lds;                                                        // n_slots accumulators, shared by the workgroup
  double* cs = lds + A.n_slots + ((size_t)wave * P + sub) * (A.max_len + 2);  // this pair's scales c[o]
  // alpha_hat and beta_hat live in registers (lane = state); a value of another state comes by cross-lane read
  // (ds_bpermute), so LDS holds only the accumulators and many waves fit a CU -- they are what hides the L2 latency
  // of the table reads.  alpha_hat[o][lane] is parked in a scratch row (one coalesced store / load per position).
  double* rows = A.alpha_scratch + ((size_t)blockIdx.x * n_waves + wave) * (size_t)(A.max_len + 1) * 64 + lane;
  for (uint32_t k = threadIdx.x; k < A.n_slots; k += blockDim.x) acc[k] = 0.0;
  __syncthreads();
  const bool on = sl < S;
  const uint32_t ln = on ? sl : 0u;  // idle lanes shadow state 0 and are masked out: the wave stays converged
  const uint64_t stride = (uint64_t)gridDim.x * n_waves * P;
  for (uint64_t q0 = ((uint64_t)blockIdx.x * n_waves + wave) * P; q0 < A.n_pairs; q0 += stride) {
    const uint64_t q = q0 + sub;
    const bool valid = q < A.n_pairs;
    const uint64_t s0 = valid ? A.seq_off[q] : 0;
    const uint32_t L = valid ? (uint32_t)(A.seq_off[q + 1] - s0) : 0u;
    const uint32_t Lmax = across_max<WD>(L);
    const uint16_t* xs = A.seq_sym + s0;
    // ---------- forward ----------
    double a = (valid && sl == A.start) ? 1.0 : 0.0;
    for (uint32_t e = 0; e < A.n_eps; ++e) {  // *e*:*e* arcs in topological order of their sources
      const double v = __shfl(a, (int)(base + A.e_src[e]), 64) * A.We[e];
      if (sl == A.e_dst[e]) a += v;
    }
    double lnz = 0.0;
    bool dead = !valid;
    {
      const double c0 = sub_sum<WD>(a);
      if (valid) {
        a /= c0;
        lnz = log(c0);
        rows[0] = a;
        if (sl == 0) cs[0] = c0;
      }
    }
    for (uint32_t o = 0; o < Lmax; ++o) {
      const bool act = !dead && o < L;
      const uint32_t x = act ? xs[o] : 0u;
      double v = 0.0;
      if (A.f_deg_u) {  // slabs of one size: scalar loop bounds, one address per position, no masks (padding rows weigh 0)
        const uint32_t deg = A.f_deg_u;
        const URec* __restrict__ p = A.f_rec + (size_t)x * deg * S + ln;
        uint32_t it0 = 0;
        for (; it0 + U_BATCH <= deg; it0 += U_BATCH) {
          URec r[U_BATCH];
#pragma unroll
          for (int j = 0; j < U_BATCH; ++j) r[j] = p[(size_t)(it0 + j) * S];
#pragma unroll
          for (int j = 0; j < U_BATCH; ++j) v += __shfl(a, (int)(base + (r[j].other_slot2 & 0x3ffu)), 64) * r[j].w;
        }
        for (; it0 < deg; ++it0) {
          const URec r = p[(size_t)it0 * S];
          v += __shfl(a, (int)(base + (r.other_slot2 & 0x3ffu)), 64) * r.w;
        }
      } else {
      const uint32_t row0 = A.f_off[x], deg = A.f_off[x + 1] - row0;  // in rows of S entries
      const uint32_t degmax = across_max<WD>(deg);
      for (uint32_t it0 = 0; it0 < degmax; it0 += U_BATCH) {  // loads of a batch issue together
        URec r[U_BATCH];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) r[j] = A.f_rec[(size_t)(row0 + min(it0 + j, deg - 1)) * S + ln];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) {
          const double as = __shfl(a, (int)(base + (r[j].other_slot2 & 0x3ffu)), 64);
          v += (it0 + j < deg) ? as * r[j].w : 0.0;
        }
      }
      }
      if (!on) v = 0.0;
      for (uint32_t e = 0; e < A.n_eps; ++e) {
        const double u = __shfl(v, (int)(base + A.e_src[e]), 64) * A.We[e];
        if (sl == A.e_dst[e]) v += u;
      }
      const double c = sub_sum<WD>(v);
      if (act) {
        if (!(c > 0.0))
          dead = true;
        else {
          a = v / c;
          lnz += log(c);
          rows[(size_t)(o + 1) * 64] = a;
          if (sl == 0) cs[o + 1] = c;
        }
      }
    }
    const double pfin = __shfl(a, (int)(base + A.fin), 64);
    const double lp = (dead || !(pfin > 0.0)) ? U_NEG_INF : lnz + log(pfin);
    if (valid && sl == 0) A.pair_logprob[A.pair_id[q]] = lp;
    const bool live = valid && lp != U_NEG_INF;
    if (!__ballot(live)) continue;
    if (A.debug_no_acc & 2u) continue;  // timing experiment: forward only
    // ---------- backward + posteriors ----------
    const double g = live ? A.pair_weight[q] / pfin : 0.0;  // "* weight / prob" (derivations.h:445)
    double b = (live && sl == A.fin) ? 1.0 : 0.0;
    // *e*:*e* arcs of the last position, in reverse order (a = alpha_hat[L] is still in the registers)
    for (uint32_t e = A.n_eps; e-- > 0;) {
      const uint32_t es = A.e_src[e], ed = A.e_dst[e];
      const double u = __shfl(b, (int)(base + ed), 64) * A.We[e];
      const double p = __shfl(a, (int)(base + es), 64) * u * g;
      if (sl == es) b += u;
      if (live && sl == 0 && p > 0.0)
        for (int j = 0; j < (int)UNROLLED_MAX_CHAIN; ++j) {
          const uint32_t t = A.e_slot[e * UNROLLED_MAX_CHAIN + j];
          if (t != UNROLLED_NO_SLOT) atomicAdd(acc + t, p);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // this wave's scratch rows before it reads them back
    double a_o = (live && L) ? rows[(size_t)(L - 1) * 64] : 0.0;
    for (uint32_t oo = 0; oo < Lmax; ++oo) {
      const bool act = live && oo < L;
      const uint32_t o = act ? L - 1 - oo : 0u;
      const uint32_t x = act ? xs[o] : 0u;
      const double a_next = (act && o) ? rows[(size_t)(o - 1) * 64] : 0.0;  // prefetch the next row
      const double co = act ? cs[o + 1] : 1.0;
      const double ag = (on && act) ? a_o * (g / co) : 0.0;
      // one loop over the out-arcs of every source (lane = source) gives both beta_hat[o] and the posteriors: the
      // term W * beta_hat[o+1][dst] is the arc's share of beta, times alpha_hat[o][src] it is the arc's posterior
      double v = 0.0;
      if (A.b_deg_u) {
        const uint32_t deg = A.b_deg_u;
        const URec* __restrict__ p = A.b_rec + (size_t)x * deg * S + ln;
#define U_BWD_TERM(R)                                                                                       \
  {                                                                                                         \
    const double term = __shfl(b, (int)(base + ((R).other_slot2 & 0x3ffu)), 64) * (R).w;                     \
    v += term;                                                                                              \
    const double pp = ag * term;                                                                            \
    if (pp > 0.0 && !(A.debug_no_acc & 1u)) {                                                               \
      const uint32_t t0 = (R).slot01 & 0xffffu, t1 = (R).slot01 >> 16, t2 = (R).other_slot2 >> 16;          \
      if (t0 != UNROLLED_NO_SLOT) atomicAdd(acc + t0, pp);                                                  \
      if (t1 != UNROLLED_NO_SLOT) atomicAdd(acc + t1, pp);                                                  \
      if (t2 != UNROLLED_NO_SLOT) atomicAdd(acc + t2, pp);                                                  \
    }                                                                                                       \
  }
        uint32_t it0 = 0;
        for (; it0 + U_BATCH <= deg; it0 += U_BATCH) {
          URec r[U_BATCH];
#pragma unroll
          for (int j = 0; j < U_BATCH; ++j) r[j] = p[(size_t)(it0 + j) * S];
#pragma unroll
          for (int j = 0; j < U_BATCH; ++j) U_BWD_TERM(r[j])
        }
        for (; it0 < deg; ++it0) {
          const URec r = p[(size_t)it0 * S];
          U_BWD_TERM(r)
        }
#undef U_BWD_TERM
      } else {
      const uint32_t row0 = A.b_off[x], deg = A.b_off[x + 1] - row0;
      const uint32_t degmax = across_max<WD>(deg);
      for (uint32_t it0 = 0; it0 < degmax; it0 += U_BATCH) {
        URec r[U_BATCH];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) r[j] = A.b_rec[(size_t)(row0 + min(it0 + j, deg - 1)) * S + ln];
#pragma unroll
        for (int j = 0; j < U_BATCH; ++j) {
          const double bd = __shfl(b, (int)(base + (r[j].other_slot2 & 0x3ffu)), 64);
          const double term = (it0 + j < deg) ? bd * r[j].w : 0.0;
          v += term;
          const double p = ag * term;
          if (p > 0.0 && !(A.debug_no_acc & 1u)) {
            const uint32_t t0 = r[j].slot01 & 0xffffu, t1 = r[j].slot01 >> 16, t2 = r[j].other_slot2 >> 16;
            if (t0 != UNROLLED_NO_SLOT) atomicAdd(acc + t0, p);
            if (t1 != UNROLLED_NO_SLOT) atomicAdd(acc + t1, p);
            if (t2 != UNROLLED_NO_SLOT) atomicAdd(acc + t2, p);
          }
        }
      }
      }
      if (act) b = on ? v / co : 0.0;
      for (uint32_t e = A.n_eps; e-- > 0;) {
        const uint32_t es = A.e_src[e], ed = A.e_dst[e];
        const double u = __shfl(b, (int)(base + ed), 64) * A.We[e];
        const double p = __shfl(a_o, (int)(base + es), 64) * u * g;
        if (act && sl == es) b += u;
        if (act && sl == 0 && p > 0.0)
          for (int j = 0; j < (int)UNROLLED_MAX_CHAIN; ++j) {
            const uint32_t t = A.e_slot[e * UNROLLED_MAX_CHAIN + j];
            if (t != UNROLLED_NO_SLOT) atomicAdd(acc + t, p);
          }
      }
      if (act) a_o = a_next;
    }
  }
  __syncthreads();
  double* out = A.partial + (size_t)blockIdx.x * A.n_slots;
  for (uint32_t k = threadIdx.x; k < A.n_slots; k += blockDim.x) out[k] = acc[k];
}

// ---------------- more than 64 states: a workgroup per pair, a thread per state ----------------
// The same scaled forward-backward with alpha_hat / beta_hat in LDS (double-buffered: a position's values are read by
// every thread while the next position's are written), the position's scale from a workgroup-wide sum, the parked
// alpha_hat rows S wide.  Two barriers per position (four when the transducer has *e*:*e* arcs).  Used for one-tape
// transducers of 65 .. 1024 states (e.g. a trigram character model under a substitution channel: 729 states).
__device__ __forceinline__ double wide_sum(double v, double* part, uint32_t n_waves) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (uint32_t k = 0; k < n_waves; ++k) t += part[k];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(1024) void unrolled_wide_kernel(UnrolledArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const uint32_t S = A.S, tid = threadIdx.x, n_waves = blockDim.x >> 6;
  const bool on = tid < S;
  const uint32_t ln = on ? tid : 0u;
  double* acc = lds;                       // n_slots accumulators
  double* va = acc + A.n_slots;            // two buffers of S values (alpha_hat, then beta_hat)
  double* vb = va + S;
  double* part = vb + S;                   // 16 partial sums
  double* cs = part + 16;                  // scales c[o] of the current pair
  const size_t Sp = (size_t)blockDim.x;    // row pitch of the parked alpha_hat rows
  double* rows = A.alpha_scratch + (size_t)blockIdx.x * (size_t)(A.max_len + 1) * Sp + tid;
  for (uint32_t k = tid; k < A.n_slots; k += blockDim.x) acc[k] = 0.0;
  __syncthreads();
  auto eps_forward = [&](double* buf, double v) -> double {  // *e*:*e* arcs in topological order of their sources
    if (!A.n_eps) return v;
    if (on) buf[tid] = v;
    __syncthreads();
    if (tid == 0)
      for (uint32_t e = 0; e < A.n_eps; ++e) buf[A.e_dst[e]] += buf[A.e_src[e]] * A.We[e];
    __syncthreads();
    const double r = on ? buf[tid] : 0.0;
    __syncthreads();
    return r;
  };
  for (uint64_t q = blockIdx.x; q < A.n_pairs; q += gridDim.x) {
    const uint64_t s0 = A.seq_off[q];
    const uint32_t L = (uint32_t)(A.seq_off[q + 1] - s0);
    const uint16_t* xs = A.seq_sym + s0;
    double* cur = va;
    double* nxt = vb;
    // ---------- forward ----------
    double a = eps_forward(nxt, tid == A.start ? 1.0 : 0.0);
    double lnz = 0.0;
    bool dead = false;
    {
      const double c0 = wide_sum(a, part, n_waves);
      a /= c0;
      lnz = log(c0);
      rows[0] = a;
      if (on) cur[tid] = a;
      if (tid == 0) cs[0] = c0;
    }
    __syncthreads();
    for (uint32_t o = 0; o < L; ++o) {
      const uint32_t x = xs[o];
      const uint32_t row0 = A.f_off[x], deg = A.f_off[x + 1] - row0;
      const URec* __restrict__ p = A.f_rec + (size_t)row0 * S + ln;
      double v = 0.0;
      for (uint32_t it0 = 0; it0 < deg; it0 += 4) {
        URec r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = p[(size_t)min(it0 + j, deg - 1) * S];
#pragma unroll
        for (int j = 0; j < 4; ++j) v += (it0 + j < deg) ? cur[r[j].other_slot2 & 0x3ffu] * r[j].w : 0.0;
      }
      if (!on) v = 0.0;
      v = eps_forward(nxt, v);
      const double c = wide_sum(v, part, n_waves);
      if (!(c > 0.0)) {
        dead = true;
        break;
      }
      a = v / c;
      lnz += log(c);
      rows[(size_t)(o + 1) * Sp] = a;
      if (on) nxt[tid] = a;
      if (tid == 0) cs[o + 1] = c;
      __syncthreads();
      double* t = cur;
      cur = nxt;
      nxt = t;
    }
    const double pfin = dead ? 0.0 : cur[A.fin];
    const double lp = (dead || !(pfin > 0.0)) ? U_NEG_INF : lnz + log(pfin);
    if (tid == 0) A.pair_logprob[A.pair_id[q]] = lp;
    __syncthreads();
    if (lp == U_NEG_INF) continue;
    // ---------- backward + posteriors ----------
    const double g = A.pair_weight[q] / pfin;
    // beta_hat in the two buffers; alpha_hat[L] is still in the register a
    double* bc = cur;  // (alpha's last buffer is free: pfin has been read by everyone behind the barrier above)
    double* bn = nxt;
    double b = (tid == A.fin) ? 1.0 : 0.0;
    auto eps_backward = [&](double* bbuf, double bval, double a_here) -> double {
      // *e*:*e* arcs inside a position, in reverse order; their posteriors go to the accumulators
      if (!A.n_eps) return bval;
      if (on) bbuf[tid] = bval;
      if (on) bn[tid] = a_here;  // alpha_hat of this position, for thread 0
      __syncthreads();
      if (tid == 0)
        for (uint32_t e = A.n_eps; e-- > 0;) {
          const uint32_t es = A.e_src[e], ed = A.e_dst[e];
          const double u = bbuf[ed] * A.We[e];
          const double pp = bn[es] * u * g;
          bbuf[es] += u;
          if (pp > 0.0)
            for (int j = 0; j < (int)UNROLLED_MAX_CHAIN; ++j) {
              const uint32_t t = A.e_slot[e * UNROLLED_MAX_CHAIN + j];
              if (t != UNROLLED_NO_SLOT) acc[t] += pp;
            }
        }
      __syncthreads();
      const double r = on ? bbuf[tid] : 0.0;
      __syncthreads();
      return r;
    };
    b = eps_backward(bc, b, a);
    if (on) bc[tid] = b;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    double a_o = L ? rows[(size_t)(L - 1) * Sp] : 0.0;
    for (uint32_t o = L; o-- > 0;) {
      const uint32_t x = xs[o];
      const double a_next = o ? rows[(size_t)(o - 1) * Sp] : 0.0;
      const double co = cs[o + 1];
      const double ag = on ? a_o * (g / co) : 0.0;
      const uint32_t row0 = A.b_off[x], deg = A.b_off[x + 1] - row0;
      const URec* __restrict__ p = A.b_rec + (size_t)row0 * S + ln;
      double v = 0.0;
      for (uint32_t it0 = 0; it0 < deg; it0 += 4) {
        URec r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = p[(size_t)min(it0 + j, deg - 1) * S];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double term = (it0 + j < deg) ? bc[r[j].other_slot2 & 0x3ffu] * r[j].w : 0.0;
          v += term;
          const double pp = ag * term;
          if (pp > 0.0) {
            const uint32_t t0 = r[j].slot01 & 0xffffu, t1 = r[j].slot01 >> 16, t2 = r[j].other_slot2 >> 16;
            if (t0 != UNROLLED_NO_SLOT) atomicAdd(acc + t0, pp);
            if (t1 != UNROLLED_NO_SLOT) atomicAdd(acc + t1, pp);
            if (t2 != UNROLLED_NO_SLOT) atomicAdd(acc + t2, pp);
          }
        }
      }
      b = on ? v / co : 0.0;
      __syncthreads();  // everyone has read bc
      b = eps_backward(bc, b, a_o);
      if (on) bc[tid] = b;
      __syncthreads();
      a_o = a_next;
    }
    __syncthreads();
  }
  __syncthreads();
  double* out = A.partial + (size_t)blockIdx.x * A.n_slots;
  for (uint32_t k = tid; k < A.n_slots; k += blockDim.x) out[k] = acc[k];
}

size_t unrolled_wide_lds_bytes(const UnrolledArgs& A) {
  return ((size_t)A.n_slots + 2 * (size_t)A.S + 16 + A.max_len + 2) * sizeof(double);
}
size_t unrolled_wide_scratch_doubles(uint32_t n_wg, uint32_t S, uint32_t max_len) {
  return (size_t)n_wg * (max_len + 1) * (((size_t)S + 63) / 64 * 64);
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
                                             const double* __restrict__ wprior, const uint32_t* __restrict__ group,
                                             const uint32_t* __restrict__ slot_of, uint32_t n) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t sl = slot_of[p];
  out[p] = (group[p] != 0u && sl != 0xffffffffu) ? counts[sl] + floor_count * uses[p] + (wprior ? wprior[p] : 0.0) : 0.0;
}
// wprior: carmel -U on a cascade -- per parameter, the initial weights of the composed arcs that use it (may be null)
hipError_t launch_unrolled_param_counts(double* out, const double* counts, const double* uses, double floor_count,
                                        const double* wprior, const uint32_t* group, const uint32_t* slot_of, uint32_t n,
                                        hipStream_t s) {
  hipLaunchKernelGGL(unrolled_param_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, counts, uses, floor_count, wprior,
                     group, slot_of, n);
  return hipGetLastError();
}

// lanes per pair / pairs per wavefront
static inline uint32_t unrolled_wd(uint32_t S) { return S <= 16 ? 16u : S <= 32 ? 32u : 64u; }
size_t unrolled_lds_bytes(const UnrolledArgs& A, uint32_t n_waves) {
  return (A.n_slots + (size_t)n_waves * (64 / unrolled_wd(A.S)) * (A.max_len + 2)) * sizeof(double);
}
// waves per workgroup (0: the accumulators alone do not fit)
uint32_t unrolled_waves(uint32_t n_slots, uint32_t max_len, uint32_t S) {
  const uint32_t P = 64 / unrolled_wd(S);
  for (uint32_t w = U_MAX_WAVES; w >= 1; --w)
    if ((n_slots + (size_t)w * P * (max_len + 2)) * sizeof(double) <= 64 * 1024) return w;
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

// many rows (the dense sweep leaves one per group of 64 strings): two levels, both in a fixed order (deterministic).
// Level 1: the rows are cut into gridDim.y contiguous chunks; a workgroup sums the rows of its chunk for 64 slots (four
// wavefronts take every fourth row) and leaves the sum in the chunk's own first row.  Level 2: the same kernel over those
// first rows (stride = rows per chunk), into counts.
__global__ __launch_bounds__(256) void unrolled_reduce_rows_kernel(double* __restrict__ partial, uint32_t n_rows, uint32_t chunk,
                                                                   uint32_t stride, uint32_t n_slots, double* __restrict__ out) {
  __shared__ double sh[4][64];
  const uint32_t k = blockIdx.x * 64 + (threadIdx.x & 63), share = threadIdx.x >> 6;
  const uint32_t r0 = blockIdx.y * chunk, r1 = min(n_rows, r0 + chunk);  // in units of `stride` rows
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (k < n_slots) {
    uint32_t w = r0 + share;
    for (; w + 12 < r1; w += 16)
      for (int j = 0; j < 4; ++j) v[j] += partial[(size_t)(w + 4 * j) * stride * n_slots + k];
    for (; w < r1; w += 4) v[0] += partial[(size_t)w * stride * n_slots + k];
  }
  sh[share][threadIdx.x & 63] = (v[0] + v[1]) + (v[2] + v[3]);
  __syncthreads();
  if (share == 0 && k < n_slots) {
    const double sum = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    if (out)
      out[k] = sum;
    else
      partial[(size_t)r0 * stride * n_slots + k] = sum;
  }
}
// (partial is used as scratch when it has many rows)
hipError_t launch_unrolled_reduce(double* partial, uint32_t n_wg, uint32_t n_slots, double* counts, hipStream_t s) {
  if (n_wg >= 256) {
    const uint32_t chunk = 64, n_chunks = (n_wg + chunk - 1) / chunk;
    hipLaunchKernelGGL(unrolled_reduce_rows_kernel, dim3((n_slots + 63) / 64, n_chunks), dim3(256), 0, s, partial, n_wg, chunk, 1u,
                       n_slots, (double*)nullptr);
    hipLaunchKernelGGL(unrolled_reduce_rows_kernel, dim3((n_slots + 63) / 64, 1), dim3(256), 0, s, partial, n_chunks, n_chunks, chunk,
                       n_slots, counts);
  } else
    hipLaunchKernelGGL(unrolled_reduce_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, partial, n_wg, n_slots, counts);
  return hipGetLastError();
}

hipError_t launch_unrolled_sweep(const UnrolledArgs& A, uint32_t n_wg, double* counts, hipStream_t s) {
  if (A.S > UNROLLED_MAX_STATES) {
    const size_t lds = unrolled_wide_lds_bytes(A);
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)unrolled_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(unrolled_wide_kernel, dim3(n_wg), dim3((A.S + 63) / 64 * 64), lds, s, A);
    hipLaunchKernelGGL(unrolled_reduce_kernel, dim3((A.n_slots + 255) / 256), dim3(256), 0, s, A.partial, n_wg, A.n_slots, counts);
    return hipGetLastError();
  }
  const uint32_t n_waves = unrolled_waves(A.n_slots, A.max_len, A.S);
  if (!n_waves) return hipErrorInvalidValue;
  const size_t lds = unrolled_lds_bytes(A, n_waves);
  switch (unrolled_wd(A.S)) {
    case 16: hipLaunchKernelGGL(unrolled_sweep_kernel<16>, dim3(n_wg), dim3(64 * n_waves), lds, s, A); break;
    case 32: hipLaunchKernelGGL(unrolled_sweep_kernel<32>, dim3(n_wg), dim3(64 * n_waves), lds, s, A); break;
    default: hipLaunchKernelGGL(unrolled_sweep_kernel<64>, dim3(n_wg), dim3(64 * n_waves), lds, s, A); break;
  }
  hipLaunchKernelGGL(unrolled_reduce_kernel, dim3((A.n_slots + 255) / 256), dim3(256), 0, s, A.partial, n_wg, A.n_slots, counts);
  return hipGetLastError();
}

}  // namespace carmel_hip
