// dense.hip: see dense.hpp.  Restates the forward-backward of derivations.h:401-470 on the implicit lattice of a one-tape
// cascade (unrolled.hpp) in the scaled linear domain, as dense S x S products.
#include "dense.hpp"

#include <cmath>
#include <cstdlib>

namespace carmel_hip {

uint32_t dense_padded_states(uint32_t S) {
  // 16 and 32 run on the matrix cores (dense_mfma_kernel); no_mfma (a source switch, for A/B) pads to the next of the vector sizes
  static const bool no_mfma = false;
  static const uint32_t vec[] = {4, 8, 12, 16, 20, 24, 28, 30, 32}, mat[] = {4, 8, 12, 16, 32};
  if (no_mfma) {
    for (uint32_t p : vec)
      if (S <= p) return p;
  } else
    for (uint32_t p : mat)
      if (S <= p) return p;
  return 0;
}

// one thread per table entry
__global__ void dense_tables_kernel(double* __restrict__ A, double* __restrict__ AT, double* __restrict__ B, uint32_t SP, uint32_t V,
                                    const uint32_t* __restrict__ a_off, const uint32_t* __restrict__ a_par,
                                    const uint32_t* __restrict__ b_off, const uint32_t* __restrict__ b_par,
                                    const uint8_t* __restrict__ a_has, const uint8_t* __restrict__ b_has,
                                    const double* __restrict__ param_logw) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t na = SP * SP, nb = V * SP;
  if (e < na) {
    double lw = 0.0;
    for (uint32_t k = a_off[e]; k < a_off[e + 1]; ++k) lw += param_logw[a_par[k]];
    const double v = a_has[e] ? exp(lw) : 0.0;
    A[e] = v;
    AT[(e % SP) * SP + e / SP] = v;
  } else if (e < na + nb) {
    const uint32_t f = e - na;
    double lw = 0.0;
    for (uint32_t k = b_off[f]; k < b_off[f + 1]; ++k) lw += param_logw[b_par[k]];
    B[f] = b_has[f] ? exp(lw) : 0.0;
  }
}
hipError_t launch_dense_tables(double* A, double* AT, double* B, uint32_t SP, uint32_t V, const uint32_t* a_off,
                               const uint32_t* a_par, const uint32_t* b_off, const uint32_t* b_par, const uint8_t* a_has,
                               const uint8_t* b_has, const double* param_logw, hipStream_t s) {
  const uint32_t n = SP * SP + V * SP;
  hipLaunchKernelGGL(dense_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, s, A, AT, B, SP, V, a_off, a_par, b_off, b_par, a_has,
                     b_has, param_logw);
  return hipGetLastError();
}

#define DENSE_CH 8
// *e*:*e* arcs inside a position (unrolled.hpp): alpha[dst] += alpha[src] * w in topological order; the state indices are
// uniform but not compile-time, so a register of the column is picked by a chain of selects
template <int SP>
__device__ __forceinline__ void dense_eps_forward(double (&a)[SP], const DenseArgs& D) {
  for (uint32_t k = 0; k < D.n_eps; ++k) {
    const uint32_t src = D.e_src[k], dst = D.e_dst[k];
    const double w = D.We[k];
    double as = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j) as = (uint32_t)j == src ? a[j] : as;
    const double add = as * w;
#pragma unroll
    for (int j = 0; j < SP; ++j) a[j] += (uint32_t)j == dst ? add : 0.0;
  }
}
template <int SP>
__device__ __forceinline__ void dense_eps_backward(double (&b)[SP], const DenseArgs& D) {
  for (uint32_t k = D.n_eps; k-- > 0;) {
    const uint32_t src = D.e_src[k], dst = D.e_dst[k];
    const double w = D.We[k];
    double bd = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j) bd = (uint32_t)j == dst ? b[j] : bd;
    const double add = bd * w;
#pragma unroll
    for (int j = 0; j < SP; ++j) b[j] += (uint32_t)j == src ? add : 0.0;
  }
}
// One wavefront per group of 64 strings (sorted by length), one string per lane.
template <int SP, bool LDSA>
__global__ __launch_bounds__(64) void dense_sweep_kernel(DenseArgs D) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // LDS: [A | AT] first (compile-time addresses: the reads of the inner loops need no address arithmetic), then the
  // accumulators, B and its slots
  const uint32_t nsl = (D.n_slots + 1u) & ~1u;
  double* acc = lds + (LDSA ? 2 * SP * SP : 0);  // [n_slots]
  double* Bl = acc + nsl;                        // [V][SP]
  uint16_t* Sl = (uint16_t*)(Bl + D.V * SP);      // [V][SP]
  // LDSA: A and AT in LDS, read with one address for all lanes (a broadcast: ds_read_b128 hands two elements to the whole
  // wave); otherwise they are streamed through the scalar unit
  double* Al = lds;
  double* ATl = lds + SP * SP;
  const int lane = threadIdx.x;
  if (LDSA)
    for (uint32_t k = lane; k < (uint32_t)(SP * SP); k += 64) {
      Al[k] = D.A[k];
      ATl[k] = D.AT[k];
    }
  for (uint32_t k = lane; k < D.n_slots; k += 64) acc[k] = 0.0;
  for (uint32_t k = lane; k < D.V * SP; k += 64) {
    Bl[k] = D.B[k];
    Sl[k] = D.Bslot[k];
  }
  __syncthreads();
  const uint32_t g = blockIdx.x;
  const uint32_t T = D.len[g * 64 + lane];
  uint32_t Tmax = T;
  for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, o, 64));
  Tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmax);  // uniform loop bounds: the loops stay scalar control flow
  const uint16_t* __restrict__ sym = D.sym + D.sym_off[g] + lane;
  double* __restrict__ zb = D.zbuf + D.sym_off[g] + lane;
  double* __restrict__ vb = D.vbuf + D.vbuf_off[g] + lane;
  // A and AT are read through the CONSTANT address space: the compiler may then assume they do not change under the
  // kernel's own stores and, the indices being uniform, fetch them with scalar loads (one s_load_dwordx16 feeds eight
  // fused multiply-adds of all 64 lanes) instead of 64 identical vector loads per element
  typedef const __attribute__((address_space(4))) double* cptr;
  cptr A = (cptr)(uintptr_t)D.A;
  cptr AT = (cptr)(uintptr_t)D.AT;
  uint32_t off = 0;
  // ---------- forward ----------
  double a[SP];
#pragma unroll
  for (int j = 0; j < SP; ++j) a[j] = (uint32_t)j == D.start ? 1.0 : 0.0;
  dense_eps_forward<SP>(a, D);
  double lp = 0.0;
  for (uint32_t t = 0; t < Tmax; ++t) {
    const bool active = t < T;
    const uint32_t c = active ? sym[(size_t)t * 64] : 0u;
    double v[SP];
#pragma unroll
    for (int j = 0; j < SP; ++j) v[j] = 0.0;
    // The matrix is the same at every position, so the optimiser would hoist all SP * SP scalar loads out of the loop
    // over positions and spill them; `off` (always 0) is laundered through an empty asm per chunk of DENSE_CH
    // columns, which keeps each chunk's loads next to its multiply-adds: ~4 chunks of 16 SGPRs in flight.
#pragma unroll
    for (int i = 0; i < SP; ++i)
#pragma unroll
      for (int jc = 0; jc < SP; jc += DENSE_CH) {
        if (LDSA) {
#pragma unroll
          for (int j = 0; j < DENSE_CH && jc + j < SP; ++j) v[jc + j] = fma(a[i], Al[i * SP + jc + j], v[jc + j]);
        } else {
          asm volatile("" : "+s"(off));
          cptr Ar = A + off + i * SP + jc;
#pragma unroll
          for (int j = 0; j < DENSE_CH && jc + j < SP; ++j) v[jc + j] = fma(a[i], Ar[j], v[jc + j]);
        }
      }
    double z = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      vb[((size_t)t * SP + j) * 64] = v[j];
      v[j] *= Bl[c * SP + j];
      z += v[j];
    }
    if (active) {
      const double zi = 1.0 / z;
#pragma unroll
      for (int j = 0; j < SP; ++j) v[j] *= zi;
      dense_eps_forward<SP>(v, D);
#pragma unroll
      for (int j = 0; j < SP; ++j) a[j] = v[j];
      lp += log(z);
      zb[(size_t)t * 64] = z;
    }
  }
  double af = 0.0;
#pragma unroll
  for (int j = 0; j < SP; ++j) af = (uint32_t)j == D.fin ? a[j] : af;
  const double pw = D.weight[g * 64 + lane];
  if (T) D.pair_logprob[D.pair[g * 64 + lane]] = lp + log(af);
  // ---------- backward + posteriors ----------
  double b[SP];
#pragma unroll
  for (int j = 0; j < SP; ++j) b[j] = 0.0;
  for (uint32_t t = Tmax; t-- > 0;) {
    const bool active = t < T;
    if (t + 1 == T) {
#pragma unroll
      for (int j = 0; j < SP; ++j) b[j] = (uint32_t)j == D.fin ? 1.0 / af : 0.0;
    }
    const uint32_t c = active ? sym[(size_t)t * 64] : 0u;
    const double zi = active ? 1.0 / zb[(size_t)t * 64] : 0.0;
    if (active) dense_eps_backward<SP>(b, D);  // beta over the *e*:*e* arcs of position t + 1
    double w[SP];
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      w[j] = Bl[c * SP + j] * b[j] * zi;
      const double gam = vb[((size_t)t * SP + j) * 64] * w[j] * pw;
      const uint32_t sl = Sl[c * SP + j];
      if (active && sl != 0xffffu && gam != 0.0) atomicAdd(acc + sl, gam);
    }
    double nb[SP];
#pragma unroll
    for (int i = 0; i < SP; ++i) nb[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j)
#pragma unroll
      for (int ic = 0; ic < SP; ic += DENSE_CH) {
        if (LDSA) {
#pragma unroll
          for (int i = 0; i < DENSE_CH && ic + i < SP; ++i) nb[ic + i] = fma(ATl[j * SP + ic + i], w[j], nb[ic + i]);
        } else {
          asm volatile("" : "+s"(off));
          cptr Ar = AT + off + j * SP + ic;
#pragma unroll
          for (int i = 0; i < DENSE_CH && ic + i < SP; ++i) nb[ic + i] = fma(Ar[i], w[j], nb[ic + i]);
        }
      }
    if (active) {
#pragma unroll
      for (int i = 0; i < SP; ++i) b[i] = nb[i];
    }
  }
  __syncthreads();
  for (uint32_t k = lane; k < D.n_slots; k += 64) D.partial[(size_t)g * D.n_slots + k] = acc[k];
}


// ---------------- the same sweep on the matrix cores ----------------
// Per position the 64 strings of a wavefront need V^T = A^T . alpha^T, an (S x S) . (S x 64) product:
// v_mfma_f64_16x16x4_f64 tiles with the constant matrix as the A operand (lane l holds A^T[16 ib + (l & 15)][4 ks + (l >> 4)],
// resident in registers for the whole kernel) and the forward values as the B operand (lane l holds, for string
// 16 nb + (l & 15), state 4 ks + (l >> 4)).  The f64 result layout (row = (lane >> 4) + 4 reg, col = lane & 15) puts state
// 16 ib + (l >> 4) + 4 r of that same string in register r of lane l -- which IS the B-operand slot ks = 4 ib + r of the next
// position: no shuffle between positions, the channel factor and the scaling are applied in place.  Four lanes (l & 15 equal)
// hold the S states of one string, so a sum over states ends with two cross-lane adds (lanes 16 and 32 apart).
typedef double dense_d4 __attribute__((ext_vector_type(4)));
// A workgroup is a group of 64 strings; each of its four wavefronts takes 16 of them (one tile of strings): the registers
// of a wavefront are the two constant operands and ONE tile's values, so several workgroups share a CU and their
// wavefronts hide each other's load and LDS latencies.
// PHASE 0: both passes in one launch; 1: the forward pass alone (leaves alpha_T[goal] per string in D.afbuf); 2: the backward
// pass alone.  Split, each launch keeps ONE of the two constant operands in registers and none of the other pass's state:
// three to four wavefronts per SIMD instead of two.
template <int NB16, int PHASE>  // SP = 16 * NB16
__global__ __launch_bounds__(256) void dense_mfma_kernel(DenseArgs D) {
  constexpr int SP = 16 * NB16, KS = 4 * NB16;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const uint32_t nsl = (D.n_slots + 1u) & ~1u;
  double* acc = lds;
  double* Bl = acc + nsl;
  uint16_t* Sl = (uint16_t*)(Bl + D.V * SP);
  const int lane = threadIdx.x & 63, nb = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  if (PHASE != 1)
    for (uint32_t k = threadIdx.x; k < D.n_slots; k += 256) acc[k] = 0.0;
  for (uint32_t k = threadIdx.x; k < D.V * SP; k += 256) {
    Bl[k] = D.B[k];
    if (PHASE != 1) Sl[k] = D.Bslot[k];
  }
  __syncthreads();
  const uint32_t g = blockIdx.x;
  const uint32_t T = D.len[g * 64 + nb * 16 + c16];
  const double pw = D.weight[g * 64 + nb * 16 + c16];
  uint32_t Tmax = T;
  for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, o, 64));
  Tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmax);
  const uint16_t* __restrict__ sym = D.sym + D.sym_off[g] + nb * 16 + c16;
  double* __restrict__ zb = D.zbuf + D.sym_off[g] + nb * 16 + c16;
  double* __restrict__ vb = D.vbuf + D.vbuf_off[g] + (size_t)nb * KS * 64 + lane;  // row (t * 4 + nb) * KS + ks
  // the constant operands: forward A^T, backward A
  double aop[PHASE != 2 ? NB16 : 1][KS], aop2[PHASE != 1 ? NB16 : 1][KS];
#pragma unroll
  for (int ib = 0; ib < NB16; ++ib)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (PHASE != 2) aop[PHASE != 2 ? ib : 0][ks] = D.A[(size_t)(ks * 4 + q) * SP + ib * 16 + c16];
      if (PHASE != 1) aop2[PHASE != 1 ? ib : 0][ks] = D.A[(size_t)(ib * 16 + c16) * SP + ks * 4 + q];
    }
  // a sum over the four lanes that share a string
  auto sum4 = [](double x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
  };
  // the value of state s (uniform) of every string, in all four of its lanes: register s / 4 of lane group s % 4
  auto state_value = [&](const double (&a)[KS], uint32_t s) {
    double x = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) x = (uint32_t)ks == (s >> 2) ? a[ks] : x;
    return __shfl(x, c16 + 16 * (int)(s & 3u), 64);
  };
  auto eps_forward = [&](double (&a)[KS]) {
    for (uint32_t k = 0; k < D.n_eps; ++k) {
      const uint32_t src = D.e_src[k], dst = D.e_dst[k];
      const double add = state_value(a, src) * D.We[k];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[ks] += ((uint32_t)ks == (dst >> 2) && (uint32_t)q == (dst & 3u)) ? add : 0.0;
    }
  };
  auto eps_backward = [&](double (&b)[KS]) {
    for (uint32_t k = D.n_eps; k-- > 0;) {
      const uint32_t src = D.e_src[k], dst = D.e_dst[k];
      const double add = state_value(b, dst) * D.We[k];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) b[ks] += ((uint32_t)ks == (src >> 2) && (uint32_t)q == (src & 3u)) ? add : 0.0;
    }
  };
  double af = 0.0;
  uint32_t c_next = 0u;
  if (PHASE != 2) {
  // ---------- forward ----------
  // ln p = sum over positions of ln z: the product of the z is kept as mantissa x 2^exponent, one log at the end
  double al[KS], lp_m = 1.0;
  int lp_e = 0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) al[ks] = ((uint32_t)(ks * 4 + q) == D.start) ? 1.0 : 0.0;
  eps_forward(al);
  c_next = Tmax ? (0 < T ? sym[0] : 0u) : 0u;  // the symbol of a position is fetched one position ahead
  for (uint32_t t = 0; t < Tmax; ++t) {
    const bool active = t < T;
    const uint32_t c = c_next;
    c_next = (t + 1 < T) ? sym[(size_t)(t + 1) * 64] : 0u;
    // two accumulators per tile: the dependent chain of multiply-accumulates is KS / 2 deep instead of KS
    dense_d4 d[NB16], d1[NB16];
#pragma unroll
    for (int ib = 0; ib < NB16; ++ib) {
      d[ib] = dense_d4{0.0, 0.0, 0.0, 0.0};
      d1[ib] = dense_d4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2)
#pragma unroll
      for (int ib = 0; ib < NB16; ++ib) {
        d[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[PHASE != 2 ? ib : 0][ks], al[ks], d[ib], 0, 0, 0);
        d1[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[PHASE != 2 ? ib : 0][ks + 1], al[ks + 1], d1[ib], 0, 0, 0);
      }
#pragma unroll
    for (int ib = 0; ib < NB16; ++ib) d[ib] += d1[ib];
    double v[KS], z = 0.0;
#pragma unroll
    for (int ib = 0; ib < NB16; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ks = 4 * ib + r;  // state 4 ks + q
        if (!(D.debug & 2u)) vb[((size_t)t * 4 * KS + ks) * 64] = d[ib][r];
        v[ks] = d[ib][r] * Bl[c * SP + ks * 4 + q];
        z += v[ks];
      }
    z = sum4(z);
    if (active) {
      const double zi = 1.0 / z;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) v[ks] *= zi;
      eps_forward(v);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) al[ks] = v[ks];
      int e1, e2;
      const double m1 = frexp(z, &e1);
      lp_m = frexp(lp_m * m1, &e2);
      lp_e += e1 + e2;
      zb[(size_t)t * 64] = zi;  // the backward pass wants the reciprocal
    }
  }
  af = state_value(al, D.fin);
  if (T && q == 0) D.pair_logprob[D.pair[g * 64 + nb * 16 + c16]] = log(lp_m) + (double)lp_e * 0.693147180559945309417 + log(af);
  if (PHASE == 1 && q == 0) D.afbuf[g * 64 + nb * 16 + c16] = af;
  } else
    af = D.afbuf[g * 64 + nb * 16 + c16];
  if (PHASE == 1) return;
  if (D.debug & 4u) Tmax = 0;
  // ---------- backward + posteriors ----------
  double be[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) be[ks] = 0.0;
  // the parked forward values, the symbol and the scale of a position are fetched one position ahead
  double vq[KS], z_next = 0.0;  // (zbuf holds 1 / z)
  c_next = 0u;
  if (Tmax) {
    const uint32_t t = Tmax - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) vq[ks] = (D.debug & 2u) ? 1.0 : vb[((size_t)t * 4 * KS + ks) * 64];
    if (t < T) {
      c_next = sym[(size_t)t * 64];
      z_next = zb[(size_t)t * 64];
    }
  }
  for (uint32_t t = Tmax; t-- > 0;) {
    const bool active = t < T;
    if (t + 1 == T) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) be[ks] = ((uint32_t)(ks * 4 + q) == D.fin) ? 1.0 / af : 0.0;
    }
    const uint32_t c = c_next;
    const double zi = active ? z_next : 0.0;
    double vt[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) vt[ks] = vq[ks];
    if (t > 0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) vq[ks] = (D.debug & 2u) ? 1.0 : vb[((size_t)(t - 1) * 4 * KS + ks) * 64];
      c_next = (t - 1 < T) ? sym[(size_t)(t - 1) * 64] : 0u;
      z_next = (t - 1 < T) ? zb[(size_t)(t - 1) * 64] : 0.0;
    }
    if (active) eps_backward(be);
    double w[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      w[ks] = Bl[c * SP + ks * 4 + q] * be[ks] * zi;
      const double gam = vt[ks] * w[ks] * pw;
      const uint32_t sl = Sl[c * SP + ks * 4 + q];
      if (active && sl != 0xffffu && gam != 0.0 && !(D.debug & 1u)) atomicAdd(acc + sl, gam);
    }
    dense_d4 d[NB16], d1[NB16];
#pragma unroll
    for (int ib = 0; ib < NB16; ++ib) {
      d[ib] = dense_d4{0.0, 0.0, 0.0, 0.0};
      d1[ib] = dense_d4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2)
#pragma unroll
      for (int ib = 0; ib < NB16; ++ib) {
        d[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop2[PHASE != 1 ? ib : 0][ks], w[ks], d[ib], 0, 0, 0);
        d1[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop2[PHASE != 1 ? ib : 0][ks + 1], w[ks + 1], d1[ib], 0, 0, 0);
      }
#pragma unroll
    for (int ib = 0; ib < NB16; ++ib) d[ib] += d1[ib];
    if (active) {
#pragma unroll
      for (int ib = 0; ib < NB16; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) be[4 * ib + r] = d[ib][r];
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < D.n_slots; k += 256) D.partial[(size_t)g * D.n_slots + k] = acc[k];
}

hipError_t launch_dense_sweep(const DenseArgs& D, uint32_t n_groups, hipStream_t s) {
  static const bool no_mfma = false;  // A/B (source switch): the vector variants
  if (!no_mfma && (D.SP == 16 || D.SP == 32)) {
    const size_t l2 = (((size_t)D.n_slots + 1) & ~(size_t)1) * 8 + (size_t)D.V * D.SP * 8 + (((size_t)D.V * D.SP + 3) / 4) * 8 + 16;
    static const bool split = true;  // (false: one launch for both passes -- measured slower)
    if (D.SP == 16) {
      if (split && D.afbuf) {
        hipLaunchKernelGGL((dense_mfma_kernel<1, 1>), dim3(n_groups), dim3(256), l2, s, D);
        hipLaunchKernelGGL((dense_mfma_kernel<1, 2>), dim3(n_groups), dim3(256), l2, s, D);
      } else
        hipLaunchKernelGGL((dense_mfma_kernel<1, 0>), dim3(n_groups), dim3(256), l2, s, D);
    } else {
      if (split && D.afbuf) {
        hipLaunchKernelGGL((dense_mfma_kernel<2, 1>), dim3(n_groups), dim3(256), l2, s, D);
        hipLaunchKernelGGL((dense_mfma_kernel<2, 2>), dim3(n_groups), dim3(256), l2, s, D);
      } else
        hipLaunchKernelGGL((dense_mfma_kernel<2, 0>), dim3(n_groups), dim3(256), l2, s, D);
    }
    return hipGetLastError();
  }
  static const bool smem = true;  // (false: the LDS-broadcast variant, measured slower)
  const size_t lds = (((size_t)D.n_slots + 1) & ~(size_t)1) * 8 + (size_t)D.V * D.SP * 8 + (((size_t)D.V * D.SP + 3) / 4) * 8 +
                     (smem ? 0 : (size_t)2 * D.SP * D.SP * 8) + 16;
#define DENSE_CASE(P)                                                                                                   \
  case P:                                                                                                               \
    if (smem) {                                                                                                         \
      if (lds > 64 * 1024)                                                                                              \
        (void)hipFuncSetAttribute((const void*)dense_sweep_kernel<P, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((dense_sweep_kernel<P, false>), dim3(n_groups), dim3(64), lds, s, D);                          \
    } else {                                                                                                            \
      if (lds > 64 * 1024)                                                                                              \
        (void)hipFuncSetAttribute((const void*)dense_sweep_kernel<P, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((dense_sweep_kernel<P, true>), dim3(n_groups), dim3(64), lds, s, D);                           \
    }                                                                                                                   \
    break;
  switch (D.SP) {
    DENSE_CASE(4)
    DENSE_CASE(8)
    DENSE_CASE(12)
    DENSE_CASE(16)
    DENSE_CASE(20)
    DENSE_CASE(24)
    DENSE_CASE(28)
    DENSE_CASE(30)
    DENSE_CASE(32)
    default: return hipErrorInvalidValue;
  }
#undef DENSE_CASE
  return hipGetLastError();
}

}  // namespace carmel_hip
