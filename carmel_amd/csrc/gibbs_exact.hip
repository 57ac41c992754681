// gibbs_exact.hip — `carmel --crp`: the reference's chain of the blocked Gibbs sampler over derivation lattices, blocks
// strictly one after another, as one persistent wavefront per sweep.
//
// Replaces (for the default chain: no annealing, no --include-self, no --expectation): /root/reference/carmel/src/gibbs.cc:306-371
// (resample_block: proposal weights, random_path), carmel/src/derivations.h:318-375 (random_path: backward sweep, walk),
// graehl/shared/random.ipp:111-127 (choose_p), graehl/shared/gibbs.hpp:835-877 (iteration), :769-792 + delta_sum.hpp:74-84
// (addc), :712-742 (cache-model probability).  gibbs.hip's gibbs_sweep_exact_kernel is the same chain with a workgroup of four
// wavefronts, a barrier per level and per bookkeeping step, and four dependent global gathers per lattice arc
// (arc -> chain -> parameter -> norm group -> counts): 64 us per block on the tutorial's tagging cascade, five times what one
// CPU core takes.  Here, as in forest_exact.hip:
//   * everything static about a lattice arc -- ends, parameters, their norm groups -- is one 24-byte record: the proposal
//     weights are ONE round of count gathers, in the linear domain (a product of one or two count ratios: no logarithm);
//   * the backward sweep runs level by level with the lanes over the level's ARCS: term = weight x beta[destination], summed
//     per source state with LDS adds (plain doubles; a start value below 1e-250 repeats the sweep with mantissa x 2^exponent
//     per state, each state's terms aligned to its largest exponent first);
//   * the walk reads a state's shares in the reference's list order and subtracts them from u x total one by one
//     (random.ipp:111-127), the uniforms keyed by the step as in gibbs.hip: the oracle's draws;
//   * counts: one round of device atomics per block -- the new sample in, the next block's previous sample out --, the
//     cache-model probability from the values the cache counts' atomics return, the time-weighted sums folded once per sweep
//     for every parameter (forest_exact.hpp: launch_forest_fold), workgroup-scope coherence (one wavefront, one L2).
#include "gibbs_exact.hpp"
#include "rng.hpp"

namespace carmel_hip {

#define GX_NONE 0xffffffffu
#define GX_SQ 2  // entries of a block's previous sample per lane that travel in registers
#define GX_WAVE_SYNC()                                   \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_wave_barrier();                     \
  } while (0)

__device__ __forceinline__ double gx_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void gx_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
__device__ __forceinline__ double gx_add(double* p, double v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void gx_lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void gx_lds_max(int* p, int v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ double gx_rl(double v, uint32_t lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (int)lane), __builtin_amdgcn_readlane(__double2loint(v), (int)lane));
}
// the value `ctrl` lanes to the left within a row of 16 lanes (0 beyond the row's start): a step of a scan without memory
template <int ctrl>
__device__ __forceinline__ double gx_row_shr(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
struct GxProd {  // a running product as mantissa x 2^exponent (one logarithm per lane and sweep)
  double m;
  long long e;
  __device__ __forceinline__ void mul(double x) {
    int t;
    m = frexp(m * x, &t);
    e += t;
  }
  __device__ __forceinline__ double ln() const { return log(m) + (double)e * 0.69314718055994530942; }
};

struct GxOld {  // a block's previous sample, on its way
  uint32_t id[GX_SQ], nr[GX_SQ], len;
};
__device__ __forceinline__ void gx_request_old(const GxArgs& A, const GxBlock& B, uint32_t b, uint32_t lane, GxOld& O) {
  O.len = A.sample_len[b];
#pragma unroll
  for (int q = 0; q < GX_SQ; ++q) {  // (read past the sample: within its capacity or the buffer's padding)
    O.id[q] = A.sample_ids[B.sample_off + lane + q * 64];
    O.nr[q] = A.sample_nrm[B.sample_off + lane + q * 64];
  }
}
__device__ __forceinline__ void gx_take_out(const GxArgs& A, const GxBlock& B, uint32_t lane, const GxOld& O) {
#pragma unroll
  for (int q = 0; q < GX_SQ; ++q)
    if (lane + q * 64 < O.len && O.nr[q] != GX_NONE) {
      gx_add(A.p_x + O.id[q], -B.wt);
      gx_add(A.normsum + O.nr[q], -B.wt);
    }
  for (uint32_t k = lane + GX_SQ * 64; k < O.len; k += 64) {
    const uint32_t n = A.sample_nrm[B.sample_off + k];
    if (n == GX_NONE) continue;
    gx_add(A.p_x + A.sample_ids[B.sample_off + k], -B.wt);
    gx_add(A.normsum + n, -B.wt);
  }
}

__global__ __launch_bounds__(64) void gibbs_exact_wave_kernel(GxArgs A) {
  __shared__ double gw[GX_ARCS];   // proposal weight the walk samples from
  __shared__ double pc[GX_ARCS];   // ... from the counts (what the proposal probability of the sample is made of)
  __shared__ double sh[GX_ARCS];   // the arc's share of its source state's total
  __shared__ uint32_t ds[GX_ARCS], par0[GX_ARCS], par1[GX_ARCS];
  __shared__ double bv[GX_STATES], bsum[GX_STATES];
  __shared__ int be[GX_STATES], emx[GX_STATES];
  __shared__ uint32_t ooff[GX_STATES + 1], lvl[GX_LEVELS + 1];
  __shared__ uint32_t ids[GX_SAMPLE], idn[GX_SAMPLE];
  const uint32_t lane = threadIdx.x;
  const uint32_t nb = A.n_blocks;
  GxProd cheap{1.0, 0}, cnum{1.0, 0}, cden{1.0, 0}, after{1.0, 0};
  unsigned long long clk[6] = {0, 0, 0, 0, 0, 0};
  GxBlock B = A.blocks[0], Bn = A.blocks[min(1u, nb - 1)];
  GxOld O;
  gx_request_old(A, B, 0, lane, O);
  if (!A.want_after) gx_take_out(A, B, lane, O);
  uint32_t n_prev = 0;
  double wt_prev = 1.0;
  (void)wt_prev;
  for (uint32_t b = 0; b < nb; ++b) {
    unsigned long long t0 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    gx_order();  // the counts are as the chain has them
    if (A.want_after) {
      // the previous block's sample scored with itself counted (the "overestimate" of gibbs.hpp:866), then this block's
      // previous sample leaves the counts
      for (uint32_t k = lane; k < n_prev; k += 64) {
        const uint32_t p = ids[k], n = idn[k];
        after.mul(n == GX_NONE ? A.p_prior[p] : gx_ld(A.p_x + p) / gx_ld(A.normsum + n));
      }
      gx_take_out(A, B, lane, O);
      gx_order();
    }
    // ---- the lattice: arc records, proposal weights (gibbs.cc:348-359, gibbs.hpp:153-157), offsets ----
    const uint4* __restrict__ rec = A.arc_rec + B.out_base;
    const uint2* __restrict__ nrm = A.arc_nrm + B.out_base;
    for (uint32_t a = lane; a < B.n_arcs; a += 64) {
      const uint4 r = rec[a];
      const uint2 n = nrm[a];
      double w = 1.0;  // (a composed arc may stand for no parameter at all)
      if (r.z != GX_NONE) w = n.x == GX_NONE ? A.p_prior[r.z] : gx_ld(A.p_x + r.z) / gx_ld(A.normsum + n.x);
      if (r.w != GX_NONE) w *= n.y == GX_NONE ? A.p_prior[r.w] : gx_ld(A.p_x + r.w) / gx_ld(A.normsum + n.y);
      pc[a] = w;
      gw[a] = A.init_logw ? exp(A.init_logw[r.y]) : w;
      ds[a] = r.x;
      par0[a] = r.z;
      par1[a] = r.w;
    }
    for (uint32_t s = lane; s <= B.n_states; s += 64) ooff[s] = A.out_off[B.off_base + s];
    for (uint32_t l = lane; l <= B.n_levels; l += 64) lvl[l] = A.level_off[B.level_base + l];
    // the next block's previous sample sets out now
    GxOld On;
    gx_request_old(A, Bn, min(b + 1, nb - 1), lane, On);
    const GxBlock Bnn = A.blocks[min(b + 2, nb - 1)];
    const double U = gibbs_uniform(A.seed, A.iter, b, lane);  // the walk's first 64 uniforms, one per lane
    unsigned long long t1 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- backward sweep (derivations.h:345-360): beta[s] = sum over out-arcs of weight x beta[destination].  A state's value
    // is read only by the levels before it, so the terms are added straight into it; the level's arc range comes from the
    // level table one level ahead of its use ----
    bool ext = false;
    for (;;) {
      for (uint32_t s = lane; s < B.n_states; s += 64) {
        bv[s] = 0.0;
        bsum[s] = 0.0;
        be[s] = 0;
        emx[s] = -(1 << 28);
      }
      GX_WAVE_SYNC();
      if (lane == 0) {
        bv[B.fin] = ext ? 0.5 : 1.0;
        bsum[B.fin] = 1.0;
        be[B.fin] = ext ? 1 : 0;
      }
      GX_WAVE_SYNC();
      uint32_t a_hi = ooff[lvl[B.n_levels]];
      uint32_t s_next = B.n_levels ? lvl[B.n_levels - 1] : 0u;
      for (uint32_t l = B.n_levels; l-- > 0;) {
        const uint32_t s_lo = s_next, s_hi = lvl[l + 1];
        const uint32_t a_lo = ooff[s_lo];
        s_next = l ? lvl[l - 1] : 0u;
        if (a_lo != a_hi) {
          if (!ext) {
            for (uint32_t a = a_lo + lane; a < a_hi; a += 64) {
              const uint32_t d = ds[a];
              const double term = gw[a] * bv[d & 0xffffu];
              sh[a] = term;
              if (term != 0.0) gx_lds_add(&bv[d >> 16], term);
            }
            GX_WAVE_SYNC();
          } else {
            // every state's terms in units of the largest exponent among them
            for (uint32_t a = a_lo + lane; a < a_hi; a += 64) {
              const uint32_t d = ds[a];
              if (gw[a] * bv[d & 0xffffu] != 0.0) gx_lds_max(&emx[d >> 16], be[d & 0xffffu]);
            }
            GX_WAVE_SYNC();
            for (uint32_t a = a_lo + lane; a < a_hi; a += 64) {
              const uint32_t d = ds[a];
              const double term = ldexp(gw[a] * bv[d & 0xffffu], max(be[d & 0xffffu] - emx[d >> 16], -1100));
              sh[a] = term;
              if (term != 0.0) gx_lds_add(&bsum[d >> 16], term);
            }
            GX_WAVE_SYNC();
            for (uint32_t s = s_lo + lane; s < s_hi; s += 64)
              if (ooff[s + 1] > ooff[s]) {
                int t;
                bv[s] = frexp(bsum[s], &t);
                be[s] = bsum[s] != 0.0 ? emx[s] + t : 0;
              }
            GX_WAVE_SYNC();
          }
        }
        a_hi = a_lo;
      }
      const double root = bv[B.start];
      if (ext || (root >= 1e-250 && root <= 1e250)) break;
      ext = true;  // plain doubles ran out: once more with exponents
    }
    // a state's total in the units of its arcs' shares
    double* const tot = ext ? bsum : bv;
    unsigned long long t2 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- walk start -> goal (derivations.h:361-374; random.ipp:111-127): at every state the shares of its out-arcs, in the
    // reference's list order (newest first: the reverse of the stored order), are subtracted from u x total until it drops
    // below zero; the chosen arc's parameters are recorded in chain order ----
    uint32_t n_ids = 0, step = 0;
    {
      uint32_t s = B.start;
      while (s != B.fin) {
        const uint32_t a0 = ooff[s], a1 = ooff[s + 1], deg = a1 - a0;
        if (!deg) break;  // (a dead end cannot be reached: pruned lattices)
        const double u = step < 64 ? gx_rl(U, step) : gibbs_uniform(A.seed, A.iter, b, step);
        ++step;
        double choice = u * tot[s];
        // lane i holds the i-th arc of the state's list (its share and what the walk records of it): the chosen one's are
        // read across lanes, not fetched again
        uint32_t p0 = GX_NONE, p1 = GX_NONE, to = B.fin;
        double pcv = 1.0;
        if (deg <= 16) {
          // the usual case: the shares' running sums by a scan over the first row of lanes (DPP row shifts, no memory);
          // the reference stops at the first arc whose running sum exceeds u x total
          const uint32_t at = lane < deg ? a1 - 1 - lane : a0;
          const double share = lane < deg ? sh[at] : 0.0;
          const double pcl = pc[at];
          const uint32_t dl = ds[at], q0 = par0[at], q1 = par1[at];
          double run = share;
          run += gx_row_shr<0x111>(run);
          run += gx_row_shr<0x112>(run);
          run += gx_row_shr<0x114>(run);
          run += gx_row_shr<0x118>(run);
          const unsigned long long passed = __ballot(lane < deg && run > choice);
          const uint32_t j = passed ? (uint32_t)__builtin_ctzll(passed) : deg - 1;
          p0 = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)j);
          p1 = (uint32_t)__builtin_amdgcn_readlane((int)q1, (int)j);
          to = (uint32_t)__builtin_amdgcn_readlane((int)dl, (int)j) & 0xffffu;
          pcv = gx_rl(pcl, j);
        } else {
        bool done = false;
        for (uint32_t base = 0; base < deg && !done; base += 64) {
          const uint32_t mine = base + lane, at = mine < deg ? a1 - 1 - mine : a0;
          const double share = mine < deg ? sh[at] : 0.0;
          const double pcl = pc[at];
          const uint32_t dl = ds[at], q0 = par0[at], q1 = par1[at];
          const uint32_t cnt = min(64u, deg - base);
          uint32_t j = 0;
          for (; j < cnt; ++j) {
            choice -= gx_rl(share, j);
            if (choice < 0) {
              done = true;
              break;
            }
          }
          if (j == cnt) j = cnt - 1;  // (never below zero: the last arc of the list, as the reference's loop leaves it)
          p0 = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)j);
          p1 = (uint32_t)__builtin_amdgcn_readlane((int)q1, (int)j);
          to = (uint32_t)__builtin_amdgcn_readlane((int)dl, (int)j) & 0xffffu;
          pcv = gx_rl(pcl, j);
        }
        }
        if (lane == 0) {
          if (p0 != GX_NONE && n_ids < GX_SAMPLE) ids[n_ids] = p0;
          if (p1 != GX_NONE && n_ids + 1 < GX_SAMPLE) ids[n_ids + 1] = p1;
        }
        n_ids += p0 == GX_NONE ? 0u : (p1 != GX_NONE ? 2u : 1u);
        cheap.mul(pcv);
        s = to;
      }
      n_ids = min(n_ids, (uint32_t)GX_SAMPLE);
    }
    GX_WAVE_SYNC();
    unsigned long long t3 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- the new sample into the counts, the next block's previous sample out (gibbs.hpp:851-871, 712-742, 769-792) ----
    for (uint32_t k = lane; k < n_ids; k += 64) {
      const uint32_t p = ids[k], n = A.p_norm[p];
      idn[k] = n;
      A.sample_ids[B.sample_off + k] = p;
      A.sample_nrm[B.sample_off + k] = n;
      if (n != GX_NONE) {
        gx_add(A.p_x + p, B.wt);
        gx_add(A.normsum + n, B.wt);
        cnum.mul(gx_add(A.ccount + p, 1.0));
        cden.mul(gx_add(A.csum + n, 1.0));
      } else
        cnum.mul(A.p_prior[p]);
    }
    if (lane == 0) A.sample_len[b] = n_ids;
    if (!A.want_after && b + 1 < nb) gx_take_out(A, Bn, lane, On);
    n_prev = n_ids;
    B = Bn;
    Bn = Bnn;
    O = On;
    GX_WAVE_SYNC();
    if (A.phase_clk) {
      const unsigned long long t4 = __builtin_readcyclecounter();
      clk[0] += t1 - t0;
      clk[1] += t2 - t1;
      clk[2] += t3 - t2;
      clk[3] += t4 - t3;
      clk[4] += 1;
    }
  }
  gx_order();
  if (A.want_after)
    for (uint32_t k = lane; k < n_prev; k += 64) {
      const uint32_t p = ids[k], n = idn[k];
      after.mul(n == GX_NONE ? A.p_prior[p] : gx_ld(A.p_x + p) / gx_ld(A.normsum + n));
    }
  // cheap is accumulated by every lane alike (the walk is wavefront-uniform): lane 0's is the sweep's
  double cache_ln = cnum.ln() - cden.ln(), after_ln = after.ln();
  for (int o = 32; o > 0; o >>= 1) {
    cache_ln += __shfl_down(cache_ln, o, 64);
    after_ln += __shfl_down(after_ln, o, 64);
  }
  if (lane == 0) {
    A.iter_out[0] = cache_ln;
    A.iter_out[1] = cheap.ln();
    A.iter_out[2] = after_ln;
    if (A.phase_clk)
      for (int k = 0; k < 5; ++k) A.phase_clk[k] += clk[k];
  }
}

hipError_t launch_gibbs_exact_wave(const GxArgs& A, hipStream_t s) {
  hipLaunchKernelGGL(gibbs_exact_wave_kernel, dim3(1), dim3(64), 0, s, A);
  return hipGetLastError();
}

}  // namespace carmel_hip
