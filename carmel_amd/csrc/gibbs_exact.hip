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
#include <algorithm>
#include "gibbs_exact.hpp"
#include "rng.hpp"

namespace carmel_hip {

#define GX_NONE 0xffffffffu
#define GX_SQ 2  // entries of a block's previous sample per lane that travel in registers
#define GX_WAVE_SYNC()                                   \
  do {                                                   \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): as the builtin, so that the compiler's own wait counts know of it */ \
    asm volatile("" ::: "memory");                       \
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

// the parallel sweep's own-sample tables: GX_OWN slots each (open addressing; a sample of at most GX_SQ * 64 entries)
#define GX_OWN 256
__device__ __forceinline__ void gx_own_put(uint32_t* keys, uint32_t* cnts, uint32_t key) {
  for (uint32_t h = (key * 2654435761u) >> 8;; ++h) {
    const uint32_t at = h & (GX_OWN - 1);
    const uint32_t old = atomicCAS(keys + at, GX_NONE, key);
    if (old == GX_NONE || old == key) {
      atomicAdd(cnts + at, 1u);
      return;
    }
  }
}
__device__ __forceinline__ uint32_t gx_own_get(const uint32_t* keys, const uint32_t* cnts, uint32_t key) {
  for (uint32_t h = (key * 2654435761u) >> 8;; ++h) {
    const uint32_t at = h & (GX_OWN - 1);
    const uint32_t k = keys[at];
    if (k == key) return cnts[at];
    if (k == GX_NONE) return 0u;
  }
}
struct GxOld {  // a block's previous sample, on its way
  uint32_t id[GX_SQ], nr[GX_SQ], len;
};
// (A.old_*: the buffers the previous sweep wrote; the exact chain rewrites them in place, the parallel sweep writes the others)
__device__ __forceinline__ void gx_request_old(const GxArgs& A, const GxBlock& B, uint32_t b, uint32_t lane, GxOld& O) {
  O.len = A.old_len[b];
#pragma unroll
  for (int q = 0; q < GX_SQ; ++q) {  // (read past the sample: within its capacity or the buffer's padding)
    O.id[q] = A.old_ids[B.sample_off + lane + q * 64];
    O.nr[q] = A.old_nrm[B.sample_off + lane + q * 64];
  }
}
__device__ __forceinline__ void gx_take_out(const GxArgs& A, const GxBlock& B, uint32_t lane, const GxOld& O) {
#pragma unroll
  for (int q = 0; q < GX_SQ; ++q)
    if (lane + q * 64 < O.len && O.nr[q] != GX_NONE) {
      gx_add(A.p_x + O.id[q], -B.wt);
      gx_add(A.normsum + O.nr[q], -B.wt);
      if (A.p_touch) A.p_touch[O.id[q]] = A.time;
    }
  for (uint32_t k = lane + GX_SQ * 64; k < O.len; k += 64) {
    const uint32_t n = A.old_nrm[B.sample_off + k];
    if (n == GX_NONE) continue;
    gx_add(A.p_x + A.old_ids[B.sample_off + k], -B.wt);
    gx_add(A.normsum + n, -B.wt);
    if (A.p_touch) A.p_touch[A.old_ids[B.sample_off + k]] = A.time;
  }
}

// what a block's sweep needs that no sweep changes -- arc records, offsets, levels -- for a block of the chain's usual size (at
// most 256 arcs, 320 states, 192 levels; every index clamped into the block, so the loads are unconditional): the exact chain
// requests the NEXT block's while this block's sweep runs
struct GxStatic {
  uint4 r4[4];
  uint2 n4[4];
  uint32_t oo[5], ll[3], sl[5], lev[2], levn[2];  // (levn: the entry after lev's, for the levels' widths)
};
__device__ __forceinline__ void gx_request_static(const GxArgs& A, const GxBlock& B, uint32_t lane, GxStatic& S) {
  const uint4* __restrict__ rec = A.arc_rec + B.out_base;
  const uint2* __restrict__ nrm = A.arc_nrm + B.out_base;
  const uint32_t nl = B.n_levels & 0x7fffffffu;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t a = (uint32_t)q * 64u + lane, at = a < B.n_arcs ? a : 0u;
    S.r4[q] = rec[at];
    S.n4[q] = nrm[at];
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    S.oo[j] = A.out_off[B.off_base + min((uint32_t)j * 64u + lane, B.n_states)];
    S.sl[j] = A.state_lev[B.off_base + min((uint32_t)j * 64u + lane, B.n_states)];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) S.ll[j] = A.level_off[B.level_base + min((uint32_t)j * 64u + lane, nl)];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    S.lev[j] = A.lev_arc[B.level_base + min((uint32_t)j * 64u + lane, nl)];
    S.levn[j] = A.lev_arc[B.level_base + min((uint32_t)j * 64u + lane + 1u, nl)];
  }
}

// PAR = false: the reference's chain, one wavefront, blocks in order, live counts.
// PAR = true:  the stale-count parallel sweep (never the default): every wavefront of the grid takes blocks
//              blockIdx, blockIdx + gridDim, ... against the counts of the PREVIOUS sweep (A.p_x / A.normsum point at the
//              snapshot) with the block's own previous sample taken out arithmetically (counterfactual counts: every entry
//              of it is compared with every lattice arc's parameters and norm groups); the new samples go to the other sample
//              buffer and gibbs.hip's recount / commit kernels rebuild the counts from them.
#ifndef GX_PAR_WAVES
#define GX_PAR_WAVES 4
#endif
// (PAR: four wavefronts a SIMD -- the sweep waits for gathers from the count tables and has a grid to hide them behind; the
// chain is one wavefront, or 64, and keeps the registers it wants)
template <bool PAR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PAR ? GX_PAR_WAVES : 1, PAR ? GX_PAR_WAVES : 3))) void gibbs_exact_wave_kernel(GxArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gx_lds[];
  if (!PAR && A.n_chains > 1) {  // chain blockIdx.x of several: its own counts, cache model, sample, results and uniforms
    const uint32_t c = blockIdx.x;
    A.p_x += (size_t)c * A.ch_params;
    A.ccount += (size_t)c * A.ch_params;
    A.normsum += (size_t)c * A.ch_norms;
    A.csum += (size_t)c * A.ch_norms;
    A.sample_len += (size_t)c * A.n_blocks;
    A.sample_ids += (size_t)c * A.ch_sample;
    A.sample_nrm += (size_t)c * A.ch_sample;
    A.old_len += (size_t)c * A.n_blocks;
    A.old_ids += (size_t)c * A.ch_sample;
    A.old_nrm += (size_t)c * A.ch_sample;
    A.iter_out += (size_t)c * 8;
    A.iter += c * A.iter_stride;
    if (c != A.init_chain) A.init_logw = nullptr;
  }
  const uint32_t CA = A.cap_arcs, CS = A.cap_states, CL = A.cap_levels, CM = A.cap_sample;
  double* gw = (double*)gx_lds;          // proposal weight the walk samples from
  double* pc = gw + CA;                  // ... from the counts (what the proposal probability of the sample is made of)
  double* sh = pc + CA;                  // the arc's share of its source state's total
  double* bv = sh + CA;                  // backward values
  double* bsum = bv + CS;
  uint32_t* ds = (uint32_t*)(bsum + CS);
  uint32_t* par0 = ds + CA;
  uint32_t* par1 = par0 + CA;
  int* be = (int*)(par1 + CA);
  int* emx = be + CS;
  uint32_t* ooff = (uint32_t*)(emx + CS);
  uint32_t* lvl = ooff + CS + 1;
  uint32_t* ids = lvl + CL + 1;
  uint32_t* idn = ids + CM;
  uint32_t* own_k = idn + CM;  // PAR: the block's previous sample as {parameter -> uses} and {norm group -> uses}
  uint32_t* own_c = own_k + 2 * GX_OWN;
  // (round 6) the arcs' norm groups, the states' levels and the path's norm groups beside its parameters: what the walk and the
  // count update used to fetch from global memory -- a dependent round trip each -- in the middle of the block
  uint32_t* nr0 = own_c + 2 * GX_OWN;
  uint32_t* nr1 = nr0 + CA;
  uint32_t* slv = nr1 + CA;
  uint32_t* idm = slv + CS;
  const uint32_t lane = threadIdx.x;
  const uint32_t nb = A.n_blocks;
  GxProd cheap{1.0, 0}, cnum{1.0, 0}, cden{1.0, 0}, after{1.0, 0};
  unsigned long long clk[6] = {0, 0, 0, 0, 0, 0}, sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GX_STAMP(v) const unsigned long long v = A.phase_clk ? __builtin_readcyclecounter() : 0;
  // PAR: this launch's blocks are A.list[0 .. nb) (a launch per LDS size class); i counts through the list
  const uint32_t b0 = PAR ? blockIdx.x : 0u, stride = PAR ? gridDim.x : 1u;
  if (b0 >= nb) return;
#define GX_BLOCK_ID(i) ((PAR && A.list) ? A.list[(i)] : (i))
  GxBlock B = A.blocks[GX_BLOCK_ID(b0)], Bn = A.blocks[GX_BLOCK_ID(min(b0 + stride, nb - 1))];
  GxOld O;
  gx_request_old(A, B, GX_BLOCK_ID(b0), lane, O);
  if (!PAR && !A.want_after) gx_take_out(A, B, lane, O);
  GxStatic S;
  if (!PAR) gx_request_static(A, B, lane, S);
  // the cache model's factors of the last block's sample (what its adds returned): multiplied in a block late, when the wait
  // for them is the wait the chain's order asks for anyway
  double pend_c = 1.0, pend_s = 1.0, pend_p = 1.0;
  uint32_t pend_kind = 0;  // 1: a counted parameter (pend_c, pend_s), 2: a fixed one (pend_p: its prior)
  uint32_t n_prev = 0;
  for (uint32_t bi = b0; bi < nb; bi += stride) {
    const uint32_t b = GX_BLOCK_ID(bi);  // the block's number in the corpus: what its uniforms and its sample are keyed by
    unsigned long long t0 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    if (!PAR) {
      gx_order();  // the counts are as the chain has them
      cnum.mul(pend_kind == 1u ? pend_c : (pend_kind == 2u ? pend_p : 1.0));
      cden.mul(pend_kind == 1u ? pend_s : 1.0);
      pend_kind = 0;
    }
    GX_STAMP(ta_)
    if (!PAR && A.want_after) {
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
    const uint32_t own_n = PAR && A.counterfactual ? O.len : 0u;
    const bool own_tab = PAR && own_n && own_n <= GX_SQ * 64;
    if (own_tab) {
      for (uint32_t i = lane; i < 2 * GX_OWN; i += 64) {
        own_k[i] = GX_NONE;
        own_c[i] = 0;
      }
      GX_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < GX_SQ; ++q)
        if (lane + q * 64 < own_n && O.nr[q] != GX_NONE) {  // (a fixed-probability parameter has no count to correct)
          gx_own_put(own_k, own_c, O.id[q]);
          gx_own_put(own_k + GX_OWN, own_c + GX_OWN, O.nr[q]);
        }
      GX_WAVE_SYNC();
    }
    // the first arc of every level, two registers a lane (round 6): a level's arc range is a v_readlane away instead of two
    // dependent LDS reads (level -> its first state -> that state's first arc); lattices of more levels keep the reads
    const bool lev_regs = (B.n_levels & 0x7fffffffu) < 128u;
    uint32_t lev_r[2] = {0u, 0u};
    bool lev_wide = false;  // some level has more than 64 arcs
    if (!PAR) {
      lev_r[0] = S.lev[0];
      lev_r[1] = S.lev[1];
      lev_wide = __builtin_amdgcn_ballot_w64(S.levn[0] - S.lev[0] > 64u || S.levn[1] - S.lev[1] > 64u) != 0;
    } else if (lev_regs) {
      bool w = false;
#pragma unroll
      for (int j = 0; j < 2; ++j) {  // (past the last level: the entry behind it, the lattice's arc count)
        const uint32_t nl = B.n_levels & 0x7fffffffu;
        lev_r[j] = A.lev_arc[B.level_base + min((uint32_t)j * 64u + lane, nl)];
        w |= A.lev_arc[B.level_base + min((uint32_t)j * 64u + lane + 1u, nl)] - lev_r[j] > 64u;
      }
      lev_wide = __builtin_amdgcn_ballot_w64(w) != 0;
    }
    // the chain's usual block (round 6): at most 256 arcs, 320 states, 190 levels -- its records, offsets and level entries came
    // with the block before (GxStatic); every count is requested at once: one round trip for the block
    const bool staged = !PAR && !A.init_logw && B.n_arcs <= 256u && B.n_states < 320u && (B.n_levels & 0x7fffffffu) < 192u;
    if (staged) {
      uint4(&r4)[4] = S.r4;
      uint2(&n4)[4] = S.n4;
      uint32_t(&oo)[5] = S.oo, (&ll)[3] = S.ll, (&sl)[5] = S.sl;
      double x0[4], s0[4], x1[4], s1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool h0 = r4[q].z != GX_NONE, h1 = r4[q].w != GX_NONE, f0 = n4[q].x == GX_NONE, f1 = n4[q].y == GX_NONE;
        x0[q] = gx_ld((f0 ? A.p_prior : (const double*)A.p_x) + (h0 ? r4[q].z : 0u));
        s0[q] = gx_ld(A.normsum + ((h0 && !f0) ? n4[q].x : 0u));
        x1[q] = gx_ld((f1 ? A.p_prior : (const double*)A.p_x) + (h1 ? r4[q].w : 0u));
        s1[q] = gx_ld(A.normsum + ((h1 && !f1) ? n4[q].y : 0u));
      }
#pragma unroll
      for (int j = 0; j < 5; ++j)
        if ((uint32_t)j * 64u + lane <= B.n_states) {
          ooff[(uint32_t)j * 64u + lane] = oo[j];
          if ((uint32_t)j * 64u + lane < B.n_states) slv[(uint32_t)j * 64u + lane] = sl[j];
        }
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if ((uint32_t)j * 64u + lane <= (B.n_levels & 0x7fffffffu)) lvl[(uint32_t)j * 64u + lane] = ll[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t a = (uint32_t)q * 64u + lane;
        if (a < B.n_arcs) {
          const bool h0 = r4[q].z != GX_NONE, h1 = r4[q].w != GX_NONE, f0 = n4[q].x == GX_NONE, f1 = n4[q].y == GX_NONE;
          const double w = ((h0 ? x0[q] : 1.0) / ((h0 && !f0) ? s0[q] : 1.0)) * ((h1 ? x1[q] : 1.0) / ((h1 && !f1) ? s1[q] : 1.0));
          pc[a] = w;
          gw[a] = w;
          ds[a] = r4[q].x;
          par0[a] = r4[q].z;
          par1[a] = r4[q].w;
          nr0[a] = n4[q].x;
          nr1[a] = n4[q].y;
        }
      }
    } else
    for (uint32_t a0_ = 0; a0_ < B.n_arcs; a0_ += 64) {
      const uint32_t a = a0_ + lane;
      const bool have = a < B.n_arcs;
      const uint4 r = have ? rec[a] : make_uint4(0, 0, GX_NONE, GX_NONE);
      const uint2 n = have ? nrm[a] : make_uint2(GX_NONE, GX_NONE);
      double x0 = 1.0, s0 = 1.0, x1 = 1.0, s1 = 1.0;
      if (r.z != GX_NONE) {
        if (n.x == GX_NONE)
          x0 = A.p_prior[r.z];
        else {
          x0 = PAR ? A.p_x[r.z] : gx_ld(A.p_x + r.z);
          s0 = PAR ? A.normsum[n.x] : gx_ld(A.normsum + n.x);
        }
      }
      if (r.w != GX_NONE) {
        if (n.y == GX_NONE)
          x1 = A.p_prior[r.w];
        else {
          x1 = PAR ? A.p_x[r.w] : gx_ld(A.p_x + r.w);
          s1 = PAR ? A.normsum[n.y] : gx_ld(A.normsum + n.y);
        }
      }
      if (PAR && own_n) {
        // counterfactual counts: this block's previous sample does not count -- how often it uses this arc's parameters and
        // their norm groups comes out of two small tables the wavefront filled from the sample (below, once per block)
        if (own_tab) {
          const uint32_t c0 = r.z != GX_NONE ? gx_own_get(own_k, own_c, r.z) : 0u, c1 = r.w != GX_NONE ? gx_own_get(own_k, own_c, r.w) : 0u;
          const uint32_t m0 = n.x != GX_NONE ? gx_own_get(own_k + GX_OWN, own_c + GX_OWN, n.x) : 0u,
                         m1 = n.y != GX_NONE ? gx_own_get(own_k + GX_OWN, own_c + GX_OWN, n.y) : 0u;
          x0 -= (double)c0 * B.wt;
          s0 -= (double)m0 * B.wt;
          x1 -= (double)c1 * B.wt;
          s1 -= (double)m1 * B.wt;
        } else {  // a sample too long for the tables: entry by entry
          uint32_t c0 = 0, m0 = 0, c1 = 0, m1 = 0;
          for (uint32_t k = 0; k < own_n; ++k) {
            const uint32_t id = A.old_ids[B.sample_off + k], nr = A.old_nrm[B.sample_off + k];
            c0 += id == r.z;
            c1 += id == r.w;
            m0 += nr == n.x;
            m1 += nr == n.y;
          }
          if (n.x != GX_NONE) {
            x0 -= (double)c0 * B.wt;
            s0 -= (double)m0 * B.wt;
          }
          if (n.y != GX_NONE) {
            x1 -= (double)c1 * B.wt;
            s1 -= (double)m1 * B.wt;
          }
        }
      }
      if (!have) continue;
      const double w = (x0 / s0) * (x1 / s1);
      pc[a] = w;
      gw[a] = A.init_logw ? exp(A.init_logw[r.y]) : w;
      ds[a] = r.x;
      par0[a] = r.z;
      par1[a] = r.w;
      nr0[a] = n.x;
      nr1[a] = n.y;
    }
    if (!staged) {
      for (uint32_t s = lane; s <= B.n_states; s += 64) ooff[s] = A.out_off[B.off_base + s];
      for (uint32_t s = lane; s < B.n_states; s += 64) slv[s] = A.state_lev[B.off_base + s];
    }
    const uint32_t n_levels = B.n_levels & 0x7fffffffu;  // (bit 31: the lattice is a trellis)
    if (!staged)
      for (uint32_t l = lane; l <= n_levels; l += 64) lvl[l] = A.level_off[B.level_base + l];
    GX_STAMP(tb_)
    // the next block's records and its previous sample set out now
    if (!PAR) gx_request_static(A, Bn, lane, S);
    GxOld On;
    gx_request_old(A, Bn, GX_BLOCK_ID(min(bi + stride, nb - 1)), lane, On);
    const GxBlock Bnn = A.blocks[GX_BLOCK_ID(min(bi + 2 * stride, nb - 1))];
    const double U = gibbs_uniform(A.seed, A.iter, b, lane);  // the walk's first 64 uniforms, one per lane
    unsigned long long t1 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- backward sweep (derivations.h:345-360): beta[s] = sum over out-arcs of weight x beta[destination].  A state's value
    // is read only by the levels before it, so the terms are added straight into it; the level's arc range comes from the
    // level table one level ahead of its use ----
    bool ext = false;
    GX_WAVE_SYNC();
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
      if (lev_regs && !ext) {
        // a level's first 64 arcs (their ends and weights) are requested a level ahead.  Every read of the loop is unconditional
        // (indices clamped into the block) and the destinations' values are requested FIRST: the compiler's wait for them is
        // then a count that leaves the requests behind them in flight (LDS returns in order) -- with the reads under a
        // condition it waited for everything, three round trips a level.  Between levels nothing waits: a wavefront's LDS
        // operations execute in the order they were issued, so the next level's reads see this level's adds.
#define GX_LEV(l) ((uint32_t)__builtin_amdgcn_readlane((int)((l) < 64u ? lev_r[0] : lev_r[1]), (int)((l) & 63u)))
#define GX_LDS_ORDER()                   \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_wave_barrier();     \
  } while (0)
        const uint32_t a_last = B.n_arcs ? B.n_arcs - 1u : 0u;
        uint32_t a_hi = B.n_arcs;
        uint32_t pd = 0;
        double pg = 0.0;
        if (n_levels) {
          const uint32_t a0 = GX_LEV(n_levels - 1);
          pd = ds[min(a0 + lane, a_last)];
          pg = gw[min(a0 + lane, a_last)];
        }
        if (!lev_wide) {
          // no branch in the loop: a lane without an arc of the level stores and adds a zero at a slot of its own in bsum
          // (this pass does not use it; the pass with exponents clears it first)
          double* const idle = bsum + min(lane, CS - 1u);
          // (a level's first arc: ONE v_readlane a level -- the one two levels ahead; the others are carried)
          uint32_t lo0 = n_levels ? GX_LEV(n_levels - 1u) : 0u, lo1 = GX_LEV(n_levels >= 2u ? n_levels - 2u : 0u);
          uint32_t nidx = min(lo1 + lane, a_last);  // where the level after's arcs are
          GX_WAVE_SYNC();  // (nothing in flight at the loop's head: its waits count what the loop itself issued)
          for (uint32_t l = n_levels; l-- > 0;) {
            const double v = bv[pd & 0xffffu];
            const uint32_t nd = ds[nidx];
            const double ng = gw[nidx];
            const uint32_t a_lo = lo0;
            const uint32_t lo2 = GX_LEV(l >= 2u ? l - 2u : 0u);
            lo0 = lo1;
            lo1 = lo2;
            nidx = min(lo2 + lane, a_last);
            const bool on = a_lo + lane < a_hi;
            const double term = on ? pg * v : 0.0;
            *(on ? &sh[a_lo + lane] : idle) = term;
            gx_lds_add(on ? &bv[pd >> 16] : idle, term);
            GX_LDS_ORDER();
            pd = nd;
            pg = ng;
            a_hi = a_lo;
          }
        } else
          for (uint32_t l = n_levels; l-- > 0;) {
            const uint32_t a_lo = GX_LEV(l);
            const uint32_t n_lo = l ? GX_LEV(l - 1) : a_lo;
            const double v = bv[pd & 0xffffu];
            const uint32_t nd = ds[min(n_lo + lane, a_last)];
            const double ng = gw[min(n_lo + lane, a_last)];
            const double term = pg * v;
            if (a_lo + lane < a_hi) {
              sh[a_lo + lane] = term;
              if (term != 0.0) gx_lds_add(&bv[pd >> 16], term);
            }
            for (uint32_t a = a_lo + 64 + lane; a < a_hi; a += 64) {  // (a level of more than 64 arcs)
              const uint32_t d = ds[a];
              const double t2 = gw[a] * bv[d & 0xffffu];
              sh[a] = t2;
              if (t2 != 0.0) gx_lds_add(&bv[d >> 16], t2);
            }
            GX_WAVE_SYNC();
            pd = nd;
            pg = ng;
            a_hi = a_lo;
          }
        GX_WAVE_SYNC();
#undef GX_LDS_ORDER
#undef GX_LEV
      } else {
      uint32_t a_hi = ooff[lvl[n_levels]];
      uint32_t s_next = n_levels ? lvl[n_levels - 1] : 0u;
      for (uint32_t l = n_levels; l-- > 0;) {
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
    if (B.n_levels & 0x80000000u) {
      // TRELLIS (every arc joins neighbouring levels -- a tagger's lattice, any epsilon-free pair): the walk visits one state
      // per level, so the uniform of a state's choice is its level's, and EVERY state makes its choice at once -- a lane per
      // state, the reference's subtraction over its list of arcs (random.ipp:111-127).  What is left of the walk is following
      // the chosen arcs from the start: one LDS read per step.  be / emx hold the chosen arc and its destination.
      for (uint32_t s = lane; s < B.n_states; s += 64) {
        const uint32_t a0 = ooff[s], a1 = ooff[s + 1];
        uint32_t pick = a0;
        if (a1 > a0) {
          double choice = gibbs_uniform(A.seed, A.iter, b, slv[s]) * tot[s];
          bool done = false;
          for (uint32_t top = a1; top > a0 && !done; top -= min(4u, top - a0)) {  // list order: newest first, four shares a round
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = top >= a0 + 1u + (uint32_t)i ? sh[top - 1u - (uint32_t)i] : 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (!done && top >= a0 + 1u + (uint32_t)i) {
                choice -= v[i];
                pick = top - 1u - (uint32_t)i;
                done = choice < 0;
              }
          }
        }
        // (the chosen arc and its destination in one word: a step of the path below is one LDS read)
        be[s] = (int)(pick | ((a1 > a0 ? (ds[pick] & 0xffffu) : B.fin) << 16));
      }
      GX_WAVE_SYNC();
      GX_STAMP(tc_)
      uint32_t s = B.start, n_path = 0;
      // a path of at most 64 arcs (one a level) stays in a register, lane t its t-th arc: no store a step, no read back
      const bool short_path = n_levels <= 64u && n_levels <= CM && B.n_states <= 320u;
      uint32_t path_r = 0;
      if (B.n_states <= 320u) {
        // (round 6) the states' words in registers, lane s % 64 of register s / 64: a step of the path is a v_readlane (the
        // state is the same in every lane), not the round trip of an LDS read
        uint32_t wr[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) wr[j] = (uint32_t)be[min((uint32_t)j * 64u + lane, max(B.n_states, 1u) - 1u)];
        GX_WAVE_SYNC();  // (arrived: the loops below wait for nothing -- their stores of the path's arcs are not waited for)
        if (short_path && B.n_states <= 64u)
          while (s != B.fin) {  // (a step: v_readlane, a compare and a select, a shift; the goal's word leads to the goal)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint32_t wd = (uint32_t)__builtin_amdgcn_readlane((int)wr[0], (int)s);
              path_r = lane == n_path ? wd : path_r;
              n_path += s != B.fin;
              s = wd >> 16;
            }
          }
        else if (short_path)
          while (s != B.fin) {
            const uint32_t c = s >> 6;
            uint32_t w = c == 1u ? wr[1] : wr[0];
            w = c == 2u ? wr[2] : w;
            w = c == 3u ? wr[3] : w;
            w = c == 4u ? wr[4] : w;
            const uint32_t wd = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)(s & 63u));
            path_r = lane == n_path ? wd : path_r;
            ++n_path;
            s = wd >> 16;
          }
        else if (B.n_states <= 64u)
          while (s != B.fin && n_path < CM) {
            const uint32_t wd = (uint32_t)__builtin_amdgcn_readlane((int)wr[0], (int)s);
            if (lane == 0) idn[n_path] = wd & 0xffffu;  // (the path's arcs; their parameters are gathered below, side by side)
            ++n_path;
            s = wd >> 16;
          }
        else
          while (s != B.fin && n_path < CM) {
            const uint32_t c = s >> 6;
            uint32_t w = c == 1u ? wr[1] : wr[0];
            w = c == 2u ? wr[2] : w;
            w = c == 3u ? wr[3] : w;
            w = c == 4u ? wr[4] : w;
            const uint32_t wd = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)(s & 63u));
            if (lane == 0) idn[n_path] = wd & 0xffffu;
            ++n_path;
            s = wd >> 16;
          }
      } else
        while (s != B.fin && n_path < CM) {
          const uint32_t wd = (uint32_t)be[s];
          if (lane == 0) idn[n_path] = wd & 0xffffu;
          ++n_path;
          s = wd >> 16;
        }
      GX_WAVE_SYNC();
      GX_STAMP(td_)
      if (A.phase_clk) {
        sub[4] += tc_ - t2;
        sub[5] += td_ - tc_;
      }
      for (uint32_t base = 0; base < n_path; base += 64) {
        const uint32_t t = base + lane;
        const bool have = t < n_path;
        const uint32_t a = have ? (short_path ? path_r & 0xffffu : idn[t]) : 0u;
        const uint32_t p0 = have ? par0[a] : GX_NONE, p1 = have ? par1[a] : GX_NONE;
        const uint32_t g0 = nr0[a], g1 = nr1[a];
        if (have) cheap.mul(pc[a]);
        // slots of the parameters in the sample: chain order within an arc, arcs in path order
        const unsigned long long m0 = __ballot(p0 != GX_NONE), m1 = __ballot(p1 != GX_NONE);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        const uint32_t at = n_ids + (uint32_t)__popcll(m0 & below) + (uint32_t)__popcll(m1 & below);
        if (p0 != GX_NONE && at < CM) {
          ids[at] = p0;
          idm[at] = g0;
        }
        if (p1 != GX_NONE && at + (p0 != GX_NONE) < CM) {
          ids[at + (p0 != GX_NONE)] = p1;
          idm[at + (p0 != GX_NONE)] = g1;
        }
        n_ids += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
      }
      n_ids = min(n_ids, CM);
      if (A.phase_clk) sub[6] += __builtin_readcyclecounter() - td_;
    } else {
      uint32_t s = B.start;
      while (s != B.fin) {
        const uint32_t a0 = ooff[s], a1 = ooff[s + 1], deg = a1 - a0;
        if (!deg) break;  // (a dead end cannot be reached: pruned lattices)
        const double u = step < 64 ? gx_rl(U, step) : gibbs_uniform(A.seed, A.iter, b, step);
        ++step;
        double choice = u * tot[s];
        // lane i holds the i-th arc of the state's list (its share and what the walk records of it): the chosen one's are
        // read across lanes, not fetched again
        uint32_t p0 = GX_NONE, p1 = GX_NONE, n0 = GX_NONE, n1 = GX_NONE, to = B.fin;
        double pcv = 1.0;
        if (deg <= 16) {
          // the usual case: the shares' running sums by a scan over the first row of lanes (DPP row shifts, no memory);
          // the reference stops at the first arc whose running sum exceeds u x total
          const uint32_t at = lane < deg ? a1 - 1 - lane : a0;
          const double share = lane < deg ? sh[at] : 0.0;
          const double pcl = pc[at];
          const uint32_t dl = ds[at], q0 = par0[at], q1 = par1[at];
          const uint32_t h0 = nr0[at], h1 = nr1[at];
          double run = share;
          run += gx_row_shr<0x111>(run);
          run += gx_row_shr<0x112>(run);
          run += gx_row_shr<0x114>(run);
          run += gx_row_shr<0x118>(run);
          const unsigned long long passed = __ballot(lane < deg && run > choice);
          const uint32_t j = passed ? (uint32_t)__builtin_ctzll(passed) : deg - 1;
          p0 = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)j);
          p1 = (uint32_t)__builtin_amdgcn_readlane((int)q1, (int)j);
          n0 = (uint32_t)__builtin_amdgcn_readlane((int)h0, (int)j);
          n1 = (uint32_t)__builtin_amdgcn_readlane((int)h1, (int)j);
          to = (uint32_t)__builtin_amdgcn_readlane((int)dl, (int)j) & 0xffffu;
          pcv = gx_rl(pcl, j);
        } else {
        bool done = false;
        for (uint32_t base = 0; base < deg && !done; base += 64) {
          const uint32_t mine = base + lane, at = mine < deg ? a1 - 1 - mine : a0;
          const double share = mine < deg ? sh[at] : 0.0;
          const double pcl = pc[at];
          const uint32_t dl = ds[at], q0 = par0[at], q1 = par1[at];
          const uint32_t h0 = nr0[at], h1 = nr1[at];
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
          n0 = (uint32_t)__builtin_amdgcn_readlane((int)h0, (int)j);
          n1 = (uint32_t)__builtin_amdgcn_readlane((int)h1, (int)j);
          to = (uint32_t)__builtin_amdgcn_readlane((int)dl, (int)j) & 0xffffu;
          pcv = gx_rl(pcl, j);
        }
        }
        if (lane == 0) {
          if (p0 != GX_NONE && n_ids < GX_SAMPLE) {
            ids[n_ids] = p0;
            idm[n_ids] = n0;
          }
          if (p1 != GX_NONE && n_ids + 1 < GX_SAMPLE) {
            ids[n_ids + 1] = p1;
            idm[n_ids + 1] = n1;
          }
        }
        n_ids += p0 == GX_NONE ? 0u : (p1 != GX_NONE ? 2u : 1u);
        if (lane == 0) cheap.mul(pcv);  // (per-lane products: the trellis path multiplies side by side)
        s = to;
      }
      n_ids = min(n_ids, (uint32_t)GX_SAMPLE);
    }
    GX_WAVE_SYNC();
    unsigned long long t3 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- the new sample into the counts, the next block's previous sample out (gibbs.hpp:851-871, 712-742, 769-792); the
    // parallel sweep only writes its sample down (gibbs.hip recounts) ----
    // (what was requested before the sweep -- the next block's records and previous sample -- has long arrived; saying so here
    // keeps the compiler from waiting for it below, behind the adds)
    if (!PAR) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    if (!PAR && n_ids <= 64u) {
      // the usual sample, an entry a lane: the adds whose results the cache model wants are issued by EVERY lane (a lane without
      // an entry, or with a fixed parameter, adds zero to a slot of A.idle) and nothing here waits for them -- their results
      // are multiplied in at the top of the next block, behind the wait the chain's order asks for anyway
      const bool on = lane < n_ids;
      const uint32_t p = ids[on ? lane : 0u], n = idm[on ? lane : 0u];
      const bool cnt = on && n != GX_NONE, fix = on && n == GX_NONE;
      if (on) {
        idn[lane] = n;
        A.sample_ids[B.sample_off + lane] = p;
        A.sample_nrm[B.sample_off + lane] = n;
      }
      if (cnt) {
        gx_add(A.p_x + p, B.wt);
        gx_add(A.normsum + n, B.wt);
        if (A.p_touch) A.p_touch[p] = A.time;
      }
      pend_c = gx_add(cnt ? A.ccount + p : A.idle + lane, cnt ? 1.0 : 0.0);
      pend_s = gx_add(cnt ? A.csum + n : A.idle + 64 + lane, cnt ? 1.0 : 0.0);
      pend_p = A.p_prior[fix ? p : 0u];
      pend_kind = cnt ? 1u : (fix ? 2u : 0u);
    } else
    for (uint32_t k = lane; k < n_ids; k += 64) {
      const uint32_t p = ids[k], n = idm[k];  // (= p_norm[p]: the arc's record says so)
      idn[k] = n;
      A.sample_ids[B.sample_off + k] = p;
      A.sample_nrm[B.sample_off + k] = n;
      if (PAR) continue;
      if (n != GX_NONE) {
        gx_add(A.p_x + p, B.wt);
        gx_add(A.normsum + n, B.wt);
        if (A.p_touch) A.p_touch[p] = A.time;
        cnum.mul(gx_add(A.ccount + p, 1.0));
        cden.mul(gx_add(A.csum + n, 1.0));
      } else
        cnum.mul(A.p_prior[p]);
    }
    if (lane == 0) A.sample_len[b] = n_ids;
    if (!PAR && !A.want_after && bi + 1 < nb) gx_take_out(A, Bn, lane, On);
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
      sub[0] += ta_ - t0;
      sub[1] += tb_ - ta_;
      sub[2] += t1 - tb_;
    }
  }
  if (PAR) {  // the sweep's proposal probability: every wavefront adds its blocks'
    double cl = cheap.ln();
    for (int o = 32; o > 0; o >>= 1) cl += __shfl_down(cl, o, 64);
    if (lane == 0) {
      unsafeAtomicAdd(A.iter_out + 1, cl);
      if (A.phase_clk)
        for (int k = 0; k < 5; ++k) atomicAdd(A.phase_clk + k, clk[k]);
    }
    return;
  }
  gx_order();
  cnum.mul(pend_kind == 1u ? pend_c : (pend_kind == 2u ? pend_p : 1.0));
  cden.mul(pend_kind == 1u ? pend_s : 1.0);
  if (A.want_after)
    for (uint32_t k = lane; k < n_prev; k += 64) {
      const uint32_t p = ids[k], n = idn[k];
      after.mul(n == GX_NONE ? A.p_prior[p] : gx_ld(A.p_x + p) / gx_ld(A.normsum + n));
    }
  double cache_ln = cnum.ln() - cden.ln(), after_ln = after.ln(), cheap_ln = cheap.ln();
  for (int o = 32; o > 0; o >>= 1) {
    cache_ln += __shfl_down(cache_ln, o, 64);
    after_ln += __shfl_down(after_ln, o, 64);
    cheap_ln += __shfl_down(cheap_ln, o, 64);
  }
  if (lane == 0) {
    A.iter_out[0] = cache_ln;
    A.iter_out[1] = cheap_ln;
    A.iter_out[2] = after_ln;
    if (A.phase_clk) {
      for (int k = 0; k < 5; ++k) A.phase_clk[k] += clk[k];
      for (int k = 0; k < 7; ++k) A.phase_clk[8 + k] += sub[k];
    }
  }
}

// ---- the same sweep with a block's arcs in REGISTERS (round 6) ----
// What the kernel above spends its time on (profiles/r6_v0_crp_phase_cycles.txt: 61 k cycles a block in the parallel sweep) is
// not arithmetic and not bandwidth but round trips: every chunk of 64 arcs waits for its records, then for the counts they point
// at (28 k cycles), and a level of the backward sweep is a chain of five dependent LDS reads (offset -> arc -> destination's value
// -> add -> sync: 670 cycles x 26 levels).  Here, for TRELLIS blocks (every arc joins neighbouring levels: a tagger's lattice, any
// epsilon-free pair) of at most 64 NQ arcs and states and 127 levels:
//   * lane i owns arcs i, i + 64, ... for the whole block: ends, parameters, norm groups, proposal weight and share live in
//     registers; a level of the sweep is ONE LDS read (the destination's value) and one LDS add per arc, the level's arc range a
//     v_readlane away (lev_arc, built with the records: first arc of every level);
//   * the NEXT block's records are requested before this block's sweep and its counts -- the parallel sweep's are the snapshot's,
//     nobody changes them -- right after it: both round trips run behind the sweep and the walk (the exact chain's counts are live:
//     it requests them at the top of the block, all chunks at once);
//   * LDS per block: shares (8 B an arc, for the states' choices), destinations (2 B), values / exponents / offsets per state, the
//     own-sample tables sized by the class's longest sample -- a quarter of the LDS kernel's, so a CU holds its 16 wavefronts;
//   * a state's choice reads its shares four at a time; the path is one word per state (chosen arc | its destination); the path's
//     parameters come back out of the registers by ds_bpermute.
// Same values, same order of every state's subtraction, same uniforms: the sample of the kernel above (its sums differ by the
// order of the LDS adds, as its own do from run to run).  Anything else -- lattices with arcs that skip levels, longer ones, the
// sweeps that sample from --init-em weights -- keeps the kernel above.
#define GXR_LQ 2  // registers of level offsets per lane: at most 64 GXR_LQ - 1 levels
__device__ __forceinline__ void gxr_own_put(uint32_t* keys, uint32_t* cnts, uint32_t mask, uint32_t key) {
  for (uint32_t h = (key * 2654435761u) >> 8;; ++h) {
    const uint32_t at = h & mask;
    const uint32_t old = atomicCAS(keys + at, GX_NONE, key);
    if (old == GX_NONE || old == key) {
      atomicAdd(cnts + at, 1u);
      return;
    }
  }
}
__device__ __forceinline__ uint32_t gxr_own_get(const uint32_t* keys, const uint32_t* cnts, uint32_t mask, uint32_t key) {
  for (uint32_t h = (key * 2654435761u) >> 8;; ++h) {
    const uint32_t at = h & mask;
    const uint32_t k = keys[at];
    if (k == key) return cnts[at];
    if (k == GX_NONE) return 0u;
  }
}
__device__ __forceinline__ uint32_t gxr_bperm(uint32_t v, uint32_t from_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from_lane << 2), (int)v);
}
template <bool PAR, int NQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PAR ? (NQ <= 4 ? 4 : 2) : 1, PAR ? (NQ <= 4 ? 4 : 2) : (NQ <= 8 ? 2 : 1)))) void gibbs_reg_wave_kernel(GxArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gx_lds[];
  if (!PAR && A.n_chains > 1) {
    const uint32_t c = blockIdx.x;
    A.p_x += (size_t)c * A.ch_params;
    A.ccount += (size_t)c * A.ch_params;
    A.normsum += (size_t)c * A.ch_norms;
    A.csum += (size_t)c * A.ch_norms;
    A.sample_len += (size_t)c * A.n_blocks;
    A.sample_ids += (size_t)c * A.ch_sample;
    A.sample_nrm += (size_t)c * A.ch_sample;
    A.old_len += (size_t)c * A.n_blocks;
    A.old_ids += (size_t)c * A.ch_sample;
    A.old_nrm += (size_t)c * A.ch_sample;
    A.iter_out += (size_t)c * 8;
    A.iter += c * A.iter_stride;
  }
  const uint32_t CA = A.cap_arcs, CS = A.cap_states, CM = A.cap_sample, OWN = PAR ? A.own_slots : 0u;
  double* sh = (double*)gx_lds;
  double* bv = sh + CA;
  double* bsum = bv + CS;
  int* be = (int*)(bsum + CS);
  int* emx = be + CS;
  uint32_t* ooff = (uint32_t*)(emx + CS);
  uint32_t* ids = ooff + CS + 1;
  uint32_t* idn = ids + CM;
  uint32_t* own_k = idn + CM;
  uint32_t* own_c = own_k + 2 * OWN;
  unsigned short* dst16 = (unsigned short*)(own_c + 2 * OWN);
  const uint32_t lane = threadIdx.x;
  const uint32_t nb = A.n_blocks;
  GxProd cheap{1.0, 0}, cnum{1.0, 0}, cden{1.0, 0}, after{1.0, 0};
  unsigned long long clk[6] = {0, 0, 0, 0, 0, 0}, sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GXR_STAMP(v) const unsigned long long v = A.phase_clk ? __builtin_readcyclecounter() : 0;
  const uint32_t b0 = PAR ? blockIdx.x : 0u, stride = PAR ? gridDim.x : 1u;
  if (b0 >= nb) return;
  GxBlock B = A.blocks[GX_BLOCK_ID(b0)], Bn = A.blocks[GX_BLOCK_ID(min(b0 + stride, nb - 1))];
  GxOld O;
  gx_request_old(A, B, GX_BLOCK_ID(b0), lane, O);
  // the block's arcs: {parameters, ends, norm groups} per chunk, the first arc of every level; n*: the next block's, on their way
  uint32_t nz[NQ], nw[NQ], nds[NQ], nnx[NQ], nny[NQ], nlev[GXR_LQ];
  double x0[NQ], s0[NQ], x1[NQ], s1[NQ];
#define GXR_REQUEST_RECORDS(BLK)                                                                       \
  {                                                                                                    \
    const uint4* __restrict__ rec_ = A.arc_rec + (BLK).out_base;                                       \
    const uint2* __restrict__ nrm_ = A.arc_nrm + (BLK).out_base;                                       \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                   \
      const uint32_t a_ = (uint32_t)q * 64u + lane;                                                    \
      const bool have_ = a_ < (BLK).n_arcs;                                                            \
      const uint4 r_ = have_ ? rec_[a_] : make_uint4(0, 0, GX_NONE, GX_NONE);                          \
      const uint2 n_ = have_ ? nrm_[a_] : make_uint2(GX_NONE, GX_NONE);                                \
      nds[q] = r_.x;                                                                                   \
      nz[q] = r_.z;                                                                                    \
      nw[q] = r_.w;                                                                                    \
      nnx[q] = n_.x;                                                                                   \
      nny[q] = n_.y;                                                                                   \
    }                                                                                                  \
    const uint32_t nl_ = (BLK).n_levels & 0x7fffffffu;                                                 \
    _Pragma("unroll") for (int j = 0; j < GXR_LQ; ++j)                                                 \
      nlev[j] = (uint32_t)j * 64u + lane <= nl_ ? A.lev_arc[(BLK).level_base + (uint32_t)j * 64u + lane] : (BLK).n_arcs; \
  }
  // the counts the records point at (the parallel sweep: the snapshot's, plain loads; the chain: live, workgroup-coherent)
#define GXR_REQUEST_COUNTS()                                                    \
  _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                              \
    x0[q] = 1.0;                                                                \
    s0[q] = 1.0;                                                                \
    x1[q] = 1.0;                                                                \
    s1[q] = 1.0;                                                                \
    if (nz[q] != GX_NONE) {                                                     \
      if (nnx[q] == GX_NONE)                                                    \
        x0[q] = A.p_prior[nz[q]];                                               \
      else {                                                                    \
        x0[q] = PAR ? A.p_x[nz[q]] : gx_ld(A.p_x + nz[q]);                      \
        s0[q] = PAR ? A.normsum[nnx[q]] : gx_ld(A.normsum + nnx[q]);            \
      }                                                                         \
    }                                                                           \
    if (nw[q] != GX_NONE) {                                                     \
      if (nny[q] == GX_NONE)                                                    \
        x1[q] = A.p_prior[nw[q]];                                               \
      else {                                                                    \
        x1[q] = PAR ? A.p_x[nw[q]] : gx_ld(A.p_x + nw[q]);                      \
        s1[q] = PAR ? A.normsum[nny[q]] : gx_ld(A.normsum + nny[q]);            \
      }                                                                         \
    }                                                                           \
  }
  GXR_REQUEST_RECORDS(B)
  if (PAR) GXR_REQUEST_COUNTS()
  if (!PAR && !A.want_after) gx_take_out(A, B, lane, O);
  uint32_t n_prev = 0;
  for (uint32_t bi = b0; bi < nb; bi += stride) {
    const uint32_t b = GX_BLOCK_ID(bi);
    unsigned long long t0 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    if (!PAR) gx_order();
    if (!PAR && A.want_after) {
      for (uint32_t k = lane; k < n_prev; k += 64) {
        const uint32_t p = ids[k], n = idn[k];
        after.mul(n == GX_NONE ? A.p_prior[p] : gx_ld(A.p_x + p) / gx_ld(A.normsum + n));
      }
      gx_take_out(A, B, lane, O);
      gx_order();
    }
    if (!PAR) GXR_REQUEST_COUNTS()  // (live counts: as the chain has them now)
    // this block's arcs take their registers
    uint32_t cz[NQ], cw[NQ], cds[NQ], lev[GXR_LQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      cz[q] = nz[q];
      cw[q] = nw[q];
      cds[q] = nds[q];
    }
#pragma unroll
    for (int j = 0; j < GXR_LQ; ++j) lev[j] = nlev[j];
    const uint32_t n_levels = B.n_levels & 0x7fffffffu;
    // offsets of the states' arcs and the level of every state (the key of its uniform): on their way while the weights are formed
    uint32_t oq[NQ + 1];
    uint32_t slv[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const uint32_t s = (uint32_t)j * 64u + lane;
      oq[j] = s <= B.n_states ? A.out_off[B.off_base + s] : 0u;
      slv[j] = s < B.n_states ? (uint32_t)A.state_lev[B.off_base + s] : 0u;
    }
    oq[NQ] = (lane == 0 && (uint32_t)NQ * 64u <= B.n_states) ? A.out_off[B.off_base + (uint32_t)NQ * 64u] : 0u;
    const uint32_t own_n = PAR && A.counterfactual ? O.len : 0u;
    if (PAR && own_n) {
      for (uint32_t i = lane; i < 2 * OWN; i += 64) {
        own_k[i] = GX_NONE;
        own_c[i] = 0;
      }
      GX_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < GX_SQ; ++q)
        if (lane + q * 64 < own_n && O.nr[q] != GX_NONE) {  // (a fixed-probability parameter has no count to correct)
          gxr_own_put(own_k, own_c, OWN - 1, O.id[q]);
          gxr_own_put(own_k + OWN, own_c + OWN, OWN - 1, O.nr[q]);
        }
      for (uint32_t k = lane + GX_SQ * 64; k < own_n; k += 64) {
        const uint32_t nr = A.old_nrm[B.sample_off + k];
        if (nr == GX_NONE) continue;
        gxr_own_put(own_k, own_c, OWN - 1, A.old_ids[B.sample_off + k]);
        gxr_own_put(own_k + OWN, own_c + OWN, OWN - 1, nr);
      }
      GX_WAVE_SYNC();
    }
    GXR_STAMP(ta)
    // ---- proposal weights (gibbs.cc:348-359, gibbs.hpp:153-157) ----
    double gwq[NQ], shq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const uint32_t a = (uint32_t)q * 64u + lane;
      double a0 = x0[q], b0_ = s0[q], a1 = x1[q], b1 = s1[q];
      if (PAR && own_n) {
        const uint32_t c0 = (cz[q] != GX_NONE && nnx[q] != GX_NONE) ? gxr_own_get(own_k, own_c, OWN - 1, cz[q]) : 0u;
        const uint32_t c1 = (cw[q] != GX_NONE && nny[q] != GX_NONE) ? gxr_own_get(own_k, own_c, OWN - 1, cw[q]) : 0u;
        const uint32_t m0 = nnx[q] != GX_NONE ? gxr_own_get(own_k + OWN, own_c + OWN, OWN - 1, nnx[q]) : 0u;
        const uint32_t m1 = nny[q] != GX_NONE ? gxr_own_get(own_k + OWN, own_c + OWN, OWN - 1, nny[q]) : 0u;
        a0 -= (double)c0 * B.wt;
        b0_ -= (double)m0 * B.wt;
        a1 -= (double)c1 * B.wt;
        b1 -= (double)m1 * B.wt;
      }
      gwq[q] = (a0 / b0_) * (a1 / b1);
      shq[q] = 0.0;
      if (a < B.n_arcs) dst16[a] = (unsigned short)(cds[q] & 0xffffu);
    }
    GXR_STAMP(tb)
#pragma unroll
    for (int j = 0; j < NQ; ++j)
      if ((uint32_t)j * 64u + lane <= B.n_states) ooff[(uint32_t)j * 64u + lane] = oq[j];
    if (lane == 0 && (uint32_t)NQ * 64u <= B.n_states) ooff[(uint32_t)NQ * 64u] = oq[NQ];
    // a state's uniform is its level's (one state per level on the walk's way)
    double us[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) us[j] = gibbs_uniform(A.seed, A.iter, b, slv[j]);
    GXR_STAMP(tc)
    // the next block's records and previous sample set out now
    GxOld On;
    gx_request_old(A, Bn, GX_BLOCK_ID(min(bi + stride, nb - 1)), lane, On);
    const GxBlock Bnn = A.blocks[GX_BLOCK_ID(min(bi + 2 * stride, nb - 1))];
    GXR_REQUEST_RECORDS(Bn)
    unsigned long long t1 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- backward sweep (derivations.h:345-360), level by level from the goal's ----
#define GXR_LEV(l) ((l) < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)lev[0], (int)(l)) : (uint32_t)__builtin_amdgcn_readlane((int)lev[1], (int)((l) - 64u)))
    bool ext = false;
    for (;;) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const uint32_t s = (uint32_t)j * 64u + lane;
        if (s < B.n_states) {
          bv[s] = 0.0;
          bsum[s] = 0.0;
          be[s] = 0;
          emx[s] = -(1 << 28);
        }
      }
      GX_WAVE_SYNC();
      if (lane == 0) {
        bv[B.fin] = ext ? 0.5 : 1.0;
        bsum[B.fin] = 1.0;
        be[B.fin] = ext ? 1 : 0;
      }
      GX_WAVE_SYNC();
      uint32_t a_hi = B.n_arcs;
      for (uint32_t l = n_levels; l-- > 0;) {
        const uint32_t a_lo = GXR_LEV(l);
        if (a_lo != a_hi) {
          const uint32_t q_lo = a_lo >> 6, q_hi = (a_hi - 1) >> 6;
          if (!ext) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              if ((uint32_t)q >= q_lo && (uint32_t)q <= q_hi) {
                const uint32_t a = (uint32_t)q * 64u + lane;
                if (a >= a_lo && a < a_hi) {
                  const uint32_t d = cds[q];
                  const double term = gwq[q] * bv[d & 0xffffu];
                  shq[q] = term;
                  if (term != 0.0) gx_lds_add(&bv[d >> 16], term);
                }
              }
            GX_WAVE_SYNC();
          } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              if ((uint32_t)q >= q_lo && (uint32_t)q <= q_hi) {
                const uint32_t a = (uint32_t)q * 64u + lane;
                if (a >= a_lo && a < a_hi) {
                  const uint32_t d = cds[q];
                  if (gwq[q] * bv[d & 0xffffu] != 0.0) gx_lds_max(&emx[d >> 16], be[d & 0xffffu]);
                }
              }
            GX_WAVE_SYNC();
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              if ((uint32_t)q >= q_lo && (uint32_t)q <= q_hi) {
                const uint32_t a = (uint32_t)q * 64u + lane;
                if (a >= a_lo && a < a_hi) {
                  const uint32_t d = cds[q];
                  const double term = ldexp(gwq[q] * bv[d & 0xffffu], max(be[d & 0xffffu] - emx[d >> 16], -1100));
                  shq[q] = term;
                  if (term != 0.0) gx_lds_add(&bsum[d >> 16], term);
                }
              }
            GX_WAVE_SYNC();
            // (the states of the level: the sources of its arcs -- from the first arc's to the last one's)
            const uint32_t s_lo = A.level_off[B.level_base + l], s_hi = A.level_off[B.level_base + l + 1];
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
    double* const tot = ext ? bsum : bv;
    // the parallel sweep's next counts set out: the snapshot's, whatever this block samples
    double pcq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      pcq[q] = gwq[q];
      const uint32_t a = (uint32_t)q * 64u + lane;
      if (a < B.n_arcs) sh[a] = shq[q];
    }
    if (PAR) GXR_REQUEST_COUNTS()
    GX_WAVE_SYNC();
    unsigned long long t2 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- every state chooses its way on (random.ipp:111-127: the shares in list order, newest first, subtracted from
    // u x total until it drops below zero), then the path is followed from the start: one word per step ----
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const uint32_t s = (uint32_t)j * 64u + lane;
      if (s < B.n_states) {
        const uint32_t a0 = ooff[s], a1 = ooff[s + 1];
        uint32_t pick = a0;
        if (a1 > a0) {
          double choice = us[j] * tot[s];
          bool done = false;
          for (uint32_t top = a1; top > a0 && !done; top -= min(4u, top - a0)) {
            // (four shares requested together; the subtractions in the reference's order)
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = top >= a0 + 1u + (uint32_t)i ? sh[top - 1u - (uint32_t)i] : 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (!done && top >= a0 + 1u + (uint32_t)i) {
                choice -= v[i];
                pick = top - 1u - (uint32_t)i;
                done = choice < 0;
              }
          }
        }
        be[s] = (int)(pick | ((a1 > a0 ? (uint32_t)dst16[pick] : B.fin) << 16));
      }
    }
    GX_WAVE_SYNC();
    GXR_STAMP(td)
    uint32_t n_ids = 0;
    unsigned long long te_ = 0;
    {
      uint32_t s = B.start, n_path = 0;
      while (s != B.fin && n_path < CM) {
        const uint32_t wd = (uint32_t)be[s];
        if (lane == 0) idn[n_path] = wd & 0xffffu;
        ++n_path;
        s = wd >> 16;
      }
      GX_WAVE_SYNC();
      te_ = A.phase_clk ? __builtin_readcyclecounter() : 0;
      for (uint32_t base = 0; base < n_path; base += 64) {
        const uint32_t t = base + lane;
        const bool have = t < n_path;
        const uint32_t a = have ? idn[t] : 0u;
        const uint32_t from = a & 63u, qa = a >> 6;
        uint32_t p0 = GX_NONE, p1 = GX_NONE, wlo = 0, whi = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {  // (every lane takes part in every exchange; each keeps its own chunk's answer)
          const uint32_t z = gxr_bperm(cz[q], from), w = gxr_bperm(cw[q], from);
          const uint32_t lo = gxr_bperm((uint32_t)__double2loint(pcq[q]), from), hi = gxr_bperm((uint32_t)__double2hiint(pcq[q]), from);
          if (qa == (uint32_t)q) {
            p0 = z;
            p1 = w;
            wlo = lo;
            whi = hi;
          }
        }
        if (!have) {
          p0 = GX_NONE;
          p1 = GX_NONE;
        }
        if (have) cheap.mul(__hiloint2double((int)whi, (int)wlo));
        const unsigned long long m0 = __ballot(p0 != GX_NONE), m1 = __ballot(p1 != GX_NONE);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        const uint32_t at = n_ids + (uint32_t)__popcll(m0 & below) + (uint32_t)__popcll(m1 & below);
        if (p0 != GX_NONE && at < CM) ids[at] = p0;
        if (p1 != GX_NONE && at + (p0 != GX_NONE) < CM) ids[at + (p0 != GX_NONE)] = p1;
        n_ids += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
      }
      n_ids = min(n_ids, CM);
    }
    GX_WAVE_SYNC();
    unsigned long long t3 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    // ---- the new sample into the counts, the next block's previous sample out (as in the kernel above) ----
    for (uint32_t k = lane; k < n_ids; k += 64) {
      const uint32_t p = ids[k], n = A.p_norm[p];
      idn[k] = n;
      A.sample_ids[B.sample_off + k] = p;
      A.sample_nrm[B.sample_off + k] = n;
      if (PAR) continue;
      if (n != GX_NONE) {
        gx_add(A.p_x + p, B.wt);
        gx_add(A.normsum + n, B.wt);
        if (A.p_touch) A.p_touch[p] = A.time;
        cnum.mul(gx_add(A.ccount + p, 1.0));
        cden.mul(gx_add(A.csum + n, 1.0));
      } else
        cnum.mul(A.p_prior[p]);
    }
    if (lane == 0) A.sample_len[b] = n_ids;
    if (!PAR && !A.want_after && bi + 1 < nb) gx_take_out(A, Bn, lane, On);
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
      sub[0] += ta - t0;
      sub[1] += tb - ta;
      sub[2] += tc - tb;
      sub[3] += t1 - tc;
      sub[4] += td - t2;
      sub[5] += te_ - td;
      sub[6] += t3 - te_;
    }
  }
#undef GXR_LEV
#undef GXR_REQUEST_RECORDS
#undef GXR_REQUEST_COUNTS
  if (PAR) {
    double cl = cheap.ln();
    for (int o = 32; o > 0; o >>= 1) cl += __shfl_down(cl, o, 64);
    if (lane == 0) {
      unsafeAtomicAdd(A.iter_out + 1, cl);
      if (A.phase_clk) {
        for (int k = 0; k < 5; ++k) atomicAdd(A.phase_clk + k, clk[k]);
        for (int k = 0; k < 7; ++k) atomicAdd(A.phase_clk + 8 + k, sub[k]);
      }
    }
    return;
  }
  gx_order();
  if (A.want_after)
    for (uint32_t k = lane; k < n_prev; k += 64) {
      const uint32_t p = ids[k], n = idn[k];
      after.mul(n == GX_NONE ? A.p_prior[p] : gx_ld(A.p_x + p) / gx_ld(A.normsum + n));
    }
  double cache_ln = cnum.ln() - cden.ln(), after_ln = after.ln(), cheap_ln = cheap.ln();
  for (int o = 32; o > 0; o >>= 1) {
    cache_ln += __shfl_down(cache_ln, o, 64);
    after_ln += __shfl_down(after_ln, o, 64);
    cheap_ln += __shfl_down(cheap_ln, o, 64);
  }
  if (lane == 0) {
    A.iter_out[0] = cache_ln;
    A.iter_out[1] = cheap_ln;
    A.iter_out[2] = after_ln;
    if (A.phase_clk) {
      for (int k = 0; k < 5; ++k) A.phase_clk[k] += clk[k];
      for (int k = 0; k < 7; ++k) A.phase_clk[8 + k] += sub[k];
    }
  }
}
size_t gibbs_reg_lds_bytes(uint32_t cap_arcs, uint32_t cap_states, uint32_t cap_sample, uint32_t own_slots) {
  return (size_t)cap_arcs * 8 + (size_t)cap_states * (2 * 8 + 2 * 4) + ((size_t)cap_states + 1) * 4 + (size_t)cap_sample * 8 +
         (size_t)own_slots * 16 + ((size_t)cap_arcs * 2 + 15) / 16 * 16;
}
template <bool PAR, int NQ>
static hipError_t launch_reg(const GxArgs& A, uint32_t grid, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)gibbs_reg_wave_kernel<PAR, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((gibbs_reg_wave_kernel<PAR, NQ>), dim3(grid), dim3(64), lds, s, A);
  return hipGetLastError();
}
hipError_t launch_gibbs_reg_wave(const GxArgs& A, uint32_t n_waves, int nq, hipStream_t s) {
  const size_t lds = gibbs_reg_lds_bytes(A.cap_arcs, A.cap_states, A.cap_sample, n_waves ? A.own_slots : 0u);
  if (n_waves) {
    switch (nq) {
      case 1: return launch_reg<true, 1>(A, n_waves, lds, s);
      case 2: return launch_reg<true, 2>(A, n_waves, lds, s);
      case 4: return launch_reg<true, 4>(A, n_waves, lds, s);
      case 8: return launch_reg<true, 8>(A, n_waves, lds, s);
    }
    return hipErrorInvalidValue;
  }
  const uint32_t grid = A.n_chains > 1 ? A.n_chains : 1u;
  switch (nq) {
    case 2: return launch_reg<false, 2>(A, grid, lds, s);
    case 4: return launch_reg<false, 4>(A, grid, lds, s);
    case 8: return launch_reg<false, 8>(A, grid, lds, s);
    case 16: return launch_reg<false, 16>(A, grid, lds, s);
  }
  return hipErrorInvalidValue;
}

// ---- the parallel sweep's recount: counts := prior + weighted uses in the new samples (the caller has set them to the priors).
// A thread per sample entry; a workgroup's entries meet in two open-addressing tables in LDS (parameter -> weight, norm group ->
// weight) and reach global memory once per workgroup and key: the popular parameters of a model (a tagger's "the" / DT) are
// used by most blocks, and 10^5 device-scope adds to one address serialise.  Keys that find no slot within eight probes are
// added directly.  (gibbs.hip's recount walks a block's sample with one thread, id by id: 8 ms of a 13 ms sweep on the tagging
// cascade x 100.)
__device__ __forceinline__ void gx_tab_add(uint32_t* keys, double* vals, uint32_t mask, uint32_t key, double v, double* global) {
  uint32_t h = (key * 2654435761u) >> 10;
  for (int probe = 0; probe < 8; ++probe, ++h) {
    const uint32_t at = h & mask;
    const uint32_t old = atomicCAS(keys + at, 0xffffffffu, key);
    if (old == 0xffffffffu || old == key) {
      gx_lds_add(vals + at, v);
      return;
    }
  }
  unsafeAtomicAdd(global + key, v);
}
__global__ __launch_bounds__(1024) void gibbs_recount_tables_kernel(const GxBlock* blocks, const uint32_t* len, const uint32_t* ids,
                                                                    const uint32_t* nrm, uint32_t n_blocks, double* new_x,
                                                                    double* new_norm, uint32_t p_slots, uint32_t n_slots,
                                                                    const uint32_t* list) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_lds[];
  double* pv = (double*)rc_lds;
  double* nv = pv + p_slots;
  uint32_t* pk = (uint32_t*)(nv + n_slots);
  uint32_t* nk = pk + p_slots;
  for (uint32_t i = threadIdx.x; i < p_slots; i += blockDim.x) {
    pk[i] = 0xffffffffu;
    pv[i] = 0.0;
  }
  for (uint32_t i = threadIdx.x; i < n_slots; i += blockDim.x) {
    nk[i] = 0xffffffffu;
    nv[i] = 0.0;
  }
  __syncthreads();
  // a wavefront per block at a time: its entries side by side
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
  for (uint32_t bi = blockIdx.x * nw + wv; bi < n_blocks; bi += gridDim.x * nw) {
    const uint32_t b = list ? list[bi] : bi;  // (list: the blocks to count, n_blocks of them)
    const GxBlock B = blocks[b];
    const uint32_t n = len[b];
    for (uint32_t k = lane; k < n; k += 64) {
      const uint32_t g = nrm[B.sample_off + k];
      if (g == GX_NONE) continue;
      gx_tab_add(pk, pv, p_slots - 1, ids[B.sample_off + k], B.wt, new_x);
      if (new_norm) gx_tab_add(nk, nv, n_slots - 1, g, B.wt, new_norm);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < p_slots; i += blockDim.x)
    if (pk[i] != 0xffffffffu) unsafeAtomicAdd(new_x + pk[i], pv[i]);
  for (uint32_t i = threadIdx.x; i < n_slots && new_norm; i += blockDim.x)
    if (nk[i] != 0xffffffffu) unsafeAtomicAdd(new_norm + nk[i], nv[i]);
}
// the norm sums of the parallel sweep from the counts, group by group: sum over a group's counted members of their new counts
// (prior + uses), a workgroup per group, in a fixed order.  A use used to add to its norm group as well as to its parameter: half
// of the recount's adds, and the hottest ones -- a tagger has 45 norm groups for five million uses a sweep.
__global__ __launch_bounds__(256) void gibbs_normsum_kernel(const double* x, const uint32_t* p_norm, const uint64_t* group_off, const uint64_t* norm_perm,
                                                            uint64_t n_groups, double* normsum) {
  __shared__ double part[4];
  for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    double v = 0.0;
    for (uint64_t j = group_off[g] + threadIdx.x; j < group_off[g + 1]; j += 256) {
      const uint64_t p = norm_perm[j];
      if (p_norm[p] != GX_NONE) v += x[p];
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) normsum[g] = (part[0] + part[1]) + (part[2] + part[3]);
  }
}
hipError_t launch_gibbs_normsum(const double* x, const uint32_t* p_norm, const uint64_t* group_off, const uint64_t* norm_perm, uint64_t n_groups,
                                double* normsum, hipStream_t s) {
  if (!n_groups) return hipSuccess;
  hipLaunchKernelGGL(gibbs_normsum_kernel, dim3((unsigned)std::min<uint64_t>(n_groups, 16384)), dim3(256), 0, s, x, p_norm, group_off, norm_perm,
                     n_groups, normsum);
  return hipGetLastError();
}
hipError_t launch_gibbs_recount_tables(const GxBlock* blocks, const uint32_t* len, const uint32_t* ids, const uint32_t* nrm,
                                       uint32_t n_blocks, double* new_x, double* new_norm, hipStream_t s, const uint32_t* list) {
  const uint32_t p_slots = 8192, n_slots = 1024;
  const size_t lds = (size_t)(p_slots + n_slots) * 12;
  (void)hipFuncSetAttribute((const void*)gibbs_recount_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t grid = std::min<uint32_t>(256u, (n_blocks + 15) / 16);
  hipLaunchKernelGGL(gibbs_recount_tables_kernel, dim3(grid ? grid : 1u), dim3(1024), lds, s, blocks, len, ids, nrm, n_blocks, new_x, new_norm,
                     p_slots, n_slots, list);
  return hipGetLastError();
}

size_t gibbs_exact_lds_bytes(uint32_t cap_arcs, uint32_t cap_states, uint32_t cap_levels, uint32_t cap_sample) {
  return (size_t)cap_arcs * (3 * 8 + 3 * 4) + (size_t)cap_states * (2 * 8 + 2 * 4) + ((size_t)cap_states + 1 + cap_levels + 1) * 4 +
         (size_t)cap_sample * 8 + (size_t)4 * GX_OWN * 4 + (size_t)cap_arcs * 8 + (size_t)cap_states * 4 + (size_t)cap_sample * 4;
}
hipError_t launch_gibbs_exact_wave(const GxArgs& A, uint32_t n_waves, hipStream_t s) {
  const size_t lds = gibbs_exact_lds_bytes(A.cap_arcs, A.cap_states, A.cap_levels, A.cap_sample);
  if (n_waves) {  // the parallel sweep
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)gibbs_exact_wave_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gibbs_exact_wave_kernel<true>, dim3(n_waves), dim3(64), lds, s, A);
  } else {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)gibbs_exact_wave_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gibbs_exact_wave_kernel<false>, dim3(A.n_chains > 1 ? A.n_chains : 1u), dim3(64), lds, s, A);
  }
  return hipGetLastError();
}
__global__ void gibbs_broadcast_kernel(double* dst, const double* src, uint64_t n, uint32_t copies) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
    const double v = src[k];
    for (uint32_t c = 0; c < copies; ++c) dst[(size_t)c * n + k] = v;
  }
}
hipError_t launch_gibbs_broadcast(double* dst, const double* src, uint64_t n, uint32_t copies, hipStream_t s) {
  if (!n || !copies) return hipSuccess;
  hipLaunchKernelGGL(gibbs_broadcast_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, dst, src, n, copies);
  return hipGetLastError();
}

}  // namespace carmel_hip
