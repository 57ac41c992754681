// gibbs.hip — blocked Gibbs sampling of derivations (`carmel --crp`) on the GPU.
//
// Replaces carmel_gibbs + gibbs_base (/root/reference/carmel/src/gibbs.cc:15-41, 114-186, 306-371;
// graehl/shared/gibbs.hpp:106-227, 589-638, 769-877) and derivations::random_path (derivations.h:318-375).
//
// One block = one training pair = one cached derivation lattice.  Per block and sweep: take the block's previous
// sample out of the CRP counts, give every lattice arc the proposal weight prod_{p in chain} count[p]/normsum[p],
// sweep backward (beta), walk start->goal choosing an out-arc with probability proportional to (w * beta[dest])^power,
// record the chosen arcs' parameters, put them back into the counts.
//
// Two schedules:
//   mode 0 (exact):    blocks are resampled strictly one after another inside ONE workgroup — the reference's
//                      Markov chain; given the same uniforms it reproduces the reference's samples.
//   mode 1 (parallel): every workgroup resamples its blocks against the counts of the previous sweep with its OWN
//                      previous sample taken out (counterfactual counts, looked up through a small LDS table) and
//                      the counts are rebuilt after the sweep.  This is a different (approximate, "stale count")
//                      chain — SURVEY.md section 8e — and is never the default.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include "engine.hpp"
#include "options.hpp"
#include "forest_exact.hpp"
#include "gibbs_exact.hpp"
#include "gibbs_lane.hpp"
#include "rng.hpp"

namespace carmel_hip {

#define G_NEG_INF (-__builtin_huge_val())
#define G_NONORM 0xffffffffu

struct GibbsArgs {
  const BundleDesc* bundles;
  const uint32_t* block_bundle;  // block -> bundle index
  const uint2* out_arcs;
  const uint32_t* out_off;
  const uint2* in_arcs;          // cyclic lattices only: the reversed graph's lists in the reference's order (lattice.cpp)
  const uint32_t* in_off;
  const uint32_t* level_off;
  const uint32_t* pair_start;
  const uint32_t* pair_final;
  const double* pair_logw;
  // composed arc -> chain of parameter ids
  const uint64_t* chain_off;   // per composed arc (already resolved through the chain id)
  const uint32_t* chain_param;
  // parameters
  const uint32_t* p_norm;      // G_NONORM = fixed probability (locked arc / NONE member): prob = p_prior
  const double* p_prior;
  double* p_x;                 // current count          (delta_sum::x)
  double* p_s;                 // time-integrated count  (delta_sum::s)
  double* p_tmax;              //                        (delta_sum::tmax)
  double* normsum;
  double* ccount;              // cache model (gibbs.hpp:678-742)
  double* csum;
  const double* snap_x;        // mode 1: counts / normsums frozen at the start of the sweep
  const double* snap_norm;
  // samples
  const uint64_t* sample_off;  // capacity offsets per block
  uint32_t* sample_len;
  uint32_t* sample_ids;
  uint32_t* new_len;           // mode 1: next sweep's samples are written beside the current ones
  uint32_t* new_ids;
  // scratch
  double* gw;                  // per lattice arc (bundle out_base + a): ln proposal weight
  double* beta;                // per lattice state (bundle off_base + s)
  double* alpha;               // --expectation: forward weights, like beta
  double* ewt;                 // --expectation: per lattice arc, its posterior in the block's current "sample"
  int expectation;
  int include_self;            // --include-self (gibbs.hpp:851-870): the block's previous counts stay in while its proposal is
                               // formed and leave just before the new ones go in
  int randomize;               // --expectation, initial sweep of a --random-start run or of a restart (gibbs.hpp:816, 860-864,
                               // 296-301): every (arc, chain element) entry's weight is scaled by its own uniform, the
                               // block's probability counts as 0
  uint32_t* old_ids;           // --include-self, sampling: the previous sample of the block being resampled (exact mode)
  const uint32_t* ent_base;    // --expectation: per lattice arc (bundle out-arc order), the number of its first entry in the
                               // reference's listing of the block's fractional counts
  const double* init_logw;     // --init-em: per composed arc, the weight the first sweep of the first run samples from
  int par_books;               // exact mode: count bookkeeping of a block by the whole workgroup (g_addc_all / g_block_probs)
  uint32_t books_cap;          // ids of one sample the workgroup's LDS scratch holds
  uint32_t stage_arcs, stage_states;  // exact mode: a block's lattice at most this large is staged in LDS (0: never)
  double* iter_out;            // [0] ln cache-model prob of the sweep, [1] ln proposal ("cheap") prob, [2] see want_after
  int want_after;              // exact mode: also the proposal prob of every block evaluated AFTER its new sample was
                               // added back (the "overestimate" of the comment at gibbs.hpp:866)
  uint32_t* overflow;          // set when a walk through a cyclic lattice outgrew the block's sample buffer
  uint64_t seed;
  uint32_t n_blocks, iter;
  double time, power;
};

__device__ __forceinline__ double g_lwadd(double a, double b) {  // weight.h:765-801
  if (a == G_NEG_INF) return b;
  if (b == G_NEG_INF) return a;
  double d = a - b;
  if (d > 36.0) return a;
  if (d < -36.0) return b;
  if (d < 0) return b + log1p(exp(d));
  return a + log1p(exp(-d));
}

// gibbs_param::addc (gibbs.hpp:210-217) + delta_sum::add_delta (delta_sum.hpp:74-84)
__device__ __forceinline__ void g_addc(const GibbsArgs& G, uint32_t p, double d) {
  const uint32_t n = G.p_norm[p];
  if (n == G_NONORM) return;
  G.normsum[n] += d;
  const double moret = G.time - G.p_tmax[p];
  if (moret > 0) {
    G.p_tmax[p] = G.time;
    G.p_s[p] += moret * G.p_x[p];
  } else if (moret < 0)
    G.p_s[p] += d * (-moret);
  G.p_x[p] += d;
}

// proposal probability of parameter p (gibbs.hpp:153-157).  own_x / own_n: counts to take out first (mode 1).
template <bool SNAP>
__device__ __forceinline__ double g_prob(const GibbsArgs& G, uint32_t p, const uint32_t* own_ids, uint32_t own_len,
                                         double own_wt) {
  const uint32_t n = G.p_norm[p];
  if (n == G_NONORM) return G.p_prior[p];
  if (!SNAP) return G.p_x[p] / G.normsum[n];
  double x = G.snap_x[p], ns = G.snap_norm[n];
  for (uint32_t k = 0; k < own_len; ++k) {  // counterfactual: this block's previous sample does not count
    const uint32_t q = own_ids[k];
    if (q == p) x -= own_wt;
    if (G.p_norm[q] == n) ns -= own_wt;
  }
  return x / ns;
}

// gibbs.hpp:769-792 for a whole sample at once, by every thread of the workgroup.  One thread doing it id by id
// spends three dependent memory round trips per id -- that was three quarters of the exact sweep's time.  The result is
// the one the sequential loop gives: the first touch of a parameter at this time folds (time - tmax) * x into the
// running sum (one thread per parameter wins an atomicMax on the time stamp, while x is still untouched); the adds of
// +-wt commute (equal addends).  Times never decrease within a run, so the moret < 0 branch cannot occur.
__device__ __forceinline__ void g_addc_all(const GibbsArgs& G, const uint32_t* ids, uint32_t n, double d) {
  const unsigned long long tb = (unsigned long long)__double_as_longlong(G.time);  // time >= 0: bit order = value order
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t p = ids[k];
    if (G.p_norm[p] == G_NONORM) continue;
    const unsigned long long old = atomicMax((unsigned long long*)(G.p_tmax + p), tb);
    if (old < tb) G.p_s[p] += (G.time - __longlong_as_double((long long)old)) * G.p_x[p];
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t p = ids[k], nn = G.p_norm[p];
    if (nn == G_NONORM) continue;
    unsafeAtomicAdd(G.p_x + p, d);
    unsafeAtomicAdd(G.normsum + nn, d);
  }
  __syncthreads();
}
// proposal ("cheap") and cache-model probability of a new sample (gibbs.hpp:712-742), by the whole workgroup: the
// k-th id sees the cache counts of its parameter and of its norm group raised by the earlier ids of the same sample
// (their number comes from a scan of the ids staged in LDS).  scratch: 2 * n uint32.  Sums in a fixed tree order.
__device__ __forceinline__ void g_block_probs(const GibbsArgs& G, const uint32_t* ids, uint32_t n, uint32_t* scratch, double* red) {
  uint32_t* sid = scratch;
  uint32_t* snn = scratch + n;
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    sid[k] = ids[k];
    snn[k] = G.p_norm[ids[k]];
  }
  __syncthreads();
  double cheap = 0.0, cache = 0.0;
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t p = sid[k], nn = snn[k];
    if (nn == G_NONORM) {
      const double q = G.p_prior[p];
      cheap += log(q);
      cache += log(q);
      continue;
    }
    uint32_t m = 0, M = 0;
    for (uint32_t j = 0; j < k; ++j) {
      m += sid[j] == p;
      M += snn[j] == nn;
    }
    cheap += log(G.p_x[p] / G.normsum[nn]);
    cache += log((G.ccount[p] + (double)m) / (G.csum[nn] + (double)M));
  }
  __syncthreads();  // every cache count has been read
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    if (snn[k] == G_NONORM) continue;
    unsafeAtomicAdd(G.ccount + sid[k], 1.0);
    unsafeAtomicAdd(G.csum + snn[k], 1.0);
  }
  for (int o = 32; o > 0; o >>= 1) {
    cheap += __shfl_down(cheap, o, 64);
    cache += __shfl_down(cache, o, 64);
  }
  __shared__ double part[2][4];
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = cheap;
    part[1][threadIdx.x >> 6] = cache;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    red[0] += (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
    red[1] += (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
  }
  __syncthreads();
}

// the proposal probability of a block's sample with the sample itself counted (whole workgroup; red[2])
__device__ __forceinline__ void g_block_after(const GibbsArgs& G, const uint32_t* ids, uint32_t n, double* red) {
  double v = 0.0;
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t p = ids[k], nn = G.p_norm[p];
    v += log(nn == G_NONORM ? G.p_prior[p] : G.p_x[p] / G.normsum[nn]);
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) red[2] += (part[0] + part[1]) + (part[2] + part[3]);
  __syncthreads();
}

// Resample one block.  Called by every thread of the workgroup; serial parts run on thread 0.
// own_ids (LDS, mode 1 only): the block's previous sample.
// --expectation: the block's previous fractional counts out of the counts (gibbs.hpp:851-852 / 869-870 over block_delta::wt):
// entry by entry in the order they went in; a block whose entries were scaled by the uniforms of the previous sweep
// (sample_len 2: the randomised initial sweep) takes the same factors again
__device__ void g_take_out_expected(const GibbsArgs& G, const BundleDesc& d, const uint2* oa, const uint32_t* ooff, bool cyc,
                                    uint32_t b, double wt) {
  const uint32_t n = G.sample_len[b];
  if (!n) return;
  const double* ewt = G.ewt + d.out_base;
  const uint32_t* eb = G.ent_base + d.out_base;
  for (uint32_t s = 0; s < d.n_states; ++s)
    for (uint32_t i = 0, deg = ooff[s + 1] - ooff[s]; i < deg; ++i) {
      const uint32_t a = cyc ? ooff[s] + i : ooff[s + 1] - 1 - i;  // list order
      const uint32_t arc = oa[a].y;
      uint32_t k = eb[a];
      for (uint64_t j = G.chain_off[arc]; j < G.chain_off[arc + 1]; ++j, ++k) {
        const double f = n == 2u ? gibbs_uniform(G.seed, G.iter - 1u, b, k) : 1.0;
        g_addc(G, G.chain_param[j], (ewt[a] * f) * -wt);
      }
    }
}
template <bool SNAP>
__device__ void g_resample_block(const GibbsArgs& G, uint32_t b, uint32_t* own_ids, uint32_t own_cap, double* red) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const BundleDesc d = G.bundles[G.block_bundle[b]];
  const uint2* oa = G.out_arcs + d.out_base;
  const uint32_t* ooff = G.out_off + d.off_base;
  const uint32_t* lvl = G.level_off + d.level_base;
  double* gw = G.gw + d.out_base;
  double* beta = G.beta + d.off_base;
  // A lattice with a cycle (derivations.h:726-728) is stored in the reference's own sweep order -- states in its forward
  // order, out lists newest first, reversed lists as it walks them (lattice.cpp) -- and is swept by one thread in that
  // order, so that what random_path sees (back edges' partial sums included) is what the reference sees.
  const bool cyc = (d.flags & 1u) != 0;
  if (!SNAP && !cyc && G.stage_arcs && d.n_arcs <= G.stage_arcs && d.n_states <= G.stage_states) {
    // exact mode: the backward sweep (a barrier per level) and the walk (one thread) are chains of dependent reads of
    // this block's lattice -- from LDS they cost a tenth of what they cost from L2
    double* sgw = (double*)(own_ids + 2 * (size_t)G.books_cap);
    double* sbeta = sgw + G.stage_arcs;
    uint2* soa = (uint2*)(sbeta + G.stage_states);
    uint32_t* soff = (uint32_t*)(soa + G.stage_arcs);
    uint32_t* slvl = soff + G.stage_states + 1;
    for (uint32_t a = tid; a < (uint32_t)d.n_arcs; a += NT) soa[a] = oa[a];
    for (uint32_t k = tid; k <= d.n_states; k += NT) soff[k] = ooff[k];
    for (uint32_t k = tid; k <= d.n_levels; k += NT) slvl[k] = lvl[k];
    __syncthreads();
    oa = soa;
    ooff = soff;
    lvl = slvl;
    gw = sgw;
    beta = sbeta;
  }
  const double wt = exp(G.pair_logw[d.pair_base]);
  const uint64_t so = G.sample_off[b];
  uint32_t* ids = G.sample_ids + so;
  uint32_t own_len = 0, old_n = 0;
  const uint32_t* own = own_ids;
  if (SNAP) {
    own_len = G.include_self ? 0u : G.sample_len[b];  // (--include-self: no counterfactual subtraction of the own uses)
    // the LDS copy holds GIBBS_OWN_CAP ids at most (a cyclic lattice's sample may revisit states: its capacity is
    // 32 * states * chain); a longer previous sample is read where it lies -- the sweep writes new_ids, not sample_ids
    if (own_len > own_cap)
      own = ids;
    else
      for (uint32_t k = tid; k < own_len; k += NT) own_ids[k] = ids[k];
    __syncthreads();
  } else {
    // 1. take the previous sample out of the counts (gibbs.hpp:851-852) -- unless --include-self: then it is set aside
    // (sampling: copied; --expectation: the arcs' previous posteriors stay where they are) and leaves in step 6
    if (G.include_self) {
      if (!G.expectation) {
        old_n = G.sample_len[b];
        for (uint32_t k = tid; k < old_n; k += NT) G.old_ids[k] = ids[k];
      }
    } else if (G.par_books && !G.expectation) {
      g_addc_all(G, ids, G.sample_len[b], -wt);
    } else if (tid == 0) {
      if (G.expectation)
        g_take_out_expected(G, d, oa, ooff, cyc, b, wt);
      else
        for (uint32_t k = 0, n = G.sample_len[b]; k < n; ++k) g_addc(G, ids[k], -wt);
    }
    __syncthreads();
  }
  // 2. proposal weight of every lattice arc (gibbs.cc:348-359)
  for (uint32_t a = tid; a < (uint32_t)d.n_arcs; a += NT) {
    const uint32_t arc = oa[a].y;
    double w = 0.0;
    if (G.init_logw)  // gibbs.cc:316-318 p_init
      w = G.init_logw[arc];
    else
      for (uint64_t j = G.chain_off[arc]; j < G.chain_off[arc + 1]; ++j)
        w += log(g_prob<SNAP>(G, G.chain_param[j], own, own_len, wt));
    gw[a] = w;
  }
  for (uint32_t s = tid; s < d.n_states; s += NT) beta[s] = G_NEG_INF;
  __syncthreads();
  if (tid == 0) beta[G.pair_final[d.pair_base]] = 0.0;
  __syncthreads();
  // 3. backward sweep: level-synchronous; a cyclic lattice in the reference's order (graph.h:391-402 over the reversed graph)
  if (cyc) {
    if (tid == 0) {
      const uint2* ia = G.in_arcs + d.in_base;
      const uint32_t* ioff = G.in_off + d.off_base;
      for (uint32_t s = d.n_states; s-- > 0;) {
        const double bs = beta[s];
        if (bs == G_NEG_INF) continue;
        for (uint32_t a = ioff[s]; a < ioff[s + 1]; ++a) {
          const uint2 r = ia[a];  // {source state, composed arc}
          double w = 0.0;
          if (G.init_logw)
            w = G.init_logw[r.y];
          else
            for (uint64_t j = G.chain_off[r.y]; j < G.chain_off[r.y + 1]; ++j)
              w += log(g_prob<SNAP>(G, G.chain_param[j], own, own_len, wt));
          beta[r.x] = g_lwadd(beta[r.x], bs + w);
        }
      }
    }
    __syncthreads();
  } else
  for (uint32_t l = d.n_levels; l-- > 0;) {
    for (uint32_t s = lvl[l] + tid; s < lvl[l + 1]; s += NT) {
      const uint32_t a0 = ooff[s], a1 = ooff[s + 1];
      if (a0 == a1) continue;
      double acc = G_NEG_INF;
      for (uint32_t a = a1; a-- > a0;) acc = g_lwadd(acc, gw[a] + beta[oa[a].x]);  // list order = newest first
      beta[s] = acc;
    }
    __syncthreads();
  }
  if (!SNAP && G.expectation) {
    // --expectation (derivations.h:381-398, gibbs.cc:367-371, gibbs.hpp:783-792): forward sweep, posterior of every
    // lattice arc, fractional counts in for the chain of every arc (the previous ones went out in step 1)
    double* alpha = G.alpha + d.off_base;
    double* ewt = G.ewt + d.out_base;
    for (uint32_t s = tid; s < d.n_states; s += NT) alpha[s] = G_NEG_INF;
    __syncthreads();
    if (tid == 0) {
      alpha[G.pair_start[d.pair_base]] = 0.0;
      for (uint32_t s = 0; s < d.n_states; ++s) {  // states are numbered by level: sources before destinations
        const double as = alpha[s];
        if (as == G_NEG_INF) continue;
        for (uint32_t i = 0, deg = ooff[s + 1] - ooff[s]; i < deg; ++i) {
          const uint32_t a = cyc ? ooff[s] + i : ooff[s + 1] - 1 - i;
          alpha[oa[a].x] = g_lwadd(alpha[oa[a].x], as + gw[a]);
        }
      }
    }
    __syncthreads();
    const double prob = alpha[G.pair_final[d.pair_base]];
    if (G.include_self) {  // the previous posteriors leave now (gibbs.hpp:869-870), before they are overwritten
      if (tid == 0) g_take_out_expected(G, d, oa, ooff, cyc, b, wt);
      __syncthreads();
    }
    for (uint32_t s = tid; s < d.n_states; s += NT)
      for (uint32_t a = ooff[s]; a < ooff[s + 1]; ++a) ewt[a] = exp(gw[a] + alpha[s] + beta[oa[a].x] - prob);
    __syncthreads();
    if (tid == 0) {
      const uint32_t* eb = G.ent_base + d.out_base;  // entry numbers: (arc, chain element) in the order the reference lists them
      for (uint32_t s = 0; s < d.n_states; ++s)
        for (uint32_t i = 0, deg = ooff[s + 1] - ooff[s]; i < deg; ++i) {
          const uint32_t a = cyc ? ooff[s] + i : ooff[s + 1] - 1 - i;
          const uint32_t arc = oa[a].y;
          uint32_t k = eb[a];
          for (uint64_t j = G.chain_off[arc]; j < G.chain_off[arc + 1]; ++j, ++k) {
            const double f = G.randomize ? gibbs_uniform(G.seed, G.iter, b, k) : 1.0;
            g_addc(G, G.chain_param[j], (ewt[a] * f) * wt);
          }
        }
      G.sample_len[b] = G.randomize ? 2u : 1u;  // "has counts in" (2: scaled entry by entry by this sweep's uniforms)
      const double bp = G.randomize ? G_NEG_INF : prob;  // (gibbs.hpp:863: bd.prob = 0)
      red[0] += bp;
      red[1] += bp;
    }
    __syncthreads();
    return;
  }
  // 4. walk start -> goal (derivations.h:361-374; random.ipp:111-127), 5. probabilities, 6. put the new sample in
  if (tid == 0) {
    uint32_t* out_ids = SNAP ? (G.new_ids + so) : ids;
    const uint32_t cap_ids = (uint32_t)(G.sample_off[b + 1] - so);
    uint32_t n = 0, step = 0;
    uint32_t s = G.pair_start[d.pair_base];
    const uint32_t fin = G.pair_final[d.pair_base];
    while (s != fin) {
      const uint32_t a0 = ooff[s], a1 = ooff[s + 1];
      // at temperature 1 the state's normaliser is its own beta: the backward sweep folded the same terms in the
      // same order
      const uint32_t deg = a1 - a0;
#define G_LIST(i) (cyc ? a0 + (i) : a1 - 1 - (i)) /* the i-th arc of the state's list (newest first) */
      double sum = beta[s];
      if (G.power != 1.0 || cyc) {  // (a cyclic lattice's beta is the reference's scatter, not this sum)
        sum = G_NEG_INF;
        for (uint32_t i = 0; i < deg; ++i) sum = g_lwadd(sum, (gw[G_LIST(i)] + beta[oa[G_LIST(i)].x]) * G.power);
      }
      if (sum == G_NEG_INF) sum = 0.0;
      // each arc's probability once (kept in LDS for the usual small out-degree), used for the total and the choice
      __shared__ double pe[32];
      const bool keep = a1 - a0 <= 32;
      double tot = 0.0;
      for (uint32_t i = 0; i < deg; ++i) {
        const uint32_t a = G_LIST(i);
        const double e = exp((gw[a] + beta[oa[a].x]) * G.power - sum);
        if (keep) pe[a - a0] = e;
        tot += e;
      }
      double choice = tot * gibbs_uniform(G.seed, G.iter, b, step++);
      uint32_t pick = a0;
      for (uint32_t i = 0; i < deg; ++i) {
        const uint32_t a = G_LIST(i);
        choice -= keep ? pe[a - a0] : exp((gw[a] + beta[oa[a].x]) * G.power - sum);
        pick = a;
        if (choice < 0) break;
      }
#undef G_LIST
      const uint32_t arc = oa[pick].y;
      for (uint64_t j = G.chain_off[arc]; j < G.chain_off[arc + 1]; ++j) {
        if (n < cap_ids) out_ids[n] = G.chain_param[j];
        ++n;
      }
      if (n > cap_ids) {  // (only a lattice with a cycle can get here: the sample buffer is full, the run is void)
        *G.overflow = 1u;
        n = cap_ids;
        break;
      }
      s = oa[pick].x;
    }
    if (!SNAP && G.par_books && n <= G.books_cap) {  // probabilities and counts by the whole workgroup, below
      G.sample_len[b] = n;
    } else {
    double cheap = 0.0, cache = 0.0;
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t p = out_ids[k];
      cheap += log(g_prob<SNAP>(G, p, own, own_len, wt));
      if (!SNAP) {  // cache model: counts restart from the priors every sweep and grow by one per use
        const uint32_t nn = G.p_norm[p];
        double q = G.p_prior[p];
        if (nn != G_NONORM) {
          q = G.ccount[p] / G.csum[nn];
          G.ccount[p] += 1.0;
          G.csum[nn] += 1.0;
        }
        cache += log(q);
      }
    }
    if (SNAP) {
      G.new_len[b] = n;
      red[0] += cheap;
    } else {
      G.sample_len[b] = G.par_books ? (n | 0x80000000u) : n;  // flag for the other threads: already booked
      for (uint32_t k = 0; k < old_n; ++k) g_addc(G, G.old_ids[k], -wt);  // (--include-self: gibbs.hpp:869-870)
      for (uint32_t k = 0; k < n; ++k) g_addc(G, out_ids[k], wt);
      red[0] += cheap;
      red[1] += cache;
    }
    }
  }
  __syncthreads();
  if (!SNAP && G.par_books) {
    const uint32_t n = G.sample_len[b];
    __syncthreads();
    if (n & 0x80000000u) {
      if (tid == 0) G.sample_len[b] = n & 0x7fffffffu;
    } else {
      g_block_probs(G, ids, n, own_ids, red);
      if (old_n) g_addc_all(G, G.old_ids, old_n, -wt);  // (--include-self)
      g_addc_all(G, ids, n, wt);
    }
    __syncthreads();
  }
  if (!SNAP && G.want_after && !G.expectation) {
    __threadfence_block();
    __syncthreads();
    g_block_after(G, ids, G.sample_len[b] & 0x7fffffffu, red);
  }
}

// mode 0: the whole sweep in one workgroup, blocks strictly in order
__global__ __launch_bounds__(256) void gibbs_sweep_exact_kernel(GibbsArgs G) {
  extern __shared__ uint32_t books[];  // 2 * G.books_cap ids: scratch of g_block_probs
  __shared__ double red[3];
  if (threadIdx.x == 0) red[0] = red[1] = red[2] = 0.0;
  __syncthreads();
  for (uint32_t b = 0; b < G.n_blocks; ++b) g_resample_block<false>(G, b, books, 0, red);
  if (threadIdx.x == 0) {
    G.iter_out[0] = red[1];
    G.iter_out[1] = red[0];
    G.iter_out[2] = red[2];
  }
}

// mode 1: blocks spread over the grid, counterfactual counts from the snapshot
#define GIBBS_OWN_CAP 8192u
__global__ __launch_bounds__(64) void gibbs_sweep_parallel_kernel(GibbsArgs G, uint32_t own_cap) {
  extern __shared__ uint32_t own_ids[];
  __shared__ double red[3];
  if (threadIdx.x == 0) red[0] = red[1] = red[2] = 0.0;
  __syncthreads();
  for (uint32_t b = blockIdx.x; b < G.n_blocks; b += gridDim.x) g_resample_block<true>(G, b, own_ids, own_cap, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(G.iter_out + 1, red[0]);
}
// mode 1, after the sweep: counts := prior + weighted uses in the new samples (atomics; sample ids are scattered)
__global__ void gibbs_recount_kernel(GibbsArgs G, double* new_x, double* new_norm) {
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < G.n_blocks; b += gridDim.x * blockDim.x) {
    const BundleDesc d = G.bundles[G.block_bundle[b]];
    const double wt = exp(G.pair_logw[d.pair_base]);
    const uint32_t* ids = G.new_ids + G.sample_off[b];
    const uint32_t n = G.new_len[b];
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t p = ids[k], nn = G.p_norm[p];
      if (nn == G_NONORM) continue;
      unsafeAtomicAdd(new_x + p, wt);
      unsafeAtomicAdd(new_norm + nn, wt);
    }
  }
}
// mode 1: fold the sweep's count change into the time-weighted sums (delta_sum::add_delta with d = new - old)
__global__ void gibbs_commit_kernel(GibbsArgs G, const double* new_x, uint64_t n_params) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_params; p += (uint64_t)gridDim.x * blockDim.x) {
    if (G.p_norm[p] == G_NONORM) continue;
    const double d = new_x[p] - G.p_x[p];
    const double moret = G.time - G.p_tmax[p];
    if (moret > 0) {
      G.p_tmax[p] = G.time;
      G.p_s[p] += moret * G.p_x[p];
    } else if (moret < 0)
      G.p_s[p] += d * (-moret);
    G.p_x[p] += d;
  }
}

// ---- prior-scale inference (gibbs.hpp:404-553) ----
// gibbs_param::scale_prior (gibbs.hpp:161-176) for every parameter, one thread per norm group walking its members in
// ascending order (the reference's order within a group; groups do not interact): prior *= f, the count and its
// time-weighted sum move with it (delta_sum::addbase), the group's normsum too; psum = the new prior mass of the group
__global__ void gibbs_scale_priors_kernel(GibbsArgs G, const uint64_t* group_off, const uint64_t* norm_perm, uint64_t n_groups,
                                          const uint32_t* meta, const double* scales, int invert, double* p_prior,
                                          double* prior_norm) {
  const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_groups) return;
  const uint32_t i = meta[n];
  double ns = G.normsum[n], ps = 0.0;
  for (uint64_t j = group_off[n]; j < group_off[n + 1]; ++j) {
    const uint64_t p = norm_perm[j];
    if (G.p_norm[p] == G_NONORM) continue;  // a locked member of the group: a fixed probability, no prior
    double pr = p_prior[p];
    if (i > 0) {
      double f = scales[i];
      if (invert) f = 1. / f;
      const double sc = f * pr, d = sc - pr;
      G.p_s[p] += d * G.p_tmax[p];
      G.p_x[p] += d;
      ns += d;
      pr = sc;
      p_prior[p] = pr;
    }
    ps += pr;
  }
  G.normsum[n] = ns;
  prior_norm[n] = ps;
}
// gibbs_base::cache_prob(recompute) (gibbs.hpp:712-742): the cache-model probability of the WHOLE current sample, block
// after block, from cache counts that start at the priors (ccount / csum were reset by the caller); iter_out[3]
__global__ __launch_bounds__(256) void gibbs_cache_prob_kernel(GibbsArgs G) {
  extern __shared__ uint32_t books[];
  __shared__ double red[3];
  if (threadIdx.x == 0) red[0] = red[1] = red[2] = 0.0;
  __syncthreads();
  for (uint32_t b = 0; b < G.n_blocks; ++b) {
    const uint32_t* ids = G.sample_ids + G.sample_off[b];
    const uint32_t n = G.sample_len[b];
    if (n <= G.books_cap) {
      g_block_probs(G, ids, n, books, red);
    } else {
      if (threadIdx.x == 0) {
        double cache = 0.0;
        for (uint32_t k = 0; k < n; ++k) {
          const uint32_t p = ids[k], nn = G.p_norm[p];
          double q = G.p_prior[p];
          if (nn != G_NONORM) {
            q = G.ccount[p] / G.csum[nn];
            G.ccount[p] += 1.0;
            G.csum[nn] += 1.0;
          }
          cache += log(q);
        }
        red[1] += cache;
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) G.iter_out[3] = red[1];
}

}  // namespace carmel_hip

using namespace carmel_hip;

// the standard normal's cdf and quantile (the reference uses boost::math::normal_distribution, gibbs.hpp:474-516 -- a
// third-party dependency absent from the tree; restated from the published algorithm: Wichura, AS 241 (PPND16), 1988)
double gibbs_norm_cdf(double z) { return 0.5 * std::erfc(-z / std::sqrt(2.0)); }
double gibbs_norm_quantile(double p) {
  const double q = p - 0.5;
  if (std::fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    return q * (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r + 4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r + 1.3314166789178437745e2) * r + 3.3871328727963666080e0) /
           (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r + 2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r + 4.2313330701600911252e1) * r + 1.0);
  }
  double r = q < 0 ? p : 1.0 - p;
  r = std::sqrt(-std::log(r));
  double v;
  if (r <= 5.0) {
    r -= 1.6;
    v = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r + 1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r + 4.63033784615654529590e0) * r + 1.42343711074968357734e0) /
        (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r + 2.05319162663775882187e0) * r + 1.0);
  } else {
    r -= 5.0;
    v = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r + 5.46378491116411436990e0) * r + 6.65790464350110377720e0) /
        (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r + 5.99832206555887937690e-1) * r + 1.0);
  }
  return q < 0 ? -v : v;
}

static constexpr int GX_WCLASSES = 8;  // launch classes of the parallel sweep's blocks: four of the register kernel (by chunks of arcs), four by LDS need
struct carmel_hip_gibbs {
  // --prior-inference-* (gibbs_opts.hpp:82-89, 148-153)
  double pi_stddev = 0;
  bool pi_restart_fresh = false;
  uint32_t pi_start = 0, pi_end = 0, n_scale = 0;
  std::vector<uint32_t> h_meta;       // metanorm: scale group of every norm group, 0 = never scaled (gibbs.hpp:404-470)
  DevBuf<uint32_t> d_meta, old_ids, ent_base;
  uint32_t obs_every = 0;  // carmel_hip_gibbs_set_observer
  carmel_hip_gibbs_observer_fn obs_fn = nullptr;
  void* obs_ctx = nullptr;
  DevBuf<double> d_scales;
  std::vector<double> cumulative;     // product of the accepted scales per scale group
  std::vector<double> h_prior0;       // the priors before any inference (--prior-inference-restart-fresh)
  std::vector<double> pi_trace;       // per sweep: {proposed, accepted, ln p1, ln p2, a2, p_accept}
  carmel_hip_trainer* t = nullptr;
  int device = 0;
  carmel_hip_gibbs_opts opt;
  LatticeSet lat;
  uint64_t n_params = 0, n_norm = 0;
  uint32_t n_blocks = 0, max_sample = 0;
  std::vector<uint32_t> h_norm;
  std::vector<double> h_prior;
  DevBuf<BundleDesc> bundles;
  DevBuf<uint2_t> out_arcs, in_arcs;
  DevBuf<uint32_t> in_off;
  DevBuf<uint32_t> out_off, level_off, pair_start, pair_final, block_bundle, chain_param, p_norm, sample_len, sample_ids,
      new_len, new_ids;
  DevBuf<uint64_t> chain_off, sample_off;
  DevBuf<double> alpha, ewt, init_logw;
  DevBuf<double> p_touch;  // (see carmel_hip_gibbs_run: the "last@t" of the count table)
  DevBuf<double> pair_logw, p_prior, p_x, p_s, p_tmax, normsum, prior_norm, ccount, csum, snap_x, snap_norm, gw, beta,
      iter_out;
  std::vector<uint64_t> h_sample_off;
  // the wavefront path of the exact chain (gibbs_exact.hip): per-block descriptors, per-lattice-arc records
  bool wave_ok = false;
  DevBuf<GxBlock> gx_blocks;
  DevBuf<uint32_t> gx_rec, gx_nrm, sample_nrm, new_nrm;
  DevBuf<uint16_t> gx_state_lev;
  DevBuf<uint32_t> gx_lev_arc;  // per level: its first arc (beside level_off)
  int reg_nq = 0;               // the exact chain on gibbs_reg_wave_kernel<false, reg_nq> (0: some block is not eligible)
  uint32_t cap_arcs = 0, cap_states = 0, cap_levels = 0, cap_sample = 0;
  // parallel sweep: two launch classes by LDS need (the blocks up to the 90th percentile of arcs; the rest)
  struct WaveClass {
    DevBuf<uint32_t> list;
    uint32_t n = 0, cap_arcs = 0, cap_states = 0, cap_levels = 0, cap_sample = 0;
    int nq = 0;  // > 0: gibbs_reg_wave_kernel<true, nq>
  } wclass[GX_WCLASSES], wrest[GX_WCLASSES];  // wrest: the same classes without the blocks that go one per lane
  // the parallel sweep with one block per lane (gibbs_lane.hip): trellis blocks within its LDS budget
  DevBuf<GlGroup> gl_groups;
  DevBuf<GlLane> gl_lanes;
  DevBuf<uint32_t> gl_recA, gl_recB, gl_arc, gl_samp[2], rest_list;
  DevBuf<double> gl_sw;
  std::vector<GlClass> gl_classes;
  uint32_t gl_ngroups = 0, n_rest = 0;
  int gl_cur = 0;  // gl_samp[gl_cur]: the paths of the last sweep
  bool ran = false;
  uint32_t best_run = 0;  // --crp-restarts: the run whose counts and sample were kept
  std::vector<double> h_final_x;  // ... its counts as finalize_cumulative_counts left them (carmel_hip_gibbs_final_counts)
  // runs as replicas (carmel_hip_gibbs_set_run_share): this sampler takes the runs r with r % run_stride == run_first
  uint32_t run_first = 0, run_stride = 1;
  bool ran_any = false;
  double best_stats[3] = {0, 0, 0};  // gibbs_stats of the kept run: allprob, finalprob, sumprob (ln)
};

extern "C" {

double carmel_hip_gibbs_uniform(uint64_t seed, uint32_t iter, uint32_t block, uint32_t step) {
  return gibbs_uniform(seed, iter, block, step);
}

double carmel_hip_gibbs_power(double high_temp, double low_temp, uint32_t iter, uint32_t sweep) {
  return gibbs_anneal_power(high_temp, low_temp, iter, sweep);
}
int carmel_hip_gibbs_create(carmel_hip_gibbs** out, carmel_hip_trainer* t, const carmel_hip_gibbs_opts* o) {
  if (!out || !t || !o) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (!t->have_norm) return fail(CARMEL_HIP_ERR_STATE, "set_norm / set_cascade first");
  if (!t->have_corpus) return fail(CARMEL_HIP_ERR_STATE, "set_corpus first");
  if (o->expectation && o->mode != 0)
    return fail(CARMEL_HIP_ERR_UNSUPPORTED,
                "--expectation is the sequential online EM of the reference (mode 0); its parallel counterpart is plain EM "
                "with --priors (carmel_hip_estimate / carmel_hip_maximize)");
  HIPCHK(hipSetDevice(t->device));
  std::unique_ptr<carmel_hip_gibbs> g(new carmel_hip_gibbs());
  g->t = t;
  g->device = t->device;
  g->opt = *o;
  hipStream_t s = t->stream;
  // ---- parameters: norm group and prior pseudo-count (gibbs.cc:114-186; gibbs.hpp:589-592) ----
  const uint64_t np = t->np();
  g->n_params = np;
  std::vector<double> lw(np);
  HIPCHK(hipMemcpyAsync(lw.data(), t->params(), np * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const std::vector<uint32_t>& pg = t->cascade ? t->h_param_group : t->w.group;
  const uint64_t ng = t->h_group_add.size();
  g->n_norm = ng;
  std::vector<double> gsum(ng, 0.0), gcnt(ng, 0.0);
  for (uint64_t p = 0; p < np; ++p) {
    uint32_t n = t->h_norm_of[p];
    if (n == 0xffffffffu || pg[p] == CARMEL_HIP_LOCKED_GROUP) continue;
    gsum[n] += std::exp(lw[p]);
    gcnt[n] += 1.0;
  }
  g->h_norm.assign(np, 0xffffffffu);
  g->h_prior.assign(np, 0.0);
  for (uint64_t p = 0; p < np; ++p) {
    uint32_t n = t->h_norm_of[p];
    if (n == 0xffffffffu || pg[p] == CARMEL_HIP_LOCKED_GROUP) {  // fixed probability = the arc's weight
      g->h_prior[p] = std::exp(lw[p]);
      continue;
    }
    double alpha = t->h_group_add[n];
    if (!(alpha > 0)) alpha = o->min_prior > 0 ? o->min_prior : 1e-2;  // gibbs.cc:390-397
    double sum = o->dirichlet_p0 ? 1.0 : gsum[n];
    g->h_norm[p] = n;
    g->h_prior[p] = o->uniform_p0 ? alpha : alpha * (std::exp(lw[p]) / sum) * gcnt[n];
  }
  // ---- composed arc -> parameter ids ----
  std::vector<uint64_t> coff(t->w.n_arcs + 1, 0);
  std::vector<uint32_t> cpar;
  uint32_t max_chain = 1;
  for (uint64_t a = 0; a < t->w.n_arcs; ++a) {
    if (t->cascade) {
      uint32_t c = t->w.group[a];
      for (uint64_t j = t->h_chain_off[c]; j < t->h_chain_off[c + 1]; ++j) cpar.push_back((uint32_t)t->h_chain_param[j]);
      max_chain = std::max<uint32_t>(max_chain, (uint32_t)(t->h_chain_off[c + 1] - t->h_chain_off[c]));
    } else
      cpar.push_back((uint32_t)a);
    coff[a + 1] = cpar.size();
  }
  // ---- one lattice per block, in corpus order ----
  BuildOptions bo;
  bo.lane_states = 0;
  bo.small_pairs = 1;
  bo.wave = false;  // the sampler walks bundles (one lattice each)
  bo.keep_state_ids = o->expectation != 0;
  bo.threads = 0;
  std::string err;
  if (!build_lattices(t->w, t->corpus, bo, g->lat, err)) return fail(CARMEL_HIP_ERR_ARG, err);
  LatticeSet& L = g->lat;
  std::vector<uint32_t> bundle_of_pair(t->corpus.n_pairs, 0xffffffffu);
  for (size_t b = 0; b < L.bundles.size(); ++b) bundle_of_pair[L.pair_id[L.bundles[b].pair_base]] = (uint32_t)b;
  std::vector<uint32_t> bb;
  for (uint64_t p = 0; p < t->corpus.n_pairs; ++p)
    if (bundle_of_pair[p] != 0xffffffffu) bb.push_back(bundle_of_pair[p]);
  g->n_blocks = (uint32_t)bb.size();
  if (!g->n_blocks) return fail(CARMEL_HIP_ERR_NO_DERIV, "No training example had a derivation - aborting training.");
  g->h_sample_off.assign(bb.size() + 1, 0);
  for (size_t b = 0; b < bb.size(); ++b) {
    // an acyclic lattice's path has at most n_levels - 1 arcs.  A lattice with a cycle (derivations.h:726-728) has no bound: the
    // walk may take a loop any number of times (the reference grows a vector).  Room for a walk 32 times the lattice's states
    // (at least 256 arcs); a longer one ends the run with an error instead of writing past the buffer (G.overflow)
    const BundleDesc& bd = L.bundles[bb[b]];
    uint32_t cap = ((bd.flags & 1u) ? std::max<uint32_t>(256u, 32u * bd.n_states) : bd.n_levels) * max_chain;
    g->max_sample = std::max(g->max_sample, cap);
    g->h_sample_off[b + 1] = g->h_sample_off[b] + cap;
  }
  HIPCHK(g->bundles.upload(L.bundles, s));
  HIPCHK(g->out_arcs.upload(L.out_arcs, s));
  HIPCHK(g->out_off.upload(L.out_off, s));
  if (L.n_cyclic) {
    HIPCHK(g->in_arcs.upload(L.in_arcs, s));
    HIPCHK(g->in_off.upload(L.in_off, s));
  }
  HIPCHK(g->level_off.upload(L.level_off, s));
  HIPCHK(g->pair_start.upload(L.pair_start, s));
  HIPCHK(g->pair_final.upload(L.pair_final, s));
  HIPCHK(g->pair_logw.upload(L.pair_logw, s));
  HIPCHK(g->block_bundle.upload(bb, s));
  HIPCHK(g->chain_off.upload(coff, s));
  HIPCHK(g->chain_param.upload(cpar, s));
  HIPCHK(g->p_norm.upload(g->h_norm, s));
  HIPCHK(g->p_prior.upload(g->h_prior, s));
  HIPCHK(g->p_x.alloc(np));
  HIPCHK(g->p_s.alloc(np));
  HIPCHK(g->p_tmax.alloc(np));
  HIPCHK(g->normsum.alloc(ng));
  HIPCHK(g->ccount.alloc(np));
  HIPCHK(g->csum.alloc(ng));
  std::vector<double> pn(ng, 0.0);
  for (uint64_t p = 0; p < np; ++p)
    if (g->h_norm[p] != 0xffffffffu) pn[g->h_norm[p]] += g->h_prior[p];
  HIPCHK(g->prior_norm.upload(pn, s));
  HIPCHK(g->sample_off.upload(g->h_sample_off, s));
  HIPCHK(g->sample_len.alloc(bb.size()));
  HIPCHK(g->sample_ids.alloc(g->h_sample_off.back() + 128));  // (+ gibbs_exact.hip's staging reads a fixed number of words ahead)
  if (o->mode == 1) {
    HIPCHK(g->new_len.alloc(bb.size()));
    HIPCHK(g->new_ids.alloc(g->h_sample_off.back() + 128));
    HIPCHK(g->snap_x.alloc(np));
    HIPCHK(g->snap_norm.alloc(ng));
  }
  HIPCHK(g->gw.alloc(L.out_arcs.size()));
  HIPCHK(g->beta.alloc(L.out_off.size()));
  if (o->expectation) {
    HIPCHK(g->alpha.alloc(L.out_off.size()));
    HIPCHK(g->ewt.alloc(L.out_arcs.size()));
    // --random-start / restarts (gibbs.hpp:296-301): every ENTRY of a block's fractional counts -- one per lattice arc and chain
    // element, listed as collect_counts_gibbs lists them (derivations.h:381-398: the lattice's states in their own numbering,
    // each state's arcs in list order) -- is scaled by its own uniform: the entry number of every out-arc's first element
    std::vector<uint32_t> eb(L.out_arcs.size(), 0u), inv;
    for (size_t b = 0; b < bb.size(); ++b) {
      const BundleDesc& d = L.bundles[bb[b]];
      const uint32_t* ooff = L.out_off.data() + d.off_base;
      const uint32_t* orig = L.state_orig.data() + d.off_base;
      const bool cyc = (d.flags & 1u) != 0;
      inv.assign(d.n_states, 0u);
      for (uint32_t s = 0; s < d.n_states; ++s) inv[orig[s]] = s;
      uint32_t k = 0;
      for (uint32_t q = 0; q < d.n_states; ++q) {
        const uint32_t s = inv[q];
        for (uint32_t i = 0, deg = ooff[s + 1] - ooff[s]; i < deg; ++i) {
          const uint32_t a = cyc ? ooff[s] + i : ooff[s + 1] - 1 - i;  // list order
          eb[d.out_base + a] = k;
          const uint32_t arc = L.out_arcs[d.out_base + a].y;
          k += (uint32_t)(coff[arc + 1] - coff[arc]);
        }
      }
    }
    HIPCHK(g->ent_base.upload(eb, s));
  }
  // ---- the exact chain's wavefront path (gibbs_exact.hip): eligible when every block is an acyclic lattice within its LDS
  // tables and every composed arc stands for at most two parameters ----
  if (!o->expectation && max_chain <= 2) {
    bool ok = true;
    for (size_t b = 0; b < bb.size() && ok; ++b) {
      const BundleDesc& d = L.bundles[bb[b]];
      ok = !(d.flags & 1u) && d.n_arcs <= GX_ARCS && d.n_states <= GX_STATES && d.n_levels <= GX_LEVELS &&
           (uint64_t)d.n_levels * max_chain <= GX_SAMPLE && d.out_base + d.n_arcs <= 0xffffffffull && d.off_base <= 0xffffffffull;
      if (ok) {  // (the goal has no way on: its backward value is the 1 it starts with)
        const uint32_t fin = L.pair_final[d.pair_base];
        ok = L.out_off[d.off_base + fin + 1] == L.out_off[d.off_base + fin];
      }
    }
    if (ok) {
      std::vector<GxBlock> gb(bb.size());
      std::vector<uint32_t> rec(4 * L.out_arcs.size(), 0xffffffffu), nrm(2 * L.out_arcs.size(), 0xffffffffu);
      std::vector<uint16_t> slev(L.out_off.size(), 0);
      std::vector<uint32_t> lev_arc(L.level_off.size(), 0u);
      std::vector<int> need_nq(bb.size(), 0);  // the register kernel's chunks for this block (0: not eligible)
      for (size_t b = 0; b < bb.size(); ++b) {
        const BundleDesc& d = L.bundles[bb[b]];
        GxBlock& B = gb[b];
        B.out_base = (uint32_t)d.out_base;
        B.off_base = (uint32_t)d.off_base;
        B.level_base = d.level_base;
        B.n_arcs = (uint32_t)d.n_arcs;
        B.n_states = d.n_states;
        B.n_levels = d.n_levels;
        B.start = L.pair_start[d.pair_base];
        B.fin = L.pair_final[d.pair_base];
        B.sample_off = g->h_sample_off[b];
        B.wt = std::exp(L.pair_logw[d.pair_base]);
        const uint32_t* ooff = L.out_off.data() + d.off_base;
        const uint32_t* lo = L.level_off.data() + d.level_base;
        for (uint32_t l = 0; l < d.n_levels; ++l)
          for (uint32_t st = lo[l]; st < lo[l + 1]; ++st) slev[d.off_base + st] = (uint16_t)l;
        bool trellis = slev[d.off_base + B.start] == 0;
        for (uint32_t st = 0; st < d.n_states; ++st)
          for (uint32_t a = ooff[st]; a < ooff[st + 1]; ++a) {
            const uint2_t oa = L.out_arcs[d.out_base + a];
            if (slev[d.off_base + oa.x] != slev[d.off_base + st] + 1) trellis = false;
            uint32_t* r = &rec[4 * (d.out_base + a)];
            uint32_t* n = &nrm[2 * (d.out_base + a)];
            r[0] = oa.x | (st << 16);
            r[1] = oa.y;
            const uint64_t c0 = coff[oa.y], c1 = coff[oa.y + 1];
            for (uint64_t j = c0; j < c1; ++j) {
              r[2 + (j - c0)] = cpar[j];
              n[j - c0] = g->h_norm[cpar[j]];
            }
          }
        if (trellis) B.n_levels |= 0x80000000u;
        for (uint32_t l = 0; l <= d.n_levels; ++l) lev_arc[d.level_base + l] = ooff[lo[l]];
        if (trellis && d.n_levels <= GX_REG_LEVELS) {
          const uint32_t m = std::max<uint32_t>((uint32_t)d.n_arcs, d.n_states);
          need_nq[b] = m <= 64 ? 1 : m <= 128 ? 2 : m <= 256 ? 4 : m <= 512 ? 8 : m <= 1024 ? 16 : 0;  // (16: the exact chain only)
        }
      }
      HIPCHK(g->gx_lev_arc.upload(lev_arc, s));
      HIPCHK(g->gx_state_lev.upload(slev, s));
      HIPCHK(g->gx_blocks.upload(gb, s));
      HIPCHK(g->gx_rec.upload(rec, s));
      HIPCHK(g->gx_nrm.upload(nrm, s));
      HIPCHK(g->sample_nrm.alloc(g->h_sample_off.back() + 128));
      if (o->mode == 1) HIPCHK(g->new_nrm.alloc(g->h_sample_off.back() + 128));
      for (size_t b = 0; b < bb.size(); ++b) {
        const BundleDesc& d = L.bundles[bb[b]];
        g->cap_arcs = std::max(g->cap_arcs, (uint32_t)d.n_arcs);
        g->cap_states = std::max(g->cap_states, d.n_states);
        g->cap_levels = std::max(g->cap_levels, d.n_levels);
        g->cap_sample = std::max(g->cap_sample, d.n_levels * max_chain);
      }
      g->cap_arcs = (g->cap_arcs + 3) / 4 * 4;
      g->cap_states = (g->cap_states + 3) / 4 * 4;
      g->cap_sample = (g->cap_sample + 3) / 4 * 4;
      g->wave_ok = gibbs_exact_lds_bytes(g->cap_arcs, g->cap_states, g->cap_levels, g->cap_sample) <= 150 * 1024;
      // the register kernel (gibbs_exact.hip, round 6): the exact chain is one launch over all blocks in order, so every block
      // must be eligible; the parallel sweep takes the eligible ones class by class.  CARMEL_HIP_GIBBS_REG=0: the LDS kernel (A/B)
      const bool reg_on = !(lib_opt("gibbs_reg") && atoi(lib_opt("gibbs_reg")) == 0);
      g->reg_nq = 0;
      // (the exact chain: measured on the tagging cascade -- whose longest sentence needs 16 chunks -- the 16-chunk instance is
      // twice as SLOW as the LDS kernel, 65 k cycles a block against 36 k: issuing a block's 16 x 2 record loads and exchanging 16 x 4
      // registers for the path's parameters cost more than the level loop saves; so only when asked for: gibbs_reg = 1)
      if (g->wave_ok && reg_on && o->mode == 0 && lib_opt("gibbs_reg") && atoi(lib_opt("gibbs_reg")) == 1) {
        int nq = 2;
        for (size_t b = 0; b < bb.size() && nq; ++b) nq = need_nq[b] ? std::max(nq, need_nq[b]) : 0;
        g->reg_nq = nq;
        if (lib_opt("timing")) {
          size_t bad = 0;
          for (size_t b = 0; b < bb.size(); ++b) bad += need_nq[b] == 0;
          fprintf(stderr, "timing: gibbs exact chain: register kernel nq %d (%zu of %zu blocks not eligible; caps arcs %u states %u levels %u)\n", nq, bad,
                  bb.size(), g->cap_arcs, g->cap_states, g->cap_levels);
        }
      }
      if (g->wave_ok && o->mode == 1) {
        // a wavefront's LDS is sized by its launch's largest block: the few long sentences must not set the occupancy of all
        std::vector<uint32_t> arcs(bb.size());
        for (size_t b = 0; b < bb.size(); ++b) arcs[b] = (uint32_t)L.bundles[bb[b]].n_arcs;
        std::vector<uint32_t> sorted;  // (of the blocks the LDS kernel takes)
        for (size_t b = 0; b < bb.size(); ++b)
          if (!(reg_on && need_nq[b] && need_nq[b] <= 8)) sorted.push_back(arcs[b]);
        if (sorted.empty()) sorted.push_back(0u);
        std::sort(sorted.begin(), sorted.end());
        // (four classes at the median, the 80th and the 95th percentile: the tagging cascade's median sentence needs a third of
        // the 90th percentile's LDS, and a wavefront that waits for its gathers wants neighbours)
        const uint32_t cuts[3] = {sorted[std::min(sorted.size() - 1, sorted.size() / 2)],
                                  sorted[std::min(sorted.size() - 1, sorted.size() * 8 / 10)],
                                  sorted[std::min(sorted.size() - 1, sorted.size() * 19 / 20)]};
        std::vector<uint32_t> lists[GX_WCLASSES];
        for (size_t b = 0; b < bb.size(); ++b) {
          int c = 0;
          if (reg_on && need_nq[b] && need_nq[b] <= 8) {  // classes 0..3: the register kernel with 1, 2, 4, 8 chunks of arcs
            c = need_nq[b] == 1 ? 0 : need_nq[b] == 2 ? 1 : need_nq[b] == 4 ? 2 : 3;
            g->wclass[c].nq = need_nq[b];
          } else {
            while (c < 3 && arcs[b] > cuts[c]) ++c;
            c += 4;
          }
          const BundleDesc& d = L.bundles[bb[b]];
          auto& W = g->wclass[c];
          lists[c].push_back((uint32_t)b);
          W.cap_arcs = std::max(W.cap_arcs, (uint32_t)d.n_arcs);
          W.cap_states = std::max(W.cap_states, d.n_states);
          W.cap_levels = std::max(W.cap_levels, d.n_levels);
          W.cap_sample = std::max(W.cap_sample, d.n_levels * max_chain);
        }
        // one block per lane where the lattices allow it (gibbs_lane.hip); CARMEL_HIP_GIBBS_LANE=0: every block on the kernels above (A/B)
        GlHost H;
        if (!(lib_opt("gibbs_lane") && atoi(lib_opt("gibbs_lane")) == 0)) gibbs_lane_build(L, bb, gb, coff, cpar, g->h_norm, H);
        if (!H.groups.empty()) {
          g->gl_ngroups = (uint32_t)H.groups.size();
          g->gl_classes = H.classes;
          HIPCHK(g->gl_groups.upload(H.groups, s));
          HIPCHK(g->gl_lanes.upload(H.lanes, s));
          HIPCHK(g->gl_recA.upload(H.recA, s));
          HIPCHK(g->gl_recB.upload(H.recB, s));
          HIPCHK(g->gl_arc.upload(H.arc_id, s));
          HIPCHK(g->gl_sw.alloc(3 * H.n_rec));
          for (int k = 0; k < 2; ++k) {
            HIPCHK(g->gl_samp[k].alloc(8 * H.n_samp));
            HIPCHK(hipMemsetAsync(g->gl_samp[k].p, 0, g->gl_samp[k].bytes(), s));
          }
          std::vector<uint32_t> rest;
          std::vector<uint32_t> rlists[GX_WCLASSES];
          for (int c = 0; c < GX_WCLASSES; ++c) {
            g->wrest[c].nq = g->wclass[c].nq;
            for (uint32_t b : lists[c])
              if (!H.taken[b]) {
                rlists[c].push_back(b);
                rest.push_back(b);
                const BundleDesc& d = L.bundles[bb[b]];
                auto& W = g->wrest[c];
                W.cap_arcs = std::max(W.cap_arcs, ((uint32_t)d.n_arcs + 3) / 4 * 4);
                W.cap_states = std::max(W.cap_states, (d.n_states + 3) / 4 * 4);
                W.cap_levels = std::max(W.cap_levels, d.n_levels);
                W.cap_sample = std::max(W.cap_sample, (d.n_levels * max_chain + 3) / 4 * 4);
              }
            g->wrest[c].n = (uint32_t)rlists[c].size();
            if (g->wrest[c].n) HIPCHK(g->wrest[c].list.upload(rlists[c], s));
          }
          g->n_rest = (uint32_t)rest.size();
          if (g->n_rest) HIPCHK(g->rest_list.upload(rest, s));
          if (lib_opt("timing")) {
            size_t taken = 0;
            for (uint8_t t8 : H.taken) taken += t8;
            fprintf(stderr, "timing: gibbs parallel sweep: %zu of %zu blocks one per lane in %u groups, %zu launch classes (LDS per wavefront:", taken, bb.size(),
                    g->gl_ngroups, H.classes.size());
            for (auto& c : H.classes) fprintf(stderr, " %zu", gibbs_lane_lds_bytes(c.W, c.LP, c.LN));
            fprintf(stderr, " bytes)\n");
          }
        }
        for (int c = 0; c < GX_WCLASSES; ++c) {
          auto& W = g->wclass[c];
          W.n = (uint32_t)lists[c].size();
          W.cap_arcs = (W.cap_arcs + 3) / 4 * 4;
          W.cap_states = (W.cap_states + 3) / 4 * 4;
          W.cap_sample = (W.cap_sample + 3) / 4 * 4;
          if (W.n) HIPCHK(W.list.upload(lists[c], s));
          if (W.n && lib_opt("timing"))
            fprintf(stderr, "timing: gibbs parallel class %d: %u blocks, nq %d, caps arcs %u states %u levels %u sample %u\n", c, W.n, W.nq, W.cap_arcs,
                    W.cap_states, W.cap_levels, W.cap_sample);
        }
      }
    }
  }
  HIPCHK(g->iter_out.alloc(6));  // (+ the overflow flag of the walks, as a word of its own)
  HIPCHK(hipStreamSynchronize(s));
  *out = g.release();
  return CARMEL_HIP_OK;
}

int carmel_hip_gibbs_set_prior_inference(carmel_hip_gibbs* g, double stddev, int global, int local, int restart_fresh,
                                         uint32_t start, uint32_t end, const int* member_priorgroup,
                                         const uint32_t* member_n_states, uint32_t n_members) {
  if (!g) return fail(CARMEL_HIP_ERR_ARG, "null sampler");
  if (stddev > 0 && (g->opt.mode != 0 || g->opt.expectation))
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "prior inference works with the cache-model probability of the exact blocked sampler only (gibbs.hpp:528-529)");
  carmel_hip_trainer* t = g->t;
  HIPCHK(hipSetDevice(t->device));
  g->pi_stddev = stddev;
  g->pi_restart_fresh = restart_fresh != 0;
  g->pi_start = start;
  g->pi_end = end;
  const size_t ng = g->n_norm;
  // The reference's norm ids (gibbs.cc:114-186 over NormGroupIter, fst.h:1362-1445): members in order; a JOINT member
  // has one group PER STATE, arcs or not; a CONDITIONAL member one per (state, input) that has arcs.  Groups without
  // parameters own no counts, but each still draws a scale that enters the acceptance ratio, so they are counted here.
  std::vector<uint32_t> refid(ng, 0u), ref_member;  // product group -> reference norm id; reference id -> member
  {
    uint32_t members = 0;
    for (size_t n = 0; n < ng; ++n) members = std::max(members, t->h_group_member[n] + 1);
    members = std::max(members, n_members);
    std::vector<uint32_t> states(members, 0u);
    for (size_t n = 0; n < ng; ++n)
      if (t->h_group_joint[n]) states[t->h_group_member[n]] = std::max(states[t->h_group_member[n]], t->h_group_src[n] + 1);
    if (member_n_states)
      for (uint32_t m = 0; m < n_members; ++m) states[m] = std::max(states[m], member_n_states[m]);
    size_t n = 0;
    for (uint32_t m = 0; m < members; ++m) {
      const uint32_t base = (uint32_t)ref_member.size();
      bool joint = false;
      size_t e = n;
      for (; e < ng && t->h_group_member[e] == m; ++e) joint = joint || t->h_group_joint[e] != 0;
      if (joint) {
        for (size_t k = n; k < e; ++k) refid[k] = base + t->h_group_src[k];
        ref_member.insert(ref_member.end(), states[m], m);
      } else {
        for (size_t k = n; k < e; ++k) refid[k] = base + t->h_group_ref_rank[k];  // (the walk order of State::index: engine.cpp)
        ref_member.insert(ref_member.end(), e - n, m);
      }
      n = e;
    }
  }
  const size_t nref = ref_member.size();
  // metanorm as add_gibbs_params builds it: --prior-groupby 0 = fixed, 1 = one scale per transducer (the default),
  // 2 = one per norm group
  std::vector<uint32_t> meta(nref, 0u);
  uint32_t nexti = 1;
  for (size_t r = 0; r < nref;) {
    const uint32_t m = ref_member[r];
    const int pg = (member_priorgroup && m < n_members) ? member_priorgroup[m] : 1;
    for (; r < nref && ref_member[r] == m; ++r) {
      meta[r] = pg == 0 ? 0u : nexti;
      if (pg == 2) ++nexti;
    }
    if (pg == 1) ++nexti;
  }
  // finish_params cuts the table back to nnorm (gibbs.hpp:572-579), and nnorm only counts up to the last norm group that
  // owns a parameter (define_param, gibbs.hpp:582-588)
  size_t nnorm = 0;
  for (uint32_t n : g->h_norm)
    if (n != 0xffffffffu) nnorm = std::max(nnorm, (size_t)refid[n] + 1);
  if (global) {  // gibbs.hpp:575-578
    nexti = 2;
    std::fill(meta.begin(), meta.end(), 1u);
  }
  if (local) {
    nexti = (uint32_t)nnorm + 1;
    for (size_t r = 0; r < nref; ++r) meta[r] = r < nnorm ? (uint32_t)r + 1 : 0u;
  }
  g->h_meta.assign(ng, 0u);
  for (size_t n = 0; n < ng; ++n) g->h_meta[n] = meta[refid[n]];
  g->n_scale = nexti - 1;
  g->cumulative.assign(g->n_scale, 1.0);
  g->h_prior0 = g->h_prior;
  HIPCHK(g->d_meta.upload(g->h_meta, t->stream));
  HIPCHK(g->d_scales.alloc(nexti));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}
int carmel_hip_gibbs_prior_trace(carmel_hip_gibbs* g, double* out6, uint32_t n_sweeps, double* cumulative, uint32_t n_cumulative) {
  if (!g) return fail(CARMEL_HIP_ERR_ARG, "null sampler");
  if (out6)
    for (size_t k = 0; k < (size_t)n_sweeps * 6; ++k) out6[k] = k < g->pi_trace.size() ? g->pi_trace[k] : 0.0;
  if (cumulative)
    for (uint32_t k = 0; k < n_cumulative; ++k) cumulative[k] = k < g->cumulative.size() ? g->cumulative[k] : 1.0;
  return CARMEL_HIP_OK;
}
uint32_t carmel_hip_gibbs_n_prior_scales(carmel_hip_gibbs* g) { return g ? g->n_scale : 0; }

int carmel_hip_gibbs_destroy(carmel_hip_gibbs* g) {
  if (g) {
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();  // the trainer (and its stream) may already be gone
    delete g;
  }
  return CARMEL_HIP_OK;
}

uint32_t carmel_hip_gibbs_n_blocks(carmel_hip_gibbs* g) { return g ? g->n_blocks : 0; }
int carmel_hip_gibbs_lattice_stats(carmel_hip_gibbs* g, carmel_hip_lattice_stats* st) {
  if (!g || !st) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  const LatticeSet& L = g->lat;
  std::memset(st, 0, sizeof *st);
  st->n_pairs = g->t->corpus.n_pairs;
  st->n_pairs_kept = L.n_kept;
  st->explored_states = L.explored_states;
  st->explored_arcs = L.explored_arcs;
  st->kept_states = L.total_states;
  st->kept_arcs = L.total_arcs;
  st->n_cyclic_pairs = L.n_cyclic;
  st->n_bundles = L.bundles.size();
  st->max_levels = L.max_levels;
  st->last_pair_explored_states = L.last_pre_states;
  st->last_pair_kept_states = L.last_post_states;
  st->last_pair_kept_arcs = L.last_post_arcs;
  st->n_windowed_pairs = 0;
  return CARMEL_HIP_OK;
}

// gibbs_base::run (gibbs.hpp:803-828): restore_p0, the initial sample (iteration 0), then iter = 1..Ni with
// time = max(0, iter - burnin); finally finalize_cumulative_counts (gibbs.hpp:626-638).
int carmel_hip_gibbs_run(carmel_hip_gibbs* g, double* iter_logprob, double* iter_cheap_logprob) {
  return carmel_hip_gibbs_run_ex(g, iter_logprob, iter_cheap_logprob, nullptr);
}
int carmel_hip_gibbs_run_ex(carmel_hip_gibbs* g, double* iter_logprob, double* iter_cheap_logprob, double* iter_after_logprob) {
  if (!g) return fail(CARMEL_HIP_ERR_ARG, "null sampler");
  if (iter_after_logprob && (g->opt.mode != 0 || g->opt.expectation))
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "the after-add-back sample probability exists in the exact sampling mode only");
  carmel_hip_trainer* t = g->t;
  HIPCHK(hipSetDevice(t->device));
  hipStream_t s = t->stream;
  const uint64_t np = g->n_params, ng = g->n_norm;
  GibbsArgs G;
  std::memset(&G, 0, sizeof G);
  G.bundles = g->bundles.p;
  G.block_bundle = g->block_bundle.p;
  G.out_arcs = (const uint2*)g->out_arcs.p;
  G.out_off = g->out_off.p;
  G.in_arcs = (const uint2*)g->in_arcs.p;
  G.in_off = g->in_off.p;
  G.level_off = g->level_off.p;
  G.pair_start = g->pair_start.p;
  G.pair_final = g->pair_final.p;
  G.pair_logw = g->pair_logw.p;
  G.chain_off = g->chain_off.p;
  G.chain_param = g->chain_param.p;
  G.p_norm = g->p_norm.p;
  G.p_prior = g->p_prior.p;
  G.p_x = g->p_x.p;
  G.p_s = g->p_s.p;
  G.p_tmax = g->p_tmax.p;
  G.normsum = g->normsum.p;
  G.ccount = g->ccount.p;
  G.csum = g->csum.p;
  G.snap_x = g->snap_x.p;
  G.snap_norm = g->snap_norm.p;
  G.sample_off = g->sample_off.p;
  G.sample_len = g->sample_len.p;
  G.sample_ids = g->sample_ids.p;
  G.new_len = g->new_len.p;
  G.new_ids = g->new_ids.p;
  G.gw = g->gw.p;
  G.beta = g->beta.p;
  G.alpha = g->alpha.p;
  G.ewt = g->ewt.p;
  G.expectation = g->opt.expectation;
  G.include_self = g->opt.include_self ? 1 : 0;
  G.randomize = 0;
  G.old_ids = nullptr;
  G.ent_base = g->ent_base.p;
  if (G.include_self && !G.expectation && g->opt.mode == 0) {  // the sample a block is resampled against, set aside
    if (!g->old_ids.n) HIPCHK(g->old_ids.alloc(std::max<uint32_t>(g->max_sample, 1u)));
    G.old_ids = g->old_ids.p;
  }
  G.par_books = 1;  // (0: one thread books a block's counts id by id -- the A/B reference)
  G.books_cap = std::min<uint32_t>(g->max_sample, 7168u);          // 56 KB of LDS at most
  G.stage_arcs = 3072u;  // 48 KB + 16 KB: blocks of up to 3072 lattice arcs /
  G.stage_states = 1024u;                                           // 1024 states sweep and walk out of LDS
  G.iter_out = g->iter_out.p;
  G.overflow = (uint32_t*)(g->iter_out.p + 4);
  G.want_after = iter_after_logprob ? 1 : 0;
  G.seed = g->opt.seed;
  G.n_blocks = g->n_blocks;
  // the wavefront path runs the default chain: sampling at temperature 1 with the block's own sample taken out first
  const bool wave_any = g->wave_ok && !g->opt.expectation && (g->opt.high_temp == 0 || g->opt.high_temp == 1) &&
                        (g->opt.low_temp == 0 || g->opt.low_temp == 1) && !lib_opt("gibbs_workgroup");
  const bool wave_run = wave_any && g->opt.mode == 0 && !g->opt.include_self;
  const bool wave_par = wave_any && g->opt.mode == 1;  // the stale-count sweep, a wavefront per block
  GxArgs GX;
  std::memset(&GX, 0, sizeof GX);
  DevBuf<unsigned long long> gx_clk;
  DevBuf<double> gx_idle;
  if (wave_run || wave_par) {
    HIPCHK(gx_idle.alloc(128));
    HIPCHK(hipMemsetAsync(gx_idle.p, 0, 128 * sizeof(double), s));
    GX.idle = gx_idle.p;
    GX.blocks = g->gx_blocks.p;
    GX.arc_rec = (const uint4*)g->gx_rec.p;
    GX.arc_nrm = (const uint2*)g->gx_nrm.p;
    GX.out_off = g->out_off.p;
    GX.level_off = g->level_off.p;
    GX.state_lev = g->gx_state_lev.p;
    GX.lev_arc = g->gx_lev_arc.p;
    GX.p_norm = g->p_norm.p;
    GX.p_prior = g->p_prior.p;
    GX.p_x = g->p_x.p;
    GX.normsum = g->normsum.p;
    GX.ccount = g->ccount.p;
    GX.csum = g->csum.p;
    GX.sample_len = g->sample_len.p;
    GX.sample_ids = g->sample_ids.p;
    GX.sample_nrm = g->sample_nrm.p;
    GX.old_len = g->sample_len.p;
    GX.old_ids = g->sample_ids.p;
    GX.old_nrm = g->sample_nrm.p;
    GX.counterfactual = g->opt.include_self ? 0 : 1;
    GX.cap_arcs = g->cap_arcs;
    GX.cap_states = g->cap_states;
    GX.cap_levels = g->cap_levels;
    GX.cap_sample = g->cap_sample;
    GX.iter_out = g->iter_out.p;
    GX.seed = g->opt.seed;
    GX.n_blocks = g->n_blocks;
    GX.want_after = iter_after_logprob ? 1 : 0;
    if (lib_opt("gibbs_clk")) {
      HIPCHK(gx_clk.alloc(16));
      HIPCHK(hipMemsetAsync(gx_clk.p, 0, 128, s));
      GX.phase_clk = gx_clk.p;
    }
  }
  // gibbs_opts::validate (gibbs_opts.hpp:253-266): --final-counts makes every sweep but the last burn-in; burnin <= iter
  const uint32_t Ni = g->opt.iter, burnin = g->opt.final_counts ? g->opt.iter : std::min(g->opt.burnin, g->opt.iter);
  const uint32_t n_runs = g->opt.restarts + 1;
  std::vector<double> lw(np), best_lw;
  double best_all = 0, best_final = 0, best_sum = 0;
  DevBuf<uint32_t> best_ids, best_len;  // sample of the best run so far (--crp-restarts)
  g->pi_trace.assign((size_t)n_runs * (Ni + 1) * 6, 0.0);
  if (g->run_stride > 1 && g->pi_stddev > 0 && !g->pi_restart_fresh)
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "runs as replicas need independent runs: the inferred priors drift from run to run unless --prior-inference-restart-fresh");
  g->ran_any = false;
  // finalize_cumulative_counts + probs_to_cascade for one finished run: counts := time-integrated counts over the post-burn-in
  // sweeps, weight = final_prob (gibbs.hpp:141-150, gibbs.cc:66-76)
  auto final_weights = [&](std::vector<double>& x, std::vector<double>& sacc, const std::vector<double>& tm, std::vector<double>& out_lw) {
    if (!(g->opt.final_counts && !g->opt.exclude_prior)) {
      const double tmax1 = ((double)Ni - (double)burnin) + 1.0;
      for (uint64_t p = 0; p < np; ++p) {
        if (g->h_norm[p] == 0xffffffffu) continue;
        if (g->opt.exclude_prior) {
          sacc[p] += -g->h_prior[p] * tm[p];
          x[p] += -g->h_prior[p];
        }
        if (!g->opt.final_counts) {
          sacc[p] += x[p] * (tmax1 - tm[p]);  // delta_sum::extend
          x[p] = sacc[p];
        }
      }
    }
    std::vector<double> ns(ng, 0.0);
    for (uint64_t p = 0; p < np; ++p)
      if (g->h_norm[p] != 0xffffffffu) ns[g->h_norm[p]] += x[p];
    for (uint64_t p = 0; p < np; ++p) {
      double pr = g->h_norm[p] == 0xffffffffu ? g->h_prior[p] : (x[p] > 0 ? x[p] / ns[g->h_norm[p]] : 0.0);
      out_lw[p] = pr > 0 ? std::log(pr) : -std::numeric_limits<double>::infinity();
    }
  };
  // ---- the runs of --crp-restarts SIDE BY SIDE (gibbs.hpp:880-914: each run starts from the priors and draws the uniforms of its
  // own sweeps -- independent chains).  The exact chain is one wavefront; a 256-CU part runs as many chains as it is given at the
  // price of one.  Chain c = this rank's c-th run: its own counts, cache model and sample (GxArgs::n_chains); the sweeps of all
  // chains advance together, one launch per sweep; the best run is kept by the sequential loop's rule (the earlier on a tie).
  // Not for: the workgroup kernel's cases, prior-scale inference (its proposals are sequential on the host), an observer.
  {
    std::vector<uint32_t> my_runs;
    for (uint32_t run = 0; run < n_runs; ++run)
      if (run % g->run_stride == g->run_first) my_runs.push_back(run);
    uint32_t cap = 64;
    if (const char* e = lib_opt("gibbs_chains")) cap = (uint32_t)std::max(1, atoi(e));  // 1: one run after the other (A/B)
    // memory of a chain: counts, their time-weighted sums and stamps, norm sums, the sweep's cache model, the sample
    const uint64_t chain_bytes = (np * 4 + ng * 2) * 8 + (uint64_t)g->sample_ids.n * 8 + (uint64_t)g->n_blocks * 4 + 64;
    cap = (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(1, (8ull << 30) / std::max<uint64_t>(chain_bytes, 1)));
    if (wave_run && my_runs.size() > 1 && cap > 1 && g->pi_stddev <= 0 && !g->obs_fn) {
      const uint64_t cs = g->sample_ids.n;  // sample ids of a chain
      DevBuf<double> mx, ms, mt, mn, mcc, mcs, mio;
      DevBuf<uint32_t> mlen, mids, mnrm;
      bool have_best = false;
      std::vector<uint32_t> best_ids_h, best_len_h;
      for (size_t b0 = 0; b0 < my_runs.size(); b0 += cap) {
        const uint32_t R = (uint32_t)std::min<size_t>(cap, my_runs.size() - b0);
        if (mx.n < (size_t)R * np) {
          HIPCHK(mx.alloc((size_t)R * np));
          HIPCHK(ms.alloc((size_t)R * np));
          HIPCHK(mt.alloc((size_t)R * np));
          HIPCHK(mn.alloc((size_t)R * ng));
          HIPCHK(mcc.alloc((size_t)R * np));
          HIPCHK(mcs.alloc((size_t)R * ng));
          HIPCHK(mio.alloc((size_t)R * 8));
          HIPCHK(mlen.alloc((size_t)R * g->n_blocks));
          HIPCHK(mids.alloc((size_t)R * cs));
          HIPCHK(mnrm.alloc((size_t)R * cs));
        }
        // restore_p0 for every chain: counts = priors, normsums = their sums, no sample
        HIPCHK(launch_gibbs_broadcast(mx.p, g->p_prior.p, np, R, s));
        HIPCHK(launch_gibbs_broadcast(mn.p, g->prior_norm.p, ng, R, s));
        HIPCHK(hipMemsetAsync(ms.p, 0, (size_t)R * np * sizeof(double), s));
        HIPCHK(hipMemsetAsync(mt.p, 0, (size_t)R * np * sizeof(double), s));
        HIPCHK(hipMemsetAsync(mlen.p, 0, (size_t)R * g->n_blocks * sizeof(uint32_t), s));
        GxArgs GC = GX;
        GC.p_x = mx.p;
        GC.normsum = mn.p;
        GC.ccount = mcc.p;
        GC.csum = mcs.p;
        GC.sample_len = mlen.p;
        GC.sample_ids = mids.p;
        GC.sample_nrm = mnrm.p;
        GC.old_len = mlen.p;
        GC.old_ids = mids.p;
        GC.old_nrm = mnrm.p;
        GC.iter_out = mio.p;
        GC.phase_clk = nullptr;
        GC.n_chains = R;
        GC.iter_stride = g->run_stride * (Ni + 1);
        GC.ch_params = np;
        GC.ch_norms = ng;
        GC.ch_sample = cs;
        std::vector<double> st_all(R, 0.0), st_final(R, 0.0), st_sum(R, -std::numeric_limits<double>::infinity());
        std::vector<double> io((size_t)R * 8);
        for (uint32_t iter = 0; iter <= Ni; ++iter) {
          const double time = iter == 0 ? 0.0 : std::max(0.0, (double)iter - (double)burnin);
          HIPCHK(hipMemsetAsync(mio.p, 0, (size_t)R * 8 * sizeof(double), s));
          HIPCHK(launch_gibbs_broadcast(mcc.p, g->p_prior.p, np, R, s));
          HIPCHK(launch_gibbs_broadcast(mcs.p, g->prior_norm.p, ng, R, s));
          HIPCHK(launch_forest_fold(ms.p, mt.p, mx.p, time, (uint64_t)R * np, s));
          GC.iter = my_runs[b0] * (Ni + 1) + iter;
          GC.init_logw = (iter == 0 && my_runs[b0] == 0 && g->init_logw.n) ? g->init_logw.p : nullptr;
          GC.init_chain = GC.init_logw ? 0u : 0xffffffffu;
          HIPCHK((g->reg_nq && !GC.init_logw) ? launch_gibbs_reg_wave(GC, 0, g->reg_nq, s) : launch_gibbs_exact_wave(GC, 0, s));
          HIPCHK(hipMemcpyAsync(io.data(), mio.p, io.size() * sizeof(double), hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
          for (uint32_t c = 0; c < R; ++c) {
            const uint32_t run = my_runs[b0 + c];
            const double plog = io[(size_t)c * 8 + 0];
            if (iter_logprob) iter_logprob[(size_t)run * (Ni + 1) + iter] = plog;
            if (iter_cheap_logprob) iter_cheap_logprob[(size_t)run * (Ni + 1) + iter] = io[(size_t)c * 8 + 1];
            if (iter_after_logprob) iter_after_logprob[(size_t)run * (Ni + 1) + iter] = io[(size_t)c * 8 + 2];
            if (iter >= burnin) {  // gibbs.hpp:942-943
              st_all[c] += plog;
              st_final[c] = plog;
              const double hi = std::max(st_sum[c], plog), lo = std::min(st_sum[c], plog);
              st_sum[c] = hi + (lo == -std::numeric_limits<double>::infinity() ? 0.0 : std::log1p(std::exp(lo - hi)));
            }
          }
        }
        // every chain's final weights; the better run by gibbs_stats::better, in run order
        std::vector<double> x(np), sacc(np), tm(np);
        for (uint32_t c = 0; c < R; ++c) {
          HIPCHK(hipMemcpyAsync(x.data(), mx.p + (size_t)c * np, np * sizeof(double), hipMemcpyDeviceToHost, s));
          HIPCHK(hipMemcpyAsync(sacc.data(), ms.p + (size_t)c * np, np * sizeof(double), hipMemcpyDeviceToHost, s));
          HIPCHK(hipMemcpyAsync(tm.data(), mt.p + (size_t)c * np, np * sizeof(double), hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
          final_weights(x, sacc, tm, lw);
          const bool better = !g->ran_any || (g->opt.argmax_final ? st_final[c] > best_final : g->opt.argmax_sum ? st_sum[c] > best_sum : st_all[c] > best_all);
          if (better) {
            g->best_run = my_runs[b0 + c];
            best_all = st_all[c];
            best_final = st_final[c];
            best_sum = st_sum[c];
            g->best_stats[0] = st_all[c];
            g->best_stats[1] = st_final[c];
            g->best_stats[2] = st_sum[c];
            best_lw = lw;
            g->h_final_x = x;
            best_ids_h.resize(cs);
            best_len_h.resize(g->n_blocks);
            HIPCHK(hipMemcpyAsync(best_ids_h.data(), mids.p + (size_t)c * cs, cs * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(best_len_h.data(), mlen.p + (size_t)c * g->n_blocks, g->n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            have_best = true;
          }
          g->ran_any = true;
        }
        if (b0 + R >= my_runs.size()) {  // what a later look at the sampler's state finds: the last run's counts
          HIPCHK(hipMemcpyAsync(g->p_x.p, mx.p + (size_t)(R - 1) * np, np * sizeof(double), hipMemcpyDeviceToDevice, s));
          HIPCHK(hipMemcpyAsync(g->p_s.p, ms.p + (size_t)(R - 1) * np, np * sizeof(double), hipMemcpyDeviceToDevice, s));
          HIPCHK(hipMemcpyAsync(g->p_tmax.p, mt.p + (size_t)(R - 1) * np, np * sizeof(double), hipMemcpyDeviceToDevice, s));
          HIPCHK(hipMemcpyAsync(g->normsum.p, mn.p + (size_t)(R - 1) * ng, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
          HIPCHK(hipStreamSynchronize(s));
        }
      }
      if (have_best) {  // the kept run's sample is the sampler's sample
        HIPCHK(hipMemcpyAsync(g->sample_ids.p, best_ids_h.data(), cs * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(g->sample_len.p, best_len_h.data(), g->n_blocks * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
      }
      g->ran = true;
      return carmel_hip_set_weights(t, best_lw.data());
    }
  }
  for (uint32_t run = 0; run < n_runs; ++run) {
  if (run % g->run_stride != g->run_first) continue;  // another replica's run
  if (g->pi_stddev > 0 && run > 0 && g->pi_restart_fresh) {  // gibbs.hpp:889-898: the priors start over
    HIPCHK(hipMemcpyAsync(g->p_prior.p, g->h_prior0.data(), np * sizeof(double), hipMemcpyHostToDevice, s));
    std::vector<double> pn(ng, 0.0);
    for (uint64_t p = 0; p < np; ++p)
      if (g->h_norm[p] != 0xffffffffu) pn[g->h_norm[p]] += g->h_prior0[p];
    HIPCHK(hipMemcpyAsync(g->prior_norm.p, pn.data(), ng * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    g->cumulative.assign(g->n_scale, 1.0);
  }
  // restore_p0: counts = priors, normsums = their sums, no sample
  HIPCHK(hipMemcpyAsync(g->p_x.p, g->p_prior.p, np * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemsetAsync(g->p_s.p, 0, np * sizeof(double), s));
  HIPCHK(hipMemsetAsync(g->p_tmax.p, 0, np * sizeof(double), s));
  // an observer may ask for the tables (--print-counts-*): the time of the last sweep that changed a count, as the reference's
  // delta_sum keeps it, is tracked beside the sums (GxArgs::p_touch; the one-wavefront chain)
  if (g->obs_fn && g->opt.mode == 0 && wave_run) {
    if (g->p_touch.n != np) HIPCHK(g->p_touch.alloc(np));
    HIPCHK(hipMemsetAsync(g->p_touch.p, 0, np * sizeof(double), s));
    GX.p_touch = g->p_touch.p;
  } else {
    g->p_touch.release();
    GX.p_touch = nullptr;
  }
  HIPCHK(hipMemcpyAsync(g->normsum.p, g->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemsetAsync(g->sample_len.p, 0, g->sample_len.bytes(), s));
  G.sample_ids = g->sample_ids.p;
  G.sample_len = g->sample_len.p;
  G.new_ids = g->new_ids.p;
  G.new_len = g->new_len.p;
  // gibbs_stats (gibbs_opts.hpp:270-296): over the sweeps from burn-in on
  double st_all = 0.0, st_final = 0.0, st_sum = -std::numeric_limits<double>::infinity();
  bool lane_run = false;
  for (uint32_t iter = 0; iter <= Ni; ++iter) {
    G.iter = run * (Ni + 1) + iter;  // of the uniforms: every run draws its own
    // gibbs.hpp:816: the initial --expectation sweep of a --random-start run, and of every restart
    G.randomize = (g->opt.expectation && iter == 0 && (g->opt.random_start || run > 0)) ? 1 : 0;
    G.init_logw = (run == 0 && iter == 0 && g->init_logw.n) ? g->init_logw.p : nullptr;
    G.power = gibbs_anneal_power(g->opt.high_temp, g->opt.low_temp, Ni, iter);
    G.time = iter == 0 ? 0.0 : std::max(0.0, (double)iter - (double)burnin);
    HIPCHK(hipMemsetAsync(g->iter_out.p, 0, 5 * sizeof(double), s));
    if (g->opt.mode == 0 && wave_run) {
      // the chain on one wavefront (gibbs_exact.hip); delta_sum's fold for every parameter at once: at the start of a sweep every
      // count is what the previous sweep left, which is what the reference folds at a parameter's first touch (delta_sum.hpp:74-84)
      HIPCHK(hipMemcpyAsync(g->ccount.p, g->p_prior.p, np * sizeof(double), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(g->csum.p, g->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
      HIPCHK(launch_forest_fold(g->p_s.p, g->p_tmax.p, g->p_x.p, G.time, np, s));
      GX.iter = G.iter;
      GX.init_logw = G.init_logw;
      GX.time = G.time;
      HIPCHK((g->reg_nq && !GX.init_logw) ? launch_gibbs_reg_wave(GX, 0, g->reg_nq, s) : launch_gibbs_exact_wave(GX, 0, s));
    } else if (g->opt.mode == 0) {
      HIPCHK(hipMemcpyAsync(g->ccount.p, g->p_prior.p, np * sizeof(double), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(g->csum.p, g->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
      {
        const size_t lds = (size_t)G.books_cap * 8 + (size_t)G.stage_arcs * 16 + (size_t)G.stage_states * 8 +
                           (size_t)(2 * G.stage_states + 2) * 4;
        if (lds > 64 * 1024)
          (void)hipFuncSetAttribute((const void*)gibbs_sweep_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gibbs_sweep_exact_kernel, dim3(1), dim3(256), lds, s, G);
      }
    } else {
      HIPCHK(hipMemcpyAsync(g->snap_x.p, g->p_x.p, np * sizeof(double), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(g->snap_norm.p, g->normsum.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
      if (wave_par) {
        // a wavefront per block (gibbs_exact.hip): proposals from the snapshot with the block's own sample taken out
        // arithmetically, new samples beside the old ones; the recount and commit below are shared with the kernel it replaces
        GX.iter = G.iter;
        GX.init_logw = G.init_logw;
        GX.p_x = g->snap_x.p;
        GX.normsum = g->snap_norm.p;
        GX.old_len = g->sample_len.p;
        GX.old_ids = g->sample_ids.p;
        GX.old_nrm = g->sample_nrm.p;
        GX.sample_len = g->new_len.p;
        GX.sample_ids = g->new_ids.p;
        GX.sample_nrm = g->new_nrm.p;
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device);
        // the blocks that go one per lane (not under prior-scale inference: its re-scoring reads the sampler's own sample format)
        lane_run = g->gl_ngroups && g->pi_stddev <= 0;
        if (lane_run) {
          GlArgs LA;
          std::memset(&LA, 0, sizeof LA);
          LA.groups = g->gl_groups.p;
          LA.lanes = g->gl_lanes.p;
          LA.recA = (const uint4*)g->gl_recA.p;
          LA.recB = (const uint4*)g->gl_recB.p;
          LA.arc_id = g->gl_arc.p;
          LA.sw = (double2*)g->gl_sw.p;
          LA.wq = g->gl_sw.p + 2 * (g->gl_sw.n / 3);
          LA.samp_old = (const uint4*)g->gl_samp[g->gl_cur].p;
          LA.samp_new = (uint4*)g->gl_samp[g->gl_cur ^ 1].p;
          LA.p_x = g->snap_x.p;
          LA.normsum = g->snap_norm.p;
          LA.p_prior = g->p_prior.p;
          LA.init_logw = G.init_logw;
          LA.iter_out = g->iter_out.p;
          LA.phase_clk = GX.phase_clk;
          LA.seed = g->opt.seed;
          LA.iter = G.iter;
          LA.have_old = (iter > 0 && !g->opt.include_self) ? 1 : 0;
          for (auto& c : g->gl_classes) {
            LA.first_group = c.first;
            LA.W = c.W;
            LA.LP = c.LP;
            LA.LN = c.LN;
            HIPCHK(launch_gibbs_lane(LA, c.count, s));
          }
        }
        for (int c = GX_WCLASSES - 1; c >= 0; --c) {  // (all on the trainer's stream, the long blocks first: the next fills the chip as one drains)
          auto& W = lane_run ? g->wrest[c] : g->wclass[c];
          if (!W.n) continue;
          GX.list = W.list.p;
          GX.n_blocks = W.n;
          GX.cap_arcs = W.cap_arcs;
          GX.cap_states = W.cap_states;
          GX.cap_levels = W.cap_levels;
          GX.cap_sample = W.cap_sample;
          if (W.nq && !GX.init_logw) {
            uint32_t own = 64;  // the own-sample tables: at most half full
            while (own < 2 * GX.cap_sample) own <<= 1;
            GX.own_slots = own;
            const size_t lds_w = gibbs_reg_lds_bytes(GX.cap_arcs, GX.cap_states, GX.cap_sample, own);
            const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(W.nq <= 4 ? 16 : 8, (160 * 1024) / std::max<size_t>(lds_w, 1)));
            HIPCHK(launch_gibbs_reg_wave(GX, std::min<uint32_t>(W.n, (uint32_t)cus * per_cu), W.nq, s));
            continue;
          }
          const size_t lds_w = gibbs_exact_lds_bytes(GX.cap_arcs, GX.cap_states, GX.cap_levels, GX.cap_sample);
          const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / std::max<size_t>(lds_w, 1)));
          HIPCHK(launch_gibbs_exact_wave(GX, std::min<uint32_t>(W.n, (uint32_t)cus * per_cu), s));
        }
      } else {
      uint32_t grid = std::min<uint32_t>(g->n_blocks, 256u * 16u);
      // the previous sample of a block in LDS up to GIBBS_OWN_CAP ids (32 KB: under the default dynamic-LDS limit); longer
      // ones -- cyclic lattices -- are read from global memory by g_resample_block
      uint32_t own_cap = std::min<uint32_t>(g->max_sample, GIBBS_OWN_CAP);
      if (const char* e = lib_opt("gibbs_own_cap")) own_cap = std::min<uint32_t>(own_cap, (uint32_t)std::max(1, atoi(e)));  // test hook
      size_t lds = (size_t)own_cap * sizeof(uint32_t);
      hipLaunchKernelGGL(gibbs_sweep_parallel_kernel, dim3(grid), dim3(64), lds, s, G, own_cap);
      }
      // counts of the new samples: start from the priors, add every use
      HIPCHK(hipMemcpyAsync(g->ccount.p, g->p_prior.p, np * sizeof(double), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(g->normsum.p, g->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
      // (the wavefront paths' recounts add the parameters' uses only; the norm sums are formed from the new counts group by group)
      if (wave_par && lane_run) {
        if (g->n_rest)
          HIPCHK(launch_gibbs_recount_tables(g->gx_blocks.p, g->new_len.p, g->new_ids.p, g->new_nrm.p, g->n_rest, g->ccount.p, nullptr, s,
                                             g->rest_list.p));
        HIPCHK(launch_gibbs_lane_recount(g->gl_groups.p, g->gl_lanes.p, (const uint4*)g->gl_recA.p, (const uint4*)g->gl_samp[g->gl_cur ^ 1].p,
                                         g->gl_ngroups, g->ccount.p, nullptr, s));
        g->gl_cur ^= 1;
      } else if (wave_par)
        HIPCHK(launch_gibbs_recount_tables(g->gx_blocks.p, g->new_len.p, g->new_ids.p, g->new_nrm.p, g->n_blocks, g->ccount.p, nullptr, s));
      if (wave_par) HIPCHK(launch_gibbs_normsum(g->ccount.p, g->p_norm.p, t->group_off.p, t->norm_perm.p, ng, g->normsum.p, s));
      else
        hipLaunchKernelGGL(gibbs_recount_kernel, dim3(std::min<uint32_t>((g->n_blocks + 255) / 256, 4096u)), dim3(256), 0, s,
                           G, g->ccount.p, g->normsum.p);
      hipLaunchKernelGGL(gibbs_commit_kernel, dim3((unsigned)std::min<uint64_t>((np + 255) / 256, 4096)), dim3(256), 0, s, G,
                         g->ccount.p, np);
      std::swap(g->sample_ids.p, g->new_ids.p);
      std::swap(g->sample_len.p, g->new_len.p);
      std::swap(g->sample_nrm.p, g->new_nrm.p);
      G.sample_ids = g->sample_ids.p;
      G.sample_len = g->sample_len.p;
      G.new_ids = g->new_ids.p;
      G.new_len = g->new_len.p;
      // the lanes' paths in the sampler's own format, for whoever reads the sample next: the end of the run, an observer
      if (lane_run && (iter == Ni || (g->obs_fn && g->obs_every && iter % g->obs_every == 0)))
        HIPCHK(launch_gibbs_lane_materialize(g->gl_groups.p, g->gl_lanes.p, (const uint4*)g->gl_recA.p, (const uint4*)g->gl_samp[g->gl_cur].p,
                                             g->gl_ngroups, g->sample_ids.p, g->sample_nrm.p, g->sample_len.p, s));
    }
    HIPCHK(hipGetLastError());
    double io[5];
    HIPCHK(hipMemcpyAsync(io, g->iter_out.p, sizeof io, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    {
      uint32_t ov;
      std::memcpy(&ov, &io[4], sizeof ov);
      if (ov)
        return fail(CARMEL_HIP_ERR_UNSUPPORTED, "a sampled path through a derivation lattice with a cycle outgrew its sample buffer (32 x the lattice's "
                                                "states): the reference's walk has no bound, this one has");
    }
    const double plog = g->opt.mode == 0 ? io[0] : io[1];
    {
      // propose_new_priors (gibbs.hpp:525-553), on the sweeps that infer (gibbs.hpp:559-563)
      const uint32_t pstart = g->pi_start ? g->pi_start : burnin;
      if (g->pi_stddev > 0 && g->n_scale && iter > 0 && pstart <= iter && (!g->pi_end || iter < g->pi_end)) {
        const double sdev = g->pi_stddev;
        const double q0 = gibbs_norm_cdf((0.0 - 1.0) / sdev), qrem = 1.0 - q0;  // scale ratios are > 0 (gibbs.hpp:488-499)
        std::vector<double> sc(g->n_scale + 1, 1.0);
        double ln_a2 = 0.0;
        for (uint32_t k = 1; k <= g->n_scale; ++k) {
          const double u = gibbs_uniform(g->opt.seed, G.iter, 0xfffffffeu, k);
          sc[k] = 1.0 + sdev * gibbs_norm_quantile(q0 + u * qrem);
          const double d_old = 1.0 / sc[k] - 1.0, d_new = sc[k] - 1.0;  // q(old | new) / q(new | old), both N(1, sdev)
          ln_a2 += (d_new * d_new - d_old * d_old) / (2.0 * sdev * sdev);
        }
        HIPCHK(hipMemcpyAsync(g->d_scales.p, sc.data(), sc.size() * sizeof(double), hipMemcpyHostToDevice, s));
        const size_t lds_c = (size_t)G.books_cap * 8;
        auto cache_prob_all = [&](double& out) -> int {
          HIPCHK(hipMemcpyAsync(g->ccount.p, g->p_prior.p, np * sizeof(double), hipMemcpyDeviceToDevice, s));
          HIPCHK(hipMemcpyAsync(g->csum.p, g->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
          hipLaunchKernelGGL(gibbs_cache_prob_kernel, dim3(1), dim3(256), lds_c, s, G);
          HIPCHK(hipGetLastError());
          HIPCHK(hipMemcpyAsync(&out, g->iter_out.p + 3, sizeof(double), hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
          return CARMEL_HIP_OK;
        };
        auto scale = [&](int invert) -> int {
          hipLaunchKernelGGL(gibbs_scale_priors_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, G, t->group_off.p,
                             t->norm_perm.p, (uint64_t)ng, g->d_meta.p, g->d_scales.p, invert, g->p_prior.p, g->prior_norm.p);
          HIPCHK(hipGetLastError());
          return CARMEL_HIP_OK;
        };
        double p1 = 0, p2 = 0;
        int rc = cache_prob_all(p1);
        if (!rc) rc = scale(0);
        if (!rc) rc = cache_prob_all(p2);
        if (rc) return rc;
        const double a = std::exp((p2 - p1) + ln_a2);
        const bool accept = gibbs_uniform(g->opt.seed, G.iter, 0xffffffffu, 0) < a;
        if (!accept) {
          rc = scale(1);
          if (rc) return rc;
        } else
          for (uint32_t k = 1; k <= g->n_scale; ++k) g->cumulative[k - 1] *= sc[k];
        double* tr = g->pi_trace.data() + ((size_t)run * (Ni + 1) + iter) * 6;
        tr[0] = 1;
        tr[1] = accept ? 1 : 0;
        tr[2] = p1;
        tr[3] = p2;
        tr[4] = std::exp(ln_a2);
        tr[5] = a;
      }
    }
    if (iter_logprob) iter_logprob[(size_t)run * (Ni + 1) + iter] = plog;
    if (iter_cheap_logprob) iter_cheap_logprob[(size_t)run * (Ni + 1) + iter] = io[1];
    if (iter_after_logprob) iter_after_logprob[(size_t)run * (Ni + 1) + iter] = io[2];
    if (iter >= burnin) {  // gibbs.hpp:942-943
      st_all += plog;
      st_final = plog;
      const double hi = std::max(st_sum, plog), lo = std::min(st_sum, plog);
      st_sum = hi + (lo == -std::numeric_limits<double>::infinity() ? 0.0 : std::log1p(std::exp(lo - hi)));
    }
    // maybe_print_periodic (gibbs.hpp:959-968): the caller looks at the sample and the counts as they stand after this sweep
    if (g->obs_fn && g->obs_every && iter % g->obs_every == 0) g->obs_fn(g->obs_ctx, run, iter, G.time);
  }
  if (gx_clk.n) {
    unsigned long long c[16];
    HIPCHK(hipMemcpy(c, gx_clk.p, sizeof c, hipMemcpyDeviceToHost));
    if (c[4] && c[8])
      fprintf(stderr, "[carmel_hip] gibbs wave kernels, parts: arrivals (+ own tables) %.0f, weights %.0f, offsets + uniforms %.0f, requests %.0f | choices %.0f, path %.0f, "
                      "parameters %.0f (per block, over ALL blocks of the sweep)\n", c[8] / (double)c[4], c[9] / (double)c[4], c[10] / (double)c[4],
              c[11] / (double)c[4], c[12] / (double)c[4], c[13] / (double)c[4], c[14] / (double)c[4]);
    if (c[4])
      fprintf(stderr, "[carmel_hip] gibbs_exact_wave cycles per block: wait+weights %.0f, backward %.0f, walk %.0f, counts %.0f (%llu blocks x sweeps)\n",
              c[0] / (double)c[4], c[1] / (double)c[4], c[2] / (double)c[4], c[3] / (double)c[4], c[4]);
    HIPCHK(hipMemset(gx_clk.p, 0, 128));
  }
  // finalize_cumulative_counts: counts := time-integrated counts over the post-burn-in sweeps
  if (g->pi_stddev > 0) {  // the priors have moved
    HIPCHK(hipMemcpyAsync(g->h_prior.data(), g->p_prior.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  std::vector<double> x(np), sacc(np), tm(np);
  HIPCHK(hipMemcpyAsync(x.data(), g->p_x.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(sacc.data(), g->p_s.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(tm.data(), g->p_tmax.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  final_weights(x, sacc, tm, lw);
  // gibbs_base::run_starts (gibbs.hpp:880-914): keep the run that is better by gibbs_stats::better (gibbs_opts.hpp:313-316)
  const bool better = !g->ran_any || (g->opt.argmax_final ? st_final > best_final : g->opt.argmax_sum ? st_sum > best_sum : st_all > best_all);
  if (better) {
    g->best_run = run;
    best_all = st_all;
    best_final = st_final;
    best_sum = st_sum;
    g->best_stats[0] = st_all;
    g->best_stats[1] = st_final;
    g->best_stats[2] = st_sum;
    best_lw = lw;
    g->h_final_x = x;
    if (n_runs > 1 && !g->opt.expectation) {
      HIPCHK(best_ids.alloc(g->sample_ids.n));
      HIPCHK(best_len.alloc(g->sample_len.n));
      HIPCHK(hipMemcpyAsync(best_ids.p, g->sample_ids.p, g->sample_ids.bytes(), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(best_len.p, g->sample_len.p, g->sample_len.bytes(), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
    }
  }
  g->ran_any = true;
  }  // runs
  if (!g->ran_any) {  // more replicas than runs: nothing to keep, the trainer's weights stay
    g->ran = true;
    return CARMEL_HIP_OK;
  }
  if (n_runs > 1 && best_ids.p) {
    HIPCHK(hipMemcpyAsync(g->sample_ids.p, best_ids.p, g->sample_ids.bytes(), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g->sample_len.p, best_len.p, g->sample_len.bytes(), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  lw = best_lw;
  g->ran = true;
  return carmel_hip_set_weights(t, lw.data());
}

int carmel_hip_gibbs_get_sample(carmel_hip_gibbs* g, uint32_t block, uint32_t* ids, uint32_t* n) {
  if (!g || !n || block >= g->n_blocks) return fail(CARMEL_HIP_ERR_ARG, "bad argument");
  if (g->opt.expectation) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "there is no single sample with --expectation (gibbs.cc:259-260)");
  HIPCHK(hipSetDevice(g->t->device));
  hipStream_t s = g->t->stream;
  uint32_t len = 0;
  HIPCHK(hipMemcpyAsync(&len, g->sample_len.p + block, sizeof len, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (ids && len)
    HIPCHK(hipMemcpyAsync(ids, g->sample_ids.p + g->h_sample_off[block], len * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *n = len;
  return CARMEL_HIP_OK;
}

int carmel_hip_gibbs_set_observer(carmel_hip_gibbs* g, uint32_t every, carmel_hip_gibbs_observer_fn fn, void* ctx) {
  if (!g) return fail(CARMEL_HIP_ERR_ARG, "null sampler");
  g->obs_every = fn ? every : 0u;
  g->obs_fn = fn;
  g->obs_ctx = ctx;
  return CARMEL_HIP_OK;
}
// gibbs_base::proposal_prob (gibbs.hpp:163-170) of every parameter from the counts as they stand: count / norm sum, the prior
// of a parameter outside the norm groups
int carmel_hip_gibbs_current_probs(carmel_hip_gibbs* g, double* prob) {
  if (!g || !prob) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(g->t->device));
  hipStream_t s = g->t->stream;
  const size_t np = g->h_norm.size();
  std::vector<double> x(np), ns(g->n_norm);
  HIPCHK(hipMemcpyAsync(x.data(), g->p_x.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  if (g->n_norm) HIPCHK(hipMemcpyAsync(ns.data(), g->normsum.p, g->n_norm * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (size_t p = 0; p < np; ++p) prob[p] = g->h_norm[p] == 0xffffffffu ? g->h_prior[p] : x[p] / ns[g->h_norm[p]];
  return CARMEL_HIP_OK;
}

// the sampler's counts as they stand (gibbs_param::sumcount, delta_sum.hpp: instantaneous count x, its time-weighted sum s, the
// time tmax it is summed up to) and the priors -- what --print-counts-* shows; inside an observer call, or after the run
int carmel_hip_gibbs_get_state(carmel_hip_gibbs* g, double* x, double* sum, double* tmax, double* prior, double* last_touch) {
  if (!g) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(g->t->device));
  hipStream_t s = g->t->stream;
  const size_t np = g->h_norm.size();
  if (x) HIPCHK(hipMemcpyAsync(x, g->p_x.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  if (sum) HIPCHK(hipMemcpyAsync(sum, g->p_s.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  if (tmax) HIPCHK(hipMemcpyAsync(tmax, g->p_tmax.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  if (prior) HIPCHK(hipMemcpyAsync(prior, g->p_prior.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  // (where the chain does not track it -- the workgroup kernels, the parallel sweep --: the stamp of the sums, i.e. the sweep's time)
  if (last_touch) HIPCHK(hipMemcpyAsync(last_touch, g->p_touch.n == np ? g->p_touch.p : g->p_tmax.p, np * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}
// ... and the kept run's counts after finalize_cumulative_counts (gibbs.hpp:626-638): what the final table shows as count * (t + 1)
int carmel_hip_gibbs_final_counts(carmel_hip_gibbs* g, double* x) {
  if (!g || !x) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (g->h_final_x.size() != g->h_norm.size()) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_gibbs_final_counts: run the sampler first");
  std::memcpy(x, g->h_final_x.data(), g->h_final_x.size() * sizeof(double));
  return CARMEL_HIP_OK;
}

uint32_t carmel_hip_gibbs_max_sample(carmel_hip_gibbs* g) { return g ? g->max_sample : 0; }
uint32_t carmel_hip_gibbs_best_run(carmel_hip_gibbs* g) { return g ? g->best_run : 0; }
int carmel_hip_gibbs_set_run_share(carmel_hip_gibbs* g, uint32_t first, uint32_t stride) {
  if (!g || stride == 0 || first >= stride) return fail(CARMEL_HIP_ERR_ARG, "bad run share");
  g->run_first = first;
  g->run_stride = stride;
  return CARMEL_HIP_OK;
}
int carmel_hip_gibbs_best_stats(carmel_hip_gibbs* g, double* out3, int* ran_any) {
  if (!g || !out3) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  for (int k = 0; k < 3; ++k) out3[k] = g->best_stats[k];
  if (ran_any) *ran_any = g->ran_any ? 1 : 0;
  return CARMEL_HIP_OK;
}
int carmel_hip_gibbs_set_init_weights(carmel_hip_gibbs* g, const double* arc_logw) {
  if (!g) return fail(CARMEL_HIP_ERR_ARG, "null sampler");
  HIPCHK(hipSetDevice(g->t->device));
  if (!arc_logw) {
    g->init_logw.release();
    return CARMEL_HIP_OK;
  }
  if (g->opt.expectation) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "--init-em has no effect with --expectation (gibbs.cc:311-314)");
  HIPCHK(g->init_logw.upload(std::vector<double>(arc_logw, arc_logw + g->t->w.n_arcs), g->t->stream));
  HIPCHK(hipStreamSynchronize(g->t->stream));
  return CARMEL_HIP_OK;
}

}  // extern "C"
