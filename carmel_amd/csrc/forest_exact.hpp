// forest_exact.hpp — the reference's forest sampler chain (forest-em --crp without --crp-parallel) as ONE persistent
// kernel per sweep (forest_exact.hip); arguments as forest.hip's carmel_hip_forests_gibbs fills them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carmel_hip {

// a forest of the register path: at most FX_NODES nodes, FX_KIDS children per node, at most FX_STACK nodes pending in the
// depth-first walk (a byte each in 128 bits of scalar registers) and a derivation of at most FX_NODES rules; everything else takes the LDS path (same chain, slower)
#define FX_NODES 128
#define FX_KIDS 4
#define FX_STACK 15

struct FExactArgs {
  // per forest, in forest order (the order of the chain): {first node record, n | heights << 16, sample offset low,
  // sample offset high (16 bits) | flags << 16 (bit 0: LDS path)}
  const uint4* xdesc;
  // per node, nodes numbered by height within the forest: {first child | children << 8 | height << 16 | AND << 31, the
  // other children (a byte each, the first to visit lowest), rule, norm group}; node ids are bytes, 0xff: none
  const uint4* xrec;
  // LDS path: the per-forest tables of the several-lanes sampler (forest.hip, FMultiArgs)
  const uint16_t* tab;     // {n, H, n_kids, -}, lvl_off[H + 1], kid_off[n + 1], kids[n_kids] (| 0x8000: back-reference)
  const uint32_t* hdr;     // per node {header row | bit 31 = AND, rule, class word, norm group}
  const uint4* slots;      // per lane slot: {tab offset, hdr offset} (64 bit each), {sample offset (64 bit), forest, n | words << 15}
  const uint32_t* lane_of_forest;
  uint32_t* sample_len;    // per forest
  uint32_t* sample_rules;  // the current sample of every forest: rule ids ...
  uint32_t* sample_nn;     // ... and their norm groups (what the next sweep takes out of the counts)
  double* p_x;             // CRP counts (gibbs_param::count, gibbs.hpp:106-227); their time-weighted sums are folded once per
                           // sweep for every parameter (forest_fold_kernel) instead of at a parameter's first touch
  double* normsum;         // per norm group
  const double* p_prior;   // per rule (a rule outside every group: its fixed probability)
  double* ccount;          // cache model of this sweep (gibbs.hpp:712-742): starts from the priors
  double* csum;
  double* iter_out;        // {ln cache-model prob, ln proposal prob} of the sweep
  unsigned long long* phase_clk;  // experiment: summed cycles per phase (null: none)
  double* idle;                   // 4 * 64 doubles nobody reads: where the lanes without a counted sample entry send their (zero) adds
  uint64_t seed;
  uint32_t iter, n_forests;
  uint32_t max_n, max_tab, max_stack, max_sample;  // LDS path's carve: nodes, table words, stack entries, sample entries
  // --crp-restarts, the runs side by side (gibbs.hpp:880-914: independent chains): workgroup c of the launch is chain c -- its
  // counts, norm sums, cache model, sample and results at c times the strides below, its uniforms those of sweep iter + c * iter_stride
  uint32_t n_chains, iter_stride;
  uint64_t ch_rules, ch_norms, ch_sample, ch_forests;
};

size_t forest_exact_lds_bytes(uint32_t max_n, uint32_t max_tab, uint32_t max_stack, uint32_t max_sample);
hipError_t launch_forest_exact(const FExactArgs& A, hipStream_t s);
// delta_sum's fold (delta_sum.hpp:74-84) for every parameter at the start of a sweep: s += (time - tmax) * x, tmax = time
hipError_t launch_forest_fold(double* p_s, double* p_tmax, const double* p_x, double time, uint64_t n, hipStream_t s);

}  // namespace carmel_hip
