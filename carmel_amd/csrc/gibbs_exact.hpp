// gibbs_exact.hpp — `carmel --crp`'s reference chain (blocks strictly in order) as one persistent wavefront per sweep
// (gibbs_exact.hip); arguments as gibbs.hip's carmel_hip_gibbs_run_ex fills them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carmel_hip {

// a block of the wavefront path: an acyclic lattice of at most GX_ARCS arcs / GX_STATES states / GX_LEVELS levels whose
// composed arcs stand for at most two parameters; anything else keeps gibbs.hip's workgroup kernel (the same chain)
#define GX_ARCS 1024
#define GX_STATES 512
#define GX_LEVELS 512
#define GX_SAMPLE 1024

struct GxBlock {  // 48 bytes per block, in corpus order (the order of the chain)
  uint32_t out_base, off_base, level_base, n_arcs;  // into the per-lattice-arc / per-state / per-level arrays
  uint32_t n_states, n_levels, start, fin;           // n_levels bit 31: every arc joins neighbouring levels (a trellis)
  uint64_t sample_off;
  double wt;  // the pair's weight (counts move by it)
};

struct GxArgs {
  const GxBlock* blocks;
  const uint32_t* list;       // parallel sweep: the blocks of this launch (null: all, in order)
  const uint4* arc_rec;       // per lattice arc: {destination | source << 16, composed arc, parameter 0, parameter 1 (0xffffffff: none)}
  const uint2* arc_nrm;       // ... their norm groups (0xffffffff: a fixed probability, the parameter's prior)
  const uint32_t* out_off;    // per state: first out-arc (bundle-relative), n_states + 1 per block
  const uint32_t* level_off;  // per level: first state, n_levels + 1 per block
  const uint16_t* state_lev;  // per state: its level
  const uint32_t* lev_arc;    // per level: its first arc (= out_off[level_off[l]]; n_levels + 1 per block, beside level_off)
  const uint32_t* p_norm;
  const double* p_prior;
  double* p_x;                // CRP counts; their time-weighted sums are folded once per sweep (launch_forest_fold)
  double* normsum;
  double* ccount;             // cache model of this sweep (gibbs.hpp:712-742)
  double* csum;
  uint32_t* sample_len;       // where this sweep's samples go
  uint32_t* sample_ids;       // per block: parameter ids along the sampled path, chain order ...
  uint32_t* sample_nrm;       // ... and their norm groups
  const uint32_t* old_len;    // the previous sweep's samples (the exact chain: the same buffers)
  const uint32_t* old_ids;
  const uint32_t* old_nrm;
  const double* init_logw;    // --init-em / --init-from-p0: per composed arc, what the first sweep samples from (null: the counts)
  double* iter_out;           // {ln cache-model prob, ln proposal prob, ln proposal prob after the add-back}
  unsigned long long* phase_clk;
  double* idle;               // 128 doubles nobody reads: where the lanes without a sample entry send their (zero) adds
  uint64_t seed;
  uint32_t iter, n_blocks;
  int want_after;
  int counterfactual;         // parallel sweep: take the block's own previous sample out of the snapshot counts
  uint32_t cap_arcs, cap_states, cap_levels, cap_sample;  // LDS carve: the largest block's arcs / states / levels / sample
  uint32_t own_slots;         // the register kernel's own-sample tables: slots each (a power of two, >= 2 x cap_sample)
  // several exact chains side by side (the runs of --crp-restarts, gibbs.hpp:880-914: independent by construction): chain c =
  // workgroup c of the launch, one wavefront, with its own counts, cache model, sample and results at these strides, drawing
  // the uniforms of sweep iter + c * iter_stride.  n_chains <= 1: the one chain of the fields above.
  uint32_t n_chains, iter_stride;
  uint32_t init_chain;        // the chain init_logw is for (the very first sweep of run 0 only); 0xffffffff: none
  uint64_t ch_params, ch_norms, ch_sample;  // doubles per chain of p_x / ccount, of normsum / csum; ids per chain of the sample buffers
  // --print-counts-*' "last@t" column (delta_sum::tmax as the reference keeps it: the time of the last sweep that changed the
  // count -- the sums here are folded for every parameter at the start of a sweep, which moves their own stamp): when non-null,
  // p_touch[p] = time wherever a count changes
  double* p_touch;
  double time;
};

size_t gibbs_exact_lds_bytes(uint32_t cap_arcs, uint32_t cap_states, uint32_t cap_levels, uint32_t cap_sample);
// n_waves = 0: the exact chain (one wavefront; GxArgs::n_chains of them side by side); > 0: the stale-count parallel sweep on that
// many wavefronts
hipError_t launch_gibbs_exact_wave(const GxArgs& A, uint32_t n_waves, hipStream_t s);
// the same sweep with a block's arcs in registers (gibbs_reg_wave_kernel): trellis blocks of at most 64 nq arcs and states and
// GX_REG_LEVELS levels; nq = 1, 2, 4, 8 (the exact chain: 2, 4, 8)
#define GX_REG_LEVELS 127
size_t gibbs_reg_lds_bytes(uint32_t cap_arcs, uint32_t cap_states, uint32_t cap_sample, uint32_t own_slots);
hipError_t launch_gibbs_reg_wave(const GxArgs& A, uint32_t n_waves, int nq, hipStream_t s);
// dst[c * n + k] = src[k] for c < copies (every chain's cache model starts a sweep at the priors)
hipError_t launch_gibbs_broadcast(double* dst, const double* src, uint64_t n, uint32_t copies, hipStream_t s);

// the parallel sweep's recount through per-workgroup LDS tables: new_x / new_norm (set to the priors by the caller) += the weighted
// uses of the samples (len, ids, nrm)
hipError_t launch_gibbs_recount_tables(const GxBlock* blocks, const uint32_t* len, const uint32_t* ids, const uint32_t* nrm,
                                       uint32_t n_blocks, double* new_x, double* new_norm, hipStream_t s, const uint32_t* list = nullptr);

// normsum[g] = the sum of x over group g's members that carry counts (p_norm != none): the parallel sweep's norm sums from its new
// counts (new_norm = nullptr in the recounts: they add parameter counts only)
hipError_t launch_gibbs_normsum(const double* x, const uint32_t* p_norm, const uint64_t* group_off, const uint64_t* norm_perm, uint64_t n_groups,
                                double* normsum, hipStream_t s);

}  // namespace carmel_hip
