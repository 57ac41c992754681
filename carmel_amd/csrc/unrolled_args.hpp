// unrolled_args.hpp: device-side argument block of the unrolled sweep (unrolled.hpp / unrolled.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "unrolled.hpp"

namespace carmel_hip {

// one table entry, one 16-byte load: the arc's linear weight, the state at its other end, its accumulator slots
struct __attribute__((aligned(16))) URec {
  double w;
  uint32_t other_slot2;  // bits 0-9: source (forward table) / destination (backward table); bits 16-31: slot 2
  uint32_t slot01;       // bits 0-15: slot 0, bits 16-31: slot 1
};

struct UnrolledArgs {
  uint32_t S, V, start, fin, n_eps, n_slots, max_len;
  uint32_t f_deg_u, b_deg_u;       // rows per slab when all slabs have one size (0: ragged slabs, see f_off / b_off)
  uint64_t n_pairs;                // pairs with a derivation
  const uint32_t* f_off;           // V + 1, in rows of S entries
  const URec* f_rec;               // forward table (by destination); padding entries have weight 0
  const uint32_t* b_off;
  const URec* b_rec;               // backward table (by source), with the accumulator slots
  const uint16_t* e_src;
  const uint16_t* e_dst;
  const double* We;
  const uint16_t* e_slot;
  const uint64_t* seq_off;
  const uint16_t* seq_sym;
  const uint32_t* pair_id;
  const double* pair_weight;
  double* pair_logprob;
  double* partial;                 // n_workgroups * n_slots
  uint32_t debug_no_acc;           // timing experiment (set in the source): skip the posterior adds
  double* alpha_scratch;           // unrolled_scratch_doubles(): per wave (max_len + 1) rows of 64
};

size_t unrolled_lds_bytes(const UnrolledArgs& A, uint32_t n_waves);
uint32_t unrolled_waves(uint32_t n_slots, uint32_t max_len, uint32_t S);
size_t unrolled_scratch_doubles(uint32_t n_wg, uint32_t n_waves, uint32_t max_len);
hipError_t launch_unrolled_weights(const uint32_t* arcs, const double* logw, double* out, uint32_t stride_doubles, uint32_t n,
                                   hipStream_t s);
hipError_t launch_unrolled_param_counts(double* out, const double* counts, const double* uses, double floor_count,
                                        const double* wprior, const uint32_t* group, const uint32_t* slot_of, uint32_t n,
                                        hipStream_t s);
hipError_t launch_unrolled_sweep(const UnrolledArgs& A, uint32_t n_wg, double* counts, hipStream_t s);
// counts[k] = sum over the n_wg rows of partial[row][k]
hipError_t launch_unrolled_reduce(double* partial, uint32_t n_wg, uint32_t n_slots, double* counts, hipStream_t s);
// more than 64 states: a workgroup per pair, a thread per state
size_t unrolled_wide_lds_bytes(const UnrolledArgs& A);
size_t unrolled_wide_scratch_doubles(uint32_t n_wg, uint32_t S, uint32_t max_len);

}  // namespace carmel_hip
