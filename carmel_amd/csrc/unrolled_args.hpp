// unrolled_args.hpp: device-side argument block of the unrolled sweep (unrolled.hpp / unrolled.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "unrolled.hpp"

namespace carmel_hip {

struct UnrolledArgs {
  uint32_t S, V, start, fin, n_eps, n_slots, max_len;
  uint64_t n_pairs;                // pairs with a derivation
  const uint32_t* f_off;           // V + 1
  const uint8_t* f_src;
  const double* Wf;                // linear weights in f_arc order (0 for padding)
  const uint32_t* b_off;
  const uint8_t* b_dst;
  const double* Wb;
  const uint16_t* b_slot;          // UNROLLED_MAX_CHAIN accumulator slots per entry
  const uint8_t* e_src;
  const uint8_t* e_dst;
  const double* We;
  const uint16_t* e_slot;
  const uint64_t* seq_off;
  const uint16_t* seq_sym;
  const uint32_t* pair_id;
  const double* pair_weight;
  double* pair_logprob;
  double* partial;                 // n_workgroups * n_slots
};

size_t unrolled_lds_bytes(const UnrolledArgs& A, uint32_t n_waves);
uint32_t unrolled_waves(uint32_t n_slots, uint32_t max_len, uint32_t S);
hipError_t launch_unrolled_weights(const uint32_t* arcs, const double* logw, double* out, uint32_t n, hipStream_t s);
hipError_t launch_unrolled_param_counts(double* out, const double* counts, const double* uses, double floor_count,
                                        const uint32_t* group, uint32_t n, hipStream_t s);
hipError_t launch_unrolled_sweep(const UnrolledArgs& A, uint32_t n_wg, double* counts, hipStream_t s);

}  // namespace carmel_hip
