// dense.hpp: the unrolled sweep (unrolled.hpp) for one-tape cascades whose composed arcs factor as
//     weight(s -> s', symbol c) = A[s][s'] * B[c][s'],
// A a product of parameters that collect no counts (a locked or un-normalised language model) and B of the trainable
// parameters (a channel model): the decipherment cascades of carmel/sample/decipher and the tutorial (character LM o
// substitution channel).  A position of a string is then a DENSE S x S vector-matrix product
//     alpha_{t+1} = (alpha_t . A) (*) B[c_t]
// (followed by the few *e*:*e* arcs inside the position, e.g. the language model's end-of-text arc) instead of a walk over
// arc tables: one string per lane, alpha in registers, A streamed through the scalar unit
// (the same value for all 64 lanes), B and the count accumulators in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carmel_hip {

static const uint32_t DENSE_MAX_STATES = 32;

struct DenseArgs {
  uint32_t S, SP, V, start, fin, n_slots, n_eps;
  uint32_t debug;           // timing experiments (set in the source): 1 no posterior adds, 2 no parked values, 4 no backward pass
  const uint16_t* e_src;    // *e*:*e* arcs in topological order of their sources (they collect no counts here)
  const uint16_t* e_dst;
  const double* We;         // their linear weights
  const double* A;          // [SP][SP] row-major, zero where there is no arc / padding
  const double* AT;         // its transpose
  const double* B;          // [V][SP]
  const uint16_t* Bslot;    // [V][SP] accumulator slot of B[c][s'], 0xffff = none
  const uint16_t* sym;      // per group of 64 strings: rows of 64 symbols, row t at sym_off[g] + t * 64
  const uint64_t* sym_off;  // n_groups + 1
  const uint32_t* len;      // [n_groups * 64] string lengths (0: empty lane)
  const uint32_t* pair;     // [n_groups * 64] pair id
  const double* weight;     // [n_groups * 64] pair weight
  double* pair_logprob;
  double* vbuf;             // per group: (t * SP + j) * 64 + lane, from vbuf_off[g]
  const uint64_t* vbuf_off;
  double* zbuf;             // same indexing as sym
  double* afbuf;            // [n_groups * 64] alpha_T[goal] per string, between the two launches of the split sweep (may be null)
  double* partial;          // [n_groups][n_slots]
};

// A, AT, B from the parameter weights: entry e is the product of the parameters list[off[e] .. off[e + 1]) (empty: 0)
hipError_t launch_dense_tables(double* A, double* AT, double* B, uint32_t SP, uint32_t V, const uint32_t* a_off,
                               const uint32_t* a_par, const uint32_t* b_off, const uint32_t* b_par, const uint8_t* a_has,
                               const uint8_t* b_has, const double* param_logw, hipStream_t s);
hipError_t launch_dense_sweep(const DenseArgs& D, uint32_t n_groups, hipStream_t s);
uint32_t dense_padded_states(uint32_t S);

}  // namespace carmel_hip
