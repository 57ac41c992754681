// unrolled.hpp: the derivation lattices of a ONE-TAPE transducer, never materialised.
//
// When every arc of the (composed) transducer reads nothing and writes one symbol -- or the mirror image -- plus
// possibly some *e*:*e* arcs that form no cycle, the derivation lattice of a pair (derivations.h:45-66, 640-704) is
// the transducer unrolled over the positions of the one non-empty string: node (o, s), an arc labelled x_o from
// (o, src) to (o+1, dst), *e*:*e* arcs inside a position.  This is the shape of the decipherment cascades
// (carmel/sample/decipher, tutorial cipher.*: character LM o substitution channel), where explicit
// lattices would hold 8e9 arcs for 200 000 lines.  The sweep kernel walks positions instead of stored arcs: all it
// needs per position is the symbol.
//
// Tables (ELL, one slab per symbol): the k-th in-arc of every destination state sits at [x][k][dst] (forward and
// posterior pass, one lane per destination), the k-th out-arc of every source at [x][k][src] (backward pass).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "lattice.hpp"

namespace carmel_hip {

static const uint32_t UNROLLED_MAX_STATES = 64;     // one lane per state (unrolled_sweep_kernel)
static const uint32_t UNROLLED_WIDE_MAX_STATES = 1024;  // one thread per state, a workgroup per pair (unrolled_wide_kernel)
static const uint32_t UNROLLED_MAX_SLOTS = 6144;    // count accumulators in LDS (48 KB of f64)
static const uint32_t UNROLLED_MAX_CHAIN = 3;       // accumulator slots per arc
static const uint32_t UNROLLED_NO_SLOT = 0xffffu;

struct UnrolledModel {
  bool ok = false;
  std::string why;          // why the model / corpus is not eligible
  int tape = 1;             // 0: the input string carries the symbols, 1: the output string
  uint32_t S = 0, V = 0, start = 0, fin = 0;
  std::vector<uint32_t> f_off;   // V + 1: entry offset of symbol x's slab (f_deg[x] * S entries)
  std::vector<uint32_t> f_arc;   // arc id or 0xffffffff (padding)
  std::vector<uint16_t> f_src;
  uint32_t f_deg_u = 0, b_deg_u = 0;  // rows per slab when every symbol's slab has the same size (0: ragged)
  std::vector<uint32_t> b_off, b_arc;
  std::vector<uint16_t> b_dst;
  std::vector<uint32_t> e_arc;   // *e*:*e* arcs in topological order of their sources
  std::vector<uint16_t> e_src, e_dst;
  // corpus: pairs with a derivation, their symbol strings (dense symbol ids)
  std::vector<uint32_t> pair_id;
  std::vector<uint64_t> seq_off;
  std::vector<uint16_t> seq_sym;
  std::vector<double> pair_weight;
  std::vector<uint8_t> has_deriv;
  uint32_t max_len = 0;
  uint64_t lattice_states = 0, lattice_arcs = 0, explored_arcs = 0;  // what the explicit lattices would hold (pruned / unpruned)
};

// slots: per arc up to UNROLLED_MAX_CHAIN accumulator ids (n_slots accumulators in all), UNROLLED_NO_SLOT = none
bool build_unrolled(const HostWfst& w, const HostCorpus& c, int threads, UnrolledModel& out);

}  // namespace carmel_hip
