// fem_export.hpp: carmel's forest-em export (--fem-forest, --fem-norm, --fem-param, --fem-alpha): the bridge from a
// WFST cascade and its training pairs to forest-em's packed forests (forest_em_main.cpp reads them back).
// Replaces /root/reference/carmel/src/cascade.h:34-51 (arc ids: 1-based over the members in order, arcs state-major),
// :60-82 (fem_alpha), :85-116 (fem_norms), :117-165 (fem_deriv), :167-178 (print_params) and
// graehl/shared/graph.h:165-194 (backrefs: a lattice state with several uses is a shared sub-forest #k).
// The derivation lattices come from the library's host lattice builder (carmel_hip_host_build, one pair per bundle):
// the same builder the GPU path uploads from, which keeps every state's out-arcs in the reference's list order.
#pragma once
#include <cstdint>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/carmel_hip.h"

namespace carmel_host {

struct FemLatticeRecord {  // BundleDesc of lattice.hpp (64 bytes), as carmel_hip_host_export hands it out
  uint64_t in_base, out_base, off_base;
  uint32_t n_states, n_levels, level_base, pair_base, n_pairs, flags;
  uint64_t n_arcs, pad;
};
static_assert(sizeof(FemLatticeRecord) == 64, "lattice record layout");

struct FemExport {
  // composed transducer, flattened (arc id = state-major position), chain id per arc and the chains as parameter ids
  uint32_t n_states = 0, final_state = 0;
  const std::vector<uint32_t>*src = nullptr, *dst = nullptr, *in = nullptr, *out = nullptr, *group = nullptr;
  const std::vector<std::vector<uint64_t> >* chains = nullptr;  // null: every arc is its own parameter

  void chain_ids(uint32_t arc, std::vector<uint64_t>& ids) const {
    ids.clear();
    if (!chains)
      ids.push_back((uint64_t)arc + 1);
    else
      for (uint64_t p : (*chains)[(*group)[arc]]) ids.push_back(p + 1);
  }

  struct Shared {
    uint32_t uses = 0, id = 0;
  };

  // one forest per pair with a derivation, in corpus order
  void write_forests(std::ostream& o, uint64_t n_pairs, const uint64_t* in_off, const uint32_t* cin, const uint64_t* out_off,
                     const uint32_t* cout, const double* weight) const {
    carmel_hip_host_lattices* h = nullptr;
    if (carmel_hip_host_build(&h, n_states, final_state, src->size(), src->data(), dst->data(), in->data(), out->data(),
                              n_pairs, in_off, cin, out_off, cout, weight, 1, 0, 1, 0, 0) != CARMEL_HIP_OK)
      throw std::runtime_error(std::string("--fem-forest: ") + carmel_hip_last_error());
    struct Free {
      carmel_hip_host_lattices* h;
      ~Free() { carmel_hip_host_free(h); }
    } guard{h};
    uint64_t dims[19];
    carmel_hip_host_dims(h, dims);
    std::vector<FemLatticeRecord> bundles(dims[0]);
    std::vector<uint32_t> in_arcs(2 * dims[2]), out_arcs(2 * dims[2]), ioff(dims[1]), ooff(dims[1]), lvl(dims[3]), pstart(dims[4]),
        pfinal(dims[4]), pid(dims[4]), classes(5 * dims[5] + 5);
    std::vector<double> plogw(dims[4]);
    std::vector<uint8_t> has(n_pairs);
    carmel_hip_host_export(h, bundles.data(), in_arcs.data(), out_arcs.data(), ioff.data(), ooff.data(), lvl.data(),
                           pstart.data(), pfinal.data(), pid.data(), plogw.data(), classes.data(), has.data());
    std::vector<uint32_t> bundle_of(n_pairs, 0xffffffffu);
    for (size_t b = 0; b < bundles.size(); ++b) bundle_of[pid[bundles[b].pair_base]] = (uint32_t)b;
    std::vector<Shared> br;
    std::vector<uint64_t> ids;
    for (uint64_t p = 0; p < n_pairs; ++p) {
      if (bundle_of[p] == 0xffffffffu) continue;
      const FemLatticeRecord& B = bundles[bundle_of[p]];
      if (B.flags & 1u) throw std::runtime_error("--fem-forest: a derivation lattice has a cycle");
      const uint32_t* oa = out_arcs.data() + 2 * B.out_base;  // {dst state, arc id}
      const uint32_t* off = ooff.data() + B.off_base;
      const uint32_t start = pstart[B.pair_base], fin = pfinal[B.pair_base];
      br.assign(B.n_states, Shared());
      uint32_t next_label = 1;
      use(oa, off, br, next_label, start);
      rec(o, oa, off, br, ids, start, fin);
      o << "\n";
    }
  }

  // graph.h:178-194.  The reference's lists are newest first: a state's out-arcs are walked from the back.
  void use(const uint32_t* oa, const uint32_t* off, std::vector<Shared>& br, uint32_t& next_label, uint32_t s) const {
    std::vector<std::pair<uint32_t, uint32_t> > stack;  // explicit stack: lattices can be thousands of states deep
    Shared& b0 = br[s];
    if (b0.uses++ > 0) {
      b0.id = next_label++;
      return;
    }
    stack.emplace_back(s, off[s + 1]);
    while (!stack.empty()) {
      auto& top = stack.back();
      if (top.second == off[top.first]) {
        stack.pop_back();
        continue;
      }
      const uint32_t a = --top.second;
      const uint32_t d = oa[2 * a];
      Shared& b = br[d];
      if (b.uses++ > 0)
        b.id = next_label++;
      else
        stack.emplace_back(d, off[d + 1]);
    }
  }
  // cascade.h:131-165
  void rec(std::ostream& o, const uint32_t* oa, const uint32_t* off, std::vector<Shared>& br, std::vector<uint64_t>& ids,
           uint32_t s, uint32_t fin) const {
    Shared& b = br[s];
    const bool defining = b.uses > 1;
    if (defining) {
      o << "#" << b.id;
      b.uses = 0;  // defined
    } else if (b.uses == 0) {
      o << "#" << b.id;
      return;
    }
    const uint32_t a0 = off[s], a1 = off[s + 1];
    const bool alternatives = a1 - a0 >= 2;
    if (alternatives) o << "(OR";
    for (uint32_t a = a1; a-- > a0;) {
      if (alternatives) o << " ";
      chain_ids(oa[2 * a + 1], ids);
      const uint32_t n = oa[2 * a];
      const bool inner = n != fin;
      const bool wrap = defining || (!ids.empty() && (ids.size() > 1 || inner));
      if (wrap) o << "(";
      bool first = true;
      for (uint64_t id : ids) {
        if (!first) o << ' ';
        first = false;
        o << id;
      }
      if (inner) {
        if (!first) o << ' ';
        rec(o, oa, off, br, ids, n, fin);
      }
      if (wrap) o << ")";
    }
    if (alternatives) o << ")";
  }
};

}  // namespace carmel_host
