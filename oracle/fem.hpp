// fem.hpp -- ORACLE (test infrastructure only).  CPU restatement of carmel's forest-em export (--fem-forest,
// --fem-norm, --fem-param, --fem-alpha): the bridge from WFST cascades to forest-em's packed forests.
// Follows /root/reference:
//   carmel/src/cascade.h:34-51    arcids: 1-based ids over the cascade's members in order, arcs state-major
//   carmel/src/cascade.h:60-82    fem_alpha: -1 for a locked arc / a member normalised by NONE, else the member's prior
//   carmel/src/cascade.h:85-116   fem_norms: "(" then per member a newline and its norm groups "( id id ... )"
//   carmel/src/cascade.h:117-165  fem_deriv: a pair's derivation lattice as a forest; a lattice state with several
//                                 in-arcs is a shared sub-forest (#k definition on first visit, #k afterwards)
//   graehl/shared/graph.h:165-194 backref / backrefs: which states are used more than once, ids in DFS order
//   carmel/src/cascade.h:167-178  print_params: the weights, one per line, in arcids order
//   carmel/src/cached_derivs.h:60-100  the forests are written on the first pass over the derivations
#pragma once
#include <sstream>

#include "cascade.hpp"
#include "deriv.hpp"

namespace oracle {

struct FemExport {
  Cascade& cascade;
  Wfst& composed;
  std::unordered_map<const Arc*, unsigned> aid;

  FemExport(Cascade& c, Wfst& x) : cascade(c), composed(x) {
    unsigned id = 1;
    for (Wfst* w : cascade.cascade)
      for (auto& st : w->states)
        for (auto& a : st) aid[&a] = id++;
  }

  std::string params() const {
    std::ostringstream o;
    for (Wfst* w : cascade.cascade)
      for (auto& st : w->states)
        for (auto& a : st) o << lw_str(a.weight) << "\n";
    return o.str();
  }
  std::string alphas(const std::vector<NormalizeMethod>& methods) const {
    std::ostringstream o;
    for (size_t i = 0; i < cascade.cascade.size(); ++i) {
      double prior = methods[i].group == NORM_NONE ? -1.0 : methods[i].add_count.getReal();
      for (auto& st : cascade.cascade[i]->states)
        for (auto& a : st) o << (a.locked() ? -1.0 : prior) << '\n';
    }
    return o.str();
  }
  std::string norms(const std::vector<NormalizeMethod>& methods) const {
    std::ostringstream o;
    o << "(";
    for (size_t i = 0; i < cascade.cascade.size(); ++i) {
      o << "\n";
      if (methods[i].group == NORM_NONE) continue;
      cascade.cascade[i]->for_each_norm_group(methods[i].group, [&](unsigned, std::vector<Arc*>& g) {
        o << '(';
        for (Arc* a : g) o << ' ' << aid.at(a);
        o << " )\n";
      });
    }
    o << ")\n";
    return o.str();
  }

  // chain of a composed arc as arc ids (trivial cascade: the arc itself, cascade.h:233-239)
  std::vector<unsigned> chain_ids(const Arc* composed_arc) const {
    std::vector<unsigned> r;
    if (cascade.trivial)
      r.push_back(aid.at(composed_arc));
    else
      for (Arc* p : cascade.chains[composed_arc->group]) r.push_back(aid.at(p));
    return r;
  }

  struct Shared {
    unsigned uses = 0, id = 0;
  };
  void use(const Derivations& d, std::vector<Shared>& ids, unsigned& next_label, unsigned s) const {
    Shared& b = ids[s];
    if (b.uses++ > 0) {
      b.id = next_label++;
      return;
    }
    const auto& arcs = d.g[s];
    for (size_t k = arcs.size(); k-- > 0;) use(d, ids, next_label, arcs[k].dest);  // list order
  }
  void deriv_rec(std::ostream& o, const Derivations& d, const ArcTable& arcs, std::vector<Shared>& br, unsigned s) const {
    Shared& b = br[s];
    const bool defining = b.uses > 1;
    if (defining) {
      o << "#" << b.id;
      b.uses = 0;  // BACKREF_DEFINED
    } else if (b.uses == 0) {
      o << "#" << b.id;
      return;
    }
    const auto& st = d.g[s];
    const bool alternatives = st.size() >= 2;
    if (alternatives) o << "(OR";
    for (size_t k = st.size(); k-- > 0;) {
      if (alternatives) o << " ";
      const GArc& a = st[k];
      std::vector<unsigned> p = chain_ids(arcs.t[a.arcid].arc);
      const unsigned n = a.dest;
      const bool inner = n != d.fin;
      const bool wrap = defining || (!p.empty() && (p.size() > 1 || inner));
      if (wrap) o << "(";
      bool first = true;
      for (unsigned id : p) {
        if (!first) o << ' ';
        first = false;
        o << id;
      }
      if (inner) {
        if (!first) o << ' ';
        deriv_rec(o, d, arcs, br, n);
      }
      if (wrap) o << ")";
    }
    if (alternatives) o << ")";
  }
  // one line per training pair that has a derivation (cached_derivs.h:60-100)
  std::string forests(Corpus& corpus) {
    std::ostringstream o;
    ArcTable arcs;
    arcs.build(composed, false, LW());
    IoIndex io;
    io.build(composed);
    for (auto& p : corpus.examples) {
      Derivations d;
      if (!d.compute(composed, io, arcs, p, true, 0)) continue;
      std::vector<Shared> br(d.g.size());
      unsigned next_label = 1;
      use(d, br, next_label, 0);
      deriv_rec(o, d, arcs, br, 0);
      o << "\n";
    }
    return o.str();
  }
};

}  // namespace oracle
