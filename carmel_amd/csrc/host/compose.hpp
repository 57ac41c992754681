// host/compose.hpp — composition of two transducers with the 3-state epsilon filter, recording for every composed
// arc the CHAIN of original arcs (parameters) it was built from.  This is the producer of the composed arc table
// and of the chain ids the GPU path consumes (carmel_hip_set_cascade).
//
// Semantics follow /root/reference/carmel/src/compose.cc:163-531 (filter states :315-324; composite states numbered
// in discovery order from a LIFO work list :193,326-328; arcs prepended to their state :128-141; the larger of the
// two operand states is indexed when it has more than -T arcs :334-338; several finals joined by locked *e*:*e*
// arcs :503-530) and cascade.h:489-599 (which original arcs a composed arc stands for).
//
// run_a() is carmel -a (compose.cc:219-313): the 2-state filter that keeps the arcs of the two operands apart -- an arc
// a:x of A leads to a "mediate" state (A's destination, B's state, x) from which B's x:c arcs leave -- so every composed
// arc stands for ONE operand arc (chains of length 0 or 1 per composition).  The reference walks A's per-state
// output-symbol hash table in bucket order there (compose.cc:240-242); so does this one -- refhash.hpp replays the table --,
// so composite and mediate states get the reference's numbers (no reference fixture holds a -a composition: the restatement
// is checked against the oracle's and a third model of the table, tests/test_refhash.py).
//
// trivial = no --train-cascade (cascade.h:566-592 with `trivial`): composed arcs carry no chains -- an arc built from a
// pair has no group, an arc copied from one operand arc keeps that arc's own group.
#pragma once
#include <algorithm>
#include <map>
#include "../../../include/carmel_hip.h"
#include "wfst.hpp"
#include "refhash.hpp"

namespace carmel_host {

// parameters = arcs of the member transducers, numbered member by member in arc-id order
struct ParamTable {
  std::vector<double> logw;
  std::vector<uint32_t> group, member, src, in;
  std::vector<size_t> member_base;  // first param id of each member
  void add_member(const Transducer& t) {
    uint32_t m = (uint32_t)member_base.size();
    member_base.push_back(logw.size());
    for (uint32_t s = 0; s < t.states.size(); ++s)
      for (auto& a : t.states[s]) {
        logw.push_back(a.logw);
        group.push_back(a.group);
        member.push_back(m);
        src.push_back(s);
        in.push_back(a.in);
      }
  }
  bool locked_one(uint64_t p) const { return group[p] == kLocked && logw[p] == 0.0; }  // cascade.h:555
};

struct ChainTable {
  std::vector<std::vector<uint64_t> > chains;  // chains[0] = nil (every parameter locked at weight 1)
  std::unordered_map<uint64_t, uint32_t> lone;  // a single original arc used on its own: one shared chain
  ChainTable() { chains.emplace_back(); }
  uint32_t intern(std::vector<uint64_t>&& c) {
    if (c.empty()) return 0;
    chains.push_back(std::move(c));
    return (uint32_t)(chains.size() - 1);
  }
};

// An operand of a composition: either an original member (arc k of state s stands for parameter base + arc index)
// or an earlier composition result (arc.group is already a chain id).
struct Operand {
  const Transducer* t;
  bool is_chain;
  std::vector<size_t> state_first;  // per state: param id of its first arc (originals only)
  void bind(const Transducer* tr, bool chain, size_t param_base) {
    t = tr;
    is_chain = chain;
    state_first.clear();
    if (!chain) {
      size_t k = param_base;
      for (auto& st : tr->states) {
        state_first.push_back(k);
        k += st.size();
      }
    }
  }
  uint64_t param_of(uint32_t s, size_t idx) const { return state_first[s] + idx; }
};

class Composer {
 public:
  Composer(const ParamTable& params, ChainTable& chains, unsigned index_threshold = 32, bool trivial = false)
      : P(params), C(chains), T(index_threshold), trivial_(trivial) {}

  // carmel -a: see the header comment.  Same conventions as run().
  bool run_a(const Operand& A, const Operand& B, Transducer& out) {
    const Transducer& a = *A.t;
    const Transducer& b = *B.t;
    out = Transducer();
    if (a.states.empty() || b.states.empty()) return false;  // an operand without states: !valid() (compose.cc:176-179)
    out.in_syms = a.in_syms;
    out.out_syms = b.out_syms;
    out.named = false;
    std::vector<uint32_t> a2b(a.out_syms.names.size(), kNoGroup);
    for (uint32_t i = 0; i < a2b.size(); ++i) b.in_syms.find(a.out_syms.names[i], a2b[i]);
    std::unordered_map<uint64_t, uint32_t> ids;       // (qa, qb, filter) -> state
    std::map<std::vector<uint32_t>, uint32_t> mids;   // (A's destination, B's state, hidden symbol) -> mediate state
    std::vector<Job> work;
    std::vector<std::vector<HArc> >& S = out.states;
    auto key = [&](uint32_t qa, uint32_t qb, int f) { return ((uint64_t)qa * b.states.size() + qb) * 3 + (uint64_t)f; };
    S.emplace_back();
    ids.emplace(key(0, 0, 0), 0u);
    work.push_back(Job{0, 0, 0, 0});
    auto emit = [&](uint32_t from, uint32_t in, uint32_t o, uint32_t qa, uint32_t qb, int f, double lw, uint32_t chain) {
      auto ins = ids.emplace(key(qa, qb, f), (uint32_t)S.size());
      if (ins.second) {
        work.push_back(Job{ins.first->second, qa, qb, f});
        S.emplace_back();
      }
      HArc x;
      x.in = in;
      x.out = o;
      x.dest = ins.first->second;
      x.logw = lw;
      x.group = chain;
      S[from].push_back(x);
    };
    while (!work.empty()) {
      const Job j = work.back();
      work.pop_back();
      const auto& la = a.states[j.qa];
      const auto& lb = b.states[j.qb];
      // A's arcs by output symbol, B's by input symbol; within a symbol newest-first, like the push_front lists of
      // State::indexBy (state.h:158-199)
      std::map<uint32_t, std::vector<size_t> > by_out, by_in;
      for (size_t k = la.size(); k-- > 0;) by_out[la[k].out].push_back(k);
      for (size_t k = lb.size(); k-- > 0;) by_in[lb[k].in].push_back(k);
      // ... and A's output symbols in the order the reference's walk over qa->index visits them (compose.cc:240-242; the
      // index is State::indexBy(kOutput): a hash table made for the state's arc count, filled arc by arc in list order --
      // refhash.hpp): the composite and mediate states are numbered as that walk creates them
      std::vector<uint32_t> outs;
      for (auto& arc : la) outs.push_back(arc.out);
      for (uint32_t sym : conditional_group_order(outs)) {
        auto& kv = *by_out.find(sym);
        if (sym == 0) {
          if (j.f == 0)  // a:*e* of A alone, B stays (filter 0 -> 0)
            for (size_t ka : kv.second) emit(j.id, la[ka].in, 0, la[ka].dest, j.qb, 0, la[ka].logw, lone_chain(A, j.qa, ka));
          continue;
        }
        if (a2b[sym] == kNoGroup) continue;
        auto mb = by_in.find(a2b[sym]);
        if (mb == by_in.end()) continue;
        for (size_t ka : kv.second) {
          const std::vector<uint32_t> mk{la[ka].dest, j.qb, sym};
          auto ins = mids.emplace(mk, (uint32_t)S.size());
          const uint32_t M = ins.first->second;
          if (ins.second) {  // new mediate state: B's matching arcs leave it, input *e*
            S.emplace_back();
            for (size_t kb : mb->second) emit(M, 0, lb[kb].out, la[ka].dest, lb[kb].dest, 0, lb[kb].logw, lone_chain(B, j.qb, kb));
          }
          HArc x;  // A's arc into the mediate state, output *e*
          x.in = la[ka].in;
          x.out = 0;
          x.dest = M;
          x.logw = la[ka].logw;
          x.group = lone_chain(A, j.qa, ka);
          S[j.id].push_back(x);
        }
      }
      auto eb = by_in.find(0u);
      if (eb != by_in.end())  // *e*:c of B alone, A stays (-> filter 1)
        for (size_t kb : eb->second) emit(j.id, 0, lb[kb].out, j.qa, lb[kb].dest, 1, lb[kb].logw, lone_chain(B, j.qb, kb));
    }
    return finish(out, ids, key(a.final_state, b.final_state, 0));
  }

  // The same composition with the product construction done on the GPU (carmel_hip_compose, csrc/compose.hip): the
  // device expands every composite state in the reference's emission order and reports, per arc, which arc of A and / or
  // of B it was built from; what is sequential by definition is done here, in one pass over that output -- composite
  // states get the numbers the reference's LIFO work list would give them (compose.cc:193, 326-328) and chains are
  // created in emission order (cascade.h:507-599).  The result is, arc for arc, that of run().
  bool run_device(const Operand& A, const Operand& B, Transducer& out, int device, double* device_seconds = 0) {
    const Transducer& a = *A.t;
    const Transducer& b = *B.t;
    out = Transducer();
    if (a.states.empty() || b.states.empty()) return false;
    out.in_syms = a.in_syms;
    out.out_syms = b.out_syms;
    out.named = false;
    std::vector<uint32_t> a2b(a.out_syms.names.size(), kNoGroup), b2a(b.in_syms.names.size(), kNoGroup);
    for (uint32_t i = 0; i < a2b.size(); ++i) b.in_syms.find(a.out_syms.names[i], a2b[i]);
    for (uint32_t i = 0; i < b2a.size(); ++i) a.out_syms.find(b.in_syms.names[i], b2a[i]);
    struct Flat {
      std::vector<uint64_t> off;
      std::vector<uint32_t> in, out, dst;
      std::vector<double> lw;
      explicit Flat(const Transducer& t) {
        off.push_back(0);
        for (auto& st : t.states) {
          for (auto& x : st) {
            in.push_back(x.in);
            out.push_back(x.out);
            dst.push_back(x.dest);
            lw.push_back(x.logw);
          }
          off.push_back(in.size());
        }
        if (in.empty()) {  // keep the pointers valid
          in.push_back(0);
          out.push_back(0);
          dst.push_back(0);
          lw.push_back(0);
        }
      }
    } fa(a), fb(b);
    carmel_hip_composition* c = 0;
    if (carmel_hip_compose(&c, device, (uint32_t)a.states.size(), fa.off.data(), fa.in.data(), fa.out.data(), fa.dst.data(),
                           fa.lw.data(), (uint32_t)b.states.size(), fb.off.data(), fb.in.data(), fb.out.data(), fb.dst.data(),
                           fb.lw.data(), a2b.data(), (uint32_t)a2b.size(), b2a.data(), (uint32_t)b2a.size(), T) != CARMEL_HIP_OK)
      throw std::runtime_error(std::string("carmel_hip_compose: ") + carmel_hip_last_error());
    const size_t ns = carmel_hip_composition_states(c), na = carmel_hip_composition_arcs(c);
    if (device_seconds) *device_seconds = carmel_hip_composition_seconds(c);
    std::vector<uint64_t> off(ns + 1);
    std::vector<uint32_t> qa(ns), qb(ns), ain(na ? na : 1), aout(na ? na : 1), adst(na ? na : 1), aka(na ? na : 1), akb(na ? na : 1);
    std::vector<uint8_t> fl(ns);
    std::vector<double> alw(na ? na : 1);
    const int rc = carmel_hip_composition_export(c, off.data(), qa.data(), qb.data(), fl.data(), ain.data(), aout.data(), adst.data(),
                                                 alw.data(), aka.data(), akb.data());
    carmel_hip_composition_free(c);
    if (rc != CARMEL_HIP_OK) throw std::runtime_error(std::string("carmel_hip_composition_export: ") + carmel_hip_last_error());
    std::vector<std::vector<HArc> >& S = out.states;
    std::vector<uint32_t> id(ns, kNoGroup), work;
    S.emplace_back();
    id[0] = 0;
    work.push_back(0);
    while (!work.empty()) {
      const uint32_t t = work.back();
      work.pop_back();
      const uint32_t cur = id[t];
      for (uint64_t e = off[t]; e < off[t + 1]; ++e) {
        const uint32_t d = adst[e];
        if (id[d] == kNoGroup) {
          id[d] = (uint32_t)S.size();
          S.emplace_back();
          work.push_back(d);
        }
        HArc x;
        x.in = ain[e];
        x.out = aout[e];
        x.dest = id[d];
        x.logw = alw[e];
        x.group = aka[e] != kNoGroup && akb[e] != kNoGroup ? pair_chain(A, qa[t], aka[e], B, qb[t], akb[e])
                  : aka[e] != kNoGroup                       ? lone_chain(A, qa[t], aka[e])
                                                             : lone_chain(B, qb[t], akb[e]);
        S[cur].push_back(x);
      }
    }
    auto key = [&](uint32_t x, uint32_t y, int f) { return ((uint64_t)x * b.states.size() + y) * 3 + (uint64_t)f; };
    std::unordered_map<uint64_t, uint32_t> finals;
    for (size_t t = 0; t < ns; ++t)
      if (qa[t] == a.final_state && qb[t] == b.final_state && id[t] != kNoGroup) finals.emplace(key(qa[t], qb[t], fl[t]), id[t]);
    return finish(out, finals, key(a.final_state, b.final_state, 0));
  }

  // returns false when the composition is empty (no final reachable)
  bool run(const Operand& A, const Operand& B, Transducer& out) {
    const Transducer& a = *A.t;
    const Transducer& b = *B.t;
    out = Transducer();
    if (a.states.empty() || b.states.empty()) return false;
    out.in_syms = a.in_syms;
    out.out_syms = b.out_syms;
    out.named = false;
    // interface alphabet: a's output symbol id -> b's input symbol id (strhash.h:253-256)
    std::vector<uint32_t> a2b(a.out_syms.names.size(), kNoGroup), b2a(b.in_syms.names.size(), kNoGroup);
    for (uint32_t i = 0; i < a2b.size(); ++i) b.in_syms.find(a.out_syms.names[i], a2b[i]);
    for (uint32_t i = 0; i < b2a.size(); ++i) a.out_syms.find(b.in_syms.names[i], b2a[i]);
    std::unordered_map<uint64_t, uint32_t> ids;
    std::vector<Job> work;
    std::vector<std::vector<HArc> >& S = out.states;
    auto key = [&](uint32_t qa, uint32_t qb, int f) { return ((uint64_t)qa * b.states.size() + qb) * 3 + (uint64_t)f; };
    S.emplace_back();
    ids.emplace(key(0, 0, 0), 0u);
    work.push_back(Job{0, 0, 0, 0});
    uint32_t cur = 0;
    auto emit = [&](uint32_t in, uint32_t o, uint32_t qa, uint32_t qb, int f, double lw, uint32_t chain) {
      auto ins = ids.emplace(key(qa, qb, f), (uint32_t)S.size());
      if (ins.second) {
        work.push_back(Job{ins.first->second, qa, qb, f});
        S.emplace_back();
      }
      HArc x;
      x.in = in;
      x.out = o;
      x.dest = ins.first->second;
      x.logw = lw;
      x.group = chain;
      S[cur].push_back(x);  // creation order; reversed at the end (the reference prepends)
    };
    // per-state symbol indexes, built on demand; each bucket lists arc positions newest-first like the
    // reference's push_front lists (state.h:158-199)
    std::vector<Index> ai(a.states.size()), bi(b.states.size());
    auto index_of = [&](const Transducer& t, uint32_t q, bool by_out, Index& ix) -> Index& {
      if (!ix.built) {
        ix.built = true;
        const auto& arcs = t.states[q];
        for (size_t k = 0; k < arcs.size(); ++k) ix.m[by_out ? arcs[k].out : arcs[k].in].push_back(k);
        for (auto& kv : ix.m) std::reverse(kv.second.begin(), kv.second.end());
      }
      return ix;
    };
    static const std::vector<size_t> none;
    auto bucket = [&](Index& ix, uint32_t sym) -> const std::vector<size_t>& {
      if (sym == kNoGroup) return none;
      auto it = ix.m.find(sym);
      return it == ix.m.end() ? none : it->second;
    };
    while (!work.empty()) {
      Job j = work.back();
      work.pop_back();
      cur = j.id;
      const auto& la = a.states[j.qa];
      const auto& lb = b.states[j.qb];
      const bool a_bigger = la.size() > lb.size();
      const size_t big = a_bigger ? la.size() : lb.size();
      auto both = [&](size_t ka, size_t kb) {  // a:x from a and x:c from b (x may be *e* on both sides)
        emit(la[ka].in, lb[kb].out, la[ka].dest, lb[kb].dest, 0, la[ka].logw + lb[kb].logw, pair_chain(A, j.qa, ka, B, j.qb, kb));
      };
      auto a_alone = [&](size_t ka) {  // a:*e* taken without b moving -> filter 1
        emit(la[ka].in, 0, la[ka].dest, j.qb, 1, la[ka].logw, lone_chain(A, j.qa, ka));
      };
      auto b_alone = [&](size_t kb) {  // *e*:c taken without a moving -> filter 2
        emit(0, lb[kb].out, j.qa, lb[kb].dest, 2, lb[kb].logw, lone_chain(B, j.qb, kb));
      };
      if (big > T && !a_bigger) {  // b indexed by input, a walked in order (compose.cc:339-385)
        Index& ix = index_of(b, j.qb, false, bi[j.qb]);
        for (size_t ka = 0; ka < la.size(); ++ka) {
          if (la[ka].out == 0) {
            if (j.f != 2) a_alone(ka);
            if (j.f == 0)
              for (size_t kb : bucket(ix, 0)) both(ka, kb);
          } else
            for (size_t kb : bucket(ix, a2b[la[ka].out])) both(ka, kb);
        }
        if (j.f != 1)
          for (size_t kb : bucket(ix, 0)) b_alone(kb);
      } else if (big > T) {  // a indexed by output, b walked in order (compose.cc:386-436)
        Index& ix = index_of(a, j.qa, true, ai[j.qa]);
        for (size_t kb = 0; kb < lb.size(); ++kb) {
          if (lb[kb].in == 0) {
            if (j.f != 1) b_alone(kb);
            if (j.f == 0)
              for (size_t ka : bucket(ix, 0)) both(ka, kb);
          } else
            for (size_t ka : bucket(ix, b2a[lb[kb].in])) both(ka, kb);
        }
        if (j.f != 2)
          for (size_t ka : bucket(ix, 0)) a_alone(ka);
      } else {  // small states: nested loops (compose.cc:437-487)
        for (size_t ka = 0; ka < la.size(); ++ka) {
          if (la[ka].out == 0) {
            if (j.f != 2) a_alone(ka);
            if (j.f == 0)
              for (size_t kb = 0; kb < lb.size(); ++kb)
                if (lb[kb].in == 0) both(ka, kb);
          } else
            for (size_t kb = 0; kb < lb.size(); ++kb)
              if (a2b[la[ka].out] == lb[kb].in) both(ka, kb);
        }
        if (j.f != 1)
          for (size_t kb = 0; kb < lb.size(); ++kb)
            if (lb[kb].in == 0) b_alone(kb);
      }
    }
    return finish(out, ids, key(a.final_state, b.final_state, 0));
  }

 private:
  // finals (compose.cc:503-530): key0 = key of (a.final, b.final, filter 0); filters are consecutive keys
  bool finish(Transducer& out, const std::unordered_map<uint64_t, uint32_t>& ids, uint64_t key0) {
    std::vector<std::vector<HArc> >& S = out.states;
    int found[3] = {-1, -1, -1}, n = 0;
    for (int f = 0; f < 3; ++f) {
      auto it = ids.find(key0 + (uint64_t)f);
      if (it != ids.end()) {
        found[f] = (int)it->second;
        out.final_state = it->second;
        ++n;
      }
    }
    if (!n) return false;
    if (n > 1) {
      out.final_state = (uint32_t)S.size();
      S.emplace_back();
      for (int f = 0; f < 3; ++f)
        if (found[f] >= 0) {
          HArc x;
          x.dest = out.final_state;
          x.group = 0;  // nil chain == locked at weight 1 (locked_1_groupid is 0 with or without chains)
          S[found[f]].push_back(x);
        }
    }
    for (auto& st : S) std::reverse(st.begin(), st.end());
    return true;
  }

  struct Job {
    uint32_t id, qa, qb;
    int f;
  };
  struct Index {
    bool built = false;
    std::unordered_map<uint32_t, std::vector<size_t> > m;
  };
  const ParamTable& P;
  ChainTable& C;
  unsigned T;
  bool trivial_;

  void prepend(std::vector<uint64_t>& chain, uint64_t p) const {
    if (!P.locked_one(p)) chain.insert(chain.begin(), p);  // cascade.h:507-510
  }
  // chain of a composed arc made from arc ka of A's state qa and arc kb of B's state qb (cascade.h:536-551)
  uint32_t pair_chain(const Operand& A, uint32_t qa, size_t ka, const Operand& B, uint32_t qb, size_t kb) {
    if (trivial_) return kNoGroup;  // cascade.h:581-583
    std::vector<uint64_t> c;
    const uint32_t ga = A.t->states[qa][ka].group, gb = B.t->states[qb][kb].group;
    if (A.is_chain) {
      if (B.is_chain) {
        c = C.chains[gb];
        for (uint64_t p : C.chains[ga]) prepend(c, p);  // a's items go in front one by one (reversed)
      } else {
        c = C.chains[ga];
        prepend(c, B.param_of(qb, kb));
      }
    } else {
      if (B.is_chain)
        c = C.chains[gb];
      else
        prepend(c, B.param_of(qb, kb));
      prepend(c, A.param_of(qa, ka));
    }
    return C.intern(std::move(c));
  }
  // chain of a composed arc that copies one operand arc (cascade.h:566-579)
  uint32_t lone_chain(const Operand& X, uint32_t q, size_t k) {
    if (trivial_ || X.is_chain) return X.t->states[q][k].group;  // cascade.h:566-569
    uint64_t p = X.param_of(q, k);
    auto it = C.lone.find(p);
    if (it != C.lone.end()) return it->second;
    std::vector<uint64_t> c;
    prepend(c, p);
    uint32_t id = C.intern(std::move(c));
    C.lone.emplace(p, id);
    return id;
  }
};

}  // namespace carmel_host
