// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// cascade.hpp: restatement of carmel's WFST composition (default 3-state epsilon filter) and of the
// cascade bookkeeping that maps each composed arc to the chain of original arcs it was built from.
// Follows /root/reference/carmel/src/:
//   compose.cc:163-531  WFST::set_compose (3-state filter :315-498; LIFO queue :193,326-328; arc prepend via
//                       COMPOSEARC_GROUP :128-141; multiple finals :503-530).  The `-a` 2-state filter
//                       (:219-313) is compose_a below; the reference walks a HashTable in bucket order there
//                       (refhash.hpp restates that order; no fixture of the reference holds a -a result).
//   compose.h:10-45     TrioKey (qa,qb,filter)
//   state.h:158-199     indexBy: per-key lists built with push_front => matches come in REVERSED arc order
//   cascade.h:489-599   record / record1 / record2 / record_eps / cons / cons_chain (chain ids = groupId)
//   cascade.h:426-479   calculate_chain_weights / update
//   cascade.h:286-364   distribute_counts / use_counts / use_counts_final / save_none / load_none
//   carmel.cc:1303-1355 composition loop (cascade.add, prepare_compose, reduce via shrink, done_composing)
#pragma once
#include "wfst.hpp"
#include <map>
#include <tuple>

namespace oracle {

struct Cascade {
  bool trivial;
  std::vector<Wfst*> cascade;
  Wfst* composed;
  typedef std::vector<Arc*> Chain;  // list order
  std::vector<Chain> chains;
  std::vector<LW> chain_weights;
  unsigned nil_chain;
  std::unordered_map<Arc*, unsigned> epsilon_chains;
  bool is_chain[2];
  std::vector<std::vector<double> > none_saves;

  explicit Cascade(bool remember = false) : trivial(!remember), composed(0), nil_chain(0) {
    is_chain[0] = is_chain[1] = false;
    if (trivial) return;
    nil_chain = (unsigned)chains.size();
    chains.push_back(Chain());  // cascade.h:395-400 canonical nil chain == locked_group 0
  }
  unsigned size() const { return trivial ? 1 : (unsigned)cascade.size(); }
  void set_trivial() {
    chain_weights.clear();
    epsilon_chains.clear();
    trivial = true;
  }
  void set_composed(Wfst* c) {  // cascade.h:207-210
    composed = c;
    if (trivial) {
      cascade.assign(1, c);
    }
  }
  void add(Wfst* w) {
    if (trivial) return;
    cascade.push_back(w);
  }
  void prepare_compose() { prepare_compose(false, false); }
  void prepare_compose(bool right_assoc) {
    if (right_assoc)
      prepare_compose(false, true);
    else
      prepare_compose(true, false);
  }
  void prepare_compose(bool a, bool b) {
    is_chain[0] = a;
    is_chain[1] = b;
  }
  static bool is_locked_1(const Arc* e) { return e->locked() && e->weight.isOne(); }
  unsigned locked_1_groupid() const { return trivial ? LOCKED_GROUP : nil_chain; }
  unsigned original_id(Arc* e) const { return is_locked_1(e) ? nil_chain : e->group; }

  // cons helpers build a fresh list (the reference shares tails; contents are identical)
  static Chain cons(Arc* a, const Chain& cdr) {
    if (is_locked_1(a)) return cdr;
    Chain r;
    r.reserve(cdr.size() + 1);
    r.push_back(a);
    r.insert(r.end(), cdr.begin(), cdr.end());
    return r;
  }
  static Chain cons2(Arc* a, Arc* b) { return cons(a, cons(b, Chain())); }
  static Chain cons_cc(const Chain& a, Chain b) {  // cascade.h:518-521: prepends a's items one by one
    for (Arc* x : a) b = cons(x, b);
    return b;
  }
  Chain cons_chain(Arc* a, Arc* b) {  // cascade.h:536-551
    if (is_chain[0]) {
      const Chain& ca = chains[a->group];
      if (is_chain[1]) return cons_cc(ca, chains[b->group]);
      return cons(b, ca);
    } else {
      if (is_chain[1]) return cons(a, chains[b->group]);
      return cons2(a, b);
    }
  }
  unsigned record_eps(Arc* e, bool chain) {  // cascade.h:566-579
    if (trivial) return e->group;
    if (chain) return original_id(e);
    auto ins = epsilon_chains.emplace(e, (unsigned)chains.size());
    if (ins.second) {
      Chain v = cons(e, Chain());
      if (v.empty()) return (ins.first->second = nil_chain);
      chains.push_back(v);
    }
    return ins.first->second;
  }
  unsigned record1(Arc* e) { return record_eps(e, is_chain[0]); }
  unsigned record2(Arc* e) { return record_eps(e, is_chain[1]); }
  unsigned record(Arc* a, Arc* b) {  // cascade.h:581-592
    if (trivial) return NO_GROUP;
    Chain v = cons_chain(a, b);
    if (v.empty()) return nil_chain;
    unsigned ret = (unsigned)chains.size();
    chains.push_back(v);
    return ret;
  }
  void done_composing(Wfst* c) {
    set_composed(c);
    if (trivial) return;
    epsilon_chains.clear();
  }

  void calculate_chain_weights() {  // cascade.h:426-433
    chain_weights.assign(chains.size(), LW::one());
    for (size_t i = 0; i < chains.size(); ++i)
      for (Arc* p : chains[i]) mul_eq(chain_weights[i], p->weight);
  }
  void update() {  // cascade.h:466-479
    if (trivial) return;
    calculate_chain_weights();
    for (auto& st : composed->states)
      for (auto& a : st) a.weight = chain_weights[a.group];
  }
  void normalize(const std::vector<NormalizeMethod>& methods) {  // cascade.h:402-405
    for (size_t i = 0; i < cascade.size(); ++i) cascade[i]->normalize(methods[i]);
  }
  void clear_counts() {  // cascade.h:269-272 zero_arcs skips locked arcs
    for (Wfst* w : cascade)
      for (auto& st : w->states)
        for (auto& a : st)
          if (!a.locked()) a.weight = LW();
  }
  void distribute_counts() {  // cascade.h:318-325 + 274-303
    if (trivial) return;
    clear_counts();
    for (auto& st : composed->states)
      for (auto& a : st)
        for (Arc* p : chains[a.group])
          if (!p->locked()) p->weight += a.weight;
  }
  void save_none(const std::vector<NormalizeMethod>& methods) {  // cascade.h:339-343
    none_saves.assign(methods.size(), {});
    for (size_t i = 0; i < std::min(methods.size(), cascade.size()); ++i)
      if (methods[i].group == NORM_NONE)
        for (auto& st : cascade[i]->states)
          for (auto& a : st) none_saves[i].push_back(a.weight.w);
  }
  void load_none(const std::vector<NormalizeMethod>& methods) {  // cascade.h:345-350
    for (size_t i = 0; i < std::min(methods.size(), cascade.size()); ++i)
      if (methods[i].group == NORM_NONE) {
        size_t k = 0;
        for (auto& st : cascade[i]->states)
          for (auto& a : st) a.weight.w = none_saves[i][k++];
        none_saves[i].clear();
      }
  }
  void use_counts(const std::vector<NormalizeMethod>& methods) {
    distribute_counts();
    normalize(methods);
  }
  void use_counts_final(const std::vector<NormalizeMethod>& methods) {
    if (trivial) return;
    save_none(methods);
    use_counts(methods);
    load_none(methods);
    update();
  }
};

// compose.cc:163-531, default 3-state filter.  index_threshold = carmel -T (default 32, carmel.cc:895).
inline void compose(Wfst& out, Cascade& cascade, Wfst& a, Wfst& b, unsigned index_threshold = 32) {
  out = Wfst();
  out.in_alph = a.in_alph;
  out.out_alph = b.out_alph;
  out.named_states = false;
  if (!(a.valid && b.valid)) {
    out.valid = false;
    return;
  }
  // strhash.h:253-256 computeMap
  std::vector<unsigned> map(a.out_alph.size()), revMap(b.in_alph.size());
  for (unsigned i = 0; i < a.out_alph.size(); ++i) {
    const unsigned* ip = b.in_alph.find(a.out_alph.names[i]);
    map[i] = ip ? *ip : ~0u;
  }
  for (unsigned i = 0; i < b.in_alph.size(); ++i) {
    const unsigned* ip = a.out_alph.find(b.in_alph.names[i]);
    revMap[i] = ip ? *ip : ~0u;
  }
  struct Trio {
    unsigned qa, qb;
    char filter;
    bool operator<(const Trio& o) const {
      if (qa != o.qa) return qa < o.qa;
      if (qb != o.qb) return qb < o.qb;
      return filter < o.filter;
    }
  };
  std::map<Trio, unsigned> stateMap;
  std::vector<std::pair<unsigned, Trio> > queue;  // LIFO (List push/top/pop at the front)
  // composed arcs are PREPENDED per state (addArc -> push_front); collect in creation order, reverse at the end
  std::vector<std::vector<Arc> >& S = out.states;
  S.clear();
  S.emplace_back();
  Trio t0{0, 0, 0};
  stateMap[t0] = 0;
  queue.push_back({0u, t0});
  // indexes built lazily; lists are push_front => reversed arc order (state.h:158-199)
  std::vector<std::unordered_map<unsigned, std::vector<Arc*> > > aidx(a.num_states()), bidx(b.num_states());
  std::vector<char> ahas(a.num_states(), 0), bhas(b.num_states(), 0);
  auto index_a = [&](unsigned q) {
    if (ahas[q]) return;
    ahas[q] = 1;
    auto& v = a.states[q];
    for (auto& arc : v) aidx[q][arc.out].push_back(&arc);
    for (auto& kv : aidx[q]) std::reverse(kv.second.begin(), kv.second.end());
  };
  auto index_b = [&](unsigned q) {
    if (bhas[q]) return;
    bhas[q] = 1;
    auto& v = b.states[q];
    for (auto& arc : v) bidx[q][arc.in].push_back(&arc);
    for (auto& kv : bidx[q]) std::reverse(kv.second.begin(), kv.second.end());
  };
  auto find = [](std::unordered_map<unsigned, std::vector<Arc*> >& m, unsigned k) -> std::vector<Arc*>* {
    auto it = m.find(k);
    return it == m.end() ? 0 : &it->second;
  };
  unsigned sourceState = 0;
  auto composearc = [&](unsigned in, unsigned o, Trio triDest, LW weight, unsigned g) {
    unsigned num;
    auto ins = stateMap.emplace(triDest, (unsigned)S.size());
    if (ins.second) {
      num = (unsigned)S.size();
      queue.push_back({num, triDest});
      S.emplace_back();
    } else
      num = ins.first->second;
    S[sourceState].push_back(Arc(in, o, num, weight, g));
  };
  const unsigned EMPTY = EPS;
  while (!queue.empty()) {
    sourceState = queue.back().first;
    Trio triSource = queue.back().second;
    queue.pop_back();
    auto& qa = a.states[triSource.qa];
    auto& qb = b.states[triSource.qb];
    Trio triDest;
    bool qa_larger = qa.size() > qb.size();
    size_t larger_size = qa_larger ? qa.size() : qb.size();
    if (larger_size > index_threshold) {
      if (!qa_larger) {  // qb (rhs) is larger: compose.cc:339-385
        index_b(triSource.qb);
        auto& qbi = bidx[triSource.qb];
        for (auto& l : qa) {
          unsigned in = l.in;
          triDest.qa = l.dest;
          if (l.out == EMPTY) {
            if (triSource.filter != 2) {
              triDest.filter = 1;
              triDest.qb = triSource.qb;
              composearc(in, EMPTY, triDest, l.weight, cascade.record1(&l));
            }
            if (triSource.filter == 0)
              if (auto* matches = find(qbi, EMPTY)) {
                triDest.filter = 0;
                for (Arc* r : *matches) {
                  triDest.qb = r->dest;
                  composearc(in, r->out, triDest, l.weight * r->weight, cascade.record(&l, r));
                }
              }
          } else {
            if (auto* matches = find(qbi, map[l.out])) {
              triDest.filter = 0;
              for (Arc* r : *matches) {
                triDest.qb = r->dest;
                composearc(in, r->out, triDest, l.weight * r->weight, cascade.record(&l, r));
              }
            }
          }
        }
        if (triSource.filter != 1)
          if (auto* matches = find(qbi, EMPTY)) {
            triDest.qa = triSource.qa;
            triDest.filter = 2;
            for (Arc* r : *matches) {
              triDest.qb = r->dest;
              composearc(EMPTY, r->out, triDest, r->weight, cascade.record2(r));
            }
          }
      } else {  // qa (lhs) is larger: compose.cc:386-436
        index_a(triSource.qa);
        auto& qai = aidx[triSource.qa];
        for (auto& r : qb) {
          unsigned o = r.out;
          triDest.qb = r.dest;
          if (r.in == EMPTY) {
            if (triSource.filter != 1) {
              triDest.filter = 2;
              triDest.qa = triSource.qa;
              composearc(EMPTY, o, triDest, r.weight, cascade.record2(&r));
            }
            if (triSource.filter == 0)
              if (auto* matches = find(qai, EMPTY)) {
                triDest.filter = 0;
                for (Arc* l : *matches) {
                  triDest.qa = l->dest;
                  composearc(l->in, o, triDest, l->weight * r.weight, cascade.record(l, &r));
                }
              }
          } else {
            triDest.filter = 0;
            if (auto* matches = find(qai, revMap[r.in])) {
              for (Arc* l : *matches) {
                triDest.qa = l->dest;
                composearc(l->in, o, triDest, l->weight * r.weight, cascade.record(l, &r));
              }
            }
          }
        }
        if (triSource.filter != 2)
          if (auto* matches = find(qai, EMPTY)) {
            triDest.qb = triSource.qb;
            triDest.filter = 1;
            for (Arc* l : *matches) {
              triDest.qa = l->dest;
              composearc(l->in, EMPTY, triDest, l->weight, cascade.record1(l));
            }
          }
      }
    } else {  // both small: compose.cc:437-487
      for (auto& l : qa) {
        unsigned in = l.in;
        triDest.qa = l.dest;
        if (l.out == EMPTY) {
          if (triSource.filter != 2) {
            triDest.filter = 1;
            triDest.qb = triSource.qb;
            composearc(in, EMPTY, triDest, l.weight, cascade.record1(&l));
          }
          if (triSource.filter == 0) {
            for (auto& r : qb)
              if (r.in == EMPTY) {
                triDest.qb = r.dest;
                triDest.filter = 0;
                composearc(in, r.out, triDest, l.weight * r.weight, cascade.record(&l, &r));
              }
          }
        } else {
          triDest.filter = 0;
          for (auto& r : qb)
            if (map[l.out] == r.in) {
              triDest.qb = r.dest;
              composearc(in, r.out, triDest, l.weight * r.weight, cascade.record(&l, &r));
            }
        }
      }
      if (triSource.filter != 1) {
        triDest.qa = triSource.qa;
        triDest.filter = 2;
        for (auto& r : qb)
          if (r.in == EMPTY) {
            triDest.qb = r.dest;
            composearc(EMPTY, r.out, triDest, r.weight, cascade.record2(&r));
          }
      }
    }
  }
  // finals: compose.cc:503-530
  Trio tf;
  tf.qa = a.final_state;
  tf.qb = b.final_state;
  unsigned nFinal = 0;
  int pFinal[3] = {-1, -1, -1};
  for (int i = 0; i < 3; ++i) {
    tf.filter = (char)i;
    auto it = stateMap.find(tf);
    if (it != stateMap.end()) {
      pFinal[i] = (int)it->second;
      ++nFinal;
      out.final_state = it->second;
    }
  }
  if (nFinal == 0) {
    out.valid = false;
    return;
  }
  if (nFinal > 1) {
    out.final_state = (unsigned)S.size();
    S.emplace_back();
    for (int i = 0; i < 3; ++i)
      if (pFinal[i] >= 0) S[pFinal[i]].push_back(Arc(EMPTY, EMPTY, out.final_state, LW::one(), cascade.locked_1_groupid()));
  }
  for (auto& st : S) std::reverse(st.begin(), st.end());  // push_front lists
}

// compose.cc:219-313, carmel -a ("preserveGroups"): 2-state filter
//   0->0 : a:c from a:b (in l) and b:c (in r), incl. b=*e*;  1->0 : the same with b != *e*;  0->1, 1->1 : *e*:c from r.
// An arc a:b of l goes to a "mediate" state (l's destination, r's state, b) as a:*e*; r's b:c arcs leave it as *e*:c.
// Every composed arc is recorded with record1 / record2: it stands for exactly one operand arc.
inline void compose_a(Wfst& out, Cascade& cascade, Wfst& a, Wfst& b) {
  out = Wfst();
  out.in_alph = a.in_alph;
  out.out_alph = b.out_alph;
  out.named_states = false;
  if (!(a.valid && b.valid)) {
    out.valid = false;
    return;
  }
  std::vector<unsigned> map(a.out_alph.size());
  for (unsigned i = 0; i < a.out_alph.size(); ++i) {
    const unsigned* ip = b.in_alph.find(a.out_alph.names[i]);
    map[i] = ip ? *ip : ~0u;
  }
  typedef std::tuple<unsigned, unsigned, int> Trio;      // qa, qb, filter
  typedef std::tuple<unsigned, unsigned, unsigned> Half;  // l_dest, r_source, hidden letter
  std::map<Trio, unsigned> stateMap;
  std::map<Half, unsigned> arcStateMap;
  std::vector<std::pair<unsigned, Trio> > queue;
  std::vector<std::vector<Arc> >& S = out.states;
  S.clear();
  S.emplace_back();
  stateMap[Trio(0, 0, 0)] = 0;
  queue.push_back({0u, Trio(0, 0, 0)});
  auto composearc = [&](unsigned from, unsigned in, unsigned o, Trio dest, LW weight, unsigned g) {
    auto ins = stateMap.emplace(dest, (unsigned)S.size());
    if (ins.second) {
      queue.push_back({ins.first->second, dest});
      S.emplace_back();
    }
    S[from].push_back(Arc(in, o, ins.first->second, weight, g));
  };
  while (!queue.empty()) {
    const unsigned source = queue.back().first;
    const Trio tri = queue.back().second;
    queue.pop_back();
    const unsigned sqa = std::get<0>(tri), sqb = std::get<1>(tri);
    const int filter = std::get<2>(tri);
    auto& qa = a.states[sqa];
    auto& qb = b.states[sqb];
    std::map<unsigned, std::vector<Arc*> > aindex, bindex;  // ascending symbol; lists newest first (push_front)
    for (size_t k = qa.size(); k-- > 0;) aindex[qa[k].out].push_back(&qa[k]);
    for (size_t k = qb.size(); k-- > 0;) bindex[qb[k].in].push_back(&qb[k]);
    // the walk over qa->index (compose.cc:240-242): A's output symbols in the bucket order of the table State::indexBy(kOutput)
    // builds -- made for the state's arc count, one insert per arc in list order (refhash.hpp)
    RefHashKeys walk((unsigned)qa.size());
    for (auto& arc : qa) walk.insert(arc.out);
    for (unsigned sym : walk.keys()) {
      auto& ll = *aindex.find(sym);
      if (ll.first == EPS) {
        if (filter == 0)
          for (Arc* la : ll.second) composearc(source, la->in, EPS, Trio(la->dest, sqb, 0), la->weight, cascade.record1(la));
        continue;
      }
      if (map[ll.first] == ~0u) continue;
      auto matches = bindex.find(map[ll.first]);
      if (matches == bindex.end()) continue;
      for (Arc* la : ll.second) {
        auto ins = arcStateMap.emplace(Half(la->dest, sqb, ll.first), (unsigned)S.size());
        const unsigned mediate = ins.first->second;
        if (ins.second) {
          S.emplace_back();
          for (Arc* ra : matches->second)
            composearc(mediate, EPS, ra->out, Trio(la->dest, ra->dest, 0), ra->weight, cascade.record2(ra));
        }
        S[source].push_back(Arc(la->in, EPS, mediate, la->weight, cascade.record1(la)));
      }
    }
    auto eb = bindex.find(EPS);
    if (eb != bindex.end())
      for (Arc* ra : eb->second) composearc(source, EPS, ra->out, Trio(sqa, ra->dest, 1), ra->weight, cascade.record2(ra));
  }
  unsigned nFinal = 0;
  int pFinal[3] = {-1, -1, -1};
  for (int i = 0; i < 3; ++i) {
    auto it = stateMap.find(Trio(a.final_state, b.final_state, i));
    if (it != stateMap.end()) {
      pFinal[i] = (int)it->second;
      ++nFinal;
      out.final_state = it->second;
    }
  }
  if (nFinal == 0) {
    out.valid = false;
    return;
  }
  if (nFinal > 1) {
    out.final_state = (unsigned)S.size();
    S.emplace_back();
    for (int i = 0; i < 3; ++i)
      if (pFinal[i] >= 0) S[pFinal[i]].push_back(Arc(EPS, EPS, out.final_state, LW::one(), cascade.locked_1_groupid()));
  }
  for (auto& st : S) std::reverse(st.begin(), st.end());
}

}  // namespace oracle
