// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// deriv.hpp: restatement of carmel's per-pair derivation lattice, forward/backward and count collection.
// Follows /root/reference/carmel/src/derivations.h and /root/reference/graehl/shared/graph.h:
//   deriv_state (i,s,o)                         derivations.h:45-66
//   wfst_io_index: per state (in,out)->[arc id] derivations.h:142-155 (arc id = visit order, state-major)
//   derive / add_arcs (recursive DFS, remove[])  derivations.h:640-704; 4 label classes tried in the order
//                                               (e,e) (e,out[o]) (in[i],e) (in[i],out[o])   :656-669
//   out-arc lists are LIFO (push_front)          graph.h:91-94
//   prune(): stable compaction                   derivations.h:572-629, graph.h:322-346
//   make_order: DFS post-order, back edges skipped  derivations.h:722-729, graph.h:267-283
//   propagate_paths_in_order                     graph.h:391-402
//   compute_fb / collect_counts                  derivations.h:400-449
//   training corpus reader                       train.cc:985-1025, train.h:134-189
#pragma once
#include "wfst.hpp"
#include <cstdint>
#include <unordered_map>
#include <functional>

namespace oracle {

struct Pair {
  std::vector<unsigned> in, out;
  double weight;
  Pair() : weight(1) {}
};

struct Corpus {  // train.h:134-189
  std::vector<Pair> examples;
  unsigned maxIn, maxOut, n_pairs;
  double totalEmpiricalWeight, n_input, n_output, w_input, w_output;
  Corpus() { clear_counts(); }
  void clear_counts() {
    maxIn = maxOut = n_pairs = 0;
    n_input = n_output = w_input = w_output = totalEmpiricalWeight = 0;
  }
  void count(const Pair& p) {
    unsigned i = (unsigned)p.in.size(), o = (unsigned)p.out.size();
    n_input += i;
    n_output += o;
    w_input += p.weight * i;
    w_output += p.weight * o;
    if (maxIn < i) maxIn = i;
    if (maxOut < o) maxOut = o;
    totalEmpiricalWeight += p.weight;
    ++n_pairs;
  }
  void count() {
    clear_counts();
    for (auto& p : examples) count(p);
  }
  void add(const std::vector<unsigned>& i, const std::vector<unsigned>& o, double w = 1.) {
    Pair p;
    p.in = i;
    p.out = o;
    p.weight = w;
    examples.push_back(p);
    count(examples.back());
  }
  void set_null() {  // train.h:171-175
    examples.clear();
    clear_counts();
    add(std::vector<unsigned>(), std::vector<unsigned>(), 1.0);
  }
};

// train.cc:985-1025
inline void read_training_corpus(Wfst& x, const std::string& text, Corpus& corpus, std::string* warn = 0) {
  size_t p = 0;
  auto getline = [&](std::string& buf) -> bool {
    if (p >= text.size()) return false;
    size_t e = text.find('\n', p);
    if (e == std::string::npos) {
      buf = text.substr(p);
      p = text.size();
    } else {
      buf = text.substr(p, e - p);
      p = e + 1;
    }
    return true;
  };
  std::string buf;
  for (;;) {
    double weight = 1;
    if (!getline(buf)) break;
    char s = buf.empty() ? '\0' : buf[0];
    if (std::isdigit((unsigned char)s) || s == '-' || s == '.' || s == 'e') {
      char* e;
      double w = std::strtod(buf.c_str(), &e);
      if (e == buf.c_str()) {
        if (warn) *warn += "Bad training example weight: " + buf + "\n";
        continue;
      }
      weight = w;
      if (!getline(buf)) {
        if (warn) *warn += "Incomplete input/output training pair\n";
        break;
      }
    }
    std::vector<unsigned> ins, outs;
    x.symbol_list(ins, buf, false);
    if (!getline(buf)) {
      if (!ins.empty() && warn) *warn += "Incomplete input/output training pair\n";
      break;
    }
    x.symbol_list(outs, buf, true);
    corpus.add(ins, outs, weight);
  }
}

// arc table: one record per WFST arc in visit order (derivations.h:86-101; train.h:28-40)
struct ArcRec {
  Arc* arc;
  unsigned src;
  LW scratch, em_weight, best_weight, counts, prior_counts;
};

struct ArcTable {
  std::vector<ArcRec> t;
  unsigned n_states;
  void build(Wfst& x, bool per_arc_prior, LW global_prior) {
    t.clear();
    n_states = x.num_states();
    for (unsigned s = 0; s < x.num_states(); ++s)
      for (auto& a : x.states[s]) {
        ArcRec r;
        r.arc = &a;
        r.src = s;
        r.prior_counts = per_arc_prior ? global_prior + a.weight : global_prior;  // derivations.h:99-100
        t.push_back(r);
      }
  }
};

struct IoIndex {  // derivations.h:142-155
  std::vector<std::unordered_map<uint64_t, std::vector<unsigned> > > st;
  static uint64_t key(unsigned i, unsigned o) { return ((uint64_t)i << 32) | o; }
  void build(const Wfst& x) {
    st.assign(x.num_states(), {});
    unsigned id = 0;
    for (unsigned s = 0; s < x.num_states(); ++s)
      for (auto& a : x.states[s]) st[s][key(a.in, a.out)].push_back(id++);
  }
};

struct GArc {
  unsigned src, dest;
  unsigned arcid;  // GraphArc::data
  double weight;   // GraphArc::weight (real prob of the WFST arc when the lattice was built; used by gibbs)
};

struct DerivStats {  // derivations.h:191-247 (states/arcs only)
  double pre_states, pre_arcs, post_states, post_arcs, N;
  DerivStats() : pre_states(0), pre_arcs(0), post_states(0), post_arcs(0), N(0) {}
};

struct Derivations {
  // g[s] holds out-arcs in INSERTION order; the reference list is push_front, so every place the reference
  // walks a state's list we walk g[s] from back to front (LIST_FOR below).
  std::vector<std::vector<GArc> > g;
  unsigned fin;
  bool no_goal;
  double weight;
  unsigned lineno;
  std::vector<unsigned> reverse_order;  // DFS post-order from state 0
  unsigned n_back_edges;

  struct DS {
    uint32_t i, s, o;
  };
  struct DSHash {
    size_t operator()(const DS& d) const {
      uint64_t h = d.i * 0x9E3779B97F4A7C15ull;
      h ^= (uint64_t)d.s * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
      h ^= (uint64_t)d.o * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
      return (size_t)h;
    }
  };
  struct DSEq {
    bool operator()(const DS& a, const DS& b) const { return a.i == b.i && a.s == b.s && a.o == b.o; }
  };

  bool empty() const { return no_goal; }
  size_t n_states() const { return g.size(); }
  size_t n_arcs() const {
    size_t n = 0;
    for (auto& s : g) n += s.size();
    return n;
  }

  // derivations.h:471-513 init_and_compute
  bool compute(Wfst& x, const IoIndex& io, const ArcTable& atab, const Pair& p, bool prune_, DerivStats* stats) {
    in_ = &p.in;
    out_ = &p.out;
    weight = p.weight;
    g.clear();
    remove_.clear();
    ids_.clear();
    goal_ = DS{(uint32_t)p.in.size(), x.final_state, (uint32_t)p.out.size()};
    pre_arcs_ = 0;
    derive(io, atab, DS{0, 0, 0});
    auto it = ids_.find(goal_);
    no_goal = (it == ids_.end());
    if (!no_goal) fin = it->second;
    if (stats) {
      stats->N += 1;
      stats->pre_states += (double)g.size();
      stats->pre_arcs += (double)pre_arcs_;
    }
    if (prune_)
      prune(stats);
    else if (stats) {
      stats->post_states += (double)g.size();
      stats->post_arcs += (double)n_arcs();
    }
    ids_.clear();
    remove_.clear();
    if (no_goal) {
      g.clear();
      return false;
    }
    return true;
  }

  // derivations.h:640-675
  unsigned derive(const IoIndex& io, const ArcTable& atab, DS d) {
    unsigned src = (unsigned)g.size();
    auto ins = ids_.emplace(d, src);
    if (!ins.second) return ins.first->second;
    g.emplace_back();
    remove_.push_back(0);
    const auto& fs = io.st[d.s];
    bool dead = !DSEq()(d, goal_);
    const std::vector<unsigned>& in = *in_;
    const std::vector<unsigned>& out = *out_;
    if (add_arcs(io, atab, EPS, EPS, d.i, d.o, fs, src)) dead = false;
    bool useO = d.o < out.size(), useI = d.i < in.size();
    unsigned o1 = d.o + 1, i1 = d.i + 1;
    if (useO)
      if (add_arcs(io, atab, EPS, out[d.o], d.i, o1, fs, src)) dead = false;
    if (useI) {
      unsigned si = in[d.i];
      if (add_arcs(io, atab, si, EPS, i1, d.o, fs, src)) dead = false;
      if (useO)
        if (add_arcs(io, atab, si, out[d.o], i1, o1, fs, src)) dead = false;
    }
    remove_[src] = dead;
    return src;
  }

  // derivations.h:678-704
  bool add_arcs(const IoIndex& io, const ArcTable& atab, unsigned s_in, unsigned s_out, unsigned i_in, unsigned i_out,
                const std::unordered_map<uint64_t, std::vector<unsigned> >& fs, unsigned source) {
    bool reachgoal = false;
    auto m = fs.find(IoIndex::key(s_in, s_out));
    if (m != fs.end())
      for (unsigned id : m->second) {
        ++pre_arcs_;
        const Arc* a = atab.t[id].arc;
        unsigned dst = derive(io, atab, DS{i_in, a->dest, i_out});
        if (!remove_[dst]) {
          GArc ga;
          ga.src = source;
          ga.dest = dst;
          ga.arcid = id;
          ga.weight = a->weight.getReal();
          g[source].push_back(ga);
          reachgoal = true;
        }
      }
    return reachgoal;
  }

  // derivations.h:572-629 + graph.h:322-346 + indices_after.hpp: stable compaction of states with remove[]
  void prune(DerivStats* stats) {
    if (no_goal) return;
    size_t n = g.size();
    std::vector<unsigned> tt(n, (unsigned)-1);
    unsigned k = 0;
    for (size_t i = 0; i < n; ++i)
      if (!remove_[i]) tt[i] = k++;
    fin = tt[fin];
    size_t kept = 0;
    std::vector<std::vector<GArc> > ng(k);
    for (size_t i = 0; i < n; ++i)
      if (~tt[i]) {
        auto& dst = ng[tt[i]];
        for (auto& a : g[i])
          if (~tt[a.dest]) {
            GArc b = a;
            b.src = tt[a.src];
            b.dest = tt[a.dest];
            dst.push_back(b);
            ++kept;
          }
      }
    g.swap(ng);
    if (stats) {
      stats->post_states += (double)k;
      stats->post_arcs += (double)kept;
    }
  }

  // derivations.h:722-729 -> graph.h:267-283 (recursive order_from; arcs in list order = reversed insertion)
  void make_order() {
    reverse_order.clear();
    reverse_order.reserve(g.size());
    n_back_edges = 0;
    std::vector<char> done(g.size(), 0), begun(g.size(), 0);
    order_from(0, done, begun);
  }
  void order_from(unsigned s, std::vector<char>& done, std::vector<char>& begun) {
    if (done[s]) return;
    if (begun[s]) {
      ++n_back_edges;
      return;
    }
    begun[s] = 1;
    const auto& arcs = g[s];
    for (size_t k = arcs.size(); k-- > 0;) order_from(arcs[k].dest, done, begun);
    done[s] = 1;
    reverse_order.push_back(s);
  }

  // reversed graph: add_reversed_arcs (graph.cc) pushes each reversed arc to the front of rev[dest]'s list
  // while walking states 0..n-1 and each state's list in list order.
  void make_reverse(std::vector<std::vector<GArc> >& r) const {
    r.assign(g.size(), {});
    for (unsigned s = 0; s < g.size(); ++s) {
      const auto& arcs = g[s];
      for (size_t k = arcs.size(); k-- > 0;) {
        const GArc& a = arcs[k];
        GArc b = a;
        b.src = a.dest;
        b.dest = a.src;
        r[a.dest].push_back(b);  // insertion order; walked back-to-front like every list here
      }
    }
  }

  // derivations.h:400-417 with weight functor wf(arcid) -> LW
  template <class WF>
  LW compute_fb(std::vector<LW>& f, std::vector<LW>& b, const WF& wf) {
    size_t nst = g.size();
    f.assign(nst, LW());
    b.assign(nst, LW());
    f[0] = LW::one();
    make_order();
    // forward: reverse_order.rbegin()..rend()  (graph.h:391-402)
    for (size_t t = reverse_order.size(); t-- > 0;) {
      unsigned src = reverse_order[t];
      const auto& arcs = g[src];
      for (size_t k = arcs.size(); k-- > 0;) {
        const GArc& a = arcs[k];
        f[a.dest] += f[src] * wf(a);
      }
    }
    LW prob = f[fin];
    std::vector<std::vector<GArc> > r;
    make_reverse(r);
    b[fin] = LW::one();
    for (size_t t = 0; t < reverse_order.size(); ++t) {
      unsigned src = reverse_order[t];
      const auto& arcs = r[src];
      for (size_t k = arcs.size(); k-- > 0;) {
        const GArc& a = arcs[k];
        b[a.dest] += b[src] * wf(a);
      }
    }
    return prob;
  }

  // derivations.h:432-449.  posteriors (optional): per lattice arc, in (state, list-order) sequence
  // counts_override: accumulate into this table instead of ArcRec::counts (per-thread tables of the OpenMP
  // baseline in oracle_capi.cpp; same arithmetic)
  LW collect_counts(ArcTable& t, std::vector<LW>* f_out = 0, std::vector<LW>* b_out = 0, LW* counts_override = 0) {
    std::vector<LW> f, b;
    auto wf = [&](const GArc& a) { return t.t[a.arcid].arc->weight; };
    LW prob = compute_fb(f, b, wf);
    for (unsigned s = 0; s < g.size(); ++s) {
      const auto& arcs = g[s];
      for (size_t k = arcs.size(); k-- > 0;) {
        const GArc& a = arcs[k];
        ArcRec& ac = t.t[a.arcid];
        LW arc_contrib = ac.arc->weight * f[a.src] * b[a.dest];
        LW& cnt = counts_override ? counts_override[a.arcid] : ac.counts;
        cnt += arc_contrib * LW::from_real(weight) / prob;
      }
    }
    if (f_out) f_out->swap(f);
    if (b_out) b_out->swap(b);
    return prob;
  }

 private:
  const std::vector<unsigned>* in_;
  const std::vector<unsigned>* out_;
  std::unordered_map<DS, unsigned, DSHash, DSEq> ids_;
  std::vector<char> remove_;
  DS goal_;
  size_t pre_arcs_;
};

}  // namespace oracle
