// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// matrix.hpp: restatement of carmel's legacy dense forward/backward (`carmel --matrix-fb`), the CPU reference the HIP
// kernel of carmel_amd/csrc/matrix_fb.hip is checked against.  Follows /root/reference/carmel/src/train.cc:
//   matrix_io_index (per state: (in, out) -> [(other end, arc id)], arc-id order)   :80-100
//   e_topo_populate + TopoSort (DFS, a state in front of everything it reaches)     :339-357, graph.h:197-239;
//        the *e*:*e* graph lists a state's arcs newest first (fst.cc:343-368, List::push)
//   matrix_compute: cell by cell, *e*:*e* arcs in that order, then the three consuming label classes   :698-745
//   the backward matrix = the same walk over reversed strings and arcs, rotated by 180 degrees          :254-261, 104-120
//   matrix_forward_prop / matrix_count                                                                  :266-296
//   estimate_matrix: prob = f[nIn][nOut][final]; scratch per arc over all cells; counts += weight / prob * scratch   :776-860
// Parity: unpinned by reference-held vectors (the tutorial's recorded runs use derivation lattices); the restatement is
// tied to them through tests/test_matrix_fb_*.py -- on transducers without *e*:*e* cycles the dense walk sums over exactly
// the derivations of the lattices, so its counts and probabilities must equal deriv.hpp's, which the recorded traces pin.
#pragma once
#include "deriv.hpp"

namespace oracle {

struct MatrixFB {
  typedef std::vector<std::pair<unsigned, unsigned> > ForIo;  // (other end, arc id)
  typedef std::unordered_map<uint64_t, ForIo> ForState;
  Wfst& x;
  ArcTable& arcs;
  unsigned n_st;
  std::vector<ForState> forward, backward;
  std::vector<unsigned> e_forward_topo, e_backward_topo;
  unsigned n_back_edges;
  typedef std::vector<std::vector<std::vector<LW> > > Mat;  // [i][o][s]
  Mat f, b;

  MatrixFB(Wfst& x_, ArcTable& a) : x(x_), arcs(a), n_st(x_.num_states()), n_back_edges(0) {
    forward.assign(n_st, ForState());
    backward.assign(n_st, ForState());
    for (unsigned i = 0; i < arcs.t.size(); ++i) {  // train.cc:90-98
      const ArcRec& ac = arcs.t[i];
      uint64_t io = IoIndex::key(ac.arc->in, ac.arc->out);
      forward[ac.src][io].push_back(std::make_pair(ac.arc->dest, i));
      backward[ac.arc->dest][io].push_back(std::make_pair(ac.src, i));
    }
    // train.cc:339-357
    std::vector<std::vector<unsigned> > eg(n_st), rg(n_st);
    for (unsigned s = 0; s < n_st; ++s)
      for (auto& a : x.states[s])
        if (a.in == 0 && a.out == 0) eg[s].insert(eg[s].begin(), a.dest);  // List::push = push_front
    for (unsigned s = 0; s < n_st; ++s)
      for (unsigned d : eg[s]) rg[d].insert(rg[d].begin(), s);  // reverseGraph pushes too (graph.cc)
    n_back_edges = topo(eg, e_forward_topo);
    topo(rg, e_backward_topo);
  }

  static unsigned topo(const std::vector<std::vector<unsigned> >& g, std::vector<unsigned>& out) {  // graph.h:197-239, order_crucial
    unsigned n = (unsigned)g.size(), back = 0;
    std::vector<char> done(n, 0), begun(n, 0);
    std::vector<unsigned> rev;  // finish order; the reference inserts at the front
    std::function<void(unsigned)> from = [&](unsigned s) {
      if (done[s]) return;
      if (begun[s]) {
        ++back;
        return;
      }
      begun[s] = 1;
      for (unsigned d : g[s]) from(d);
      done[s] = 1;
      rev.push_back(s);
    };
    for (unsigned i = 0; i < n; ++i)
      if (!g[i].empty()) from(i);
    out.assign(rev.rbegin(), rev.rend());
    return back;
  }

  static const ForIo* find(const ForState& fs, unsigned in, unsigned out) {
    auto it = fs.find(IoIndex::key(in, out));
    return it == fs.end() ? nullptr : &it->second;
  }

  void forward_prop(Mat& m, const ForIo* fio, unsigned s, unsigned i, unsigned o, unsigned d_i, unsigned d_o) {  // train.cc:266-284
    if (!fio) return;
    for (auto& dw : *fio) {
      LW& to = m[i + d_i][o + d_o][dw.first];
      to += m[i][o][s] * arcs.t[dw.second].arc->weight;
    }
  }

  void compute(unsigned nIn, const std::vector<unsigned>& inLet, unsigned nOut, const std::vector<unsigned>& outLet, unsigned start,
               Mat& w, std::vector<ForState>& io, const std::vector<unsigned>& eTopo) {  // train.cc:698-745
    w.assign(nIn + 1, std::vector<std::vector<LW> >(nOut + 1, std::vector<LW>(n_st)));
    w[0][0][start] = LW::from_ln(0.0);
    for (unsigned i = 0; i <= nIn; ++i)
      for (unsigned o = 0; o <= nOut; ++o) {
        for (unsigned s : eTopo) forward_prop(w, find(io[s], 0, 0), s, i, o, 0, 0);
        for (unsigned s = 0; s < n_st; ++s) {
          if (w[i][o][s].w == -std::numeric_limits<double>::infinity()) continue;
          if (o < nOut) {
            forward_prop(w, find(io[s], 0, outLet[o]), s, i, o, 0, 1);
            if (i < nIn) forward_prop(w, find(io[s], inLet[i], outLet[o]), s, i, o, 1, 1);
          }
          if (i < nIn) forward_prop(w, find(io[s], inLet[i], 0), s, i, o, 1, 0);
        }
      }
  }

  // one pair: fills f, b; returns prob (train.cc:747-759, 254-261)
  LW fb(const Pair& p) {
    unsigned nIn = (unsigned)p.in.size(), nOut = (unsigned)p.out.size();
    compute(nIn, p.in, nOut, p.out, 0, f, forward, e_forward_topo);
    std::vector<unsigned> rin(p.in.rbegin(), p.in.rend()), rout(p.out.rbegin(), p.out.rend());
    Mat r;
    compute(nIn, rin, nOut, rout, x.final_state, r, backward, e_backward_topo);
    b.assign(nIn + 1, std::vector<std::vector<LW> >(nOut + 1));  // matrix_reverse_io: w.ij <- w.(I-i)(J-j)
    for (unsigned i = 0; i <= nIn; ++i)
      for (unsigned o = 0; o <= nOut; ++o) b[i][o].swap(r[nIn - i][nOut - o]);
    return f[nIn][nOut][x.final_state];
  }

  void count(const ForIo* fio, unsigned s, unsigned i, unsigned o, unsigned d_i, unsigned d_o) {  // train.cc:288-296
    if (!fio) return;
    for (auto& dw : *fio) {
      ArcRec& a = arcs.t[dw.second];
      a.scratch += f[i][o][s] * a.arc->weight * b[i + d_i][o + d_o][dw.first];
    }
  }

  // train.cc:776-860 (pairs are not erased here: pair_logprob = -inf marks them)
  LW estimate(const Corpus& c, LW& unweighted_corpus_prob, std::vector<double>* pair_logprob) {
    for (auto& a : arcs.t) a.counts = LW();
    LW ret = LW::from_ln(0.0);
    unweighted_corpus_prob = LW::from_ln(0.0);
    if (pair_logprob) pair_logprob->assign(c.examples.size(), -std::numeric_limits<double>::infinity());
    for (size_t k = 0; k < c.examples.size(); ++k) {
      const Pair& p = c.examples[k];
      unsigned nIn = (unsigned)p.in.size(), nOut = (unsigned)p.out.size();
      LW fin = fb(p);
      if (!(fin.w > -std::numeric_limits<double>::infinity())) continue;  // warn_no_derivations; the example is dropped
      ret = ret * fin.pow(p.weight);
      unweighted_corpus_prob = unweighted_corpus_prob * fin;
      if (pair_logprob) (*pair_logprob)[k] = fin.w;
      for (auto& a : arcs.t) a.scratch = LW();
      for (unsigned i = 0; i <= nIn; ++i)
        for (unsigned o = 0; o <= nOut; ++o)
          for (unsigned s = 0; s < n_st; ++s) {
            const ForState& fs = forward[s];
            if (i < nIn) {
              if (o < nOut) count(find(fs, p.in[i], p.out[o]), s, i, o, 1, 1);
              count(find(fs, p.in[i], 0), s, i, o, 1, 0);
            }
            if (o < nOut) count(find(fs, 0, p.out[o]), s, i, o, 0, 1);
            count(find(fs, 0, 0), s, i, o, 0, 0);
          }
      LW mult = LW::from_real(p.weight) / fin;
      for (auto& a : arcs.t)
        if (a.scratch.w > -std::numeric_limits<double>::infinity()) a.counts += mult * a.scratch;
    }
    return ret;
  }
};

}  // namespace oracle
