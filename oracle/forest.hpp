// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// forest.hpp: restatement of forest-em's packed AND/OR derivation forests: text reader, inside, normalised
// outside, expected rule counts, EM over normalisation groups, and the Gibbs sampler glue.
// Follows /root/reference/forest-em and graehl/shared:
//   forest.hpp:60-76      ForestNode: preorder array, `next` = one past the subtree, label = rule id (0 = OR) or a
//                         pointer to a shared sub-forest defined earlier (#k(...) defines, #k refers)
//   forest.hpp:925-1034   text reader
//   forest.hpp:636-697    inside_rec (AND = rule weight x children, OR = sum of children, back-reference copies;
//                         records (parent, child) ancestry pairs in visit order)
//   forest.hpp:439-491    compute_norm_outside (walks the ancestry list backwards)
//   forest.hpp:417-438    visit_inside_norm_outside: count[rule] += inside * norm_outside for every AND node
//   forest.hpp:725-758    choose_random (OR: choice = random01(), subtract inside^power / norm; AND: record the
//                         rule, recurse into every child; a back-reference recurses with power 1 — kept)
//   forest.hpp:768-816    compute_inside(W): inside with proposal probabilities
//   forest-em.hpp:446-458, 511-578, 626-655   EM: counts start at prior_count * n_forests, avg log prob, maximize
//   forest-em.hpp:694-766 Gibbs glue: parameters = rules, prior = alpha * p * |group| (normalised p), fixed if no group
//   normalize.hpp:123-164, 245-260  w = count / (sum + add_k); zero group -> uniform (default) or zero
//   em.hpp:107-216        overrelaxed_em loop (relative change of the average log prob < epsilon)
// The reference computes in `float` unless --double-precision (forest-em-params.cpp:13-18); this restatement is
// the double-precision instantiation.  Its count "overflow" side table (forest.hpp:360-407) only matters for
// float and is not restated.  PARITY UNPINNED: the reference ships forest inputs but no expected outputs
// (SURVEY.md section 8c), so this oracle is pinned only by hand-derived cases in tests/test_forest_oracle.py.
#pragma once
#include "gibbs.hpp"
#include "lw.hpp"
#include <cctype>
#include <functional>
#include <stdexcept>
#include <cstdio>
#include <string>
#include <vector>

namespace oracle {

struct FNode {
  unsigned label;  // 0 = OR, > 0 = rule id (AND); meaningless for a back-reference
  int ref;         // >= 0: index of the shared node this refers to
  unsigned next;   // one past the last node of the subtree
};

struct Forest {
  std::vector<FNode> nodes;
  unsigned max_rule = 0;

  // forest.hpp:925-1034
  static bool parse(const std::string& s, size_t& p, Forest& f) {
    f.nodes.clear();
    std::vector<unsigned> open;
    std::vector<int> backrefs;
    bool follows_paren = false;
    auto skipws = [&]() {
      while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
    };
    skipws();
    if (p >= s.size()) return false;
    for (;;) {
      skipws();
      if (p >= s.size()) throw std::runtime_error("forest: unexpected end of input");
      char c = s[p++];
      if (c == '#') {
        if (follows_paren) throw std::runtime_error("Unexpected '#' following paren in Forest");
        size_t e = p;
        while (e < s.size() && std::isdigit((unsigned char)s[e])) ++e;
        unsigned id = (unsigned)std::stoul(s.substr(p, e - p));
        p = e;
        char d = p < s.size() ? s[p] : '\0';
        if (d == '(') {
          ++p;
          if (backrefs.size() <= id) backrefs.resize(id + 1, -1);
          backrefs[id] = (int)f.nodes.size();
          follows_paren = true;
          open.push_back((unsigned)f.nodes.size());
        } else {
          if (id >= backrefs.size() || backrefs[id] < 0) throw std::runtime_error("forest: undefined back-reference");
          FNode n;
          n.label = 0;
          n.ref = backrefs[id];
          n.next = (unsigned)f.nodes.size() + 1;
          f.nodes.push_back(n);
          if (open.empty()) break;
        }
      } else if (c == '(') {
        follows_paren = true;
        open.push_back((unsigned)f.nodes.size());
      } else if (c >= '1' && c <= '9') {
        size_t e = p;
        while (e < s.size() && std::isdigit((unsigned char)s[e])) ++e;
        unsigned rule = (unsigned)std::stoul(s.substr(p - 1, e - p + 1));
        p = e;
        if (rule > f.max_rule) f.max_rule = rule;
        FNode n;
        n.label = rule;
        n.ref = -1;
        n.next = (unsigned)f.nodes.size() + 1;
        f.nodes.push_back(n);
        if (!follows_paren) {
          if (open.empty()) break;
        } else
          follows_paren = false;
      } else if (c == 'O') {
        if (p >= s.size() || s[p] != 'R') throw std::runtime_error("forest: expected OR");
        ++p;
        if (!follows_paren) throw std::runtime_error("OR not following paren in Forest");
        follows_paren = false;
        FNode n;
        n.label = 0;
        n.ref = -1;
        n.next = (unsigned)f.nodes.size() + 1;
        f.nodes.push_back(n);
      } else if (c == ')') {
        if (open.empty()) throw std::runtime_error("forest: unbalanced ')'");
        if (open.back() != f.nodes.size()) f.nodes[open.back()].next = (unsigned)f.nodes.size();
        open.pop_back();
        if (open.empty()) break;
      } else
        throw std::runtime_error(std::string("Forest: unexpected char ") + c);
    }
    return true;
  }

  struct Anc {
    int parent, child;
  };

  // inside with weight functor w(rule) -> LW; fills `anc` like inside_rec does
  void inside_rec(unsigned b, const std::function<LW(unsigned)>& w, std::vector<LW>& ins, std::vector<Anc>* anc) const {
    const FNode& n = nodes[b];
    unsigned e = n.next;
    if (n.ref >= 0) {
      ins[b] = ins[n.ref];
      return;
    }
    if (n.label == 0) {
      unsigned c = b + 1;
      unsigned nx = nodes[c].next;
      inside_rec(c, w, ins, anc);
      ins[b] = ins[c];
      for (c = nx; c < e; c = nx) {
        nx = nodes[c].next;
        inside_rec(c, w, ins, anc);
        ins[b] += ins[c];
      }
    } else {
      ins[b] = w(n.label);
      unsigned nx;
      for (unsigned c = b + 1; c < e; c = nx) {
        nx = nodes[c].next;
        inside_rec(c, w, ins, anc);
        mul_eq(ins[b], ins[c]);
      }
    }
    if (anc)
      for (unsigned c = b + 1; c < e; c = nodes[c].next) anc->push_back(Anc{(int)b, nodes[c].ref >= 0 ? nodes[c].ref : (int)c});
  }

  // forest.hpp:439-491
  bool norm_outside(const std::vector<LW>& ins, const std::vector<Anc>& anc, std::vector<LW>& out) const {
    out.assign(nodes.size(), LW());
    if (!(ins[0] > LW())) return false;
    out[0] = LW::from_ln(-ins[0].w);
    for (size_t i = anc.size(); i-- > 0;) {
      int p = anc[i].parent, c = anc[i].child;
      if (nodes[p].label == 0)
        out[c] += out[p];
      else if (!ins[p].isZero())
        out[c] += out[p] * ins[p] / ins[c];
    }
    return true;
  }

  // forest.hpp:514-574 viterbi_rec: inside with max for OR -- the first child starts as the best, a later one replaces
  // it only if strictly better (:547) -- and viterbi[or node] = its best child
  void viterbi_rec(unsigned b, const std::function<LW(unsigned)>& w, std::vector<LW>& ins, std::vector<unsigned>& vit) const {
    const FNode& n = nodes[b];
    unsigned e = n.next;
    if (n.ref >= 0) {
      ins[b] = ins[n.ref];
      return;
    }
    if (n.label == 0) {
      unsigned c = b + 1;
      unsigned nx = nodes[c].next;
      viterbi_rec(c, w, ins, vit);
      ins[b] = ins[c];
      vit[b] = c;
      for (c = nx; c < e; c = nx) {
        nx = nodes[c].next;
        viterbi_rec(c, w, ins, vit);
        if (ins[b] < ins[c]) {
          ins[b] = ins[c];
          vit[b] = c;
        }
      }
    } else {
      ins[b] = w(n.label);
      unsigned nx;
      for (unsigned c = b + 1; c < e; c = nx) {
        nx = nodes[c].next;
        viterbi_rec(c, w, ins, vit);
        mul_eq(ins[b], ins[c]);
      }
    }
  }
  // forest.hpp:590-632 write_viterbi_rec: a leaf rule prints as "rule", a rule with children as "(rule child child)"
  void write_viterbi_rec(unsigned b, const std::vector<unsigned>& vit, std::string& o) const {
    const FNode& n = nodes[b];
    if (n.ref >= 0) {
      write_viterbi_rec((unsigned)n.ref, vit, o);
    } else if (n.label == 0) {
      write_viterbi_rec(vit[b], vit, o);
    } else if (b + 1 == n.next) {
      o += std::to_string(n.label);
    } else {
      o += '(';
      o += std::to_string(n.label);
      for (unsigned c = b + 1; c < n.next; c = nodes[c].next) {
        o += ' ';
        write_viterbi_rec(c, vit, o);
      }
      o += ')';
    }
  }

  // forest.hpp:725-758; u() = random01(); record(rule)
  void choose_random(unsigned b, const std::vector<LW>& ins, const std::function<double()>& u,
                     const std::function<void(unsigned)>& record, double power) const {
    const FNode& n = nodes[b];
    unsigned e = n.next;
    if (n.ref >= 0) {
      choose_random((unsigned)n.ref, ins, u, record, 1.0);  // the reference drops `power` here
      return;
    }
    if (n.label == 0) {
      LW norm;
      for (unsigned i = b + 1; i != e; i = nodes[i].next) norm += ins[i].pow(power);
      unsigned i = b + 1;
      double choice = u();
      for (;;) {
        choice -= (ins[i].pow(power) / norm).getReal();
        if (choice < 0) break;
        unsigned nx = nodes[i].next;
        if (nx == e) break;
        i = nx;
      }
      choose_random(i, ins, u, record, power);
    } else {
      record(n.label);
      for (unsigned c = b + 1; c < e; c = nodes[c].next) choose_random(c, ins, u, record, power);
    }
  }
};

typedef std::vector<std::vector<unsigned> > NormGroups;  // ((1 2 3) (5 8))

inline NormGroups parse_normgroups(const std::string& s) {
  NormGroups g;
  size_t p = 0;
  int depth = 0;
  while (p < s.size()) {
    char c = s[p];
    if (c == '(') {
      ++depth;
      if (depth == 2) g.emplace_back();
      ++p;
    } else if (c == ')') {
      --depth;
      ++p;
    } else if (std::isdigit((unsigned char)c)) {
      size_t e = p;
      while (e < s.size() && std::isdigit((unsigned char)s[e])) ++e;
      if (depth == 2) g.back().push_back((unsigned)std::stoul(s.substr(p, e - p)));
      p = e;
    } else
      ++p;
  }
  return g;
}

inline std::vector<Forest> parse_forests(const std::string& s) {
  std::vector<Forest> fs;
  size_t p = 0;
  for (;;) {
    Forest f;
    if (!Forest::parse(s, p, f)) break;
    fs.push_back(f);
  }
  return fs;
}

// rule weights are indexed by rule id (1-based); index 0 unused
struct ForestEm {
  std::vector<Forest> forests;
  NormGroups groups;
  std::vector<LW> w;       // rule_weights
  std::vector<LW> counts;
  double prior_count = 0;  // --prior-counts
  double add_k = 0;        // --add-k-smoothing
  bool zero_zerocounts = false;

  void init(unsigned rulespace, const std::vector<double>* init_ln = 0) {
    w.assign(rulespace, LW::one());
    if (init_ln)
      for (size_t i = 0; i < init_ln->size() && i < w.size(); ++i) w[i] = LW::from_ln((*init_ln)[i]);
    counts.assign(rulespace, LW());
  }
  // FForests::init_rule_weights without an initial parameter file (forest-em.hpp:297-318): every parameter 1 (-u), or every
  // norm group uniform -- NormalizeGroups::init_uniform = init(w, set_one()) (normalize.hpp:212-234): each member set to 1
  // and divided by the group's sum -- with the rules of no group left at the default weight, ZERO (weight.h:339)
  void init_rule_weights(bool ones) {
    w.assign(w.size(), ones ? LW::one() : LW());
    if (!ones)
      for (auto& g : groups) {
        LW sum;
        for (unsigned r : g) {
          w[r] = LW::one();
          sum += w[r];
        }
        if (sum > LW())
          for (unsigned r : g) w[r] = w[r] / sum;
      }
  }
  // FForests::randomize -> NormalizeGroups::init_random (forest-em.hpp:393-399, normalize.hpp:235-238): a random positive
  // fraction per member (fraction(rule) supplied by the caller), then the group divided by its sum
  void randomize(const std::function<double(unsigned)>& fraction) {
    for (auto& g : groups) {
      LW sum;
      for (unsigned r : g) {
        w[r] = LW::from_real(fraction(r));
        sum += w[r];
      }
      if (sum > LW())
        for (unsigned r : g) w[r] = w[r] / sum;
    }
  }
  // forest-em.hpp:546-550 + forest.hpp:581-585: "best/sum=PERCENT% tree" for one forest (weights print like parameters)
  std::string viterbi_line(const Forest& f, int mode, double* best_ln = 0) const {
    std::vector<LW> ins(f.nodes.size());
    f.inside_rec(0, [&](unsigned r) { return w[r]; }, ins, 0);
    const LW sum = ins[0];
    std::vector<unsigned> vit(f.nodes.size(), 0);
    f.viterbi_rec(0, [&](unsigned r) { return w[r]; }, ins, vit);
    if (best_ln) *best_ln = ins[0].w;
    char pct[64];
    std::snprintf(pct, sizeof pct, "%g", 100 * (ins[0] / sum).getReal());
    std::string o = lw_str(ins[0], mode) + "/" + lw_str(sum, mode) + "=" + pct + "% ";
    f.write_viterbi_rec(0, vit, o);
    return o;
  }
  // normalize.hpp:123-164 (source = counts or weights, dest = weights); returns max |delta|
  LW normalize_groups(std::vector<LW>& src) {
    LW maxdiff;
    for (auto& g : groups) {
      LW sum;
      for (unsigned r : g) sum += src[r];
      if (sum > LW()) {
        sum += LW::from_real(add_k);
        for (unsigned r : g) {
          LW prev = w[r];
          w[r] = src[r] / sum;
          LW d = absdiff(w[r], prev);
          if (maxdiff < d) maxdiff = d;
        }
      } else {
        LW setto = zero_zerocounts ? LW() : LW::from_real(1.0 / (double)g.size());
        for (unsigned r : g) {
          LW d = absdiff(w[r], setto);
          if (maxdiff < d) maxdiff = d;
          w[r] = setto;
        }
      }
    }
    return maxdiff;
  }
  // forest-em.hpp:561-578: returns the average log prob over forests with non-zero probability
  double estimate(std::vector<double>* per_forest = 0) {
    LW wp = LW::from_real(prior_count * (double)forests.size());
    for (auto& c : counts) c = wp;
    double total = 0;
    size_t nz = 0;
    for (auto& f : forests) {
      std::vector<LW> ins(f.nodes.size()), out;
      std::vector<Forest::Anc> anc;
      f.inside_rec(0, [&](unsigned r) { return w[r]; }, ins, &anc);
      if (f.norm_outside(ins, anc, out))
        for (size_t i = 0; i < f.nodes.size(); ++i)
          if (f.nodes[i].ref < 0 && f.nodes[i].label != 0) counts[f.nodes[i].label] += ins[i] * out[i];
      if (per_forest) per_forest->push_back(ins[0].w);
      if (ins[0].isZero())
        ++nz;
      else
        total += ins[0].w;
    }
    return total / (double)(forests.size() - nz);
  }
  LW maximize() { return normalize_groups(counts); }

  // em.hpp:107-216 (one start, learning rate 1).  Returns best average log prob; trace = per-iteration values.
  double run_em(unsigned max_iter, double rel_eps, double converge_param_delta, std::vector<double>* trace = 0) {
    double best = -HUGE_VAL, last = -HUGE_VAL;
    std::vector<LW> best_w = w;
    bool first = true;
    for (unsigned it = 1; it <= max_iter; ++it) {
      double alp = estimate();
      if (trace) trace->push_back(alp);
      if (alp > best || first) {
        best = alp;
        best_w = w;
      }
      double rel = HUGE_VAL;
      if (!first) {
        double la = std::fabs(last);
        if (la < 1e-5) la = 1e-5;  // LOGPROB_EPSILON
        rel = (alp - last) / la;
      }
      first = false;
      if (rel < rel_eps) break;
      LW d = maximize();
      if (d.getReal() <= converge_param_delta) break;
      last = alp;
    }
    w = best_w;
    return best;
  }
};

// forest-em's Gibbs (forest-em.hpp:694-766) on top of the gibbs_base restatement in gibbs.hpp
struct ForestGibbs {
  ForestEm& fe;
  GibbsOpts gopt;
  double alpha;
  std::vector<GibbsParam> gps;  // indexed by rule id
  std::vector<double> normsum;
  std::vector<std::vector<unsigned> > sample;
  double time = 0;

  // alphas: --alpha=FILE, indexed by rule id; negative = locked (forest-em.hpp:681-709); rules beyond it use `alpha`
  ForestGibbs(ForestEm& fe, const GibbsOpts& g, double alpha, const std::vector<double>& alphas = std::vector<double>())
      : fe(fe), gopt(g), alpha(alpha) {
    fe.normalize_groups(fe.w);  // define_gibbs(true): normalize() first
    gps.assign(fe.w.size(), GibbsParam());
    for (size_t r = 0; r < gps.size(); ++r) {  // rules outside every group keep their weight (fixed probability)
      gps[r].norm = GibbsParam::NONORM;
      gps[r].prior = fe.w[r].getReal();
    }
    for (size_t gi = 0; gi < fe.groups.size(); ++gi)
      for (unsigned r : fe.groups[gi]) {
        const double a = r < alphas.size() ? alphas[r] : alpha;
        if (a < 0) continue;  // locked: stays a fixed-probability parameter (forest-em.hpp:704-706)
        gps[r].norm = (unsigned)gi;
        double p = fe.w[r].getReal(), N = (double)fe.groups[gi].size();
        gps[r].prior = gopt.uniformp0 ? a : a * p * N;  // gibbs.hpp:589-592
      }
    sample.assign(fe.forests.size(), {});
    // prior-scale groups as forest-em builds them (forest-em.hpp:723-734 to_gibbs; normalize.hpp:194-210): the norm ids
    // handed to define_param_id start at ONE (visit_norm_param's normi), while to_gibbs registers scale groups for ids
    // 0 .. G-1: norm group g (id g + 1) is scaled with group g + 1's factor -- scale index g + 2 --, the first factor is
    // drawn for nobody, and finish_params' resize(nnorm) (gibbs.hpp:572-579) gives the LAST norm group the never-scaled
    // index 0.  Every drawn factor enters q(old|new)/q(new|old) all the same.  Restated as it is.
    const unsigned G = (unsigned)fe.groups.size();
    unsigned nnorm = 0;
    for (auto& g : gps)
      if (g.has_norm()) nnorm = std::max(nnorm, g.norm + 2);  // reference id = norm + 1
    metanorm.assign(nnorm, 0u);
    for (unsigned i = 0; i < nnorm && i < G; ++i) metanorm[i] = i + 1;
    nexti = G + 1;
    if (gopt.prior_inference_global) {
      nexti = 2;
      std::fill(metanorm.begin(), metanorm.end(), 1u);
    }
    if (gopt.prior_inference_local) {
      nexti = nnorm + 1;
      for (unsigned i = 0; i < nnorm; ++i) metanorm[i] = i + 1;
    }
    cumulative.assign(nexti - 1, 1.0);
  }
  std::vector<unsigned> metanorm;  // by reference norm id (= our group index + 1)
  unsigned nexti = 1;
  std::vector<double> cumulative;
  void scale_priors(const std::vector<double>& sc, bool invert) {  // gibbs.hpp:161-176, 430-443
    for (unsigned i = 1; i < nexti; ++i) {
      if (invert) cumulative[i - 1] /= sc[i]; else cumulative[i - 1] *= sc[i];
    }
    for (auto& g : gps)
      if (g.has_norm()) {
        const unsigned i = metanorm[g.norm + 1];
        if (i > 0) {
          double f = sc[i];
          if (invert) f = 1. / f;
          const double s2 = f * g.prior, d = s2 - g.prior;
          g.sum.s += d * g.sum.tmax;
          g.sum.x += d;
          normsum[g.norm] += d;
          g.prior = s2;
        }
      }
  }
  LW cache_prob_all() const {  // gibbs.hpp:712-722
    std::vector<double> ccount(gps.size(), 0.0), csum(fe.groups.size(), 0.0);
    for (size_t i = 0; i < gps.size(); ++i)
      if (gps[i].has_norm()) csum[gps[i].norm] += (ccount[i] = gps[i].prior);
    LW w = LW::one();
    for (auto& blk : sample)
      for (unsigned r : blk) {
        const GibbsParam& g = gps[r];
        const double q = g.has_norm() ? (ccount[r]++ / csum[g.norm]++) : g.prior;
        mul_eq(w, LW::from_real(q));
      }
    return w;
  }
  void propose_new_priors(const std::function<double(unsigned, unsigned)>& u, double* tr6) {  // gibbs.hpp:525-553
    const double sdev = gopt.prior_inference_stddev;
    const double q0 = normal_cdf(0, 1, sdev), qrem = 1 - q0;
    std::vector<double> sc(nexti, 1.0);
    LW q2_1 = LW::one(), q1_2 = LW::one();
    for (unsigned i = 1; i < nexti; ++i) {
      sc[i] = normal_quantile(q0 + u(0xfffffffeu, i) * qrem, 1, sdev);
      mul_eq(q2_1, LW::from_real(normal_pdf(sc[i], 1, sdev)));
      mul_eq(q1_2, LW::from_real(normal_pdf(1 / sc[i], 1, sdev)));
    }
    const LW a2 = q1_2 / q2_1, p1 = cache_prob_all();
    scale_priors(sc, false);
    const LW p2 = cache_prob_all(), a = (p2 / p1) * a2;
    const bool accept = u(0xffffffffu, 0) < a.getReal();
    if (!accept) scale_priors(sc, true);
    tr6[0] = 1;
    tr6[1] = accept;
    tr6[2] = p1.w;
    tr6[3] = p2.w;
    tr6[4] = a2.getReal();
    tr6[5] = a.getReal();
  }
  double proposal_prob(unsigned r) const {
    const GibbsParam& g = gps[r];
    return g.has_norm() ? g.sum.x / normsum[g.norm] : g.prior;
  }
  void addc(const std::vector<unsigned>& b, double d) {
    for (unsigned r : b) {
      GibbsParam& g = gps[r];
      if (g.has_norm()) {
        normsum[g.norm] += d;
        g.sum.add_delta(d, time);
      }
    }
  }
  void run(const std::function<double(unsigned, unsigned, unsigned)>& u, GibbsTrace* tr = 0) {
    normsum.assign(fe.groups.size(), 0.0);
    for (auto& g : gps)
      if (g.has_norm()) {
        normsum[g.norm] += g.prior;
        g.sum.clear(g.prior);
      }
    for (auto& s : sample) s.clear();
    const unsigned Ni = gopt.iter;
    for (unsigned iter = 0; iter <= Ni; ++iter) {
      time = iter == 0 ? 0.0 : std::max(0.0, (double)iter - (double)gopt.burnin);
      LW p = LW::one(), pc = LW::one();
      std::vector<double> ccount(gps.size(), 0.0), csum(fe.groups.size(), 0.0);
      for (size_t i = 0; i < gps.size(); ++i)
        if (gps[i].has_norm()) csum[gps[i].norm] += (ccount[i] = gps[i].prior);
      for (unsigned b = 0; b < fe.forests.size(); ++b) {
        const Forest& f = fe.forests[b];
        addc(sample[b], -1.0);  // block_weight = 1 (gibbs.hpp:829-832)
        sample[b].clear();
        std::vector<LW> ins(f.nodes.size());
        f.inside_rec(0, [&](unsigned r) { return LW::from_real(proposal_prob(r)); }, ins, 0);
        unsigned step = 0;
        f.choose_random(0, ins, [&]() { return u(iter, b, step++); }, [&](unsigned r) { sample[b].push_back(r); }, gopt.power(iter));
        LW bp = LW::one(), bc = LW::one();
        for (unsigned r : sample[b]) mul_eq(bp, LW::from_real(proposal_prob(r)));
        for (unsigned r : sample[b]) {
          const GibbsParam& g = gps[r];
          double q = g.has_norm() ? (ccount[r]++ / csum[g.norm]++) : g.prior;
          mul_eq(bc, LW::from_real(q));
        }
        mul_eq(p, bp);
        mul_eq(pc, bc);
        addc(sample[b], 1.0);
      }
      double tr6[6] = {0, 0, 0, 0, 0, 0};
      {
        const unsigned start = gopt.prior_inference_start ? gopt.prior_inference_start : gopt.burnin;  // gibbs.hpp:559-563
        if (iter > 0 && gopt.prior_inference_stddev > 0 && start <= iter && (!gopt.prior_inference_end || iter < gopt.prior_inference_end))
          propose_new_priors([&](unsigned b, unsigned k) { return u(iter, b, k); }, tr6);
      }
      if (tr) {
        tr->prior_trace.insert(tr->prior_trace.end(), tr6, tr6 + 6);
        tr->iter_logprob.push_back(pc.w);
        tr->iter_cheap_logprob.push_back(p.w);
      }
    }
    if (tr) tr->cumulative = cumulative;
    if (tr) tr->last_sample = sample;
    if (!(gopt.final_counts && !gopt.exclude_prior)) {
      double tmax1 = ((double)Ni - (double)gopt.burnin) + 1;
      if (gopt.exclude_prior)  // gibbs.hpp:629-631
        for (auto& g : gps)
          if (g.has_norm()) {
            g.sum.s += -g.prior * g.sum.tmax;
            g.sum.x += -g.prior;
          }
      if (!gopt.final_counts)
        for (auto& g : gps)
          if (g.has_norm()) {
            g.sum.extend(tmax1);
            g.sum.x = g.sum.s;
          }
      normsum.assign(fe.groups.size(), 0.0);
      for (auto& g : gps)
        if (g.has_norm()) normsum[g.norm] += g.sum.x;
    }
    for (size_t r = 0; r < gps.size(); ++r) {  // from_gibbs forest-em.hpp:736-742
      const GibbsParam& g = gps[r];
      double pr = g.has_norm() ? (g.sum.x > 0 ? g.sum.x / normsum[g.norm] : 0.0) : g.prior;
      fe.w[r] = LW::from_real(pr);
    }
  }
};

}  // namespace oracle
