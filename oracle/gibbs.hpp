// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// gibbs.hpp: restatement of carmel's blocked Gibbs sampler over cached derivation lattices (`carmel --crp`).
// Follows /root/reference:
//   carmel/src/gibbs.cc:114-186   add_gibbs_params: one parameter per member arc; per norm group the prior
//                                 pseudo-count is alpha * p0 * |group| (alpha = --priors, p0 = arc weight / group sum;
//                                 just alpha with --uniform-p0); locked arcs and NONE members are fixed-probability
//   carmel/src/gibbs.cc:306-371   resample_block -> derivations::random_path; proposal weight of a lattice arc =
//                                 product of count/normsum over its chain; choose_arc records the chain's param ids
//   carmel/src/derivations.h:318-375  random_path: backward sweep, lazy per-state normalisation, choose_p walk
//   graehl/shared/random.ipp:111-127  choose_p (sum, choice = sum * random01(), subtract until negative)
//   graehl/shared/gibbs.hpp:106-227   gibbs_param (count, norm, time-weighted running sum)
//   graehl/shared/gibbs.hpp:803-877   run / iteration: per block remove old sample, resample, prob, add new
//   graehl/shared/gibbs.hpp:626-638   finalize_cumulative_counts; :153-157 proposal_prob; :141-150 final_prob
//   graehl/shared/delta_sum.hpp:49-106
//
// PARITY UNPINNED for the random stream: the reference draws from boost::lagged_fibonacci607 + uniform_01
// (random.hpp:139-157) and no reference test records a sampled sequence, so the uniforms are INJECTED here
// (callback u(iter, block, step)); everything downstream of the uniforms is restated exactly.
#pragma once
#include "cascade.hpp"
#include "deriv.hpp"
#include <functional>

namespace oracle {

struct DeltaSum {  // delta_sum.hpp:49-106
  double x, tmax, s;
  DeltaSum() : x(0), tmax(0), s(0) {}
  void clear(double x0) {
    x = x0;
    s = tmax = 0;
  }
  void add_delta(double d, double t) {
    double moret = t - tmax;
    if (moret > 0) {
      tmax = t;
      s += moret * x;
    } else if (moret < 0)
      s += d * (-moret);
    x += d;
  }
  void extend(double t) {
    double moret = t - tmax;
    if (moret < 0) throw std::runtime_error("delta_sum asked for forgotten sum at 0<t<tmax");
    tmax = t;
    s += x * moret;
  }
};

struct GibbsParam {
  static const unsigned NONORM = (unsigned)-1;
  double prior;
  unsigned norm;
  DeltaSum sum;
  bool has_norm() const { return norm != NONORM; }
};

struct GibbsOpts {
  unsigned iter = 0;    // -M / --crp=N : number of resampling sweeps after the initial sample
  unsigned burnin = 0;  // --burnin
  bool uniformp0 = false, dirichlet_p0 = false, final_counts = false, exclude_prior = false;
  unsigned restarts = 0;                           // --crp-restarts (gibbs.hpp:880-914)
  bool argmax_final = false, argmax_sum = false;   // --crp-argmax-final / --crp-argmax-sum (gibbs_opts.hpp:313-316)
  bool expectation = false;  // --expectation (gibbs_opts.hpp:125,166): fractional counts from a full forward/backward
                             // over the block instead of one sampled derivation ("online EM")
  bool include_self = false;  // --include-self (gibbs_opts.hpp:40-41, 162; gibbs.hpp:851-870): the block's own counts stay in
                              // while its proposal is formed; they leave just before the new ones go in
  bool random_start = false;  // --random-start (gibbs_opts.hpp:127-128, 167; gibbs.hpp:816, 860-864, 296-301): --expectation's
                              // initial per-entry counts are scaled by random01() each (implied for restarts)
  // prior-scale inference (gibbs_opts.hpp:82-89, 148-153; fst.h:553-600 --prior-groupby per member: 0 fixed, 1 single, 2 local)
  double prior_inference_stddev = 0;
  bool prior_inference_global = false, prior_inference_local = false, prior_inference_restart_fresh = false;
  unsigned prior_inference_start = 0, prior_inference_end = 0;
  std::vector<int> priorgroup;
  double high_temp = 1, low_temp = 1;  // --high-temp / --low-temp (gibbs_opts.hpp:50-53)
  // gibbs_opts.hpp:206-211 + time_series.hpp:90-141: the temperature runs from high_temp at sweep 0 to low_temp at
  // sweep `iter` along clamped_time_series(.., curvature = linear = -1e8); gibbs.hpp:838-839: power = 1/temperature
  double power(unsigned sweep) const {
    double temp;
    if (iter == 0 || high_temp == low_temp)
      temp = high_temp;
    else {
      const double x_origin = low_temp * -100000000.0, x0 = high_temp - x_origin, k = (low_temp - x_origin) / x0;
      const double t = (double)sweep, t_max = (double)iter;
      temp = t <= 0 ? x0 + x_origin : t >= t_max ? x0 * k + x_origin : x0 * std::pow(k, t / t_max) + x_origin;
    }
    return temp > 0 ? 1. / temp : 1.;
  }
};

struct GibbsTrace {
  std::vector<double> iter_logprob;               // ln cache-model prob of the whole sample (the default log line,
                                                  // gibbs_opts.hpp cache_prob = true; gibbs.hpp:712-742)
  std::vector<double> iter_cheap_logprob;         // ln of the product over blocks of the proposal prob (--sample-prob)
  std::vector<double> iter_after_logprob;         // ... with every block's prob taken AFTER its sample was added back: the
                                                  // "overestimate" of the comment at gibbs.hpp:866 -- what the older binary
                                                  // behind carmel-tutorial/commands.trace logged as its sample prob
  std::vector<std::vector<unsigned> > last_sample;  // per block: param ids of the final sample
  std::vector<double> prior_trace;                // per sweep: {proposed, accepted, ln p1, ln p2, a2, p_accept}
  std::vector<double> cumulative;                 // metanorm::cumulative after the last run
  // what --print-counts-* / --print-norms-* show (gibbs.hpp:970-1078), when want_state: after every sweep of every run, per
  // parameter in define_param order {x, s, tmax, prior} (gibbs_param::sumcount, prior), then after the runs the kept run's
  // counts as finalize_cumulative_counts left them and its final_prob
  bool want_state = false;
  std::vector<double> state;        // [(run, sweep)][param][4]
  std::vector<double> final_x, final_prob;
};

// boost::math::normal_distribution's cdf / quantile (gibbs.hpp:474-516; boost is absent from the tree): the cdf from erfc,
// the quantile by bisection on it polished with Newton steps -- deliberately not the rational approximation the product uses
inline double normal_cdf(double x, double mean, double sdev) { return 0.5 * std::erfc(-(x - mean) / (sdev * std::sqrt(2.0))); }
inline double normal_pdf(double x, double mean, double sdev) {
  const double z = (x - mean) / sdev;
  return std::exp(-0.5 * z * z) / (sdev * std::sqrt(2.0 * 3.14159265358979323846));
}
inline double normal_quantile(double p, double mean, double sdev) {
  double lo = -40, hi = 40;
  for (int i = 0; i < 200; ++i) {
    double mid = 0.5 * (lo + hi);
    if (normal_cdf(mid, 0, 1) < p) lo = mid; else hi = mid;
  }
  double z = 0.5 * (lo + hi);
  for (int i = 0; i < 3; ++i) {
    double pdf = normal_pdf(z, 0, 1);
    if (!(pdf > 1e-300)) break;
    double step = (normal_cdf(z, 0, 1) - p) / pdf;
    if (!(std::fabs(step) < 1e-3)) break;
    z -= step;
  }
  return mean + sdev * z;
}

struct CarmelGibbs {
  Wfst& composed;
  Cascade& cascade;
  Corpus& corpus;
  std::vector<NormalizeMethod> methods;
  GibbsOpts gopt;
  std::vector<GibbsParam> gps;
  std::vector<double> normsum;
  unsigned nnorm = 0;
  std::vector<std::vector<unsigned> > chain_params;  // composed arc id -> param ids in chain order
  std::vector<Derivations> derivs;                    // cached lattices of the pairs that have a derivation
  std::vector<std::vector<unsigned> > sample;
  std::vector<std::vector<double> > sample_wt;  // --expectation: block_delta::wt
  std::unordered_map<const Arc*, unsigned> param_of;  // member arc -> param id (the reference overwrites groupId)
  ArcTable arcs;
  double time = 0;

  CarmelGibbs(Wfst& composed, Cascade& cascade, Corpus& corpus, const std::vector<NormalizeMethod>& m, const GibbsOpts& g)
      : composed(composed), cascade(cascade), corpus(corpus), methods(m), gopt(g) {
    // WFST::train_gibbs (gibbs.cc:386-397): non-positive --priors become min_prior 0.01
    for (auto& x : methods)
      if (!(x.add_count > LW())) x.add_count = LW::from_real(1e-2);
    cascade.set_composed(&composed);
    arcs.build(composed, false, LW());
    IoIndex io;
    io.build(composed);
    for (auto& p : corpus.examples) {
      derivs.emplace_back();
      if (!derivs.back().compute(composed, io, arcs, p, true, 0)) derivs.pop_back();
    }
    unsigned norm = 0;
    for (size_t i = 0; i < cascade.cascade.size(); ++i)
      norm = add_gibbs_params(norm, *cascade.cascade[i], methods[i], i < gopt.priorgroup.size() ? gopt.priorgroup[i] : 1);
    // finish_params (gibbs.hpp:572-579)
    metanorm.resize(nnorm, 0u);
    if (gopt.prior_inference_global) {
      nexti = 2;
      std::fill(metanorm.begin(), metanorm.end(), 1u);
    }
    if (gopt.prior_inference_local) {
      nexti = nnorm + 1;
      for (unsigned i = 0; i < nnorm; ++i) metanorm[i] = i + 1;
    }
    cumulative.assign(nexti - 1, 1.0);
    // chains of the composed arcs as param ids (trivial cascade: each arc is its own chain, cascade.h:233-239)
    for (auto& r : arcs.t) {
      std::vector<unsigned> ch;
      if (cascade.trivial)
        ch.push_back(param_of.at(r.arc));
      else
        for (Arc* p : cascade.chains[r.arc->group]) ch.push_back(param_of.at(p));
      chain_params.push_back(ch);
    }
    sample.assign(derivs.size(), {});
  }

  unsigned define_param(unsigned norm, double prior) {
    if (norm != GibbsParam::NONORM && norm + 1 > nnorm) nnorm = norm + 1;
    GibbsParam g;
    g.prior = prior;
    g.norm = norm;
    gps.push_back(g);
    return (unsigned)gps.size() - 1;
  }
  // metanorm (gibbs.hpp:404-470): scale group of every norm group; 0 = never scaled
  std::vector<unsigned> metanorm;
  unsigned nexti = 1;
  std::vector<double> cumulative;
  // gibbs.cc:114-186
  unsigned add_gibbs_params(unsigned id, Wfst& w, const NormalizeMethod& nm, int pgroup) {
    if (nm.group == NORM_NONE) {
      for (auto& st : w.states)
        for (auto& a : st) param_of[&a] = define_param(GibbsParam::NONORM, a.weight.getReal());
      return id;
    }
    double alpha = nm.add_count.getReal();
    bool cond = nm.group == NORM_CONDITIONAL;
    w.for_each_norm_group(nm.group, [&](unsigned, std::vector<Arc*>& g) {
      LW sum;
      std::vector<Arc*> unlocked;
      for (Arc* a : g) {
        if (a->locked())
          param_of[a] = define_param(GibbsParam::NONORM, a->weight.getReal());
        else {
          unlocked.push_back(a);
          sum += a->weight;
        }
      }
      double N = (double)unlocked.size();
      if (gopt.dirichlet_p0) sum = LW::one();
      if (cond) std::reverse(unlocked.begin(), unlocked.end());
      for (Arc* a : unlocked) {
        double p0 = (a->weight / sum).getReal();
        param_of[a] = define_param(id, gopt.uniformp0 ? alpha : alpha * p0 * N);
      }
      if (metanorm.size() <= id) metanorm.resize(id + 1, 0u);
      metanorm[id] = pgroup == 0 ? 0u : nexti;  // gibbs.cc:132-137
      if (pgroup == 2) ++nexti;
      ++id;
    });
    if (pgroup == 1) ++nexti;  // gibbs.cc:184
    return id;
  }
  // metanorm::scale_priors + gibbs_param::scale_prior (gibbs.hpp:161-176, 430-443)
  void scale_priors(const std::vector<double>& sc, bool invert) {
    for (unsigned i = 1; i < nexti; ++i) {
      if (invert) cumulative[i - 1] /= sc[i]; else cumulative[i - 1] *= sc[i];
    }
    for (auto& g : gps)
      if (g.has_norm()) {
        unsigned i = metanorm[g.norm];
        if (i > 0) {
          double f = sc[i];
          if (invert) f = 1. / f;
          double s2 = f * g.prior, d = s2 - g.prior;
          g.sum.s += d * g.sum.tmax;  // delta_sum::addbase
          g.sum.x += d;
          normsum[g.norm] += d;
          g.prior = s2;
        }
      }
  }
  // cache_prob(recompute) (gibbs.hpp:712-722): the whole current sample under cache counts that restart at the priors
  LW cache_prob_all() {
    std::vector<double> ccount(gps.size(), 0.0), csum(nnorm, 0.0);
    for (size_t i = 0; i < gps.size(); ++i)
      if (gps[i].has_norm()) csum[gps[i].norm] += (ccount[i] = gps[i].prior);
    LW w = LW::one();
    for (auto& blk : sample)
      for (unsigned pid : blk) {
        const GibbsParam& g = gps[pid];
        double q = g.has_norm() ? (ccount[pid]++ / csum[g.norm]++) : g.prior;
        mul_eq(w, LW::from_real(q));
      }
    return w;
  }
  bool inferring(unsigned iter) const {  // gibbs.hpp:559-563
    unsigned start = gopt.prior_inference_start ? gopt.prior_inference_start : gopt.burnin;
    return gopt.prior_inference_stddev > 0 && start <= iter && (!gopt.prior_inference_end || iter < gopt.prior_inference_end);
  }
  // propose_new_priors (gibbs.hpp:525-553); u(0xfffffffe, k) draws scale group k, u(0xffffffff, 0) the acceptance
  void propose_new_priors(const std::function<double(unsigned, unsigned)>& u, double* tr6) {
    const double sdev = gopt.prior_inference_stddev;
    const double q0 = normal_cdf(0, 1, sdev), qrem = 1 - q0;
    std::vector<double> sc(nexti, 1.0);
    LW q2_1 = LW::one(), q1_2 = LW::one();
    for (unsigned i = 1; i < nexti; ++i) {
      sc[i] = normal_quantile(q0 + u(0xfffffffeu, i) * qrem, 1, sdev);
      mul_eq(q2_1, LW::from_real(normal_pdf(sc[i], 1, sdev)));
      mul_eq(q1_2, LW::from_real(normal_pdf(1 / sc[i], 1, sdev)));
    }
    LW a2 = q1_2 / q2_1;
    LW p1 = cache_prob_all();
    scale_priors(sc, false);
    LW p2 = cache_prob_all();
    LW a = (p2 / p1) * a2;
    bool accept = u(0xffffffffu, 0) < a.getReal();
    if (!accept) scale_priors(sc, true);
    if (tr6) {
      tr6[0] = 1;
      tr6[1] = accept;
      tr6[2] = p1.w;
      tr6[3] = p2.w;
      tr6[4] = a2.getReal();
      tr6[5] = a.getReal();
    }
  }

  double proposal_prob(unsigned p) const {  // gibbs.hpp:153-157
    const GibbsParam& g = gps[p];
    return g.has_norm() ? g.sum.x / normsum[g.norm] : g.prior;
  }
  double final_prob(unsigned p) const {  // gibbs.hpp:141-150
    const GibbsParam& g = gps[p];
    if (!g.has_norm()) return g.prior;
    return g.sum.x > 0 ? g.sum.x / normsum[g.norm] : 0;
  }
  void addc(const std::vector<unsigned>& b, double d) {  // gibbs.hpp:769-792, 210-217
    for (unsigned p : b) {
      GibbsParam& g = gps[p];
      if (g.has_norm()) {
        normsum[g.norm] += d;
        g.sum.add_delta(d, time);
      }
    }
  }
  // gibbs.hpp:783-792 with block_delta::wt (--expectation): every id carries its own fractional weight
  void addc_weighted(const std::vector<unsigned>& b, const std::vector<double>& w, double scale) {
    for (size_t i = 0; i < b.size(); ++i) {
      GibbsParam& g = gps[b[i]];
      if (g.has_norm()) {
        normsum[g.norm] += w[i] * scale;
        g.sum.add_delta(w[i] * scale, time);
      }
    }
  }
  // derivations.h:381-398 collect_counts_gibbs + carmel_gibbs::choose_arc(a, wt) gibbs.cc:367-371: forward/backward
  // with the proposal weights, then (param id, posterior of the lattice arc) for every chain element of every
  // lattice arc, states in order, each state's list order
  LW collect_counts_gibbs(Derivations& d, std::vector<unsigned>& ids, std::vector<double>& wts) {
    std::vector<LW> f, b;
    auto wf = [&](const GArc& a) { return arc_weight(a); };
    LW prob = d.compute_fb(f, b, wf);
    for (unsigned s = 0; s < d.g.size(); ++s) {
      const auto& arcs_ = d.g[s];
      for (size_t k = arcs_.size(); k-- > 0;) {
        const GArc& a = arcs_[k];
        LW contrib = arc_weight(a) * f[a.src] * b[a.dest];
        double wt = (contrib / prob).getReal();
        for (unsigned pid : chain_params[a.arcid]) {
          ids.push_back(pid);
          wts.push_back(wt);
        }
      }
    }
    return prob;
  }
  // --init-em (gibbs.cc:306-383 p_init, 386-430): ln weight per composed arc (arcs-table order) the first sweep of
  // the first run samples from; use_init is set for that sweep only
  std::vector<double> init_logw;
  bool use_init = false;
  LW arc_weight(const GArc& a) const {  // gibbs.cc:348-359
    if (use_init) return LW::from_ln(init_logw[a.arcid]);
    LW prob = LW::one();
    for (unsigned p : chain_params[a.arcid]) mul_eq(prob, LW::from_real(proposal_prob(p)));
    return prob;
  }

  // derivations.h:345-375 random_path; u(step) supplies random01()
  void random_path(Derivations& d, std::vector<unsigned>& out, const std::function<double(unsigned)>& u, double power) {
    size_t nst = d.g.size();
    d.make_order();
    std::vector<std::vector<GArc> > r;
    d.make_reverse(r);
    std::vector<LW> b(nst);
    b[d.fin] = LW::one();
    for (size_t t = 0; t < d.reverse_order.size(); ++t) {  // propagate_paths_in_order_wt over the reversed graph
      unsigned src = d.reverse_order[t];
      const auto& arcs_ = r[src];
      for (size_t k = arcs_.size(); k-- > 0;) b[arcs_[k].dest] += b[src] * arc_weight(arcs_[k]);
    }
    unsigned s = 0, step = 0;
    std::vector<char> normed(nst, 0);
    std::vector<std::vector<double> > probs(nst);
    while (s != d.fin) {
      auto& out_arcs = d.g[s];
      if (!normed[s]) {  // pfor::global_normalize derivations.h:318-337
        normed[s] = 1;
        LW sum;
        std::vector<LW> nw;
        for (size_t k = out_arcs.size(); k-- > 0;) {
          LW v = (arc_weight(out_arcs[k]) * b[out_arcs[k].dest]).pow(power);
          sum += v;
          nw.push_back(v);
        }
        if (sum.isZero()) sum = LW::one();
        for (auto& v : nw) probs[s].push_back((v / sum).getReal());  // list order
      }
      const auto& p = probs[s];
      double tot = 0;
      for (double x : p) tot += x;
      double choice = tot * u(step++);
      size_t pick = 0;
      for (size_t i = 0;;) {  // choose_p random.ipp:111-127
        choice -= p[i];
        size_t rr = i;
        ++i;
        if (choice < 0 || i == p.size()) {
          pick = rr;
          break;
        }
      }
      const GArc& a = out_arcs[out_arcs.size() - 1 - pick];  // list order = reverse insertion
      for (unsigned pid : chain_params[a.arcid]) out.push_back(pid);
      s = a.dest;
    }
  }

  // gibbs_stats (gibbs_opts.hpp:270-316)
  struct Stats {
    LW sumprob, allprob = LW::one(), finalprob = LW::one();
    double N = 0;
    void record(LW p) {
      N += 1;
      sumprob += p;
      mul_eq(allprob, p);
      finalprob = p;
    }
    bool better(const Stats& o, const GibbsOpts& g) const {
      return g.argmax_final ? finalprob > o.finalprob : (g.argmax_sum ? sumprob > o.sumprob : allprob > o.allprob);
    }
  };
  unsigned best_run = 0;
  // gibbs_base::run_starts (gibbs.hpp:880-914): restarts + 1 runs from the priors; the best by Stats::better gives the
  // final probabilities and sample.  The uniforms of run r, sweep i are those of sweep r * (iter + 1) + i.
  void run(const std::function<double(unsigned, unsigned, unsigned)>& u, GibbsTrace* tr = 0) {
    Stats best;
    std::vector<double> best_prob;
    std::vector<std::vector<unsigned> > best_sample;
    std::vector<double> priors0;
    const bool restart_priors = gopt.restarts > 0 && gopt.prior_inference_restart_fresh;  // gibbs.hpp:889-898
    if (restart_priors)
      for (auto& g : gps) priors0.push_back(g.prior);
    for (unsigned r = 0; r <= gopt.restarts; ++r) {
      if (r > 0 && restart_priors) {
        for (size_t i = 0; i < gps.size(); ++i) gps[i].prior = priors0[i];
        cumulative.assign(nexti - 1, 1.0);
      }
      Stats st = run_one(u, tr, r);
      if (r == 0 || st.better(best, gopt)) {
        best_run = r;
        best = st;
        finalize_cumulative_counts();
        best_prob.resize(gps.size());
        for (size_t i = 0; i < gps.size(); ++i) best_prob[i] = final_prob((unsigned)i);
        if (tr && tr->want_state) {
          tr->final_x.resize(gps.size());
          for (size_t i = 0; i < gps.size(); ++i) tr->final_x[i] = gps[i].sum.x;
          tr->final_prob = best_prob;
        }
        best_sample = sample;
      }
    }
    sample = best_sample;
    if (tr) tr->cumulative = cumulative;
    if (tr && !gopt.expectation) tr->last_sample = sample;  // (gibbs.cc:259-260: no single sample with --expectation)
    else if (tr) tr->last_sample.assign(sample.size(), {});
    // probs_to_cascade gibbs.cc:66-76
    for (Wfst* w : cascade.cascade)
      for (auto& st : w->states)
        for (auto& a : st) a.weight = LW::from_real(best_prob[param_of.at(&a)]);
  }
  // gibbs.hpp:803-877 (one run; no prior inference)
  Stats run_one(const std::function<double(unsigned, unsigned, unsigned)>& u, GibbsTrace* tr, unsigned run_index) {
    Stats stats;
    normsum.assign(nnorm, 0.0);
    for (auto& g : gps)
      if (g.has_norm()) {
        normsum[g.norm] += g.prior;
        g.sum.clear(g.prior);
      }
    for (auto& s : sample) s.clear();
    sample_wt.assign(sample.size(), {});
    const unsigned Ni = gopt.iter;
    for (unsigned iter = 0; iter <= Ni; ++iter) {
      time = iter == 0 ? 0.0 : std::max(0.0, (double)iter - (double)gopt.burnin);
      use_init = run_index == 0 && iter == 0 && !init_logw.empty();
      LW p = LW::one(), pc = LW::one(), pself = LW::one();
      // cache model (gibbs.hpp:678-742): counts restart from the priors every iteration and grow by one per use
      std::vector<double> ccount(gps.size(), 0.0), csum(nnorm, 0.0);
      for (size_t i = 0; i < gps.size(); ++i)
        if (gps[i].has_norm()) csum[gps[i].norm] += (ccount[i] = gps[i].prior);
      for (unsigned b = 0; b < derivs.size(); ++b) {
        double wt = derivs[b].weight;
        // gibbs.hpp:851-857: the old counts leave now, or (--include-self) are set aside and leave after the resampling
        std::vector<unsigned> self_ids;
        std::vector<double> self_wt;
        if (gopt.expectation) {  // gibbs.hpp:849-871 with gopt.expectation, gibbs.cc:311-314
          if (!gopt.include_self) addc_weighted(sample[b], sample_wt[b], -wt);
          else {
            self_ids.swap(sample[b]);
            self_wt.swap(sample_wt[b]);
          }
          sample[b].clear();
          sample_wt[b].clear();
          LW bprob = collect_counts_gibbs(derivs[b], sample[b], sample_wt[b]);
          // gibbs.hpp:816, 860-864: the initial sample's weights scaled by one random01() each; its probability is logged as 0
          if (iter == 0 && (gopt.random_start || run_index > 0)) {
            for (size_t k = 0; k < sample_wt[b].size(); ++k) sample_wt[b][k] *= u(run_index * (Ni + 1) + iter, b, (unsigned)k);
            bprob = LW::zero();
          }
          mul_eq(p, bprob);  // "sum-all-derivations" prob (gibbs.hpp:927-941)
          mul_eq(pc, bprob);
          if (gopt.include_self) addc_weighted(self_ids, self_wt, -wt);
          addc_weighted(sample[b], sample_wt[b], wt);
          continue;
        }
        if (!gopt.include_self) addc(sample[b], -wt);
        else self_ids.swap(sample[b]);
        sample[b].clear();
        random_path(derivs[b], sample[b], [&](unsigned step) { return u(run_index * (Ni + 1) + iter, b, step); }, gopt.power(iter));
        LW bp = LW::one();
        for (unsigned pid : sample[b]) mul_eq(bp, LW::from_real(proposal_prob(pid)));
        mul_eq(p, bp);
        LW bc = LW::one();
        for (unsigned pid : sample[b]) {
          const GibbsParam& g = gps[pid];
          double q = g.has_norm() ? (ccount[pid]++ / csum[g.norm]++) : g.prior;
          mul_eq(bc, LW::from_real(q));
        }
        mul_eq(pc, bc);
        if (gopt.include_self) addc(self_ids, -wt);
        addc(sample[b], wt);
        for (unsigned pid : sample[b]) mul_eq(pself, LW::from_real(proposal_prob(pid)));
      }

      double tr6[6] = {0, 0, 0, 0, 0, 0};
      if (iter > 0 && inferring(iter)) {  // gibbs.hpp:874-875
        if (gopt.expectation) throw std::runtime_error("prior inference not yet supported for expectation");
        const unsigned sweep = run_index * (Ni + 1) + iter;
        propose_new_priors([&](unsigned b, unsigned k) { return u(sweep, b, k); }, tr6);
      }
      if (tr && tr->want_state)
        for (auto& g : gps) {
          const double v[4] = {g.sum.x, g.sum.s, g.sum.tmax, g.prior};
          tr->state.insert(tr->state.end(), v, v + 4);
        }
      if (tr) {
        tr->prior_trace.insert(tr->prior_trace.end(), tr6, tr6 + 6);
        tr->iter_logprob.push_back(pc.w);
        tr->iter_cheap_logprob.push_back(p.w);
        tr->iter_after_logprob.push_back(pself.w);
      }
      if (iter >= gopt.burnin) stats.record(pc);  // gibbs.hpp:942-943 (the logged probability)
    }
    return stats;
  }
  // finalize_cumulative_counts gibbs.hpp:626-638
  void finalize_cumulative_counts() {
    const unsigned Ni = gopt.iter;
    if (!(gopt.final_counts && !gopt.exclude_prior)) {
      double tmax1 = ((double)Ni - (double)gopt.burnin) + 1;
      if (gopt.exclude_prior)
        for (auto& g : gps)
          if (g.has_norm()) {
            g.sum.s += -g.prior * g.sum.tmax;  // addbase(-prior)
            g.sum.x += -g.prior;
          }
      if (!gopt.final_counts)
        for (auto& g : gps)
          if (g.has_norm()) {
            g.sum.extend(tmax1);
            g.sum.x = g.sum.s;
          }
      normsum.assign(nnorm, 0.0);
      for (auto& g : gps)
        if (g.has_norm()) normsum[g.norm] += g.sum.x;
    }
  }
};

}  // namespace oracle
