// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// train.hpp: restatement of carmel's EM driver (WFST::train) and its E/M-step object (forward_backward).
// Follows /root/reference/carmel/src/:
//   train.cc:119-221   for_arcs::* per-arc M-step functors (prep_new_weights, overrelax, max_change, save_*)
//   train.cc:224-460   forward_backward (estimate_cached :315-322, operator() :326-332, save_best :449-455)
//   train.cc:503-678   WFST::train iteration control, log lines, convergence tests
//   train.cc:763-773   estimate;  :893-923 maximize
//   cached_derivs.h:60-138  lattice lifetime: cached once (-? / -:) or rebuilt per iteration; pairs without
//                      a derivation are dropped on the first pass and corpus stats recounted
//   fst.h:999-1044     random_restart_acceptor (only the no-restart path is pinned; restarts need Boost RNG)
//   fst.h:1080-1095    train_opts defaults: max_iter 500 (carmel.cc:896-897 -e 1e-4, -X .999)
#pragma once
#include "deriv.hpp"
#include "cascade.hpp"
#include <ostream>
#include <iostream>

namespace oracle {

struct TrainOpts {
  unsigned max_iter;  // (unsigned)-1 == "-M" without number? no: carmel sets -1 when -M given w/o value
  double learning_rate_growth_factor;
  unsigned ran_restarts;
  // random_restart_acceptor (fst.h:999-1044; carmel.cc:1426-1430): 0 = unset (tolerance infinite, final = tolerance,
  // final_restart = ran_restarts)
  double restart_tolerance = 0, final_restart_tolerance = 0;
  unsigned final_restart = 0;
  unsigned long long restart_seed;  // of the counter-based generator below (the reference's Boost stream is unpinned)
  bool cache_derivations;  // -? / -: (both are "cache" here; the reverse graph is always rebuilt)
  bool prune;
  TrainOpts() : max_iter(500), learning_rate_growth_factor(1), ran_restarts(0), restart_seed(0), cache_derivations(false), prune(true) {}
};

struct IterRecord {  // what one log line of train.cc:587-613 carries
  unsigned iter;
  double log2_prob;       // unweighted corpus prob, log2
  double ppx_symbol_log2; // per-symbol perplexity log2, N = max(n_in, n_out)
  double ppx_example_log2;
  double n_symbol, n_example;
  bool new_best;
  double rel_ppx_ratio_ln;  // ln of relative-perplexity-ratio (nan on first iteration)
  double last_change;       // max{d(weight)} printed on this line = change made by the previous M-step
};

struct ForwardBackward {
  Wfst& x;
  Cascade& cascade;
  Corpus& corpus;
  TrainOpts opts;
  ArcTable arcs;
  IoIndex io;
  bool io_built;
  std::vector<Derivations> derivs;  // cached lattices (pairs with no derivation dropped)
  bool first;
  DerivStats stats;
  LW weighted_corpus_prob;
  std::vector<double> last_pair_logprob;  // ln p(pair) in surviving-pair order, from the last estimate

  ForwardBackward(Wfst& x, Cascade& cascade, bool per_arc_prior, LW global_prior, const TrainOpts& o, Corpus& c)
      : x(x), cascade(cascade), corpus(c), opts(o), io_built(false), first(true) {
    // cached_derivs ctor runs before arcs_table in forward_backward's init list (train.cc:369-372) but both
    // only read x; cache_derivations (cached_derivs.h:104-138) recounts the corpus over surviving pairs.
    arcs.build(x, per_arc_prior, global_prior);
    cascade.set_composed(&x);
    if (opts.cache_derivations) cache_derivations();
  }

  void cache_derivations() {
    io.build(x);
    io_built = true;
    derivs.clear();
    for (auto& p : corpus.examples) {
      derivs.emplace_back();
      // QUIRK kept (cached_derivs.h:121): clear_counts() sits INSIDE the per-pair loop, so with -? / -: the
      // corpus statistics (n_pairs, totalEmpiricalWeight, n_input/n_output) end up describing only the last
      // pair.  Only the printed N= / per-example-perplexity are affected: relative_perplexity_ratio is
      // invariant to the normaliser (weight.h:247-249) and best-perplexity comparisons are monotone in it.
      corpus.clear_counts();
      if (!derivs.back().compute(x, io, arcs, p, opts.prune, &stats)) {
        derivs.pop_back();
      } else {
        corpus.count(p);
      }
    }
    // the reference keeps no-derivation pairs in corpus.examples when caching (only the lattice list skips them)
  }

  // train.cc:763-773 (+315-332, cached_derivs.h:60-101)
  LW estimate(LW& unweighted_corpus_prob) {
    for (auto& a : arcs.t) a.counts = LW();
    unweighted_corpus_prob = LW::one();
    weighted_corpus_prob = LW::one();
    last_pair_logprob.clear();
    if (opts.cache_derivations) {
      for (auto& d : derivs) visit(d, unweighted_corpus_prob);
    } else {
      if (!io_built) {
        io.build(x);
        io_built = true;
      }
      std::vector<Pair> keep;
      for (auto& p : corpus.examples) {
        Derivations d;
        if (d.compute(x, io, arcs, p, opts.prune, first ? &stats : 0)) {
          visit(d, unweighted_corpus_prob);
          if (first) keep.push_back(p);
        } else if (first && !opts.prune)
          keep.push_back(p);
      }
      if (first) {
        corpus.examples.swap(keep);
        corpus.count();
      }
    }
    first = false;
    if (corpus.examples.empty())  // train.cc:241-252 tests examples.empty(), not the counts
      throw std::runtime_error("No training example had a derivation - aborting training.");
    return weighted_corpus_prob;
  }
  void visit(Derivations& d, LW& unweighted) {  // train.cc:326-332
    LW prob = d.collect_counts(arcs);
    mul_eq(unweighted, prob);
    mul_eq(weighted_corpus_prob, prob.pow(d.weight));
    last_pair_logprob.push_back(prob.w);
  }

  // train.cc:893-923
  LW maximize(const std::vector<NormalizeMethod>& methods, double delta_scale) {
    cascade.save_none(methods);
    for (auto& a : arcs.t)  // prep_new_weights(1.0) train.cc:134-153
      if (!a.arc->locked()) {
        a.scratch = a.arc->weight;
        a.arc->weight = a.counts + a.prior_counts * LW::from_real(1.0);
      }
    if (cascade.trivial)
      cascade.cascade[0]->normalize(methods[0]);  // use_counts: distribute is a no-op, then normalize
    else
      cascade.use_counts(methods);
    cascade.load_none(methods);
    if (cascade.trivial) {
      for (auto& a : arcs.t) {  // overrelax train.cc:157-171
        a.em_weight = a.arc->weight;
        if (delta_scale > 1.)
          if (!a.arc->locked())
            if (a.scratch.isPositive()) a.arc->weight = a.scratch * ((a.em_weight / a.scratch).pow(delta_scale));
      }
      if (delta_scale > 1.) x.normalize(methods[0]);
      LW maxChange;  // max_change train.cc:173-182
      for (auto& a : arcs.t)
        if (!a.arc->locked()) {
          LW change = absdiff(a.arc->weight, a.scratch);
          if (change > maxChange) maxChange = change;
        }
      return maxChange;
    }
    return LW::from_real(10);
  }
  void save_best() {  // train.cc:449-455
    if (!cascade.trivial)
      for (auto& a : arcs.t) a.best_weight = a.em_weight;
    else
      for (auto& a : arcs.t) a.best_weight = a.arc->weight;
  }
  void load_best() {
    for (auto& a : arcs.t) a.arc->weight = a.best_weight;
  }
};

// The restart generator: the library's counter-based uniform u(seed, restart, parameter, 0) restated (splitmix64
// finaliser over the counters; the reference's Boost lagged_fibonacci607 stream is not pinned by any of its tests, so
// restart sequences are comparable with this build's GPU path only).  Parameters are numbered member by member in
// arc visit order; locked arcs and members normalised by NONE keep their weights (fst.h:976, cascade.h:398-401).
inline unsigned long long restart_mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double restart_uniform(unsigned long long seed, unsigned restart, unsigned param) {
  unsigned long long h = restart_mix64(seed ^ 0xD1B54A32D192ED03ull);
  h = restart_mix64(h ^ ((unsigned long long)restart << 32 | param));
  h = restart_mix64(h ^ 0ull);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
inline void restart_randomize(Cascade& cascade, const std::vector<NormalizeMethod>& methods, unsigned long long seed,
                              unsigned restart) {
  unsigned p = 0;
  for (size_t i = 0; i < cascade.cascade.size(); ++i)
    for (auto& st : cascade.cascade[i]->states)
      for (auto& a : st) {
        if (!a.locked() && methods[i].group != NORM_NONE) a.weight = LW::from_real(1.0 - restart_uniform(seed, restart, p));
        ++p;
      }
}

// train.cc:503-678.  Returns best per-example perplexity.  `log` receives the reference's log lines; `trace`
// (optional) receives one IterRecord per iteration.
inline LW train(Wfst& x, Cascade& cascade, Corpus& corpus, const std::vector<NormalizeMethod>& methods,
                bool weight_is_prior_count, LW smoothFloor, LW converge_arc_delta, LW converge_perplexity_ratio,
                const TrainOpts& opts, std::ostream* logp = 0, std::vector<IterRecord>* trace = 0) {
  std::ostream nul(0);
  std::ostream& log = logp ? *logp : nul;
  cascade.set_composed(&x);
  cascade.normalize(methods);
  unsigned ran_restarts = opts.ran_restarts;
  double learning_rate_growth_factor = opts.learning_rate_growth_factor;
  ForwardBackward fb(x, cascade, weight_is_prior_count, smoothFloor, opts, corpus);
  LW corpus_p;
  auto print_ppx_symbol = [&](LW p) {  // weight.h:314-329
    double n_symbol = std::max(corpus.n_output, corpus.n_input);
    log << "probability=" << lw_base2(p);
    if (n_symbol) log << " per-symbol-perplexity(N=" << n_symbol << ")=" << lw_base2(p.ppxper(n_symbol));
    if (corpus.n_pairs) log << " per-example-perplexity(N=" << corpus.n_pairs << ")=" << lw_base2(p.ppxper(corpus.n_pairs));
  };
  if (opts.max_iter + 1 == 0) return fb.estimate(corpus_p).ppxper(corpus.totalEmpiricalWeight);
  if (opts.max_iter == 0 || (opts.max_iter == 1 && opts.ran_restarts == 0)) {  // train.cc:520-538
    cascade.update();
    LW p = fb.estimate(corpus_p);
    log << "Corpus ";
    print_ppx_symbol(corpus_p);
    if (opts.max_iter == 0) {
      for (auto& a : fb.arcs.t)
        if (!a.arc->locked()) {
          a.scratch = a.arc->weight;
          a.arc->weight = a.counts + a.prior_counts;
        }
      cascade.distribute_counts();
    } else {
      fb.maximize(methods, 1);
      cascade.use_counts_final(methods);
    }
    log << "\n";
    return p.ppxper(corpus.totalEmpiricalWeight);
  }
  LW bestPerplexity = LW::inf();
  bool using_cascade = !cascade.trivial;
  if (using_cascade && learning_rate_growth_factor != 1) learning_rate_growth_factor = 1;
  bool have_good_weights = false;
  LW best_start = LW::inf();
  const double MAX_LEARNING_RATE_EXP = 20;  // fst.h MAX_LEARNING_RATE_EXP
  for (unsigned restart_no = 0;; ++restart_no) {
    unsigned train_iter = 0;
    LW lastChange = LW::from_real(10);
    LW lastPerplexity = LW::inf();
    double learning_rate = 1;
    bool last_was_reset = false;
    for (;;) {
      const bool first_time = train_iter == 0;
      ++train_iter;
      bool cascade_counts = using_cascade && !first_time;
      if (cascade_counts)
        for (auto& a : fb.arcs.t) a.em_weight = a.arc->weight;  // save_counts train.cc:123-125
      cascade.update();
      if (~opts.max_iter && train_iter > opts.max_iter && have_good_weights) {
        log << "Maximum number of iterations (" << opts.max_iter
            << ") reached before convergence criteria was met - greatest arc weight change was "
            << lw_str(lastChange) << "\n";
        break;
      }
      LW p = fb.estimate(corpus_p);
      LW newPerplexity = p.ppxper(corpus.totalEmpiricalWeight);
      log << "i=" << train_iter << " (rate=" << learning_rate << "): ";
      print_ppx_symbol(corpus_p);
      IterRecord rec;
      rec.iter = train_iter;
      rec.log2_prob = corpus_p.w / std::log(2.0);
      rec.n_symbol = std::max(corpus.n_output, corpus.n_input);
      rec.n_example = corpus.n_pairs;
      rec.ppx_symbol_log2 = corpus_p.ppxper(rec.n_symbol).w / std::log(2.0);
      rec.ppx_example_log2 = corpus_p.ppxper(rec.n_example).w / std::log(2.0);
      rec.new_best = false;
      rec.rel_ppx_ratio_ln = std::numeric_limits<double>::quiet_NaN();
      rec.last_change = lastChange.getReal();
      if (newPerplexity < bestPerplexity && (!using_cascade || cascade_counts)) {
        log << " (new best)";
        rec.new_best = true;
        bestPerplexity = newPerplexity;
        have_good_weights = true;
        fb.save_best();
      }
      LW pp_ratio_scaled;
      if (first_time) {
        log << std::endl;
        // random_restart_acceptor::accept (fst.h:1017-1040): restart 0 is always accepted and sets the yardstick
        if (restart_no == 0) {
          best_start = newPerplexity;
          log << "Initial best start point ppx=" << lw_base2(newPerplexity) << "\n";
        } else {
          // likelihood_ratio (fst.h:1017-1021): tolerance moves to final_tolerance over restarts 1..N
          LW tol = opts.restart_tolerance > 0 ? LW::from_real(opts.restart_tolerance) : LW::inf();
          LW fin = opts.final_restart_tolerance > 0 ? LW::from_real(opts.final_restart_tolerance) : tol;
          const double N = opts.final_restart ? opts.final_restart : opts.ran_restarts;
          LW lr = restart_no >= N ? fin : tol.isInfinity() ? tol : tol * (fin / tol).pow((restart_no - 1) / (N - 1));
          LW ppr = relative_perplexity_ratio(newPerplexity, best_start);
          const bool ok = lr > ppr;
          log << "For restart " << restart_no << ", " << (ok ? "accepting" : "rejecting") << " worse random start of "
              << lw_base2(newPerplexity) << " compared to " << lw_base2(best_start) << " with relative ppx ratio="
              << lw_str(ppr) << " compared to target of " << (lr.isInfinity() ? std::string("inf") : lw_str(lr)) << "\n";
          if (!ok) {
            log << "Random start was insufficiently promising; trying another." << std::endl;
            if (trace) trace->push_back(rec);
            break;  // to the next random restart
          }
        }
        pp_ratio_scaled = LW();
      } else {
        pp_ratio_scaled = relative_perplexity_ratio(newPerplexity, lastPerplexity);
        rec.rel_ppx_ratio_ln = pp_ratio_scaled.w;
        log << " (relative-perplexity-ratio=" << lw_str(pp_ratio_scaled) << ")";
        if (lastChange < LW::from_real(1)) log << ", max {d(weight)}=" << lw_str(lastChange);
        log << std::endl;
      }
      if (trace) trace->push_back(rec);
      if (!last_was_reset) {
        if (pp_ratio_scaled >= converge_perplexity_ratio) {
          if (learning_rate > 1) {
            log << "Failed to improve (relaxation rate too high); starting again at learning rate 1" << std::endl;
            learning_rate = 1;
            for (auto& a : fb.arcs.t) a.arc->weight = a.em_weight;  // keep_em_weight
            last_was_reset = true;
            continue;
          }
          log << "Converged - per-example perplexity ratio exceeds " << lw_str(converge_perplexity_ratio) << " after "
              << train_iter << " iterations.\n";
          if (!have_good_weights)
            log << "Because of the --train-cascade implementation, we need another iteration even though "
                   "we've converged.\n";
          else
            break;
        } else {
          if (learning_rate < MAX_LEARNING_RATE_EXP) learning_rate *= learning_rate_growth_factor;
        }
      } else
        last_was_reset = false;
      lastChange = fb.maximize(methods, learning_rate);
      if (lastChange <= converge_arc_delta && have_good_weights) {
        log << "Converged - maximum weight change less than " << lw_str(converge_arc_delta) << " after " << train_iter
            << " iterations.\n";
        break;
      }
      lastPerplexity = newPerplexity;
    }
    if (ran_restarts > 0) {  // train.cc:660-663; cascade.h:398-411 random_restart = randomSet + normalize
      --ran_restarts;
      restart_randomize(cascade, methods, opts.restart_seed, restart_no + 1);
      cascade.normalize(methods);
      log << "\nRandom restart - " << ran_restarts << " remaining.\n";
    } else
      break;
  }
  log << "Setting weights to model with lowest per-example-perplexity ( = "
         "prod[modelprob(example)]^(-1/num_examples) = 2^(-log_2(p_model(corpus))/N) = "
      << lw_base2(bestPerplexity) << std::endl;
  fb.load_best();
  cascade.use_counts_final(methods);
  return bestPerplexity;
}

}  // namespace oracle
