// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// C entry points over the CPU restatement so tests/ (ctypes), __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive it with the same flat arrays the product's C-ABI (include/carmel_hip.h) takes.
// Arc arrays are in the reference's arc-id order: state-major, each state's arcs in list order
// (derivations.h:86-101, fst.h:1331-1334).
#include "train.hpp"
#include "matrix.hpp"
#include "fem.hpp"
#include "gibbs.hpp"
#include "forest.hpp"
#include <pthread.h>
#include <functional>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oracle;

namespace {
struct BigStack {
  std::function<void()> f;
  std::string err;
};
void* tramp(void* p) {
  BigStack* b = (BigStack*)p;
  try {
    b->f();
  } catch (std::exception& e) {
    b->err = e.what();
  } catch (...) {
    b->err = "unknown exception";
  }
  return 0;
}
thread_local std::string g_err;
// the lattice builder recurses like the reference (derivations.h:640-704): run on a deep stack
int run_big_stack(std::function<void()> f) {
  BigStack b;
  b.f = f;
  pthread_attr_t attr;
  pthread_attr_init(&attr);
  pthread_attr_setstacksize(&attr, (size_t)1 << 30);
  pthread_t th;
  if (pthread_create(&th, &attr, tramp, &b)) {
    g_err = "pthread_create failed";
    return -1;
  }
  pthread_join(th, 0);
  if (!b.err.empty()) {
    g_err = b.err;
    return -1;
  }
  return 0;
}
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

struct orc_wfst {
  Wfst w;
};
struct orc_corpus {
  Corpus c;
};

orc_wfst* orc_wfst_from_arrays(uint32_t n_states, uint32_t final_state, uint64_t n_arcs, const uint32_t* src,
                               const uint32_t* dst, const uint32_t* in, const uint32_t* out, const double* logw,
                               const uint32_t* group) {
  orc_wfst* h = new orc_wfst();
  h->w.states.assign(n_states, {});
  h->w.final_state = final_state;
  uint32_t max_in = 1, max_out = 1;
  for (uint64_t k = 0; k < n_arcs; ++k) {
    h->w.states[src[k]].push_back(Arc(in[k], out[k], dst[k], LW::from_ln(logw[k]), group ? group[k] : NO_GROUP));
    if (in[k] > max_in) max_in = in[k];
    if (out[k] > max_out) max_out = out[k];
  }
  for (uint32_t i = 2; i <= max_in; ++i) h->w.in_alph.index_of("i" + std::to_string(i));
  for (uint32_t i = 2; i <= max_out; ++i) h->w.out_alph.index_of("o" + std::to_string(i));
  return h;
}
orc_wfst* orc_wfst_parse(const char* text, int always_named) {
  orc_wfst* h = new orc_wfst();
  if (!h->w.read_legible(text, always_named != 0)) {
    g_err = "bad WFST text";
    delete h;
    return 0;
  }
  return h;
}
void orc_wfst_free(orc_wfst* h) { delete h; }
void orc_wfst_dims(orc_wfst* h, uint32_t* n_states, uint64_t* n_arcs, uint32_t* final_state) {
  *n_states = h->w.num_states();
  *n_arcs = h->w.num_arcs();
  *final_state = h->w.final_state;
}
void orc_wfst_export(orc_wfst* h, uint32_t* src, uint32_t* dst, uint32_t* in, uint32_t* out, double* logw,
                     uint32_t* group) {
  uint64_t k = 0;
  for (uint32_t s = 0; s < h->w.num_states(); ++s)
    for (auto& a : h->w.states[s]) {
      src[k] = s;
      dst[k] = a.dest;
      in[k] = a.in;
      out[k] = a.out;
      logw[k] = a.weight.w;
      group[k] = a.group;
      ++k;
    }
}
void orc_wfst_set_logw(orc_wfst* h, const double* logw) {
  uint64_t k = 0;
  for (auto& st : h->w.states)
    for (auto& a : st) a.weight.w = logw[k++];
}
void orc_wfst_reduce(orc_wfst* h) { h->w.reduce(); }
void orc_wfst_normalize(orc_wfst* h, int group, double add_count_real) {
  NormalizeMethod m;
  m.group = group;
  m.add_count = LW::from_real(add_count_real);
  h->w.normalize(m);
}
// text of the transducer in carmel's output syntax (wfstio.cc:594-625); caller frees with orc_free_str
char* orc_wfst_write(orc_wfst* h, int full, int onearc, int wmode) {
  std::string s = h->w.write_legible(full != 0, onearc != 0, wmode);
  char* r = (char*)std::malloc(s.size() + 1);
  std::memcpy(r, s.c_str(), s.size() + 1);
  return r;
}
void orc_free_str(char* s) { std::free(s); }
uint32_t orc_wfst_alphabet_size(orc_wfst* h, int output) { return output ? h->w.out_alph.size() : h->w.in_alph.size(); }

orc_corpus* orc_corpus_from_arrays(uint64_t n_pairs, const uint64_t* in_off, const uint32_t* in_sym,
                                   const uint64_t* out_off, const uint32_t* out_sym, const double* weight) {
  orc_corpus* h = new orc_corpus();
  for (uint64_t p = 0; p < n_pairs; ++p) {
    std::vector<unsigned> i(in_sym + in_off[p], in_sym + in_off[p + 1]);
    std::vector<unsigned> o(out_sym + out_off[p], out_sym + out_off[p + 1]);
    h->c.add(i, o, weight ? weight[p] : 1.0);
  }
  return h;
}
orc_corpus* orc_corpus_parse(orc_wfst* w, const char* text) {
  orc_corpus* h = new orc_corpus();
  read_training_corpus(w->w, text, h->c);
  return h;
}
void orc_corpus_free(orc_corpus* h) { delete h; }
void orc_corpus_dims(orc_corpus* h, uint64_t* n_pairs, uint64_t* n_in, uint64_t* n_out) {
  *n_pairs = h->c.examples.size();
  uint64_t a = 0, b = 0;
  for (auto& p : h->c.examples) {
    a += p.in.size();
    b += p.out.size();
  }
  *n_in = a;
  *n_out = b;
}
void orc_corpus_export(orc_corpus* h, uint64_t* in_off, uint32_t* in_sym, uint64_t* out_off, uint32_t* out_sym,
                       double* weight) {
  uint64_t a = 0, b = 0, k = 0;
  for (auto& p : h->c.examples) {
    in_off[k] = a;
    out_off[k] = b;
    for (unsigned s : p.in) in_sym[a++] = s;
    for (unsigned s : p.out) out_sym[b++] = s;
    weight[k] = p.weight;
    ++k;
  }
  in_off[k] = a;
  out_off[k] = b;
}

// One E-step over the dense (input, output, state) matrix (carmel --matrix-fb; matrix.hpp).  counts_ln / pair_logprob as
// orc_estimate; sums[3] = sum ln p, sum weight * ln p, number of back edges of the *e*:*e* graph.
int orc_estimate_matrix(orc_wfst* wh, orc_corpus* ch, double* counts_ln, double* pair_logprob, double* sums) {
  return run_big_stack([&]() {
    Wfst& x = wh->w;
    ArcTable arcs;
    arcs.build(x, false, LW());
    MatrixFB m(x, arcs);
    LW un;
    std::vector<double> pl;
    LW ret = m.estimate(ch->c, un, &pl);
    if (counts_ln)
      for (size_t k = 0; k < arcs.t.size(); ++k) counts_ln[k] = arcs.t[k].counts.w;
    if (pair_logprob)
      for (size_t k = 0; k < pl.size(); ++k) pair_logprob[k] = pl[k];
    if (sums) {
      sums[0] = un.w;
      sums[1] = ret.w;
      sums[2] = m.n_back_edges;
    }
    return 0;
  });
}

// One E-step over the whole corpus at the transducer's current weights (train.cc:763-773 / derivations.h:432-449).
//   counts_ln[n_arcs]      ln of the expected count per WFST arc (-inf = none)
//   pair_logprob[n_pairs]  ln p(pair); -inf where the pair has no derivation
//   has_deriv[n_pairs]
//   stats[4]               lattice states/arcs explored, states/arcs kept (derivations.h:191-247)
//   sums[2]                sum ln p, sum weight*ln p over pairs with a derivation
// n_threads <= 1: serial, corpus order, exactly the reference's accumulation order.  n_threads > 1: OpenMP over
// pairs with per-thread log-domain count tables merged at the end (same values up to log-add rounding).
int orc_estimate(orc_wfst* wh, orc_corpus* ch, int prune, double* counts_ln, double* pair_logprob, uint8_t* has_deriv,
                 double* stats, double* sums, int n_threads) {
  return run_big_stack([&]() {
    Wfst& x = wh->w;
    ArcTable arcs;
    arcs.build(x, false, LW());
    IoIndex io;
    io.build(x);
    size_t np = ch->c.examples.size();
    DerivStats st;
    double s0 = 0, s1 = 0;
    if (n_threads <= 1) {
      for (auto& a : arcs.t) a.counts = LW();
      for (size_t p = 0; p < np; ++p) {
        Derivations d;
        bool ok = d.compute(x, io, arcs, ch->c.examples[p], prune != 0, &st);
        if (has_deriv) has_deriv[p] = ok;
        double lp = -std::numeric_limits<double>::infinity();
        if (ok) {
          LW prob = d.collect_counts(arcs);
          lp = prob.w;
          s0 += lp;
          s1 += lp * d.weight;
        }
        if (pair_logprob) pair_logprob[p] = lp;
      }
      if (counts_ln)
        for (size_t k = 0; k < arcs.t.size(); ++k) counts_ln[k] = arcs.t[k].counts.w;
    } else {
#ifdef _OPENMP
      omp_set_num_threads(n_threads);
#endif
      std::vector<std::vector<LW> > tc;
      std::vector<DerivStats> tst;
#pragma omp parallel
      {
#pragma omp single
        {
          int nt = 1;
#ifdef _OPENMP
          nt = omp_get_num_threads();
#endif
          tc.assign(nt, std::vector<LW>(arcs.t.size()));
          tst.assign(nt, DerivStats());
        }
        int me = 0;
#ifdef _OPENMP
        me = omp_get_thread_num();
#endif
        ArcTable mine = arcs;  // private count column
        for (auto& a : mine.t) a.counts = LW();
        double ls0 = 0, ls1 = 0;
#pragma omp for schedule(dynamic, 64)
        for (long p = 0; p < (long)np; ++p) {
          Derivations d;
          bool ok = d.compute(x, io, mine, ch->c.examples[p], prune != 0, &tst[me]);
          if (has_deriv) has_deriv[p] = ok;
          double lp = -std::numeric_limits<double>::infinity();
          if (ok) {
            LW prob = d.collect_counts(mine);
            lp = prob.w;
            ls0 += lp;
            ls1 += lp * d.weight;
          }
          if (pair_logprob) pair_logprob[p] = lp;
        }
        for (size_t k = 0; k < mine.t.size(); ++k) tc[me][k] = mine.t[k].counts;
#pragma omp critical
        {
          s0 += ls0;
          s1 += ls1;
        }
      }
      for (auto& t : tst) {
        st.N += t.N;
        st.pre_states += t.pre_states;
        st.pre_arcs += t.pre_arcs;
        st.post_states += t.post_states;
        st.post_arcs += t.post_arcs;
      }
      if (counts_ln)
        for (size_t k = 0; k < arcs.t.size(); ++k) {
          LW c;
          for (auto& t : tc) c += t[k];
          counts_ln[k] = c.w;
        }
    }
    if (stats) {
      stats[0] = st.pre_states;
      stats[1] = st.pre_arcs;
      stats[2] = st.post_states;
      stats[3] = st.post_arcs;
    }
    if (sums) {
      sums[0] = s0;
      sums[1] = s1;
    }
  });
}

// Lattice of one pair in the reference's own numbering: states in DFS pre-order after stable pruning, each
// state's out-arcs in list order (derivations.h:640-704, 572-629).  Call with null arrays to get the sizes.
int orc_lattice(orc_wfst* wh, orc_corpus* ch, uint64_t pair, int prune, uint32_t* n_states, uint64_t* n_arcs,
                uint32_t* fin, uint32_t* a_src, uint32_t* a_dst, uint32_t* a_arcid, uint32_t* order, uint32_t* n_back) {
  return run_big_stack([&]() {
    Wfst& x = wh->w;
    ArcTable arcs;
    arcs.build(x, false, LW());
    IoIndex io;
    io.build(x);
    Derivations d;
    bool ok = d.compute(x, io, arcs, ch->c.examples[pair], prune != 0, 0);
    if (!ok) {
      *n_states = 0;
      *n_arcs = 0;
      return;
    }
    *n_states = (uint32_t)d.n_states();
    *n_arcs = d.n_arcs();
    *fin = d.fin;
    d.make_order();
    if (n_back) *n_back = d.n_back_edges;
    if (order)
      for (size_t i = 0; i < d.reverse_order.size(); ++i) order[i] = d.reverse_order[i];
    if (a_src) {
      uint64_t k = 0;
      for (unsigned s = 0; s < d.g.size(); ++s)
        for (size_t j = d.g[s].size(); j-- > 0;) {
          a_src[k] = d.g[s][j].src;
          a_dst[k] = d.g[s][j].dest;
          a_arcid[k] = d.g[s][j].arcid;
          ++k;
        }
    }
  });
}

// Full EM run for a single transducer (trivial cascade): WFST::train (train.cc:503-678).
// trace rows: [iter, log2 P, log2 ppx/symbol, log2 ppx/example, new_best, ln rel-ppx-ratio, last_change, n_example]
static double g_rate_growth = 1.0;  // carmel -o (train_opts::learning_rate_growth_factor), set before orc_train
void orc_set_rate_growth(double g) { g_rate_growth = g < 1 ? 1 : g; }
static double g_high_temp = 1.0, g_low_temp = 1.0;  // --high-temp / --low-temp, set before orc_gibbs / orc_forests_gibbs
double orc_gibbs_power(double high, double low, uint32_t iter, uint32_t sweep) {
  GibbsOpts go;
  go.iter = iter;
  go.high_temp = high;
  go.low_temp = low;
  return go.power(sweep);
}
static int g_expectation = 0;  // --expectation, set before orc_gibbs_run
void orc_set_gibbs_expectation(int on) { g_expectation = on; }
static int g_include_self = 0, g_random_start = 0;  // --include-self / --random-start, set before orc_gibbs_run
void orc_set_gibbs_self_start(int include_self, int random_start) {
  g_include_self = include_self;
  g_random_start = random_start;
}
static unsigned g_crp_restarts = 0;  // --crp-restarts / --crp-argmax-final / --crp-argmax-sum, set before orc_gibbs_run
static int g_argmax_final = 0, g_argmax_sum = 0, g_best_run = 0;
void orc_set_gibbs_restarts(unsigned n, int argmax_final, int argmax_sum) {
  g_crp_restarts = n;
  g_argmax_final = argmax_final;
  g_argmax_sum = argmax_sum;
}
int orc_gibbs_best_run() { return g_best_run; }
static unsigned g_init_em = 0;  // --init-em=N / --em-p0 (gibbs.cc:386-430), set before orc_gibbs_run
static int g_em_p0 = 0;
void orc_set_gibbs_init_em(unsigned n, int em_p0) {
  g_init_em = n;
  g_em_p0 = em_p0;
}
static int g_init_from_p0 = 0;  // --init-from-p0 (gibbs.cc:405-421)
void orc_set_gibbs_init_from_p0(int on) { g_init_from_p0 = on; }
// --prior-inference-* / --prior-groupby, set before orc_gibbs_run; the trace of the last run
static double g_pi_stddev = 0;
static int g_pi_global = 0, g_pi_local = 0, g_pi_fresh = 0;
static unsigned g_pi_start = 0, g_pi_end = 0;
static std::vector<int> g_pi_groupby;
static std::vector<double> g_last_prior_trace, g_last_cumulative;
void orc_set_gibbs_prior_inference(double stddev, int global, int local, int restart_fresh, uint32_t start, uint32_t end,
                                   const int* groupby, uint32_t n) {
  g_pi_stddev = stddev;
  g_pi_global = global;
  g_pi_local = local;
  g_pi_fresh = restart_fresh;
  g_pi_start = start;
  g_pi_end = end;
  g_pi_groupby.assign(groupby, groupby + (groupby ? n : 0));
}
uint32_t orc_gibbs_last_prior_trace(double* out6, uint32_t n_sweeps, double* cumulative, uint32_t n_cum) {
  for (size_t k = 0; k < (size_t)n_sweeps * 6; ++k) out6[k] = k < g_last_prior_trace.size() ? g_last_prior_trace[k] : 0.0;
  for (uint32_t k = 0; k < n_cum; ++k) cumulative[k] = k < g_last_cumulative.size() ? g_last_cumulative[k] : 1.0;
  return (uint32_t)g_last_cumulative.size();
}
// --print-counts-* / --print-norms-*: the sampler's per-parameter state after every sweep (GibbsTrace::state), kept when asked
static int g_want_state = 0;
static std::vector<double> g_last_state, g_last_final;  // state: [(run, sweep)][param in MEMBER-ARC order][4]; final: [param][2] = {x, prob}
static std::vector<int32_t> g_last_ids;                 // per member-arc: {define_param id, norm id or -1, scale group (metanorm) or 0}
void orc_set_gibbs_state_trace(int on) { g_want_state = on; }
uint64_t orc_gibbs_last_state(double* state, uint64_t n_state, double* fin, uint64_t n_fin, int32_t* ids, uint64_t n_ids) {
  for (uint64_t i = 0; i < n_state && i < g_last_state.size(); ++i) state[i] = g_last_state[i];
  for (uint64_t i = 0; i < n_fin && i < g_last_final.size(); ++i) fin[i] = g_last_final[i];
  for (uint64_t i = 0; i < n_ids && i < g_last_ids.size(); ++i) ids[i] = g_last_ids[i];
  return g_last_state.size();
}
static std::vector<double> g_last_after;  // GibbsTrace::iter_after_logprob of the last orc_gibbs_run
void orc_gibbs_last_after(double* out, uint32_t n) {
  for (uint32_t i = 0; i < n && i < g_last_after.size(); ++i) out[i] = g_last_after[i];
}
void orc_set_gibbs_temps(double high, double low) {
  g_high_temp = high;
  g_low_temp = low;
}
int orc_train(orc_wfst* wh, orc_corpus* ch, int norm_group, double add_count, int weight_is_prior_count,
              double smooth_floor, double converge_arc_delta, double converge_ppx_ratio, int max_iter, int cache,
              int prune, double* trace, int max_trace, int* n_trace, double* best_ppx_ln) {
  return run_big_stack([&]() {
    Cascade cascade(false);
    std::vector<NormalizeMethod> nms(1);
    nms[0].group = norm_group;
    nms[0].add_count = LW::from_real(add_count);
    TrainOpts opts;
    opts.max_iter = (unsigned)max_iter;
    opts.cache_derivations = cache != 0;
    opts.prune = prune != 0;
    opts.learning_rate_growth_factor = g_rate_growth;
    std::vector<IterRecord> tr;
    LW best = train(wh->w, cascade, ch->c, nms, weight_is_prior_count != 0, LW::from_real(smooth_floor),
                    LW::from_real(converge_arc_delta), LW::from_real(converge_ppx_ratio), opts, 0, &tr);
    if (best_ppx_ln) *best_ppx_ln = best.w;
    int n = 0;
    for (auto& r : tr) {
      if (n >= max_trace) break;
      double* row = trace + 8 * n;
      row[0] = r.iter;
      row[1] = r.log2_prob;
      row[2] = r.ppx_symbol_log2;
      row[3] = r.ppx_example_log2;
      row[4] = r.new_best;
      row[5] = r.rel_ppx_ratio_ln;
      row[6] = r.last_change;
      row[7] = r.n_example;
      ++n;
    }
    if (n_trace) *n_trace = n;
  });
}

// CPU baseline for bench.py: lattices are built once (like carmel -:), then `iters` EM iterations
// (estimate over the cached lattices + maximize) are timed.  threads <= 1: the scalar restatement, corpus order.
// threads > 1: OpenMP over cached lattices with per-thread log-domain count tables.
// out[0] = seconds per iteration, out[1] = lattice arcs swept per iteration, out[2] = lattice build seconds,
// out[3] = ln corpus prob of the last estimate
int orc_bench_em(orc_wfst* wh, orc_corpus* ch, int norm_group, int iters, int threads, double* out) {
  return run_big_stack([&]() {
    Wfst& x = wh->w;
    Cascade cascade(false);
    std::vector<NormalizeMethod> nms(1);
    nms[0].group = norm_group;
    cascade.set_composed(&x);
    cascade.normalize(nms);
    TrainOpts opts;
    opts.cache_derivations = true;
    auto t0 = std::chrono::steady_clock::now();
    ForwardBackward fb(x, cascade, false, LW(), opts, ch->c);
    auto t1 = std::chrono::steady_clock::now();
    double arcs = 0;
    for (auto& d : fb.derivs) arcs += (double)d.n_arcs();
    LW corpus_p;
    auto run_iter = [&]() {
      if (threads <= 1) {
        fb.estimate(corpus_p);
      } else {
#ifdef _OPENMP
        omp_set_num_threads(threads);
#endif
        for (auto& a : fb.arcs.t) a.counts = LW();
        size_t na = fb.arcs.t.size();
        double sum = 0;
        std::vector<std::vector<LW> > tc;
#pragma omp parallel
        {
#pragma omp single
          {
            int nt = 1;
#ifdef _OPENMP
            nt = omp_get_num_threads();
#endif
            tc.assign(nt, std::vector<LW>(na));
          }
          int me = 0;
#ifdef _OPENMP
          me = omp_get_thread_num();
#endif
          double ls = 0;
#pragma omp for schedule(dynamic, 64)
          for (long p = 0; p < (long)fb.derivs.size(); ++p) ls += fb.derivs[p].collect_counts(fb.arcs, 0, 0, tc[me].data()).w;
#pragma omp critical
          sum += ls;
#pragma omp for schedule(static)
          for (long k = 0; k < (long)na; ++k) {
            LW c;
            for (auto& t : tc) c += t[k];
            fb.arcs.t[k].counts = c;
          }
        }
        corpus_p = LW::from_ln(sum);
      }
      fb.maximize(nms, 1.0);
    };
    auto t2 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) run_iter();
    auto t3 = std::chrono::steady_clock::now();
    out[0] = std::chrono::duration<double>(t3 - t2).count() / (iters > 0 ? iters : 1);
    out[1] = arcs;
    out[2] = std::chrono::duration<double>(t1 - t0).count();
    out[3] = corpus_p.w;
  });
}

// CPU baseline with the fixed and the per-lattice-arc cost told apart.  One EM iteration of the reference costs
//   clear counts (O(|WFST arcs|)) + per pair forward/backward/counts (O(lattice arcs)) + maximize (O(|WFST arcs|)),
// so the seconds per iteration on a SAMPLE of the corpus say little about the full corpus unless the two parts are
// separated: the E-step is timed over the first quarter of the cached lattices and over all of them, maximize on its
// own, once single-threaded (the reference is single-threaded) and once with OpenMP over `threads` cores (atomic adds
// into one linear-domain count table, parallel clear / prep_new_weights / per-state normalisation -- weights checked
// against the serial normalize).  out[0] build seconds, [1] lattice arcs of the quarter, [2] of all, then per leg (serial at 3,
// threaded at 7): E-step seconds on the quarter, on all, maximize seconds, ln corpus prob of the last full E-step.
// orc_bench_em_fit_check: the same, and the answers of the serial leg's last full E-step are handed back so that the caller
// can check the GPU against them on the very sample that was timed: counts_ln[n_arcs] (ln expected count per arc, as
// arc_counts::counts holds it) and pair_lp[n surviving pairs, corpus order] (ln p of each pair); either may be null.
int orc_bench_em_fit_check(orc_wfst* wh, orc_corpus* ch, int norm_group, int iters, int threads, double* out, double* counts_ln,
                           double* pair_lp);
int orc_bench_em_fit(orc_wfst* wh, orc_corpus* ch, int norm_group, int iters, int threads, double* out) {
  return orc_bench_em_fit_check(wh, ch, norm_group, iters, threads, out, nullptr, nullptr);
}
int orc_bench_em_fit_check(orc_wfst* wh, orc_corpus* ch, int norm_group, int iters, int threads, double* out, double* counts_ln,
                           double* pair_lp) {
  return run_big_stack([&]() {
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    Wfst& x = wh->w;
    Cascade cascade(false);
    std::vector<NormalizeMethod> nms(1);
    nms[0].group = norm_group;
    cascade.set_composed(&x);
    cascade.normalize(nms);
    TrainOpts opts;
    opts.cache_derivations = true;
    auto t0 = clk::now();
    ForwardBackward fb(x, cascade, false, LW(), opts, ch->c);
    out[0] = secs(t0, clk::now());
    const size_t n_all = fb.derivs.size(), n_q = std::max<size_t>(1, n_all / 4);
    double arcs_q = 0, arcs_all = 0;
    for (size_t p = 0; p < n_all; ++p) {
      arcs_all += (double)fb.derivs[p].n_arcs();
      if (p < n_q) arcs_q += (double)fb.derivs[p].n_arcs();
    }
    out[1] = arcs_q;
    out[2] = arcs_all;
    const size_t na = fb.arcs.t.size();
    if (iters < 1) iters = 1;
    // ---- serial leg: the reference's own loop ----
    bool record_lp = true;  // (the E-steps after the first maximize run on other weights: their ln p are not the answers)
    auto estep_serial = [&](size_t n) {
      for (auto& a : fb.arcs.t) a.counts = LW();
      double sum = 0;
      for (size_t p = 0; p < n; ++p) {
        const double lpp = fb.derivs[p].collect_counts(fb.arcs).w;
        if (pair_lp && record_lp) pair_lp[p] = lpp;
        sum += lpp;
      }
      return sum;
    };
    double lp = 0;
    auto t1 = clk::now();
    for (int i = 0; i < iters; ++i) estep_serial(n_q);
    auto t2 = clk::now();
    for (int i = 0; i < iters; ++i) lp = estep_serial(n_all);
    auto t3 = clk::now();
    if (counts_ln)
      for (size_t k = 0; k < na; ++k) counts_ln[k] = fb.arcs.t[k].counts.w;
    record_lp = false;
    // maximize changes the weights; the E-steps above all ran on the same ones
    for (int i = 0; i < iters; ++i) {
      if (i) estep_serial(n_all);  // fresh counts for a repeat (not timed below)
      auto ta = clk::now();
      fb.maximize(nms, 1.0);
      out[5] += secs(ta, clk::now()) / iters;
    }
    out[3] = secs(t1, t2) / iters;
    out[4] = secs(t2, t3) / iters;
    out[6] = lp;
    if (threads <= 1) return;
#ifdef _OPENMP
    omp_set_num_threads(threads);
    // per-thread copies of a 10^7-entry count table would cost O(threads x |WFST arcs|) per iteration in clearing and
    // reducing alone; the threaded leg adds into ONE table of linear-domain doubles with atomic adds instead
    std::vector<double> lin(na);
    auto estep_par = [&](size_t n) {
      double sum = 0;
#pragma omp parallel
      {
#pragma omp for schedule(static)
        for (long k = 0; k < (long)na; ++k) lin[k] = 0.0;

        double ls = 0;
        std::vector<LW> f, b;
#pragma omp for schedule(dynamic, 64)
        for (long p = 0; p < (long)n; ++p) {
          Derivations& d = fb.derivs[p];
          auto wf = [&](const GArc& a) { return fb.arcs.t[a.arcid].arc->weight; };
          LW prob = d.compute_fb(f, b, wf);
          ls += prob.w;
          for (unsigned st = 0; st < d.g.size(); ++st)
            for (const GArc& a : d.g[st]) {  // derivations.h:439-447 with the sum taken in the linear domain
              const double v = (fb.arcs.t[a.arcid].arc->weight * f[a.src] * b[a.dest] * LW::from_real(d.weight) / prob).getReal();
#pragma omp atomic
              lin[a.arcid] += v;
            }
        }
#pragma omp critical
        sum += ls;
#pragma omp barrier
#pragma omp for schedule(static)
        for (long k = 0; k < (long)na; ++k) fb.arcs.t[k].counts = lin[k] > 0 ? LW::from_real(lin[k]) : LW();  // (log(0) traps: slow)
      }
      return sum;
    };
    // maximize with the per-state work spread over the cores: prep_new_weights (train.cc:134-153), then every state's
    // norm groups (no ties here: fst.cc:196-230 reduces to w = (1 - locked) * w / sum per group), then max_change
    std::vector<size_t> first_arc(x.num_states() + 1, 0);
    for (unsigned st = 0; st < x.num_states(); ++st) first_arc[st + 1] = first_arc[st] + x.states[st].size();
    auto maximize_par = [&]() {
#pragma omp parallel for schedule(static)
      for (long st = 0; st < (long)x.num_states(); ++st) {
        auto& arcs = x.states[st];
        for (size_t k = 0; k < arcs.size(); ++k) {
          ArcRec& a = fb.arcs.t[first_arc[st] + k];
          if (!arcs[k].locked()) {
            a.scratch = arcs[k].weight;
            arcs[k].weight = a.counts + a.prior_counts;
          }
        }
        std::vector<char> done(arcs.size(), 0);
        for (size_t k = 0; k < arcs.size(); ++k) {
          if (done[k]) continue;
          LW normal, locked;
          for (size_t j = k; j < arcs.size(); ++j)
            if (norm_group == NORM_JOINT || arcs[j].in == arcs[k].in) (arcs[j].locked() ? locked : normal) += arcs[j].weight;
          LW remain = LW::one() - locked;
          for (size_t j = k; j < arcs.size(); ++j)
            if (norm_group == NORM_JOINT || arcs[j].in == arcs[k].in) {
              done[j] = 1;
              if (!arcs[j].locked()) arcs[j].weight = (!remain.isZero() && !normal.isZero()) ? remain * arcs[j].weight / normal : LW();
            }
        }
      }
    };
    bool plain = norm_group != NORM_NONE;
    for (auto& st : x.states)
      for (auto& a : st)
        if (a.tied()) plain = false;
    auto t4 = clk::now();
    for (int i = 0; i < iters; ++i) estep_par(n_q);
    auto t5 = clk::now();
    for (int i = 0; i < iters; ++i) lp = estep_par(n_all);
    auto t6 = clk::now();
    out[7] = secs(t4, t5) / iters;
    out[8] = secs(t5, t6) / iters;
    for (int i = 0; i < iters; ++i) {
      if (i) estep_par(n_all);
      auto ta = clk::now();
      if (plain)
        maximize_par();
      else
        fb.maximize(nms, 1.0);
      out[9] += secs(ta, clk::now()) / iters;
    }
    out[10] = lp;
    if (plain) {  // the spread-out maximize must give what the reference's gives: redo the last one serially and compare
      std::vector<double> got;
      for (auto& st : x.states)
        for (auto& a : st) got.push_back(a.weight.w);
      size_t k = 0;
      for (auto& r : fb.arcs.t) r.arc->weight = r.scratch;  // back to the weights before it
      fb.maximize(nms, 1.0);
      for (auto& st : x.states)
        for (auto& a : st) {
          const double d = std::fabs(std::exp(a.weight.w) - std::exp(got[k++]));
          if (d > 1e-12) throw std::runtime_error("bench: parallel maximize disagrees with WFST::normalize");
        }
    }
#endif
  });
}

// Cascade EM over text inputs (compose.cc + cascade.h + train.cc), as `carmel --train-cascade corpus a b ...`.
// normby: one char per transducer (J/C/N).  Trained member texts are returned '\0'-joined in *out_texts
// (caller frees with orc_free_str); trace as in orc_train.
int orc_train_cascade_text(int n_wfst, const char** wfst_texts, const char* corpus_text, const char* normby,
                           const double* priors, int max_iter, double converge_arc_delta, double converge_ppx_ratio,
                           int cache, int full, int onearc, char** out_texts, uint64_t* out_len, double* trace,
                           int max_trace, int* n_trace, uint32_t* composed_dims) {
  return run_big_stack([&]() {
    std::vector<Wfst> chain(n_wfst);
    for (int i = 0; i < n_wfst; ++i) {
      if (!chain[i].read_legible(wfst_texts[i], true)) throw std::runtime_error("bad WFST text");
      if (n_wfst > 1) chain[i].named_states = false;
    }
    Cascade cascade(true);
    Wfst* result = &chain[0];
    result->reduce();
    if (n_wfst < 2) cascade.set_trivial();
    cascade.add(result);
    std::vector<Wfst*> owned;
    bool any = false;
    for (int i = 1; i < n_wfst; ++i) {
      cascade.add(&chain[i]);
      if (i == 1)
        cascade.prepare_compose();
      else
        cascade.prepare_compose(false);
      Wfst* next = new Wfst();
      compose(*next, cascade, *result, chain[i]);
      owned.push_back(next);
      result = next;
      if (!result->valid) throw std::runtime_error("empty composition");
      if (composed_dims) {
        composed_dims[0] = result->num_states();
        composed_dims[1] = (uint32_t)result->num_arcs();
      }
      result->reduce();
      cascade.done_composing(result);
      any = true;
    }
    if (!any) cascade.set_composed(result);
    std::vector<NormalizeMethod> nms(n_wfst);
    for (int i = 0; i < n_wfst; ++i) {
      char c = normby && (int)std::strlen(normby) > i ? normby[i] : 'C';
      nms[i].group = (c == 'J' || c == 'j') ? NORM_JOINT : (c == 'N' || c == 'n') ? NORM_NONE : NORM_CONDITIONAL;
      if (priors) nms[i].add_count = LW::from_real(priors[i]);
    }
    Corpus corpus;
    read_training_corpus(*result, corpus_text, corpus);
    TrainOpts opts;
    opts.max_iter = (unsigned)max_iter;
    opts.cache_derivations = cache != 0;
    std::vector<IterRecord> tr;
    train(*result, cascade, corpus, nms, false, LW(), LW::from_real(converge_arc_delta),
          LW::from_real(converge_ppx_ratio), opts, 0, &tr);
    int n = 0;
    for (auto& r : tr) {
      if (n >= max_trace) break;
      double* row = trace + 8 * n;
      row[0] = r.iter;
      row[1] = r.log2_prob;
      row[2] = r.ppx_symbol_log2;
      row[3] = r.ppx_example_log2;
      row[4] = r.new_best;
      row[5] = r.rel_ppx_ratio_ln;
      row[6] = r.last_change;
      row[7] = r.n_example;
      ++n;
    }
    if (n_trace) *n_trace = n;
    if (out_texts) {
      std::string all;
      for (int i = 0; i < n_wfst; ++i) {
        all += chain[i].write_legible(full != 0, onearc != 0);
        all.push_back('\0');
      }
      *out_texts = (char*)std::malloc(all.size());
      std::memcpy(*out_texts, all.data(), all.size());
      *out_len = all.size();
    }
    for (Wfst* w : owned) delete w;
  });
}

// ---- cascade as flat arrays (for driving the product's carmel_hip_set_cascade from the tests) ----
struct orc_cascade {
  std::vector<Wfst> chain;
  Cascade cascade{true};
  explicit orc_cascade(bool remember = true) : cascade(remember) {}
  Wfst* result = 0;
  std::vector<Wfst*> owned;
  ~orc_cascade() {
    for (Wfst* w : owned) delete w;
  }
};
// composes the transducers left to right exactly as `carmel --train-cascade` does (carmel.cc:1303-1355)
orc_cascade* orc_cascade_compose_text_ex(int n_wfst, const char** wfst_texts, int remember, int dash_a);
orc_cascade* orc_cascade_compose_text(int n_wfst, const char** wfst_texts) {
  return orc_cascade_compose_text_ex(n_wfst, wfst_texts, 1, 0);
}
// remember = 0: plain `carmel a b` (trivial cascade: no chains, cascade.h:566-592); dash_a: carmel -a (compose.cc:219-313)
orc_cascade* orc_cascade_compose_text_ex(int n_wfst, const char** wfst_texts, int remember, int dash_a) {
  orc_cascade* h = new orc_cascade(remember != 0);
  int rc = run_big_stack([&]() {
    h->chain.resize(n_wfst);
    for (int i = 0; i < n_wfst; ++i) {
      if (!h->chain[i].read_legible(wfst_texts[i], true)) throw std::runtime_error("bad WFST text");
      if (n_wfst > 1) h->chain[i].named_states = false;
    }
    h->result = &h->chain[0];
    h->result->reduce();
    if (n_wfst < 2) h->cascade.set_trivial();
    h->cascade.add(h->result);
    bool any = false;
    for (int i = 1; i < n_wfst; ++i) {
      h->cascade.add(&h->chain[i]);
      if (i == 1)
        h->cascade.prepare_compose();
      else
        h->cascade.prepare_compose(false);
      Wfst* next = new Wfst();
      if (dash_a)
        compose_a(*next, h->cascade, *h->result, h->chain[i]);
      else
        compose(*next, h->cascade, *h->result, h->chain[i]);
      h->owned.push_back(next);
      h->result = next;
      if (!h->result->valid) throw std::runtime_error("empty composition");
      h->result->reduce();
      h->cascade.done_composing(h->result);
      any = true;
    }
    if (!any) h->cascade.set_composed(h->result);
  });
  if (rc) {
    delete h;
    return 0;
  }
  return h;
}
void orc_cascade_free(orc_cascade* h) { delete h; }
// the composed transducer as an orc_wfst (a copy; arc group = chain id)
orc_wfst* orc_cascade_composed(orc_cascade* h) {
  orc_wfst* w = new orc_wfst();
  w->w = *h->result;
  return w;
}
// dims: n_members, n_params, n_chains, n_chain_entries
void orc_cascade_dims(orc_cascade* h, uint64_t* dims) {
  dims[0] = h->chain.size();
  uint64_t np = 0;
  for (auto& w : h->chain) np += w.num_arcs();
  dims[1] = np;
  dims[2] = h->cascade.chains.size();
  uint64_t ne = 0;
  for (auto& c : h->cascade.chains) ne += c.size();
  dims[3] = ne;
}
void orc_cascade_member_states(orc_cascade* h, uint32_t* n_states) {
  for (size_t m = 0; m < h->chain.size(); ++m) n_states[m] = h->chain[m].num_states();
}
// parameters = member arcs concatenated in visit order; chains reference them by index
void orc_cascade_export(orc_cascade* h, double* param_logw, uint32_t* param_group, uint32_t* param_member,
                        uint32_t* param_src, uint32_t* param_in, uint64_t* chain_off, uint64_t* chain_param) {
  std::unordered_map<const Arc*, uint64_t> id;
  uint64_t k = 0;
  for (size_t m = 0; m < h->chain.size(); ++m)
    for (uint32_t s = 0; s < h->chain[m].num_states(); ++s)
      for (auto& a : h->chain[m].states[s]) {
        id[&a] = k;
        param_logw[k] = a.weight.w;
        param_group[k] = a.group;
        param_member[k] = (uint32_t)m;
        param_src[k] = s;
        param_in[k] = a.in;
        ++k;
      }
  uint64_t e = 0;
  for (size_t c = 0; c < h->cascade.chains.size(); ++c) {
    chain_off[c] = e;
    for (Arc* p : h->cascade.chains[c]) chain_param[e++] = id.at(p);
  }
  chain_off[h->cascade.chains.size()] = e;
}
// corpus parsed against the composed transducer's alphabets
orc_corpus* orc_cascade_corpus(orc_cascade* h, const char* text) {
  orc_corpus* c = new orc_corpus();
  read_training_corpus(*h->result, text, c->c);
  return c;
}
// write member m with the given parameter weights (n_params values, concatenated order)
char* orc_cascade_write_member(orc_cascade* h, int m, const double* param_logw, int full, int onearc) {
  uint64_t k = 0;
  for (int i = 0; i < m; ++i) k += h->chain[i].num_arcs();
  for (auto& st : h->chain[m].states)
    for (auto& a : st) a.weight.w = param_logw[k++];
  std::string s = h->chain[m].write_legible(full != 0, onearc != 0);
  char* r = (char*)std::malloc(s.size() + 1);
  std::memcpy(r, s.c_str(), s.size() + 1);
  return r;
}

// ---- a uniform stream that needs no callback into Python: the splitmix64 finaliser (Steele, Lea & Flood 2014) over
// (seed, sweep, block, step), 53 bits to [0,1).  The product's generator (carmel_hip_gibbs_uniform) is the same published
// function; tests/test_gibbs_host.py checks the two agree value for value, so a chain driven by this one is the chain the
// GPU sampler is compared with draw for draw.  The seed is set once per run (orc_set_native_uniform_seed).
static uint64_t g_native_seed = 0;
static inline uint64_t orc_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void orc_set_native_uniform_seed(uint64_t seed) { g_native_seed = seed; }
double orc_native_uniform(uint32_t iter, uint32_t block, uint32_t step) {
  uint64_t h = orc_mix64(g_native_seed ^ 0xD1B54A32D192ED03ull);
  h = orc_mix64(h ^ ((uint64_t)iter << 32 | block));
  h = orc_mix64(h ^ (uint64_t)step);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// ---- blocked Gibbs over a composed cascade (carmel --crp), uniforms injected through a callback ----
// out_param_logw[n_params]: ln final_prob per member arc, concatenated member order (probs_to_cascade)
// out_samples / out_sample_off[n_blocks+1]: final sample of every block as MEMBER-ARC indices (concatenated order)
typedef double (*orc_uniform_fn)(uint32_t iter, uint32_t block, uint32_t step);
int orc_gibbs_run(orc_cascade* h, orc_corpus* c, const char* normby, const double* priors, uint32_t iter,
                  uint32_t burnin, int uniform_p0, int dirichlet_p0, int final_counts, int exclude_prior,
                  orc_uniform_fn u, double* iter_logprob, double* iter_cheap_logprob, double* out_param_logw,
                  uint32_t* out_samples, uint64_t* out_sample_off, uint64_t max_samples, uint32_t* n_blocks) {
  return run_big_stack([&]() {
    size_t n = h->chain.size();
    std::vector<NormalizeMethod> nms(n);
    for (size_t i = 0; i < n; ++i) {
      char ch = normby && std::strlen(normby) > i ? normby[i] : 'C';
      nms[i].group = (ch == 'J' || ch == 'j') ? NORM_JOINT : (ch == 'N' || ch == 'n') ? NORM_NONE : NORM_CONDITIONAL;
      if (priors) nms[i].add_count = LW::from_real(priors[i]);
    }
    GibbsOpts go;
    go.iter = iter;
    go.burnin = burnin < iter ? burnin : iter;
    go.uniformp0 = uniform_p0 != 0;
    go.high_temp = g_high_temp;
    go.low_temp = g_low_temp;
    go.expectation = g_expectation != 0;
    go.include_self = g_include_self != 0;
    go.random_start = g_random_start != 0;
    go.restarts = g_crp_restarts;
    go.argmax_final = g_argmax_final != 0;
    go.argmax_sum = g_argmax_sum != 0;
    go.dirichlet_p0 = dirichlet_p0 != 0;
    go.final_counts = final_counts != 0;
    go.exclude_prior = exclude_prior != 0;
    if (go.final_counts) go.burnin = go.iter;  // gibbs_opts.hpp validate()
    go.prior_inference_stddev = g_pi_stddev;
    go.prior_inference_global = g_pi_global != 0;
    go.prior_inference_local = g_pi_local != 0;
    go.prior_inference_restart_fresh = g_pi_fresh != 0;
    go.prior_inference_start = g_pi_start;
    go.prior_inference_end = g_pi_end;
    go.priorgroup = g_pi_groupby;
    std::vector<double> init_logw;
    if (g_init_em > 0) {  // WFST::train_gibbs (gibbs.cc:400-423): EM without priors for the first sample's weights
      std::vector<std::vector<LW> > saved;
      for (auto& w : h->chain) {
        saved.emplace_back();
        for (auto& st : w.states)
          for (auto& a : st) saved.back().push_back(a.weight);
      }
      std::vector<NormalizeMethod> m2 = nms;
      for (auto& m : m2) m.add_count = LW();
      TrainOpts t2;
      t2.max_iter = g_init_em;
      h->cascade.set_composed(h->result);
      train(*h->result, h->cascade, c->c, m2, false, LW(), LW(), LW::one(), t2);
      for (auto& st : h->result->states)
        for (auto& a : st) init_logw.push_back(a.weight.w);
      if (!g_em_p0) {
        size_t i = 0;
        for (auto& w : h->chain) {
          size_t k = 0;
          for (auto& st : w.states)
            for (auto& a : st) a.weight = saved[i][k++];
          ++i;
        }
      }
    }
    if (g_init_from_p0 && g_init_em <= 0) {
      // --init-from-p0 (gibbs.cc:405-421): the first sample is drawn from the composed transducer's own weights, not from
      // the cache.  cascade.normalize(m2) normalises the MEMBERS (add_count 0) and the composed weights are saved without
      // a cascade.update(): for a real cascade they are the products made at composition time; a single transducer is
      // its own cascade, so there they are the normalised weights.  The members' weights are restored afterwards.
      if (h->cascade.trivial) {
        std::vector<NormalizeMethod> m2 = nms;
        for (auto& m : m2) m.add_count = LW();
        Wfst copy = *h->result;
        copy.normalize(m2[0]);
        for (auto& st : copy.states)
          for (auto& a : st) init_logw.push_back(a.weight.w);
      } else
        for (auto& st : h->result->states)
          for (auto& a : st) init_logw.push_back(a.weight.w);
    }
    CarmelGibbs g(*h->result, h->cascade, c->c, nms, go);
    g.init_logw = init_logw;
    GibbsTrace tr;
    tr.want_state = g_want_state != 0;
    g.run([&](unsigned it, unsigned b, unsigned st) { return u(it, b, st); }, &tr);
    g_best_run = (int)g.best_run;
    g_last_after = tr.iter_after_logprob;
    g_last_prior_trace = tr.prior_trace;
    g_last_cumulative = tr.cumulative;
    for (uint32_t i = 0; i < (iter + 1) * (go.restarts + 1); ++i) {
      if (iter_logprob) iter_logprob[i] = tr.iter_logprob[i];
      if (iter_cheap_logprob) iter_cheap_logprob[i] = tr.iter_cheap_logprob[i];
    }
    // oracle param id -> member-arc index
    std::vector<uint32_t> arc_index(g.gps.size(), 0);
    uint32_t k = 0;
    for (auto& w : h->chain)
      for (auto& st : w.states)
        for (auto& a : st) {
          arc_index[g.param_of.at(&a)] = k;
          if (out_param_logw) out_param_logw[k] = a.weight.w;
          ++k;
        }
    if (tr.want_state) {  // to member-arc order, with the ids the tables print
      const size_t np = g.gps.size(), nsw = np ? tr.state.size() / (4 * np) : 0;
      g_last_state.assign(tr.state.size(), 0.0);
      g_last_final.assign(2 * np, 0.0);
      g_last_ids.assign(3 * np, 0);
      for (size_t pid = 0; pid < np; ++pid) {
        const size_t a = arc_index[pid];
        for (size_t q = 0; q < nsw; ++q)
          for (int f = 0; f < 4; ++f) g_last_state[(q * np + a) * 4 + f] = tr.state[(q * np + pid) * 4 + f];
        g_last_final[2 * a] = tr.final_x[pid];
        g_last_final[2 * a + 1] = tr.final_prob[pid];
        g_last_ids[3 * a] = (int32_t)pid;
        g_last_ids[3 * a + 1] = g.gps[pid].has_norm() ? (int32_t)g.gps[pid].norm : -1;
        g_last_ids[3 * a + 2] = g.gps[pid].has_norm() && g.gps[pid].norm < g.metanorm.size() ? (int32_t)g.metanorm[g.gps[pid].norm] : 0;
      }
    }
    if (n_blocks) *n_blocks = (uint32_t)tr.last_sample.size();
    if (out_sample_off) {
      uint64_t o = 0;
      for (size_t b = 0; b < tr.last_sample.size(); ++b) {
        out_sample_off[b] = o;
        for (unsigned pid : tr.last_sample[b]) {
          if (o >= max_samples) throw std::runtime_error("sample buffer too small");
          out_samples[o++] = arc_index[pid];
        }
      }
      out_sample_off[tr.last_sample.size()] = o;
    }
  });
}

// carmel --fem-forest / --fem-norm / --fem-param / --fem-alpha (oracle/fem.hpp).  which: 0 forests, 1 norm groups,
// 2 params, 3 alphas.  Returns the text length (the text is truncated to cap - 1 bytes), -1 on error.
long orc_fem_export(orc_cascade* h, orc_corpus* c, const char* normby, const double* priors, int which, char* buf,
                    unsigned long cap) {
  long len = -1;
  int rc = run_big_stack([&]() {
    size_t n = h->chain.size();
    std::vector<NormalizeMethod> nms(n);
    for (size_t i = 0; i < n; ++i) {
      char ch = normby && std::strlen(normby) > i ? normby[i] : 'C';
      nms[i].group = (ch == 'J' || ch == 'j') ? NORM_JOINT : (ch == 'N' || ch == 'n') ? NORM_NONE : NORM_CONDITIONAL;
      if (priors) nms[i].add_count = LW::from_real(priors[i]);
    }
    h->cascade.set_composed(h->result);
    FemExport fe(h->cascade, *h->result);
    std::string txt = which == 0 ? fe.forests(c->c) : which == 1 ? fe.norms(nms) : which == 2 ? fe.params() : fe.alphas(nms);
    len = (long)txt.size();
    if (buf && cap) {
      size_t k = std::min<size_t>(txt.size(), cap - 1);
      std::memcpy(buf, txt.data(), k);
      buf[k] = 0;
    }
  });
  return rc == 0 ? len : -1;
}

// ---- forest-em ----
struct orc_forests {
  ForestEm fe;
};
orc_forests* orc_forests_parse(const char* forests_text, const char* normgroups_text) {
  orc_forests* h = new orc_forests();
  try {
    h->fe.forests = parse_forests(forests_text);
    h->fe.groups = parse_normgroups(normgroups_text);
    unsigned mx = 0;
    for (auto& f : h->fe.forests) mx = std::max(mx, f.max_rule);
    for (auto& g : h->fe.groups)
      for (unsigned r : g) mx = std::max(mx, r);
    h->fe.init(mx + 1);
  } catch (std::exception& e) {
    g_err = e.what();
    delete h;
    return 0;
  }
  return h;
}
void orc_forests_free(orc_forests* h) { delete h; }
// dims: n_forests, total_nodes, rulespace, n_groups, n_group_entries
void orc_forests_dims(orc_forests* h, uint64_t* dims) {
  dims[0] = h->fe.forests.size();
  uint64_t n = 0;
  for (auto& f : h->fe.forests) n += f.nodes.size();
  dims[1] = n;
  dims[2] = h->fe.w.size();
  dims[3] = h->fe.groups.size();
  uint64_t e = 0;
  for (auto& g : h->fe.groups) e += g.size();
  dims[4] = e;
}
void orc_forests_export(orc_forests* h, uint64_t* node_off, uint32_t* label, int32_t* ref, uint32_t* next,
                        uint64_t* group_off, uint32_t* group_rule) {
  uint64_t k = 0, fi = 0;
  for (auto& f : h->fe.forests) {
    node_off[fi++] = k;
    for (auto& n : f.nodes) {
      label[k] = n.label;
      ref[k] = n.ref;
      next[k] = n.next;
      ++k;
    }
  }
  node_off[fi] = k;
  uint64_t e = 0, gi = 0;
  for (auto& g : h->fe.groups) {
    group_off[gi++] = e;
    for (unsigned r : g) group_rule[e++] = r;
  }
  group_off[gi] = e;
}
void orc_forests_set_weights(orc_forests* h, const double* lw) {
  for (size_t i = 0; i < h->fe.w.size(); ++i) h->fe.w[i] = LW::from_ln(lw[i]);
}
void orc_forests_get_weights(orc_forests* h, double* lw) {
  for (size_t i = 0; i < h->fe.w.size(); ++i) lw[i] = h->fe.w[i].w;
}
double orc_forests_estimate(orc_forests* h, double prior_count, double* counts_ln, double* per_forest) {
  h->fe.prior_count = prior_count;
  std::vector<double> pf;
  double a = h->fe.estimate(&pf);
  if (counts_ln)
    for (size_t i = 0; i < h->fe.counts.size(); ++i) counts_ln[i] = h->fe.counts[i].w;
  if (per_forest)
    for (size_t i = 0; i < pf.size(); ++i) per_forest[i] = pf[i];
  return a;
}
void orc_forests_init_rule_weights(orc_forests* h, int ones) { h->fe.init_rule_weights(ones != 0); }
// fraction[rule]: the random positive fraction drawn for each rule (the caller's generator)
void orc_forests_randomize(orc_forests* h, const double* fraction) {
  h->fe.randomize([&](unsigned r) { return fraction[r]; });
}
// the -v line of forest `forest` into buf (mode: LwPrintMode); returns its length, best_ln = ln of the best derivation
int orc_forests_viterbi_line(orc_forests* h, uint64_t forest, int mode, char* buf, int cap, double* best_ln) {
  const std::string s = h->fe.viterbi_line(h->fe.forests[forest], mode, best_ln);
  if ((int)s.size() + 1 > cap) return -(int)s.size();
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}
double orc_forests_maximize(orc_forests* h, double add_k, int zero_zero) {
  h->fe.add_k = add_k;
  h->fe.zero_zerocounts = zero_zero != 0;
  return h->fe.maximize().getReal();
}
static int g_forest_exclude_prior = 0;  // --crp-exclude-prior for the next orc_forests_gibbs
void orc_forests_set_exclude_prior(int on) { g_forest_exclude_prior = on; }
static std::vector<double> g_forest_alphas;  // --alpha=FILE for the next orc_forests_gibbs (empty: scalar alpha)
void orc_forests_set_alphas(const double* a, uint32_t n) { g_forest_alphas.assign(a, a + (a ? n : 0)); }
// forest-em --print-counts-* / --print-norms-* (gibbs.hpp:970-1078): per rule {finalized count, norm group or -1} of the last run
static std::vector<double> g_forest_final;
uint64_t orc_forests_gibbs_last_final(double* out, uint64_t n) {
  for (uint64_t i = 0; i < n && i < g_forest_final.size(); ++i) out[i] = g_forest_final[i];
  return g_forest_final.size();
}
int orc_forests_gibbs(orc_forests* h, uint32_t iter, uint32_t burnin, int uniform_p0, int final_counts, double alpha,
                      orc_uniform_fn u, double* iter_logprob, double* iter_cheap_logprob, uint32_t* out_samples,
                      uint64_t* out_sample_off, uint64_t max_samples) {
  return run_big_stack([&]() {
    GibbsOpts go;
    go.iter = iter;
    go.burnin = burnin < iter ? burnin : iter;
    go.uniformp0 = uniform_p0 != 0;
    go.high_temp = g_high_temp;
    go.low_temp = g_low_temp;
    go.final_counts = final_counts != 0;
    go.exclude_prior = g_forest_exclude_prior != 0;
    if (go.final_counts) go.burnin = go.iter;
    go.prior_inference_stddev = g_pi_stddev;
    go.prior_inference_global = g_pi_global != 0;
    go.prior_inference_local = g_pi_local != 0;
    go.prior_inference_start = g_pi_start;
    go.prior_inference_end = g_pi_end;
    ForestGibbs g(h->fe, go, alpha, g_forest_alphas);
    GibbsTrace tr;
    g.run([&](unsigned it, unsigned b, unsigned st) { return u(it, b, st); }, &tr);
    g_last_prior_trace = tr.prior_trace;
    g_last_cumulative = tr.cumulative;
    g_forest_final.assign(2 * g.gps.size(), 0.0);
    for (size_t r = 0; r < g.gps.size(); ++r) {
      g_forest_final[2 * r] = g.gps[r].has_norm() ? g.gps[r].sum.x : 0.0;
      g_forest_final[2 * r + 1] = g.gps[r].has_norm() ? (double)g.gps[r].norm : -1.0;
    }
    for (uint32_t i = 0; i <= iter; ++i) {
      if (iter_logprob) iter_logprob[i] = tr.iter_logprob[i];
      if (iter_cheap_logprob) iter_cheap_logprob[i] = tr.iter_cheap_logprob[i];
    }
    if (out_sample_off) {
      uint64_t o = 0;
      for (size_t b = 0; b < tr.last_sample.size(); ++b) {
        out_sample_off[b] = o;
        for (unsigned r : tr.last_sample[b]) {
          if (o >= max_samples) throw std::runtime_error("sample buffer too small");
          out_samples[o++] = r;
        }
      }
      out_sample_off[tr.last_sample.size()] = o;
    }
  });
}

}  // extern "C"
