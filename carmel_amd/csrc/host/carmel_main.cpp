// carmel_main.cpp — `carmel`-compatible command line for the training path, running on the GPU through the C-ABI
// (include/carmel_hip.h).  Accepts the training subset of carmel's switches (carmel.cc:929-1066):
//
//   carmel [-t] [--train-cascade] [-M n] [-e d] [-X r] [-f w] [-U] [-u | -j] [-? | -:] [-q] [-d] [-K] [-m] [-T n] [-a]
//          [-F out] [-H] [-J] [-Z] [-D] [-B] [-2] [-+ alpha] [--normby=JCN..] [--priors=a,b,..] [--digamma=a,,b] [--gpu=n]
//          corpus transducer [transducer ...]
//   carmel -S pairs transducer [transducer ...]      sum-of-paths probability of every pair (carmel.cc:1393-1410)
//
// A switch or option this front end does not implement is REFUSED (exit -12, like carmel's "No inputs supplied",
// carmel.cc:1137) rather than accepted and ignored: a drop-in that silently computes something else is worse than one
// that says no.
//
// With one transducer the trained transducer goes to stdout (or -F file); with --train-cascade every member is
// written to <file>.trained (cascade.h:23-32).  EM log lines on stderr have the reference's wording
// (train.cc:587-613, 639-657, 669-671).  Exit codes follow carmel.cc (-2 bad transducer, -9 unreadable file,
// -11 on an error caught at top level).
//
// The iteration control below restates WFST::train (train.cc:503-678); every E-step and M-step runs on the GPU.
#include <algorithm>
#include <chrono>
#include <cctype>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <functional>
#include <limits>
#include <memory>
#include <sstream>
#include <dlfcn.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include "../../../include/carmel_hip.h"
#include "compose.hpp"
#include "fem_export.hpp"
#include "refhash.hpp"
#include "env_options.hpp"
#include "wfst.hpp"

using namespace carmel_host;

static std::string slurp(const char* fn) {
  std::ifstream f(fn, std::ios::binary);
  if (!f) throw std::runtime_error(std::string("File ") + fn + " could not be opened for input.");
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
static void hip_check(int rc, const char* what) {
  if (rc != CARMEL_HIP_OK) throw std::runtime_error(std::string(what) + ": " + carmel_hip_last_error());
}
static std::string base2(double ln_value) {  // weight.h:529-532,603 as_base(2) at the stream's default precision
  char buf[64];
  std::snprintf(buf, sizeof buf, "2^%.6g", ln_value / std::log(2.0));
  return buf;
}

struct UsageError : std::runtime_error {
  explicit UsageError(const std::string& m) : std::runtime_error(m) {}
};

struct Options {
  bool flags[256] = {false};
  bool train_cascade = false;
  long restarts = 0;         // -! (train_opts::ran_restarts)
  // random_restart_acceptor (fst.h:999-1044; carmel.cc:1426-1430, 1741-1749); 0 = unset
  double restart_tolerance = 0, final_restart_tolerance = 0;
  long final_restart = 0;
  double rate_growth = 1.0;  // -o (train_opts::learning_rate_growth_factor, fst.h:1083)
  long max_iter = 500;  // train_opts default (fst.h:1080-1095); -1 == "-M" without a number
  double converge = 1e-4, converge_ppx_ratio = .999, smooth_floor = 0;
  int norm = CARMEL_HIP_NORM_CONDITIONAL;
  std::string normby, priors, out_file, digamma;
  bool have_digamma = false;       // --digamma=... (carmel.cc:495)
  bool plus_alpha_set = false;     // -+ a (carmel.cc:1009-1013): mean-field scale of the single transducer's method
  double plus_alpha = 0;
  int index_threshold = 32, gpu = 0;
  int gpus = 1;  // --gpus=N: corpus-sharded EM, one process per GPU (not a carmel option: carmel is single-process)
  std::string comm_plugin;  // --comm-plugin=LIB.so: a transport of the caller's own instead of RCCL (carmel_hip_comm_create_custom);
                            // every rank then runs on the device --gpu names (the transport decides where the data travels)
  bool random_set = false;  // --random-set (carmel.cc:609-612, 786-789): a new weight on (0..1] for every unlocked arc before training
  int exchange_form = 0;    // --exchange=auto|allreduce|collectives|direct (carmel_hip_exchange_plan's form)
  int exchange_chunks = 0;  // --exchange-chunks=K: arc-range chunks of the sharded count exchange (0: the library's default)
  // --crp (carmel.cc:255-304)
  bool expectation = false;  // --expectation (gibbs_opts.hpp:125)
  long crp_restarts = 0;     // --crp-restarts (carmel.cc:271-273)
  long init_em = 0;          // --init-em=N, --em-p0 (carmel.cc:276-277; gibbs.cc:400-423)
  bool em_p0 = false;
  bool init_from_p0 = false;   // --init-from-p0 (carmel.cc:298; gibbs.cc:405-421)
  bool cache_no_prune = false;     // --cache-no-prune
  bool stream_lattices = false;    // --disk-cache-derivations (carmel.cc:243-246): do not keep every pair's lattice resident
  uint64_t resident_bytes = 0;     // --disk-cache-bufsize=SIZE[K|M|G]: how much lattice memory may be resident at a time (0: 64 GB)
  bool matrix_fb = false;          // --matrix-fb (carmel.cc:238)
  bool gpu_compose = false;        // --gpu-compose: the product construction of the composition on the GPU (compose.hip)
  // prior-scale inference (carmel.cc:291-294, 497; gibbs.hpp:525-563)
  double pi_stddev = 0;
  bool pi_global = false, pi_restart_fresh = false, pi_show = false;
  std::string prior_groupby;
  long number_from = 0;            // --number-from=N (carmel.cc:768, 802-806)
  std::string write_loaded;        // --write-loaded=suffix (carmel.cc:758, 807)
  bool have_write_loaded = false;
  bool sample_prob_after = false;  // --sample-prob-after: log the add-back proposal probability (carmel_hip_gibbs_run_ex)
  bool crp_argmax_final = false, crp_argmax_sum = false;
  bool include_self = false, random_start = false;  // gibbs_opts.hpp:40-41, 127-128
  long print_every = 0;                              // gibbs_opts.hpp:78-79
  // the sampler's tables (gibbs_opts.hpp:64-77, 142-146, 197-203; gibbs.hpp:970-1078): parameter ids [from, to) of the count
  // table, norm-group ids [from, to) of the norm sums; 4294967295 = to the end
  unsigned long print_counts_from = 0, print_counts_to = 0, print_norms_from = 0, print_norms_to = 0;
  double print_counts_sparse = 0;
  bool rich_counts = false, norm_order = false;
  long width = 7;
  std::string fem_forest, fem_norm, fem_param, fem_alpha;  // forest-em export (carmel.cc:756-769, 818-831)
  long print_from = 0, print_to = 0;  // --print-from=m --print-to=n (gibbs_opts.hpp; gibbs.cc:258-296): the final sample's
                                      // path through input transducers m .. n-1, one line each, on stdout
  std::string fem_early_param;                             // --fem-early-param: the weights as loaded / normalised (carmel.cc:801)
  std::string load_fem_param;                              // --load-fem-param (carmel.cc:790-799; cascade.h:180-202)
  bool crp = false, crp_parallel = false, uniform_p0 = false, dirichlet_p0 = false, final_counts = false,
       exclude_prior = false;
  long crp_iters = -1, burnin = 0;
  double high_temp = 1, low_temp = 1;  // --high-temp / --low-temp (carmel.cc:289-290)
  unsigned long long seed = 1;
  std::vector<const char*> files;
};

static Options parse_args(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("--", 0) == 0) {
      std::string k = a.substr(2), v;
      size_t e = k.find('=');
      if (e != std::string::npos) {
        v = k.substr(e + 1);
        k = k.substr(0, e);
      }
      if (k == "train-cascade")
        o.train_cascade = true;
      else if (k == "normby")
        o.normby = v;
      else if (k == "priors")
        o.priors = v;
      else if (k == "gpu")
        o.gpu = std::atoi(v.c_str());
      else if (k == "gpus")
        o.gpus = std::max(1, std::atoi(v.c_str()));
      else if (k == "comm-plugin")
        o.comm_plugin = v;
      else if (k == "random-set")
        o.random_set = true;
      else if (k == "exchange") {
        const char* names[] = {"auto", "allreduce", "collectives", "direct"};
        o.exchange_form = -1;
        for (int f = 0; f < 4; ++f)
          if (v == names[f]) o.exchange_form = f;
        if (o.exchange_form < 0) throw UsageError("--exchange is auto, allreduce, collectives or direct");
      } else if (k == "exchange-chunks")
        o.exchange_chunks = std::max(0, std::atoi(v.c_str()));
      else if (k == "crp") {
        o.crp = true;
        if (!v.empty() && std::atol(v.c_str()) > 1) o.crp_iters = std::atol(v.c_str());
      } else if (k == "burnin")
        o.burnin = std::atol(v.c_str());
      else if (k == "high-temp")
        o.high_temp = std::atof(v.c_str());
      else if (k == "low-temp")
        o.low_temp = std::atof(v.c_str());
      else if (k == "uniform-p0")
        o.uniform_p0 = true;
      else if (k == "dirichlet-p0")
        o.dirichlet_p0 = true;
      else if (k == "fem-forest") {
        o.fem_forest = v;
        o.train_cascade = true;  // force_cascade_derivs (carmel.cc:230-233, 764-767)
        o.flags[(unsigned)'t'] = true;
      } else if (k == "fem-norm")
        o.fem_norm = v;
      else if (k == "fem-param")
        o.fem_param = v;
      else if (k == "fem-alpha")
        o.fem_alpha = v;
      else if (k == "fem-early-param")
        o.fem_early_param = v;
      else if (k == "print-from")
        o.print_from = std::atol(v.c_str());
      else if (k == "print-to")
        o.print_to = std::atol(v.c_str());
      else if (k == "print-every")  // gibbs_opts.hpp:78-79, gibbs.hpp:959-968
        o.print_every = std::atol(v.c_str());
      else if (k == "print-counts-from")
        o.print_counts_from = std::strtoul(v.c_str(), 0, 10);
      else if (k == "print-counts-to")
        o.print_counts_to = std::strtoul(v.c_str(), 0, 10);
      else if (k == "print-norms-from")
        o.print_norms_from = std::strtoul(v.c_str(), 0, 10);
      else if (k == "print-norms-to")
        o.print_norms_to = std::strtoul(v.c_str(), 0, 10);
      else if (k == "print-counts-sparse")
        o.print_counts_sparse = std::atof(v.c_str());
      else if (k == "print-counts-rich")
        o.rich_counts = true;
      else if (k == "norm-order")
        o.norm_order = true;
      else if (k == "width") {
        o.width = std::atol(v.c_str());
        if (o.width < 4) o.width = 20;  // gibbs_opts.hpp:255
      }
      else if (k == "sample-prob" || k == "no-prob" || k == "cache-prob" || k == "cheap-prob" || k == "progress-every") {
        // (--progress-every: the dots gibbs.hpp:845-848 writes into the log while a sweep runs; a sweep is one launch here)
        // inert in carmel itself: gibbs_opts::cache_prob is true and never cleared (carmel.cc:296-298, gibbs_opts.hpp:240,
        // 255-258), so the cache-model probability is what is logged whatever these say
      }
      else if (k == "load-fem-param")
        o.load_fem_param = v;
      else if (k == "restart-tolerance")
        o.restart_tolerance = std::atof(v.c_str());
      else if (k == "final-restart-tolerance")
        o.final_restart_tolerance = std::atof(v.c_str());
      else if (k == "final-restart")
        o.final_restart = std::atol(v.c_str());
      else if (k == "final-counts")
        o.final_counts = true;
      else if (k == "expectation")
        o.expectation = true;
      else if (k == "init-em")
        o.init_em = std::atol(v.c_str());
      else if (k == "em-p0")
        o.em_p0 = true;
      else if (k == "init-from-p0")
        o.init_from_p0 = true;
      else if (k == "gpu-compose")
        o.gpu_compose = true;
      else if (k == "disk-cache-derivations") {
        // carmel.cc:243-246, fst.h:1057-1076: the reference spills its derivation cache to disk when it outgrows memory (and
        // without -? rebuilds every pair's derivations in every iteration, cached_derivs.h:60-101).  Here: when the lattices of
        // the corpus would take more than --disk-cache-bufsize of GPU memory they are NOT kept resident -- every iteration walks
        // the corpus in shards, each shard's lattices rebuilt on the GPU (0.15 s per million pairs), swept and dropped, the
        // shards' counts added up on the device (carmel_hip_accumulate_counts).  No file is created; same results.
        o.stream_lattices = true;
      } else if (k == "disk-cache-bufsize") {
        char* end = nullptr;
        double x = std::strtod(v.c_str(), &end);
        if (end && (*end == 'K' || *end == 'k')) x *= 1024.0;
        else if (end && (*end == 'M' || *end == 'm')) x *= 1024.0 * 1024.0;
        else if (end && (*end == 'G' || *end == 'g')) x *= 1024.0 * 1024.0 * 1024.0;
        if (!(x > 0)) throw UsageError("--disk-cache-bufsize needs a positive size (bytes; K, M, G suffixes)");
        o.resident_bytes = (uint64_t)x;
      } else if (k == "matrix-fb") {
        // carmel.cc:238, train.cc:254-266, 698-860: forward/backward over the dense (input position x output position x
        // state) matrix instead of derivation lattices (carmel_hip_set_matrix_fb, csrc/matrix_fb.hip)
        o.matrix_fb = true;
      } else if (k == "cache-no-prune")  // carmel.cc:241: keep states that cannot reach the goal in the cached lattices
        o.cache_no_prune = true;
      else if (k == "sample-prob-after")  // not a carmel option (its old builds logged this as "sample prob")
        o.sample_prob_after = true;
      else if (k == "crp-restarts")
        o.crp_restarts = std::atol(v.c_str());
      else if (k == "crp-argmax-final")
        o.crp_argmax_final = true;
      else if (k == "crp-argmax-sum")
        o.crp_argmax_sum = true;
      else if (k == "include-self")
        o.include_self = true;
      else if (k == "random-start")
        o.random_start = true;
      else if (k == "crp-exclude-prior")
        o.exclude_prior = true;
      else if (k == "crp-parallel")  // not a carmel option: the stale-count parallel sweep (gibbs.hip mode 1)
        o.crp_parallel = true;
      else if (k == "prior-inference-stddev")
        o.pi_stddev = std::atof(v.c_str());
      else if (k == "prior-inference-global")
        o.pi_global = true;
      else if (k == "prior-inference-restart-fresh")
        o.pi_restart_fresh = true;
      else if (k == "prior-inference-show")
        o.pi_show = true;
      else if (k == "prior-groupby")
        o.prior_groupby = v;
      else if (k == "number-from")
        o.number_from = std::atol(v.c_str());
      else if (k == "write-loaded") {
        o.write_loaded = v;
        o.have_write_loaded = true;
      }
      else if (k == "prior-inference-start" || k == "prior-inference-end" || k == "prior-inference-local")
        // gibbs_opts.hpp:85-89 documents them and forest-em reads them; carmel.cc:291-294 never does, so carmel runs as
        // if they were not given.  Same here (the library has them: carmel_hip_gibbs_set_prior_inference).
        std::cerr << "--" << k << " is not read by carmel (carmel.cc:291-294); ignored\n";
      else if (k == "digamma") {
        o.digamma = v;
        o.have_digamma = true;
      } else if (k == "help") {
        o.flags[(unsigned)'h'] = true;
      } else
        throw UsageError("option --" + k + " is not implemented by the GPU training front end");
      continue;
    }
    if (a.size() > 1 && a[0] == '-') {
      for (size_t j = 1; j < a.size(); ++j) {
        unsigned char c = (unsigned char)a[j];
        o.flags[c] = true;
        if (c == 'j') o.norm = CARMEL_HIP_NORM_JOINT;
        if (c == 'u') o.norm = CARMEL_HIP_NORM_NONE;
        if (c == 'M') o.max_iter = -1;
      }
      // a switch that takes a value consumes the next argument (carmel.cc:929-1000)
      auto value = [&]() -> const char* {
        if (i + 1 >= argc) throw std::runtime_error("missing value after " + a);
        return argv[++i];
      };
      for (size_t j = 1; j < a.size(); ++j) switch (a[j]) {
          case 'M':  // "-M n"; a bare -M means "report the corpus perplexity only" (train.cc:516-517)
            if (i + 1 < argc && (std::isdigit((unsigned char)argv[i + 1][0]) || argv[i + 1][0] == '-') &&
                std::strspn(argv[i + 1], "-0123456789") == std::strlen(argv[i + 1]))
              o.max_iter = std::atol(value());
            break;
          case 'e': o.converge = std::atof(value()); break;
          case 'X': o.converge_ppx_ratio = std::atof(value()); break;
          case 'f': o.smooth_floor = std::atof(value()); break;
          case 'T': o.index_threshold = std::atoi(value()); break;
          case 'F': o.out_file = value(); break;
          case 'R': o.seed = std::strtoull(value(), 0, 10); break;
          case '!':  // random restarts (carmel.cc:944-946)
            o.restarts = std::atol(value());
            break;
          case 'o':  // learning rate growth factor of over-relaxed EM (carmel.cc:940-943)
            o.rate_growth = std::max(1.0, std::atof(value()));
            break;
          case '+':  // pseudo-Dirichlet-process normalisation exp(digamma(alpha + w)) (carmel.cc:1009-1013)
            o.plus_alpha = std::atof(value());
            o.plus_alpha_set = true;
            break;
          default:
            // switches without a value that this front end implements; everything else carmel knows (k-best, generation,
            // projection, pruning, OpenFst, ...) is outside the training path
            // (O I Q W E @: WFST::path_print, fst.h:60-160 -- how --print-to writes the sampled paths)
            if (!std::strchr("tUujnlqdKmHJZDB2?:caShOIQWE@1", a[j]))
              throw UsageError(std::string("switch -") + a[j] + " is not implemented by the GPU training front end");
            break;
        }
      continue;
    }
    o.files.push_back(argv[i]);
  }
  if (o.crp) {  // force_cascade_derivs (carmel.cc:230-233)
    o.train_cascade = true;
    if (o.crp_iters > 1) o.max_iter = o.crp_iters;
  }
  if (o.train_cascade) o.flags[(unsigned)'t'] = true;
  return o;
}

// Weight::ppxper (weight.h:311, 435-440): the n-th root of 1/p -- except that the root of a ZERO weight is ZERO
// (WEIGHT_CORRECT_ZERO), so a corpus of probability 0 reports perplexity 2^-inf and counts as "best"; kept, because
// the reference's iteration control then behaves the same way
static inline double ppxper(double ln_p, double n) {
  return ln_p == -std::numeric_limits<double>::infinity() ? ln_p : -ln_p / n;
}

struct CorpusStats {  // training_corpus counters over the pairs that have a derivation (train.h:151-168)
  double n_pairs = 0, total_weight = 0, n_input = 0, n_output = 0;
};

static std::vector<pid_t> g_kids;  // --gpus: the other ranks (rank 0 only)
static int g_rank = 0;
static int g_err_fd = -1;  // ranks > 0: the job's stderr, for the one message that says why the rank failed
static std::string g_session;  // --comm-plugin: the name the ranks' transports meet under
static volatile sig_atomic_t g_waiting = 0;  // rank 0 is in its final waitpid loop (children may exit normally)
// A rank that dies leaves the others waiting in a collective for ever (RCCL blocks; round-2 advisor finding): rank 0
// watches its children and ends the whole job the moment one of them exits abnormally.
static void on_sigchld(int) {
  if (g_waiting) return;
  int st = 0;
  for (pid_t p : g_kids) {
    const pid_t r = waitpid(p, &st, WNOHANG);
    if (r == p && (!WIFEXITED(st) || WEXITSTATUS(st) != 0)) {
      static const char msg[] = "ERROR: a rank of --gpus ended abnormally; ending the job\n";
      if (write(2, msg, sizeof msg - 1) < 0) {}
      for (pid_t q : g_kids)
        if (q != p) kill(q, SIGTERM);
      _exit(245);  // -11
    }
  }
}

static int run(int argc, char** argv) {
  Options o = parse_args(argc, argv);
  const bool training = o.flags[(unsigned)'t'];
  const bool scoring = !training && o.flags[(unsigned)'S'];  // carmel.cc:1134: -t overrides -S
  const bool with_pairs = training || scoring;
  if (o.flags[(unsigned)'h']) {
    std::cout << "carmel (MI355X training front end): -t / --train-cascade / --crp / -S over carmel's transducer and corpus "
                 "files; switches: -t -M -e -X -f -U -u -j -n -o -! -1 -a -S -q -d -K -m -T -F -R -H -J -Z -D -B -2 -+ -? -: -c; "
                 "options: --train-cascade --normby= --priors= --digamma= --random-set --disk-cache-derivations= --matrix-fb; "
                 "the sampler: --crp[=N] --burnin= --crp-restarts= --print-every= --print-from= --print-to= --print-counts-from= "
                 "--print-counts-to= --print-norms-from= --print-norms-to= --width= ... ; several GPUs: --gpus=N --exchange=; "
                 "the full list and what each replaces: INTEGRATION.md\n";
    return 0;
  }
  if (o.files.empty() || (with_pairs && o.files.size() < 2)) {
    std::cerr << "usage: carmel -t [--train-cascade] [-M n] [-e d] [-X r] [-f w] [-U] [-u|-j] [-HJZD] [-F out] "
                 "corpus transducer [transducer ...]\n"
                 "       carmel [-HJZD] transducer [transducer ...]     (compose and print; host only)\n";
    return -12;
  }
  // ---- --gpus=N: N processes, one per GPU, forked before anything touches a GPU.  Every rank reads the same files and
  // composes the same cascade; rank r keeps the r-th contiguous block of the training pairs, builds its lattices, and the
  // expected counts are summed across ranks once per iteration (carmel_hip_allreduce_counts: RCCL over xGMI, on the
  // trainer's stream between the count pass and the M-step).  The M-step is replicated, so every rank holds the same
  // weights and takes the same decisions; rank 0 alone logs and writes the results. ----
  // (--crp: the runs of --crp-restarts are independent chains; with --gpus=N rank r takes the runs r, r + N, ... on the whole
  // corpus and the ranks agree on the run to keep -- gibbs_base::run_starts with its runs side by side)
  int rank = 0, world = training ? ((o.crp && o.crp_restarts <= 0) ? 1 : o.gpus) : 1;
  std::vector<int> id_pipes;  // rank 0: write ends towards the other ranks
  int id_read = -1;
  std::vector<pid_t>& kids = g_kids;
  if (world > 1 && (!o.fem_forest.empty() || !o.fem_norm.empty() || !o.fem_param.empty() || !o.fem_alpha.empty()))
    throw UsageError("--gpus with the --fem-* exports is not supported (the export walks the whole corpus)");
  if (world > 1) {
    {
      char buf[96];
      std::snprintf(buf, sizeof buf, "carmel_%d_%ld", (int)getpid(), (long)time(nullptr));
      g_session = buf;
    }
    for (int r = 1; r < world; ++r) {
      int fd[2];
      if (pipe(fd) != 0) throw std::runtime_error("pipe() failed");
      pid_t pid = fork();
      if (pid < 0) throw std::runtime_error("fork() failed");
      if (pid == 0) {
        rank = r;
        close(fd[1]);
        id_read = fd[0];
        for (int w : id_pipes) close(w);
        id_pipes.clear();
        kids.clear();
        // a rank other than 0 says nothing unless something goes wrong (its log lines are rank 0's): its streams are dropped, but
        // a copy of stderr is kept for the reason it died, should it die; and no rank outlives rank 0
        g_err_fd = dup(2);
        if (!std::freopen("/dev/null", "w", stdout) || !std::freopen("/dev/null", "w", stderr)) return -11;
        prctl(PR_SET_PDEATHSIG, SIGTERM);
        g_rank = r;
        break;
      }
      close(fd[0]);
      id_pipes.push_back(fd[1]);
      kids.push_back(pid);
    }
    if (rank == 0) {
      struct sigaction sa;
      std::memset(&sa, 0, sizeof sa);
      sa.sa_handler = on_sigchld;
      sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
      sigaction(SIGCHLD, &sa, nullptr);
    }
  } else if (o.gpus > 1)
    std::cerr << "--gpus=" << o.gpus << " applies to EM training (-t / --train-cascade) and to the runs of --crp --crp-restarts=R; running on one GPU\n";
  const bool quiet = o.flags[(unsigned)'q'] || rank > 0;
  if (!with_pairs) o.files.insert(o.files.begin(), (const char*)0);  // no corpus argument
  const size_t nw = o.files.size() - 1;
  std::string corpus_text = with_pairs ? slurp(o.files[0]) : std::string();
  // weight output (carmel.cc:76-101): -Z always / -D never in log form; a weight in log form is e^x, `x ln` (-2) or
  // `x log` base 10 (-B)
  int wstyle = o.flags[(unsigned)'Z'] ? W_ALWAYS_LOG : W_SOMETIMES_LOG;
  if (o.flags[(unsigned)'D']) wstyle = W_NEVER_LOG;
  if (o.flags[(unsigned)'B'])
    wstyle |= W_BASE_LOG10;
  else if (o.flags[(unsigned)'2'])
    wstyle |= W_BASE_LN;
  std::vector<Transducer> member(nw);
  for (size_t i = 0; i < nw; ++i) {
    try {
      member[i].parse(slurp(o.files[i + 1]), !o.flags[(unsigned)'K']);  // carmel.cc:1197
    } catch (std::exception& e) {
      std::cerr << e.what() << "\nBad format of transducer file: " << o.files[i + 1] << "\n";
      return -2;
    }
    if (!o.flags[(unsigned)'m'] && nw > 1) member[i].drop_state_names();
  }
  if (!o.load_fem_param.empty()) {  // fem_in (carmel.cc:790-799): the members' weights, one after the other, from a file
    std::cerr << "Reading cascade weights from --load-fem-param=" << o.load_fem_param << std::endl;
    std::ifstream in(o.load_fem_param.c_str());
    if (!in) throw std::runtime_error("Missing --load-fem-param file.\n");
    for (size_t i = 0; i < nw; ++i) {
      std::vector<double> w;
      for (auto& st : member[i].states)
        for (size_t k = 0; k < st.size(); ++k) {
          std::string tok;
          double lw;
          if (!(in >> tok) || !parse_weight_token(tok, lw))
            throw std::runtime_error("--load-fem-param file doesn't have enough params; make sure it was --fem-param saved for "
                                     "the same cascade");
          w.push_back(lw);
        }
      member[i].set_weights(w.data());
    }
  }
  // ---- normalisation methods per member (carmel.cc:488-499) ----
  std::vector<int> norms(nw, o.norm);
  std::vector<double> addc(nw, o.pi_stddev != 0 ? 1.0 : 0.0);  // carmel.cc:491-492: inferred priors start from 1
  std::vector<int> priorgroup(nw, 1);
  for (size_t i = 0; i < o.prior_groupby.size() && i < nw; ++i) {  // fst.h:586-598
    const char ch = o.prior_groupby[i];
    if (ch < '0' || ch > '2')
      throw std::runtime_error("prior-groupby characters must be 0 (no scaling), 1 (same scaling for whole xdcr), or 2 "
                               "(separate scaling for each normgroup)");
    priorgroup[i] = ch - '0';
  }
  for (size_t i = 0; i < o.normby.size() && i < nw; ++i) {
    char ch = o.normby[i];
    norms[i] = (ch == 'J' || ch == 'j') ? CARMEL_HIP_NORM_JOINT
               : (ch == 'N' || ch == 'n') ? CARMEL_HIP_NORM_NONE
                                          : CARMEL_HIP_NORM_CONDITIONAL;
  }
  {
    std::stringstream ss(o.priors);
    std::string tok;
    size_t i = 0;
    while (std::getline(ss, tok, ',') && i < nw) addc[i++] = std::atof(tok.c_str());
  }
  // --digamma=0,,0.5: one component per member, empty = the usual linear normalisation (carmel.cc:495); -+ a sets it for
  // the single method (carmel.cc:1009-1013)
  std::vector<double> dig_alpha(nw, 0.0);
  std::vector<uint8_t> dig_on(nw, 0);
  if (o.plus_alpha_set)
    for (size_t i = 0; i < nw; ++i) {
      dig_alpha[i] = o.plus_alpha;
      dig_on[i] = 1;
    }
  if (o.have_digamma) {
    size_t i = 0, p0 = 0;
    const std::string& d = o.digamma;
    while (i < nw) {  // split on ',' keeping empty fields
      size_t c = d.find(',', p0);
      std::string tok = d.substr(p0, c == std::string::npos ? std::string::npos : c - p0);
      if (!tok.empty()) {
        dig_alpha[i] = std::atof(tok.c_str());
        dig_on[i] = 1;
      }
      ++i;
      if (c == std::string::npos) break;
      p0 = c + 1;
    }
  }
  const bool any_digamma = std::find(dig_on.begin(), dig_on.end(), (uint8_t)1) != dig_on.end();
  // fem_in (carmel.cc:785-808).  --random-set (:786-789, cascade.h:398-401; -1 below is WFST::randomScale, fst.h:973-975, on the
  // same draws): every unlocked arc of every member not normalised by NONE gets a new weight on (0..1] -- drawn from this
  // build's counter-based generator, numbered member by member in arc order as the random restarts number them (the
  // reference's Boost stream is not pinned by anything it holds); training starts by normalising (train.cc:509).
  if (o.random_set || o.flags[(unsigned)'1']) {
    std::cerr << "Using random seed -R " << o.seed << std::endl;  // show_seed, carmel.cc:65-69
    uint32_t p = 0;
    for (size_t i = 0; i < nw; ++i)
      for (auto& st : member[i].states)
        for (auto& a : st) {
          if (a.group != kLocked && norms[i] != CARMEL_HIP_NORM_NONE) {
            const double lu = std::log(1.0 - carmel_hip_gibbs_uniform(o.seed, 0, p, 0));
            a.logw = o.random_set ? lu : a.logw + lu;
          }
          ++p;
        }
  }
  // with --normby the INPUT transducers are normalised before anything is composed
  if (!o.normby.empty()) {
    std::cerr << "Normalizing input transducers by --normby=" << o.normby << std::endl;
    for (size_t i = 0; i < nw; ++i) member[i].normalize(norms[i], addc[i], dig_on[i] != 0, dig_alpha[i]);
  }
  if (!o.fem_early_param.empty()) {  // fem_out_param(fem_early_outparam), carmel.cc:801, 810-817
    std::cerr << "Writing cascade weights to --fem-param=" << o.fem_early_param << std::endl;
    std::ofstream of(o.fem_early_param.c_str());
    for (size_t i = 0; i < nw; ++i)
      for (auto& st : member[i].states)
        for (auto& a : st) of << format_weight(a.logw, W_SOMETIMES_LOG) << "\n";
  }
  if (o.number_from > 0) {
    std::cerr << "Assigning unique group ids to each arc in input cascade starting at " << o.number_from << ".\n";
    uint32_t label = (uint32_t)o.number_from;
    for (size_t i = 0; i < nw; ++i) label = member[i].number_arcs_from(label);
  }
  if (o.have_write_loaded) {  // cascade.h:23-32
    for (size_t i = 0; i < nw; ++i) {
      std::string fn = o.write_loaded.empty() ? std::string(o.files[i + 1]) : std::string(o.files[i + 1]) + "." + o.write_loaded;
      if (const char* dir = std::getenv("CARMEL_TRAINED_DIR")) {
        std::string bname = o.files[i + 1];
        size_t sl = bname.rfind('/');
        if (sl != std::string::npos) bname = bname.substr(sl + 1);
        fn = std::string(dir) + "/" + bname + (o.write_loaded.empty() ? "" : "." + o.write_loaded);
      }
      std::cerr << "Writing " << o.write_loaded << ' ' << o.files[i + 1] << " to " << fn << std::endl;
      std::ofstream of(fn.c_str());
      of << member[i].to_text(o.flags[(unsigned)'J'], o.flags[(unsigned)'H'], wstyle);
    }
  }
  // ---- composition chain, left to right (carmel.cc:1287-1355) ----
  if (!o.flags[(unsigned)'d']) member[0].prune_useless();
  ParamTable params;
  ChainTable chains;
  std::unique_ptr<Transducer> composed;
  Transducer* result = &member[0];
  const bool cascade = o.train_cascade && nw > 1;
  if (nw > 1) {
    for (size_t i = 0; i < nw; ++i) params.add_member(member[i]);
    Composer comp(params, chains, (unsigned)o.index_threshold, /*trivial=*/!o.train_cascade);
    Operand A, B;
    for (size_t i = 1; i < nw; ++i) {
      A.bind(result, i > 1, params.member_base[0]);
      B.bind(&member[i], false, params.member_base[i]);
      std::unique_ptr<Transducer> next(new Transducer());
      double dev_s = 0;
      const bool ok = o.flags[(unsigned)'a'] ? comp.run_a(A, B, *next)  // carmel.cc:1318
                      : o.gpu_compose        ? comp.run_device(A, B, *next, o.gpu + (o.comm_plugin.empty() ? rank : 0), &dev_s)
                                             : comp.run(A, B, *next);
      if (o.gpu_compose && !o.flags[(unsigned)'a'] && std::getenv("CARMEL_TIMING"))
        std::cerr << "timing: composition on the GPU " << dev_s << " s\n";
      if (!ok) {
        std::cerr << ")\nEmpty or invalid result of composition with transducer \"" << o.files[i + 1] << "\".\n";
        return -3;
      }
      size_t st = next->states.size(), ar = next->num_arcs();
      if (!o.flags[(unsigned)'d']) next->prune_useless();
      if (!quiet) {
        std::cerr << "\n\t(" << st << " states / " << ar << " arcs";
        if (next->states.size() != st || next->num_arcs() != ar)
          std::cerr << " reduce-> " << next->states.size() << "/" << next->num_arcs();
        std::cerr << ")";
      }
      composed = std::move(next);
      result = composed.get();
    }
    if (!quiet) std::cerr << std::endl;
  }
  if (!with_pairs) {  // plain `carmel a b ...`: print the (reduced) composition — no GPU involved
    const int ws = wstyle;
    if (o.flags[(unsigned)'c'])
      std::cout << "Number of states in result: " << result->states.size() << "\nNumber of arcs in result: "
                << result->num_arcs() << "\n";
    else
      std::cout << result->to_text(o.flags[(unsigned)'J'], o.flags[(unsigned)'H'], ws);
    return 0;
  }
  // ---- corpus ----
  HostPairs pairs;
  std::string warn;
  parse_corpus(*result, corpus_text, pairs, &warn, /*weight_lines=*/!scoring);
  std::cerr << warn;
  if (pairs.size() == 0) {  // corpus.set_null() (carmel.cc:1421)
    pairs.weight.push_back(1.0);
    pairs.in_off.push_back(0);
    pairs.out_off.push_back(0);
  }
  // ---- GPU trainer ----
  std::vector<uint32_t> src, dst, in, out, group;
  std::vector<double> logw;
  result->flatten(src, dst, in, out, logw, group);
  carmel_hip_trainer* t = 0;
  // (--comm-plugin: the caller's transport carries the sums; every rank runs on the device --gpu names -- single-GPU boxes, tests)
  const bool one_device = !o.comm_plugin.empty();
  const int my_device = o.gpu + (one_device ? 0 : rank);
  hip_check(carmel_hip_create(&t, my_device, (uint32_t)result->states.size(), result->final_state, logw.size(), src.data(),
                              dst.data(), in.data(), out.data(), logw.data(), group.data()),
            "carmel_hip_create");
  struct Guard {
    carmel_hip_trainer* t;
    ~Guard() { carmel_hip_destroy(t); }
  } guard{t};
  carmel_hip_comm* comm = 0;
  if (world > 1 && !o.comm_plugin.empty()) {
    // the plugin exports  int carmel_hip_transport_open(const char* session, int rank, int world, int device,
    // carmel_hip_transport* out);  the session name is the same on every rank (made before the ranks were forked)
    void* h = dlopen(o.comm_plugin.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) throw std::runtime_error(std::string("--comm-plugin: ") + dlerror());
    typedef int (*open_fn)(const char*, int, int, int, carmel_hip_transport*);
    open_fn op = (open_fn)dlsym(h, "carmel_hip_transport_open");
    if (!op) throw std::runtime_error("--comm-plugin: the library does not export carmel_hip_transport_open");
    carmel_hip_transport tr;
    std::memset(&tr, 0, sizeof tr);
    if (op(g_session.c_str(), rank, world, my_device, &tr) != 0) throw std::runtime_error("--comm-plugin: carmel_hip_transport_open failed");
    hip_check(carmel_hip_comm_create_custom(&comm, my_device, rank, world, &tr), "carmel_hip_comm_create_custom");
    if (carmel_hip_sendrecv_fn sr = (carmel_hip_sendrecv_fn)dlsym(h, "carmel_hip_transport_sendrecv"))  // optional: point-to-point groups
      hip_check(carmel_hip_comm_set_sendrecv(comm, sr), "carmel_hip_comm_set_sendrecv");
    for (int w : id_pipes) close(w);
    if (id_read >= 0) close(id_read);
  } else if (world > 1) {
    unsigned char id[128];
    if (rank == 0) {
      hip_check(carmel_hip_comm_unique_id(id), "carmel_hip_comm_unique_id");
      for (int w : id_pipes) {
        if (write(w, id, sizeof id) != (ssize_t)sizeof id) throw std::runtime_error("could not hand the communicator id to a rank");
        close(w);
      }
    } else {
      size_t got = 0;
      while (got < sizeof id) {
        ssize_t n = read(id_read, id + got, sizeof id - got);
        if (n <= 0) throw std::runtime_error("rank 0 went away before the communicator id arrived");
        got += (size_t)n;
      }
      close(id_read);
    }
    hip_check(carmel_hip_comm_create(&comm, my_device, rank, world, id), "carmel_hip_comm_create");
  }
  if (world > 1 && !o.crp) {
    // this rank's block of the training pairs
    const size_t n = pairs.size(), lo = n * (size_t)rank / (size_t)world, hi = n * (size_t)(rank + 1) / (size_t)world;
    HostPairs mine;
    mine.in_off.assign(1, 0);
    mine.out_off.assign(1, 0);
    for (size_t p = lo; p < hi; ++p) {
      mine.in_sym.insert(mine.in_sym.end(), pairs.in_sym.begin() + pairs.in_off[p], pairs.in_sym.begin() + pairs.in_off[p + 1]);
      mine.out_sym.insert(mine.out_sym.end(), pairs.out_sym.begin() + pairs.out_off[p], pairs.out_sym.begin() + pairs.out_off[p + 1]);
      mine.in_off.push_back(mine.in_sym.size());
      mine.out_off.push_back(mine.out_sym.size());
      mine.weight.push_back(pairs.weight[p]);
    }
    pairs = mine;
    if (!quiet) std::cerr << "Corpus sharded over " << world << " GPUs: rank 0 keeps " << pairs.size() << " of " << n << " pairs\n";
  }
  struct CommGuard {
    carmel_hip_comm*& c;
    ~CommGuard() { carmel_hip_comm_destroy(c); }
  } comm_guard{comm};
  // --disk-cache-derivations with lattices beyond --disk-cache-bufsize: the corpus in shards of pairs [stream_cut[k], stream_cut[k+1]),
  // never more than one shard's lattices resident (set by train_em; empty: everything is resident)
  std::vector<size_t> stream_cut;
  bool streaming = false, stream_prune = true;
  auto set_corpus_range = [&](size_t lo, size_t hi) {
    std::vector<uint64_t> io(1, 0), oo(1, 0);
    for (size_t p = lo; p < hi; ++p) {
      io.push_back(pairs.in_off[p + 1] - pairs.in_off[lo]);
      oo.push_back(pairs.out_off[p + 1] - pairs.out_off[lo]);
    }
    hip_check(carmel_hip_set_corpus(t, hi - lo, io.data(), pairs.in_sym.data() + pairs.in_off[lo], oo.data(),
                                    pairs.out_sym.data() + pairs.out_off[lo], pairs.weight.data() + lo),
              "carmel_hip_set_corpus");
  };
  // A corpus every pair of which has probability 1: the reference's log-domain arithmetic lands on ln P = 0 exactly (a count
  // divided by itself, weight.h:737-830) and its convergence test becomes the quotient of two zeros (weight.h:247-249,
  // train.cc:611,630: the run goes on to -M); sums of exponentials land within an ulp or two of it, on either side, and the
  // same test then sees a ratio of -1e16 or 1.  A corpus probability within two ulps per pair of 1 therefore counts as 1.
  auto snap_certain = [](carmel_hip_estimate_result* er) {
    const double tol = 4.45e-16 * (double)std::max<uint64_t>(er->n_pairs, 1);
    if (std::fabs(er->sum_logprob) <= tol) er->sum_logprob = 0.0;
    if (std::fabs(er->sum_weighted_logprob) <= tol) er->sum_weighted_logprob = 0.0;
  };
  // one E-step over the whole corpus: the count pass on this rank's shard, then (N > 1) the sum over the ranks
  auto estimate_all_raw = [&](carmel_hip_estimate_result* er) {
    if (streaming) {
      // shard by shard: lattices rebuilt, swept, dropped; the count buffers (counts + corpus scalars) added up on the device.
      // A shard none of whose pairs has a derivation adds zeros: the reference gives up only when NO pair of the corpus has one
      // (train.cc:241-252; train_em checks the summed count of kept pairs)
      hip_check(carmel_hip_accumulate_counts(t, 0), "carmel_hip_accumulate_counts");
      for (size_t k = 0; k + 1 < stream_cut.size(); ++k) {
        set_corpus_range(stream_cut[k], stream_cut[k + 1]);
        hip_check(carmel_hip_build_lattices(t, stream_prune ? 1 : 0, 0, nullptr, nullptr), "carmel_hip_build_lattices");
        hip_check(carmel_hip_estimate_async(t), "carmel_hip_estimate_async");
        hip_check(carmel_hip_accumulate_counts(t, 1), "carmel_hip_accumulate_counts");
      }
      hip_check(carmel_hip_accumulate_counts(t, 2), "carmel_hip_accumulate_counts");
      if (comm) hip_check(carmel_hip_allreduce_counts(t, comm), "carmel_hip_allreduce_counts");
      hip_check(carmel_hip_read_scalars(t, er), "carmel_hip_read_scalars");
      return;
    }
    if (!comm) {
      hip_check(carmel_hip_estimate(t, er, 0), "carmel_hip_estimate");
      return;
    }
    hip_check(carmel_hip_estimate_async(t), "carmel_hip_estimate_async");
    hip_check(carmel_hip_allreduce_counts(t, comm), "carmel_hip_allreduce_counts");
    hip_check(carmel_hip_read_scalars(t, er), "carmel_hip_read_scalars");
  };
  auto estimate_all = [&](carmel_hip_estimate_result* er) {
    estimate_all_raw(er);
    snap_certain(er);
  };
  std::vector<uint64_t> coff(1, 0), cpar;
  if (cascade) {
    for (auto& c : chains.chains) {
      cpar.insert(cpar.end(), c.begin(), c.end());
      coff.push_back(cpar.size());
    }
    if (cpar.empty()) cpar.push_back(0);
  }
  auto set_methods = [&](const std::vector<double>& add) {  // the members' normalisation methods with these --priors
    if (cascade)
      hip_check(carmel_hip_set_cascade(t, params.logw.size(), params.logw.data(), params.group.data(),
                                       params.member.data(), params.src.data(), params.in.data(), (uint32_t)nw,
                                       norms.data(), add.data(), chains.chains.size(), coff.data(), cpar.data()),
                "carmel_hip_set_cascade");
    else
      hip_check(carmel_hip_set_norm(t, norms[0], add[0]), "carmel_hip_set_norm");
    if (any_digamma)
      hip_check(carmel_hip_set_digamma(t, (uint32_t)(cascade ? nw : 1), dig_alpha.data(), dig_on.data()), "carmel_hip_set_digamma");
  };
  if (scoring) {
    // carmel -S (carmel.cc:1393-1410): for every pair the sum over all its derivations with the weights as they stand
    // (WFST::sumOfAllPaths, train.cc:925-945 = derivations::init_and_compute + prob): one forward sweep per pair on the GPU
    hip_check(carmel_hip_set_corpus(t, pairs.size(), pairs.in_off.data(), pairs.in_sym.data(), pairs.out_off.data(),
                                    pairs.out_sym.data(), pairs.weight.data()),
              "carmel_hip_set_corpus");
    std::vector<uint8_t> has(pairs.size(), 0);
    carmel_hip_lattice_stats ls;
    hip_check(carmel_hip_build_lattices(t, 1, 0, has.data(), &ls), "carmel_hip_build_lattices");
    std::vector<double> lp(pairs.size(), kNegInf);
    if (ls.n_pairs_kept) {
      carmel_hip_estimate_result er;
      hip_check(carmel_hip_estimate(t, &er, lp.data()), "carmel_hip_estimate");
    }
    double prod = 0;
    for (size_t p = 0; p < pairs.size(); ++p) {
      std::cout << format_weight(has[p] ? lp[p] : kNegInf, wstyle) << std::endl;
      prod += has[p] ? lp[p] : kNegInf;
    }
    std::cerr << "-S corpus product of probs=" << format_weight(prod, wstyle) << ", probability=" << base2(prod);
    if (pairs.size()) std::cerr << " per-line-perplexity(N=" << pairs.size() << ")=" << base2(ppxper(prod, (double)pairs.size()));
    std::cerr << std::endl;
    return 0;
  }
  set_methods(addc);
  // arcs_table priors (derivations.h:96-101) are captured when forward_backward is constructed (train.cc:513): after
  // cascade.normalize (train.cc:509), which for a real cascade normalises the MEMBERS only -- the composed arcs still
  // carry their composition-time products until the first cascade.update() (train.cc:576).  So -U on a cascade takes its
  // prior counts from the weights as composed; a single transducer is its own cascade and gives its normalised weights.
  const bool want_prior = !o.crp && (!cascade || o.smooth_floor > 0 || o.flags[(unsigned)'U']);
  if (want_prior && cascade) hip_check(carmel_hip_set_prior(t, o.smooth_floor, o.flags[(unsigned)'U'] ? 1 : 0), "carmel_hip_set_prior");
  if (!o.crp) hip_check(carmel_hip_normalize(t), "carmel_hip_normalize");  // train.cc:509 (not for --crp, gibbs.cc:403)
  if (want_prior && !cascade) hip_check(carmel_hip_set_prior(t, o.smooth_floor, o.flags[(unsigned)'U'] ? 1 : 0), "carmel_hip_set_prior");
  hip_check(carmel_hip_set_corpus(t, pairs.size(), pairs.in_off.data(), pairs.in_sym.data(), pairs.out_off.data(),
                                  pairs.out_sym.data(), pairs.weight.data()),
            "carmel_hip_set_corpus");
  if (!o.fem_forest.empty()) {  // cached_derivs.h:44-50, 60-100: written on the first pass over the derivations
    carmel_host::FemExport fe;
    fe.n_states = (uint32_t)result->states.size();
    fe.final_state = result->final_state;
    fe.src = &src;
    fe.dst = &dst;
    fe.in = &in;
    fe.out = &out;
    fe.group = &group;
    fe.chains = cascade ? &chains.chains : nullptr;
    std::ofstream of(o.fem_forest.c_str());
    if (!of) throw std::runtime_error("could not create --fem-forest=" + o.fem_forest);
    fe.write_forests(of, pairs.size(), pairs.in_off.data(), pairs.in_sym.data(), pairs.out_off.data(), pairs.out_sym.data(),
                     pairs.weight.data());
  }
  // `log << derivations::global_stats` of cache_derivations (cached_derivs.h:137; derivations.h:197-247), printed whenever
  // the derivations are cached (-? -: --crp).  What the reference prints as "Pre pruning: (S states, A arcs)" is A summed
  // over all pairs but S of the LAST pair, and "Post pruning" is the last pair that has a derivation (see
  // carmel_hip_lattice_stats).  Its two "Avg # of paths" lines are not produced.
  auto log_lattice_stats = [](const carmel_hip_lattice_stats& ls, size_t n) {
    const double s0 = (double)ls.last_pair_explored_states, a0 = (double)ls.explored_arcs, s1 = (double)ls.last_pair_kept_states,
                 a1 = (double)ls.last_pair_kept_arcs;
    std::cerr << "\nTotal for " << n << " cached derivations:\nPre pruning: (" << s0 << " states, " << a0
              << " arcs)\nPost pruning: (" << s1 << " states, " << a1 << " arcs)\nPortion kept: (" << (s0 ? s1 / s0 : 0.0)
              << " states, " << (a0 ? a1 / a0 : 1.0) << " arcs)\n";
  };
  // ---- WFST::train (train.cc:503-678) over the trainer `t` with the iteration controls of `o`; also the --init-em pass
  // of the sampler (gibbs.cc:411-416) ----
  std::ostream& log = std::cerr;
  auto train_em = [&](const Options& o) {
  std::vector<uint8_t> has(pairs.size(), 0);
  carmel_hip_lattice_stats ls;
  stream_cut.clear();
  streaming = false;
  stream_prune = !o.cache_no_prune;
  const bool may_stream = o.stream_lattices && !o.crp && !o.matrix_fb && pairs.size() > 1;  // (--matrix-fb keeps no lattices)
  if (may_stream) {
    // how much GPU memory do this corpus' lattices take?  A probe of its first pairs says (explicit lattices: the shards' count
    // buffers must mean the same thing, carmel_hip_accumulate_counts)
    hip_check(carmel_hip_set_layout_policy(t, 0), "carmel_hip_set_layout_policy");
    const size_t n = pairs.size(), probe = std::min<size_t>(n, 16384);
    set_corpus_range(0, probe);
    carmel_hip_lattice_stats ps;
    hip_check(carmel_hip_build_lattices(t, stream_prune ? 1 : 0, 0, nullptr, &ps), "carmel_hip_build_lattices");
    const double per_pair = (double)ps.device_bytes / (double)probe;
    const double cap = o.resident_bytes ? (double)o.resident_bytes : 64.0 * 1024 * 1024 * 1024;
    // the ranks decide TOGETHER: a rank that streams keeps explicit lattices, plans no exchange and issues the plain all-reduce
    // once per iteration, so if one shard of the corpus is over its budget every rank streams (its own lattices in as many
    // shards as its own probe says, one if they fit) -- ranks on either side of the threshold would otherwise wait in
    // different collectives, or add up count buffers that mean different things
    double any_over[1] = {per_pair * (double)n > cap ? 1.0 : 0.0};
    if (comm) hip_check(carmel_hip_comm_allreduce_host(comm, any_over, 1, 1), "carmel_hip_comm_allreduce_host");
    streaming = any_over[0] != 0.0;
    if (streaming) {
      const size_t per_shard = std::max<size_t>(64, (size_t)std::min<double>(cap / std::max(per_pair, 1.0), 1e18));
      for (size_t lo = 0; lo < n; lo += per_shard) stream_cut.push_back(lo);
      stream_cut.push_back(n);
      if (!quiet)
        std::cerr << "Derivation lattices of " << n << " pairs would take about " << (uint64_t)(per_pair * (double)n) << " bytes of GPU memory; with "
                  << (uint64_t)cap << " allowed they are rebuilt every iteration in " << stream_cut.size() - 1 << " shards of " << per_shard << " pairs\n";
    }
  }
  if (streaming) {
    // first pass: which pairs have a derivation, and the statistics of all shards
    std::memset(&ls, 0, sizeof ls);
    for (size_t k = 0; k + 1 < stream_cut.size(); ++k) {
      set_corpus_range(stream_cut[k], stream_cut[k + 1]);
      carmel_hip_lattice_stats ps;
      hip_check(carmel_hip_build_lattices(t, stream_prune ? 1 : 0, 0, has.data() + stream_cut[k], &ps), "carmel_hip_build_lattices");
      ls.n_pairs += ps.n_pairs;
      ls.n_pairs_kept += ps.n_pairs_kept;
      ls.explored_states += ps.explored_states;
      ls.explored_arcs += ps.explored_arcs;
      ls.kept_states += ps.kept_states;
      ls.kept_arcs += ps.kept_arcs;
      ls.n_cyclic_pairs += ps.n_cyclic_pairs;
      ls.n_bundles += ps.n_bundles;
      ls.max_levels = std::max(ls.max_levels, ps.max_levels);
      ls.device_bytes = std::max(ls.device_bytes, ps.device_bytes);
      ls.build_seconds += ps.build_seconds;
      ls.last_pair_explored_states = ps.last_pair_explored_states;
      if (ps.n_pairs_kept) {
        ls.last_pair_kept_states = ps.last_pair_kept_states;
        ls.last_pair_kept_arcs = ps.last_pair_kept_arcs;
      }
      ls.n_windowed_pairs += ps.n_windowed_pairs;
    }
  } else {
    if (may_stream) {  // (the probe left its own corpus and layout policy behind)
      hip_check(carmel_hip_set_layout_policy(t, 1), "carmel_hip_set_layout_policy");
      hip_check(carmel_hip_set_corpus(t, pairs.size(), pairs.in_off.data(), pairs.in_sym.data(), pairs.out_off.data(),
                                      pairs.out_sym.data(), pairs.weight.data()),
                "carmel_hip_set_corpus");
    }
    hip_check(carmel_hip_build_lattices(t, o.cache_no_prune ? 0 : 1, 0, has.data(), &ls), "carmel_hip_build_lattices");
  }
  if (std::getenv("CARMEL_TIMING"))
    std::cerr << "timing: lattices pairs_kept=" << ls.n_pairs_kept << " states=" << ls.kept_states << " arcs=" << ls.kept_arcs
              << " layout=" << (carmel_hip_lattice_layout(t) == 2 ? "unrolled_dense" : carmel_hip_lattice_layout(t) == 1 ? "unrolled" : "explicit") << " device_bytes=" << ls.device_bytes
              << " build_seconds=" << ls.build_seconds << std::endl;
  if (comm && !o.crp && !streaming) {
    // every rank must hold its lattices in the same layout (a shard with one over-long pair would keep explicit lattices
    // while the others unroll, and the count buffers being summed would mean different things): agree, or rebuild all
    // with explicit lattices; then plan the exchange (sharded where the model allows it, csrc/exchange.cpp)
    double lay[2] = {(double)carmel_hip_lattice_layout(t), -(double)carmel_hip_lattice_layout(t)};
    hip_check(carmel_hip_comm_allreduce_host(comm, lay, 2, 1), "carmel_hip_comm_allreduce_host");
    if (lay[0] != -lay[1]) {
      if (!quiet) std::cerr << "The ranks' shards chose different lattice layouts; rebuilding every rank with explicit lattices\n";
      hip_check(carmel_hip_set_layout_policy(t, 0), "carmel_hip_set_layout_policy");
      hip_check(carmel_hip_build_lattices(t, o.cache_no_prune ? 0 : 1, 0, has.data(), &ls), "carmel_hip_build_lattices");
    }
  }
  if (o.matrix_fb) {  // train.cc:381-383
    if (rank == 0) std::cerr << "Using (input,state,output) full matrix, not derivation lattice.  Usually slower.\n";
    hip_check(carmel_hip_set_matrix_fb(t, 1), "carmel_hip_set_matrix_fb");
  }
  // (after --matrix-fb: the matrix E-step leaves no arc-range-ordered count pass to hang reduce-scatters on, so its exchange is
  // planned as the one all-reduce; csrc/exchange.cpp)
  // (streamed lattices: every shard has its own buckets, so the exchange stays the plain all-reduce of the summed buffer)
  if (comm && !o.crp && !streaming) {
    // the direct form rests on the transport's point-to-point groups: one such group between all ranks, checked, before the plan
    // is made (carmel_hip_comm_selftest); a transport that fails it on any rank keeps the ring collectives
    int form = o.exchange_form;
    if (form == 0 || form == 3) {
      double bad[1] = {carmel_hip_comm_selftest(comm, 0) == CARMEL_HIP_OK ? 0.0 : 1.0};
      hip_check(carmel_hip_comm_allreduce_host(comm, bad, 1, 1), "carmel_hip_comm_allreduce_host");
      if (bad[0] != 0.0) {
        if (form == 3) throw std::runtime_error("--exchange=direct: the transport's point-to-point self-test failed");
        if (rank == 0) std::cerr << "carmel: the transport's point-to-point self-test failed; planning the exchange over the collectives\n";
        form = 2;
      }
    }
    hip_check(carmel_hip_exchange_plan(t, comm, (uint32_t)o.exchange_chunks, form), "carmel_hip_exchange_plan");
  }
  if (o.flags[(unsigned)'?'] || o.flags[(unsigned)':']) log_lattice_stats(ls, pairs.size());
  CorpusStats cs;
  for (size_t p = 0; p < pairs.size(); ++p) {
    if (!has[p]) {
      std::cerr << "No derivations in transducer for input/output #" << (p + 1) << "\n";  // cached_derivs.h:54-58
      continue;
    }
    cs.n_pairs += 1;
    cs.total_weight += pairs.weight[p];
    cs.n_input += (double)(pairs.in_off[p + 1] - pairs.in_off[p]);
    cs.n_output += (double)(pairs.out_off[p + 1] - pairs.out_off[p]);
  }
  if (ls.n_cyclic_pairs)
    std::cerr << "Warning: at least one cycle in derivations for " << ls.n_cyclic_pairs
              << " example(s).  Forward/backward will miss some paths.\n";  // derivations.h:726-728
  if (comm && !o.crp) {  // the counters of training_corpus over ALL ranks' surviving pairs (--crp: every rank has the whole corpus)
    double v[5] = {cs.n_pairs, cs.total_weight, cs.n_input, cs.n_output, (double)ls.n_cyclic_pairs};
    hip_check(carmel_hip_comm_allreduce_host(comm, v, 5, 0), "carmel_hip_comm_allreduce_host");
    cs.n_pairs = v[0];
    cs.total_weight = v[1];
    cs.n_input = v[2];
    cs.n_output = v[3];
  }
  if (cs.n_pairs == 0) throw std::runtime_error("No training example had a derivation - aborting training.");
  auto print_ppx = [&](double ln_p) {  // weight.h:314-329 print_ppx_symbol
    double n_sym = std::max(cs.n_output, cs.n_input);
    log << "probability=" << base2(ln_p);
    if (n_sym) log << " per-symbol-perplexity(N=" << n_sym << ")=" << base2(ppxper(ln_p, n_sym));
    if (cs.n_pairs) log << " per-example-perplexity(N=" << cs.n_pairs << ")=" << base2(ppxper(ln_p, cs.n_pairs));
  };
  // ---- WFST::train (train.cc:503-678) ----
  carmel_hip_estimate_result er;
  if (o.max_iter == -1) {  // "-M" alone: just the corpus perplexity (train.cc:516-517)
    estimate_all(&er);
    log << "Corpus ";
    print_ppx(er.sum_logprob);
    log << "\n";
  } else if (o.max_iter == 0 || (o.max_iter == 1 && o.restarts == 0)) {  // train.cc:520-538
    if (o.max_iter == 0)
      log << "0 iterations specified for training; output weights will be unnormalized fractional counts (except locked "
             "arcs).\n";
    estimate_all(&er);
    log << "Corpus ";
    print_ppx(er.sum_logprob);
    if (o.max_iter == 0)  // prep_new_weights(1.0) + cascade.distribute_counts()
      hip_check(carmel_hip_fractional_counts(t), "carmel_hip_fractional_counts");
    else {
      double mc;
      hip_check(carmel_hip_maximize(t, 1.0, &mc), "carmel_hip_maximize");
    }
    log << "\n";
  } else {
    const bool timing = std::getenv("CARMEL_TIMING") != nullptr;  // per-iteration wall clock on stderr
    double best = std::numeric_limits<double>::infinity(), best_start = best;
    bool have_good = false;
    double growth = o.rate_growth;
    if (cascade && growth != 1.0) {  // train.cc:545-549
      log << "Overrelaxed EM not supported for --train-cascade (compose with -a and train, instead?).  Disabling (growth factor=1)." << std::endl;
      growth = 1.0;
    }
    long restarts_left = o.restarts;
    for (unsigned restart_no = 0;; ++restart_no) {  // train.cc:552-667
    double last_ppx = std::numeric_limits<double>::infinity(), last_change = 10.0;
    bool last_was_reset = false;
    long iter = 0;
    double learning_rate = 1.0;
    for (;;) {
      const bool first_time = iter == 0;
      ++iter;
      const bool cascade_counts = cascade && !first_time;
      if (cascade_counts) hip_check(carmel_hip_save_counts(t), "carmel_hip_save_counts");
      if (iter > o.max_iter && have_good) {
        log << "Maximum number of iterations (" << o.max_iter
            << ") reached before convergence criteria was met - greatest arc weight change was "
            << format_weight(std::log(last_change), W_SOMETIMES_LOG) << "\n";
        break;
      }
      const auto t_e0 = std::chrono::steady_clock::now();
      if (iter > 2 * o.max_iter + 2 && !have_good) {
        // the reference keeps iterating until some iteration was accepted as best (train.cc:577); with a corpus
        // probability of zero or NaN that never happens: stop instead of spinning
        throw std::runtime_error("no iteration produced a usable corpus probability; giving up after " + std::to_string(iter - 1) + " iterations");
      }
      estimate_all(&er);
      if (timing) {
        double sweep_ms = 0;
        carmel_hip_last_sweep_ms(t, &sweep_ms);
        log << "timing: i=" << iter << " estimate " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_e0).count()
            << " ms (kernels " << sweep_ms << " ms)" << std::endl;
      }
      const double new_ppx = ppxper(er.sum_weighted_logprob, cs.total_weight);  // ln p.ppxper(totalEmpiricalWeight)
      log << "i=" << iter << " (rate=" << learning_rate << "): ";
      print_ppx(er.sum_logprob);
      if (new_ppx < best && (!cascade || cascade_counts)) {
        log << " (new best)";
        best = new_ppx;
        have_good = true;
        hip_check(carmel_hip_save_best(t), "carmel_hip_save_best");
      }
      double ratio_ln = kNegInf;
      if (first_time) {
        log << std::endl;
        if (restart_no == 0) {
          best_start = new_ppx;
          log << "Initial best start point ppx=" << base2(new_ppx) << "\n";
        } else {  // random_restart_acceptor::accept (fst.h:1017-1040)
          const double inf = std::numeric_limits<double>::infinity();
          const double tol = o.restart_tolerance > 0 ? std::log(o.restart_tolerance) : inf;  // ln domain
          const double fin = o.final_restart_tolerance > 0 ? std::log(o.final_restart_tolerance) : tol;
          const double N = o.final_restart ? (double)o.final_restart : (double)o.restarts;
          const double lr = restart_no >= N ? fin : tol == inf ? tol : tol + (fin - tol) * ((restart_no - 1) / (N - 1));
          const double ppr = (new_ppx - best_start) / std::fabs(new_ppx);  // weight.h:247-249
          const bool ok = lr > ppr;
          log << "For restart " << restart_no << ", " << (ok ? "accepting" : "rejecting") << " worse random start of "
              << base2(new_ppx) << " compared to " << base2(best_start) << " with relative ppx ratio="
              << format_weight(ppr, W_SOMETIMES_LOG) << " compared to target of "
              << (lr == inf ? std::string("inf") : format_weight(lr, W_SOMETIMES_LOG)) << "\n";
          if (!ok) {
            log << "Random start was insufficiently promising; trying another." << std::endl;
            break;  // to the next random restart
          }
        }
      } else {
        ratio_ln = (new_ppx - last_ppx) / std::fabs(new_ppx);  // weight.h:247-249
        log << " (relative-perplexity-ratio=" << format_weight(ratio_ln, W_SOMETIMES_LOG) << ")";
        if (last_change < 1) log << ", max {d(weight)}=" << format_weight(std::log(last_change), W_SOMETIMES_LOG);
        log << std::endl;
      }
      if (!last_was_reset) {
        if (ratio_ln >= std::log(o.converge_ppx_ratio)) {
          if (learning_rate > 1) {  // train.cc:639-643
            log << "Failed to improve (relaxation rate too high); starting again at learning rate 1" << std::endl;
            learning_rate = 1;
            hip_check(carmel_hip_keep_em_weights(t), "carmel_hip_keep_em_weights");
            last_was_reset = true;
            continue;
          }
          log << "Converged - per-example perplexity ratio exceeds "
              << format_weight(std::log(o.converge_ppx_ratio), W_SOMETIMES_LOG) << " after " << iter << " iterations.\n";
          if (!have_good)
            log << "Because of the --train-cascade implementation, we need another iteration even though we've "
                   "converged.\n";
          else
            break;
        } else if (learning_rate < 20) {  // MAX_LEARNING_RATE_EXP (train.cc:647)
          learning_rate *= growth;
        }
      } else
        last_was_reset = false;
      const auto t_m0 = std::chrono::steady_clock::now();
      hip_check(carmel_hip_maximize(t, learning_rate, &last_change), "carmel_hip_maximize");
      if (timing)
        log << "timing: i=" << iter << " maximize " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_m0).count() << " ms" << std::endl;
      if (last_change <= o.converge && have_good) {
        log << "Converged - maximum weight change less than " << format_weight(std::log(o.converge), W_SOMETIMES_LOG)
            << " after " << iter << " iterations.\n";
        break;
      }
      last_ppx = new_ppx;
    }
    if (restarts_left > 0) {  // train.cc:660-663
      --restarts_left;
      hip_check(carmel_hip_random_restart(t, o.seed, restart_no + 1), "carmel_hip_random_restart");
      log << "\nRandom restart - " << restarts_left << " remaining.\n";
    } else
      break;
    }
    log << "Setting weights to model with lowest per-example-perplexity ( = "
           "prod[modelprob(example)]^(-1/num_examples) = 2^(-log_2(p_model(corpus))/N) = "
        << base2(best) << std::endl;
    hip_check(carmel_hip_load_best(t), "carmel_hip_load_best");
  }
  };
  if (o.crp) {  // WFST::train_gibbs (gibbs.cc:386-430)
    carmel_hip_gibbs_opts go;
    std::memset(&go, 0, sizeof go);
    go.iter = (uint32_t)(o.max_iter > 0 ? o.max_iter : 0);
    go.burnin = (uint32_t)o.burnin;
    go.seed = o.seed;
    go.mode = o.crp_parallel ? 1 : 0;
    go.expectation = o.expectation;
    go.restarts = (uint32_t)std::max(0L, o.crp_restarts);
    go.argmax_final = o.crp_argmax_final;
    go.argmax_sum = o.crp_argmax_sum;
    go.include_self = o.include_self;
    go.random_start = o.random_start;
    go.uniform_p0 = o.uniform_p0;
    go.dirichlet_p0 = o.dirichlet_p0;
    go.final_counts = o.final_counts;
    go.exclude_prior = o.exclude_prior;
    go.min_prior = 1e-2;
    go.high_temp = o.high_temp;
    go.low_temp = o.low_temp;
    for (size_t i = 0; i < nw; ++i)
      if (addc[i] <= 0)
        std::cerr << "Gibbs sampling requires positive --priors for base model / initial sample.  Setting to 0.01\n";
    std::vector<double> init_arc_logw;
    if (o.init_em > 0) {
      // gibbs.cc:400-423: EM without priors gives the weights the first sample is drawn from; the base distribution
      // stays the given one unless --em-p0
      std::vector<double> p0(cascade ? params.logw.size() : logw.size());
      hip_check(carmel_hip_get_weights(t, p0.data()), "carmel_hip_get_weights");
      std::vector<double> zero(nw, 0.0);
      set_methods(zero);
      hip_check(carmel_hip_normalize(t), "carmel_hip_normalize");
      hip_check(carmel_hip_set_prior(t, 0.0, 0), "carmel_hip_set_prior");
      Options em = o;
      em.max_iter = o.init_em;
      em.converge = 0;
      em.converge_ppx_ratio = 1;
      em.restarts = 0;
      em.rate_growth = 1;
      train_em(em);
      init_arc_logw.resize(logw.size());
      hip_check(carmel_hip_get_arc_weights(t, init_arc_logw.data()), "carmel_hip_get_arc_weights");
      std::vector<double> em_w(p0.size());
      hip_check(carmel_hip_get_weights(t, em_w.data()), "carmel_hip_get_weights");
      set_methods(addc);
      hip_check(carmel_hip_set_weights(t, o.em_p0 ? em_w.data() : p0.data()), "carmel_hip_set_weights");
    }
    if (o.init_from_p0 && o.init_em <= 0) {
      // gibbs.cc:405-421: the first sample comes from the composed transducer's own weights instead of the cache.  For a
      // real cascade those are the products made at composition time (cascade.normalize normalises the members, nothing
      // updates the composed arcs); a single transducer is its own cascade: its weights normalised without priors.
      init_arc_logw = logw;
      if (!cascade) {
        std::vector<double> p0(logw.size());
        hip_check(carmel_hip_get_weights(t, p0.data()), "carmel_hip_get_weights");
        std::vector<double> zero(nw, 0.0);
        set_methods(zero);
        hip_check(carmel_hip_normalize(t), "carmel_hip_normalize");
        hip_check(carmel_hip_get_arc_weights(t, init_arc_logw.data()), "carmel_hip_get_arc_weights");
        set_methods(addc);
        hip_check(carmel_hip_set_weights(t, p0.data()), "carmel_hip_set_weights");
      }
    }
    // gibbs_base::print_all -> carmel_gibbs::print_sample (gibbs.hpp:1066-1078; gibbs.cc:258-296): per block, for every input
    // transducer in [a, b) the arcs of the sampled path that belong to it, through WFST::path_print; an arc's weight is
    // proposal_prob of its parameter at the time of printing (plw: ln of it, per parameter)
    const int ws = wstyle;
    auto print_paths = [&](const std::vector<std::vector<uint32_t> >& smp, const std::vector<double>& plw, long a, long b) {
    // parameter id -> (member, source state, arc)
    std::vector<const Transducer*> mem;
    std::vector<size_t> base;
    if (cascade)
      for (size_t i = 0; i < nw; ++i) {
        mem.push_back(&member[i]);
        base.push_back(params.member_base[i]);
      }
    else {
      mem.push_back(result);
      base.push_back(0);
    }
    std::vector<uint32_t> p_src;
    std::vector<const HArc*> p_arc;
    std::vector<uint32_t> p_mem;
    for (size_t i = 0; i < mem.size(); ++i)
      for (uint32_t st = 0; st < mem[i]->states.size(); ++st)
        for (auto& arc : mem[i]->states[st]) {
          p_src.push_back(st);
          p_arc.push_back(&arc);
          p_mem.push_back((uint32_t)i);
        }
    const bool fO = o.flags[(unsigned)'O'], fI = o.flags[(unsigned)'I'], fQ = o.flags[(unsigned)'Q'], fAT = o.flags[(unsigned)'@'],
               fW = o.flags[(unsigned)'W'], fE = o.flags[(unsigned)'E'];
    auto unquote = [](const std::string& x) {
      return (x.size() >= 2 && x[0] == '"' && x[x.size() - 1] == '"') ? x.substr(1, x.size() - 2) : x;
    };
    for (auto& blk : smp)
      for (long i = a; i < b; ++i) {
        const Transducer& W = *mem[(size_t)i];
        bool first = true;
        double lw_path = 0.0;
        std::vector<uint32_t> outs;
        auto sp = [&]() {
          if (!first) std::cout << ' ';
          first = false;
        };
        for (uint32_t pid : blk) {
          if (pid >= p_mem.size() || p_mem[pid] != (uint32_t)i) continue;
          const HArc& arc = *p_arc[pid];
          lw_path += plw[pid];
          if (fAT) {
            if (arc.out != 0) outs.push_back(arc.out);
            if (arc.in != 0) {
              sp();
              std::cout << W.in_syms.names[arc.in];
            }
          } else if (fO || fI) {
            const uint32_t id = fO ? arc.out : arc.in;
            if (!(fE && id == 0)) {
              sp();
              const std::string& nm = fO ? W.out_syms.names[id] : W.in_syms.names[id];
              std::cout << (fQ ? unquote(nm) : nm);
            }
          } else {
            sp();
            std::cout << '(' << W.state_name(p_src[pid]) << " -> " << W.state_name(arc.dest) << ' ' << W.in_syms.names[arc.in] << " : "
                      << W.out_syms.names[arc.out] << " / " << format_weight(plw[pid], ws) << ")";
          }
        }
        if (fAT) {
          std::cout << std::endl;
          bool f2 = true;
          for (uint32_t id : outs) {
            if (!f2) std::cout << ' ';
            f2 = false;
            std::cout << W.out_syms.names[id];
          }
          std::cout << std::endl;
        } else {
          if (!fW) {
            sp();
            std::cout << format_weight(lw_path, ws);
          }
          std::cout << std::endl;
        }
      }
    };
    carmel_hip_gibbs* gs = 0;
    hip_check(carmel_hip_gibbs_create(&gs, t, &go), "carmel_hip_gibbs_create");
    carmel_hip_lattice_stats gls;
    std::memset(&gls, 0, sizeof gls);
    if (carmel_hip_gibbs_lattice_stats(gs, &gls) == CARMEL_HIP_OK) log_lattice_stats(gls, pairs.size());
    if (!init_arc_logw.empty())
      hip_check(carmel_hip_gibbs_set_init_weights(gs, init_arc_logw.data()), "carmel_hip_gibbs_set_init_weights");
    const uint32_t n_runs = go.restarts + 1, per_run = go.iter + 1;
    std::vector<uint32_t> member_states(nw, (uint32_t)result->states.size());
    if (cascade)
      for (size_t i = 0; i < nw; ++i) member_states[i] = (uint32_t)member[i].states.size();
    if (o.pi_stddev > 0)
      hip_check(carmel_hip_gibbs_set_prior_inference(gs, o.pi_stddev, o.pi_global, 0, o.pi_restart_fresh, 0, 0, priorgroup.data(),
                                                     member_states.data(), (uint32_t)nw),
                "carmel_hip_gibbs_set_prior_inference");
    std::vector<double> lp((size_t)per_run * n_runs), lp_after(o.sample_prob_after ? lp.size() : 0);
    // --print-every=N (gibbs_opts.hpp:78-79; gibbs.hpp:959-968 maybe_print_periodic): after sweeps 0, N, 2N, ... a comment line
    // and, with --print-to, every block's sampled path -- the arcs carry the proposal probabilities of that moment
    // (gibbs.cc:272-286); the count / norm tables of --print-counts-* / --print-norms-* follow it (below).
    // With --gpus the runs are spread over the ranks (replicas): every rank keeps what its runs print, run by run, and rank 0
    // prints all of it in run order afterwards -- what one process running the runs one after the other prints.
    //
    // --print-counts-from/-to, --print-norms-from/-to (gibbs.hpp:970-1078; carmel's order gibbs.cc:42-64): the tables are keyed by
    // the ids define_param hands out (gibbs.cc:113-190): member by member, norm group by norm group in NormGroupIter's order (the
    // order --fem-norm lists them in: refhash.hpp), a group's locked arcs first as they come, then its free arcs -- a CONDITIONAL
    // group's in reversed list order --; a member normalised by NONE gets ids only.  Norm ids run on across the members, JOINT
    // states without arcs included; the prior-scale group of a norm group as metanorm assigns it (gibbs.hpp:404-470).
    const size_t n_par = cascade ? params.logw.size() : logw.size();
    std::vector<uint32_t> ref_id(n_par, 0), ref_meta;
    std::vector<int64_t> ref_norm(n_par, -1);
    std::vector<std::vector<uint32_t> > norm_members;  // by reference norm id: the trainer's parameter ids
    const bool want_counts = o.print_counts_to > o.print_counts_from, want_norms = o.print_norms_to > o.print_norms_from;
    std::vector<const Transducer*> tmem;
    if (cascade)
      for (size_t i = 0; i < nw; ++i) tmem.push_back(&member[i]);
    else
      tmem.push_back(result);
    if (want_counts || want_norms) {
      uint32_t gid = 0, nexti = 1;
      size_t p0 = 0;
      for (size_t i = 0; i < tmem.size(); ++i) {
        const Transducer& m = *tmem[i];
        const int pg = priorgroup[i < priorgroup.size() ? i : 0];
        if (norms[i] == CARMEL_HIP_NORM_NONE) {
          for (auto& st : m.states)
            for (size_t k = 0; k < st.size(); ++k) ref_id[p0++] = gid++;
          continue;
        }
        for (uint32_t st = 0; st < m.states.size(); ++st) {
          const auto& arcs = m.states[st];
          auto group = [&](const std::vector<size_t>& g) {  // (arc indices within the state, in the iterator's order)
            std::vector<size_t> free_arcs;
            for (size_t j : g)
              if (arcs[j].group == kLocked)
                ref_id[p0 + j] = gid++;
              else
                free_arcs.push_back(j);
            if (norms[i] == CARMEL_HIP_NORM_CONDITIONAL) std::reverse(free_arcs.begin(), free_arcs.end());
            const uint32_t nid = (uint32_t)norm_members.size();
            norm_members.emplace_back();
            for (size_t j : free_arcs) {
              ref_id[p0 + j] = gid++;
              ref_norm[p0 + j] = nid;
              norm_members.back().push_back((uint32_t)(p0 + j));
            }
            ref_meta.push_back(pg == 0 ? 0u : nexti);  // gibbs.cc:132-137
            if (pg == 2) ++nexti;
          };
          if (norms[i] == CARMEL_HIP_NORM_JOINT) {
            std::vector<size_t> g(arcs.size());
            for (size_t j = 0; j < g.size(); ++j) g[j] = j;
            group(g);
          } else if (!arcs.empty()) {
            std::vector<uint32_t> syms;
            for (auto& a : arcs) syms.push_back(a.in);
            for (uint32_t sym : carmel_host::conditional_group_order(syms)) {
              std::vector<size_t> g;
              for (size_t j = arcs.size(); j-- > 0;)
                if (arcs[j].in == sym) g.push_back(j);
              group(g);
            }
          }
          p0 += arcs.size();
        }
        if (pg == 1) ++nexti;  // gibbs.cc:184
      }
      if (o.pi_global) std::fill(ref_meta.begin(), ref_meta.end(), 1u);  // finish_params: set_global (gibbs.hpp:572-579)
    }
    // print_width (graehl/shared/print_width.hpp:98-130): a number in at most `width` characters
    auto print_width = [&](std::ostream& os, double d) {
      const int width0 = (int)o.width;
      if (width0 >= 20 || d == 0. || width0 <= 0) {
        os << d;
        return;
      }
      const std::ios::fmtflags f = os.flags();
      const std::streamsize pr = os.precision();
      int width = width0;
      double pa = d;
      if (d < 0) {
        pa = -d;
        --width;
      }
      auto sig_for_exp = [](int w, int e) {
        const int r = w - (e < 100 ? 2 : 3) - 3;
        return r > 0 ? r : 0;
      };
      const double wholes = std::log10(pa * (1 + 1e-8));
      if (wholes <= width && d == (double)(int)d)
        os << d;
      else if (pa < 1) {
        const int a = (int)-wholes, need = 2 + a;
        if (need >= width)
          os << std::scientific << std::setprecision(sig_for_exp(width, a) - 1) << d;
        else
          os << std::setprecision(width - 2 - a) << d;
      } else {
        const int a = (int)wholes, need = 1 + a;
        if (need > width)
          os << std::scientific << std::setprecision(sig_for_exp(width, a) - 1) << d;
        else
          os << std::fixed << std::setprecision(need + 1 < width ? width - need - 1 : 0) << d;
      }
      os.flags(f);
      os.precision(pr);
    };
    // print_norms (gibbs.hpp:970-981): the norm sums of groups [from, to) -- a group's sum is the sum of its members' counts
    auto print_norms = [&](uint32_t iter, double time, const std::vector<double>& x) {
      if (!want_norms) return;
      const unsigned long from = o.print_norms_from, to = std::min<unsigned long>(o.print_norms_to, norm_members.size());
      if (!(to > from)) return;
      std::cout << "\n# group\tnormalization group sums i=" << iter << " t=" << time << "\n(\n";
      for (unsigned long n = from; n < to; ++n) {
        double sum = 0;
        for (uint32_t pp : norm_members[n]) sum += x[pp];
        std::cout << ' ' << sum << "\n";
      }
      std::cout << ")\n";
    };
    // print_counts (gibbs.hpp:986-1064): x, s, tm = gibbs_param::sumcount; final: x holds the finalized counts, prob the weights
    auto print_counts = [&](bool final, const char* name, uint32_t iter, double time, const std::vector<double>& x,
                            const std::vector<double>& sacc, const std::vector<double>& tm, const std::vector<double>& prior,
                            const std::vector<double>& prob, const std::vector<double>& touch) {
      if (!want_counts) return;
      const double ta = time + 1;
      std::cout << "\n#id\tgroup\tcount\tprob";
      if (!final) std::cout << "\tavg@" << ta << "\tlast@t\tprior\tgroupby";
      if (o.rich_counts) std::cout << "\tparam name";
      if (!final) std::cout << "\titer=" << iter;
      std::cout << "\t" << name << '\n';
      const unsigned long from = o.print_counts_from, to = std::min<unsigned long>(o.print_counts_to, n_par);
      auto field = [&](double d) {
        std::cout << '\t';
        print_width(std::cout, d);
      };
      // the trainer's parameter p <-> (member, source state, arc)
      auto row = [&](size_t pp, size_t mi, uint32_t src, const HArc& arc) {
        const uint32_t gi = ref_id[pp];
        if (!(gi >= from && gi < to)) return;
        // (a parameter without a norm group -- a locked arc, a member normalised by NONE -- never counts: its sumcount stays 0)
        const bool has = ref_norm[pp] >= 0;
        const double xx = has ? x[pp] : 0.0, sx_ = has ? sacc[pp] : 0.0, tx = has ? tm[pp] : 0.0;
        const double avg = final ? xx / ta : (ta > 0 ? (sx_ + xx * (ta - tx)) / ta : xx);  // delta_sum::avg(ta)
        if (!(o.print_counts_sparse == 0 || avg >= prior[pp] + o.print_counts_sparse)) return;
        std::cout << gi << '\t';
        if (ref_norm[pp] >= 0)
          std::cout << ref_norm[pp];
        else
          std::cout << "LOCKED";
        field(final ? avg : xx);
        field(prob[pp]);
        if (!final) {
          field(avg);
          field(has ? touch[pp] : 0.0);  // delta_sum::tmax as the reference keeps it: the last sweep that changed the count
          field(prior[pp]);
          const uint32_t meta = ref_norm[pp] >= 0 ? ref_meta[(size_t)ref_norm[pp]] : 0u;
          std::cout << '\t';
          if (meta > 0)
            std::cout << meta;
          else
            std::cout << "FIXED";
        }
        if (o.rich_counts) {  // carmel_gibbs::print_param (gibbs.cc:206-212): member index, then WFST::printArc without the weight
          const Transducer& W = *tmem[mi];
          std::cout << '\t' << mi << '(' << W.state_name(src) << " -> " << W.state_name(arc.dest) << ' ' << W.in_syms.names[arc.in]
                    << " : " << W.out_syms.names[arc.out] << ')';
        }
        std::cout << '\n';
      };
      if (o.norm_order) {  // ids in order (gibbs.hpp:1050-1055)
        std::vector<uint32_t> by_id(n_par);
        std::vector<uint32_t> p_src(n_par), p_mem(n_par);
        std::vector<const HArc*> p_arc(n_par);
        size_t pp = 0;
        for (size_t mi = 0; mi < tmem.size(); ++mi)
          for (uint32_t st = 0; st < tmem[mi]->states.size(); ++st)
            for (auto& arc : tmem[mi]->states[st]) {
              by_id[ref_id[pp]] = (uint32_t)pp;
              p_src[pp] = st;
              p_mem[pp] = (uint32_t)mi;
              p_arc[pp] = &arc;
              ++pp;
            }
        for (unsigned long gi = from; gi < to; ++gi) row(by_id[gi], p_mem[by_id[gi]], p_src[by_id[gi]], *p_arc[by_id[gi]]);
      } else {  // "print counts in fst file order, not normgroups order" (gibbs.cc:58-64)
        size_t pp = 0;
        for (size_t mi = 0; mi < tmem.size(); ++mi)
          for (uint32_t st = 0; st < tmem[mi]->states.size(); ++st)
            for (auto& arc : tmem[mi]->states[st]) row(pp++, mi, st, arc);
      }
      std::cout << "\n";
    };
    std::vector<std::string> periodic_text(world > 1 ? (size_t)go.restarts + 1 : 0);
    std::streambuf* const cout_buf = std::cout.rdbuf();
    std::function<void(uint32_t, uint32_t, double)> periodic = [&](uint32_t run, uint32_t iter, double time) {
      std::ostringstream cap;
      struct Redirect {  // (print_paths writes to std::cout)
        std::streambuf* old;
        bool on;
        Redirect(std::ostream& to, bool on_) : old(std::cout.rdbuf()), on(on_) {
          if (on) std::cout.rdbuf(to.rdbuf());
        }
        ~Redirect() {
          if (on) std::cout.rdbuf(old);
        }
      } redirect(cap, world > 1);
      struct Keep {
        std::ostringstream& c;
        std::string* dst;
        ~Keep() {
          if (dst) *dst += c.str();
        }
      } keep{cap, (world > 1 && run < periodic_text.size()) ? &periodic_text[run] : nullptr};
      (void)cout_buf;
      // the tables' state: counts as they stand, their time-weighted sums and stamps, the priors, the proposal probabilities
      std::vector<double> sx, ss, st_, sp, spr, stouch;
      if (want_counts || want_norms) {
        sx.resize(n_par);
        ss.resize(n_par);
        st_.resize(n_par);
        sp.resize(n_par);
        spr.resize(n_par);
        stouch.resize(n_par);
        hip_check(carmel_hip_gibbs_get_state(gs, sx.data(), ss.data(), st_.data(), sp.data(), stouch.data()), "carmel_hip_gibbs_get_state");
        hip_check(carmel_hip_gibbs_current_probs(gs, spr.data()), "carmel_hip_gibbs_current_probs");
        for (size_t pp = 0; pp < n_par; ++pp)  // final_prob (gibbs.hpp:144-151): 0 for a count of 0
          if (ref_norm[pp] >= 0 && !(sx[pp] > 0)) spr[pp] = 0;
      }
      if (iter == 0 && o.print_counts_sparse == 0) {  // gibbs_base::run's prologue (gibbs.hpp:811-814): the priors as counts
        std::cout << "# ";
        if (want_counts) {
          std::vector<double> pprob(n_par);
          for (size_t pp = 0; pp < n_par; ++pp) {
            double ns = 0;
            if (ref_norm[pp] >= 0)
              for (uint32_t q : norm_members[(size_t)ref_norm[pp]]) ns += sp[q];
            pprob[pp] = ref_norm[pp] >= 0 ? (sp[pp] > 0 ? sp[pp] / ns : 0.0) : sp[pp];
          }
          print_counts(true, "(prior counts)", 0, 0.0, sp, ss, st_, sp, pprob, stouch);
        }
      }
      std::cout << "# Gibbs i=" << iter << " ";
      if (go.high_temp != go.low_temp && (go.high_temp > 0 || go.low_temp > 0)) {  // gibbs.hpp:945-955 itername
        const double pw_ = carmel_hip_gibbs_power(go.high_temp, go.low_temp, go.iter, iter);
        std::cout << "temperature=" << 1.0 / pw_ << " power=" << pw_ << " ";
      }
      std::cout << "t=" << time << "\n";
      struct Tables {  // print_all (gibbs.hpp:1066-1078): the sample, then the norm sums, then the counts
        std::function<void()> f;
        ~Tables() { f(); }
      } tables{[&]() {
        print_norms(iter, time, sx);
        print_counts(false, "", iter, time, sx, ss, st_, sp, spr, stouch);
      }};
      if (!(o.print_to > o.print_from)) return;
      if (go.expectation) throw std::runtime_error("can't print sample when using expectation because there is no single sample.\n");
      const size_t n_members = cascade ? nw : 1;
      long a = o.print_from, b = o.print_to;
      if (!(b > a && a < (long)n_members)) return;
      if (b > (long)n_members) b = (long)n_members;
      const uint32_t nbk = carmel_hip_gibbs_n_blocks(gs);
      std::vector<uint32_t> buf(std::max<uint32_t>(1, carmel_hip_gibbs_max_sample(gs)));
      std::vector<std::vector<uint32_t> > smp(nbk);
      for (uint32_t bk = 0; bk < nbk; ++bk) {
        uint32_t n = 0;
        hip_check(carmel_hip_gibbs_get_sample(gs, bk, buf.data(), &n), "carmel_hip_gibbs_get_sample");
        smp[bk].assign(buf.begin(), buf.begin() + n);
      }
      std::vector<double> pr(cascade ? params.logw.size() : logw.size());
      hip_check(carmel_hip_gibbs_current_probs(gs, pr.data()), "carmel_hip_gibbs_current_probs");
      for (double& x : pr) x = x > 0 ? std::log(x) : -std::numeric_limits<double>::infinity();
      print_paths(smp, pr, a, b);
    };
    if (o.print_every > 0) {
      hip_check(carmel_hip_gibbs_set_observer(gs, (uint32_t)o.print_every,
                                              [](void* ctx, uint32_t run, uint32_t iter, double time) {
                                                (*(std::function<void(uint32_t, uint32_t, double)>*)ctx)(run, iter, time);
                                              },
                                              &periodic),
                "carmel_hip_gibbs_set_observer");
    }
    if (world > 1 && (want_counts || want_norms))
      throw UsageError("--print-counts-* / --print-norms-* with --gpus: the tables are one process's (the runs are spread over the ranks)");
    if (world > 1) hip_check(carmel_hip_gibbs_set_run_share(gs, (uint32_t)rank, (uint32_t)world), "carmel_hip_gibbs_set_run_share");
    const auto t_g0 = std::chrono::steady_clock::now();
    int rc = carmel_hip_gibbs_run_ex(gs, lp.data(), 0, o.sample_prob_after ? lp_after.data() : 0);
    uint32_t nblocks = carmel_hip_gibbs_n_blocks(gs);
    if (std::getenv("CARMEL_TIMING") && rc == CARMEL_HIP_OK) {  // (bench.py --config crp)
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_g0).count();
      std::vector<uint32_t> buf(std::max<uint32_t>(1, carmel_hip_gibbs_max_sample(gs)));
      uint64_t sampled = 0;
      if (!go.expectation)
        for (uint32_t b = 0; b < nblocks; ++b) {
          uint32_t n = 0;
          if (carmel_hip_gibbs_get_sample(gs, b, buf.data(), &n) == CARMEL_HIP_OK) sampled += n;
        }
      std::cerr << "timing: gibbs mode=" << (go.mode ? "parallel" : "exact") << " sweeps=" << (uint64_t)per_run * n_runs << " blocks=" << nblocks
                << " lattice_states=" << gls.kept_states << " lattice_arcs=" << gls.kept_arcs << " sampled_params=" << sampled
                << " seconds=" << sec << std::endl;
    }
    uint32_t best_run = carmel_hip_gibbs_best_run(gs);
    double my_stats[3] = {0, 0, 0};
    int my_ran = 0;
    if (rc == CARMEL_HIP_OK) hip_check(carmel_hip_gibbs_best_stats(gs, my_stats, &my_ran), "carmel_hip_gibbs_best_stats");
    std::vector<double> ptrace((size_t)per_run * n_runs * 6, 0.0), pcum(carmel_hip_gibbs_n_prior_scales(gs), 1.0);
    if (o.pi_stddev > 0 && rc == CARMEL_HIP_OK)
      hip_check(carmel_hip_gibbs_prior_trace(gs, ptrace.data(), per_run * n_runs, pcum.data(), (uint32_t)pcum.size()),
                "carmel_hip_gibbs_prior_trace");
    // --print-to: the kept run's sample, block by block (parameter ids along the path, chain order)
    std::vector<std::vector<uint32_t> > final_sample;
    const bool printing = o.print_to > o.print_from;
    if (printing && rc == CARMEL_HIP_OK) {
      if (go.expectation) throw std::runtime_error("can't print sample when using expectation because there is no single sample.\n");
      std::vector<uint32_t> buf(std::max<uint32_t>(1, carmel_hip_gibbs_max_sample(gs)));
      final_sample.resize(nblocks);
      for (uint32_t b = 0; b < nblocks; ++b) {
        uint32_t n = 0;
        hip_check(carmel_hip_gibbs_get_sample(gs, b, buf.data(), &n), "carmel_hip_gibbs_get_sample");
        final_sample[b].assign(buf.begin(), buf.begin() + n);
      }
    }
    std::vector<double> final_x;  // the kept run's counts as finalize_cumulative_counts left them: the final table's
    if ((want_counts || want_norms) && rc == CARMEL_HIP_OK) {
      final_x.resize(n_par);
      hip_check(carmel_hip_gibbs_final_counts(gs, final_x.data()), "carmel_hip_gibbs_final_counts");
    }
    carmel_hip_gibbs_destroy(gs);
    hip_check(rc, "carmel_hip_gibbs_run");
    if (world > 1 && o.print_every > 0) {
      // the runs' periodic output to rank 0, in run order: lengths first, then the bytes (every rank fills its own runs' slots
      // of one vector of doubles -- carmel_hip_comm_allreduce_host is the host-side collective there is)
      std::vector<double> len(periodic_text.size(), 0.0);
      for (size_t r = 0; r < len.size(); ++r) len[r] = (double)periodic_text[r].size();
      hip_check(carmel_hip_comm_allreduce_host(comm, len.data(), (uint32_t)len.size(), 0), "carmel_hip_comm_allreduce_host");
      size_t total = 0;
      std::vector<size_t> at(len.size() + 1, 0);
      for (size_t r = 0; r < len.size(); ++r) at[r + 1] = (total += (size_t)len[r]);
      std::vector<double> bytes(total, 0.0);
      for (size_t r = 0; r < len.size(); ++r)
        for (size_t k = 0; k < periodic_text[r].size(); ++k) bytes[at[r] + k] = (double)(unsigned char)periodic_text[r][k];
      for (size_t k0 = 0; k0 < total; k0 += 1u << 16)
        hip_check(carmel_hip_comm_allreduce_host(comm, bytes.data() + k0, (uint32_t)std::min<size_t>(1u << 16, total - k0), 0),
                  "carmel_hip_comm_allreduce_host");
      if (rank == 0) {
        std::string all(total, ' ');
        for (size_t k = 0; k < total; ++k) all[k] = (char)(unsigned char)bytes[k];
        std::cout << all;
      }
    }
    if (world > 1) {
      // every rank's traces (zeros for the runs it did not take) add up to the whole log; the kept run is the best of the
      // ranks' bests by gibbs_stats::better, the earlier run on a tie -- what the sequential loop would have kept
      hip_check(carmel_hip_comm_allreduce_host(comm, lp.data(), (uint32_t)lp.size(), 0), "carmel_hip_comm_allreduce_host");
      if (!lp_after.empty())
        hip_check(carmel_hip_comm_allreduce_host(comm, lp_after.data(), (uint32_t)lp_after.size(), 0), "carmel_hip_comm_allreduce_host");
      if (o.pi_stddev > 0)
        hip_check(carmel_hip_comm_allreduce_host(comm, ptrace.data(), (uint32_t)ptrace.size(), 0), "carmel_hip_comm_allreduce_host");
      std::vector<double> all((size_t)world * 5, 0.0);
      all[(size_t)rank * 5] = my_ran;
      all[(size_t)rank * 5 + 1] = my_stats[0];
      all[(size_t)rank * 5 + 2] = my_stats[1];
      all[(size_t)rank * 5 + 3] = my_stats[2];
      all[(size_t)rank * 5 + 4] = best_run;
      hip_check(carmel_hip_comm_allreduce_host(comm, all.data(), (uint32_t)all.size(), 0), "carmel_hip_comm_allreduce_host");
      int winner = -1;
      for (int r = 0; r < world; ++r) {
        if (all[(size_t)r * 5] == 0) continue;
        if (winner < 0) {
          winner = r;
          continue;
        }
        const int k = go.argmax_final ? 2 : go.argmax_sum ? 3 : 1;
        const double mine = all[(size_t)r * 5 + k], best = all[(size_t)winner * 5 + k];
        if (mine > best || (mine == best && all[(size_t)r * 5 + 4] < all[(size_t)winner * 5 + 4])) winner = r;
      }
      best_run = (uint32_t)all[(size_t)winner * 5 + 4];
      std::vector<double> wts(cascade ? params.logw.size() : logw.size(), 0.0);
      if (rank == winner) hip_check(carmel_hip_get_weights(t, wts.data()), "carmel_hip_get_weights");
      // (ln weights: -inf from the winner plus 0 from the others stays -inf)
      hip_check(carmel_hip_comm_allreduce_host(comm, wts.data(), (uint32_t)wts.size(), 0), "carmel_hip_comm_allreduce_host");
      hip_check(carmel_hip_set_weights(t, wts.data()), "carmel_hip_set_weights");
      if (printing) {  // --print-to: the kept run's sample lives on the rank that ran it; it travels to rank 0 the same way
        std::vector<double> bl(nblocks, 0.0);
        if (rank == winner)
          for (uint32_t b = 0; b < nblocks; ++b) bl[b] = (double)final_sample[b].size();
        hip_check(carmel_hip_comm_allreduce_host(comm, bl.data(), (uint32_t)bl.size(), 0), "carmel_hip_comm_allreduce_host");
        size_t total = 0;
        for (double v : bl) total += (size_t)v;
        std::vector<double> ids(total, 0.0);
        if (rank == winner) {
          size_t k = 0;
          for (uint32_t b = 0; b < nblocks; ++b)
            for (uint32_t id : final_sample[b]) ids[k++] = (double)id;
        }
        for (size_t k0 = 0; k0 < total; k0 += 1u << 16)
          hip_check(carmel_hip_comm_allreduce_host(comm, ids.data() + k0, (uint32_t)std::min<size_t>(1u << 16, total - k0), 0),
                    "carmel_hip_comm_allreduce_host");
        final_sample.assign(nblocks, std::vector<uint32_t>());
        size_t k = 0;
        for (uint32_t b = 0; b < nblocks; ++b)
          for (size_t j = 0; j < (size_t)bl[b]; ++j) final_sample[b].push_back((uint32_t)ids[k++]);
      }
      if (rank > 0) return 0;
    }
    double n_sym = 0;  // gibbs_base::init(derivs.n_output(), derivs.size())
    for (size_t p = 0; p < pairs.size(); ++p) n_sym += (double)(pairs.out_off[p + 1] - pairs.out_off[p]);
    for (uint32_t r = 0; r < n_runs; ++r) {
      if (go.restarts) std::cerr << "(random restart " << r << " of " << go.restarts << "): \n";  // gibbs.hpp:897
      for (uint32_t i = 0; i <= go.iter; ++i) {  // gibbs.hpp:927-955, gibbs_opts.hpp:298-312
        const double v = o.sample_prob_after ? lp_after[(size_t)r * per_run + i] : lp[(size_t)r * per_run + i];
        std::cerr << "Gibbs i=" << i << " ";
        const double* pt = ptrace.data() + ((size_t)r * per_run + i) * 6;
        if (pt[0] != 0)  // propose_new_priors' line (gibbs.hpp:539-547); the scales shown are the final ones
          std::cerr << (pt[1] != 0 ? "accepted" : "rejected") << " new priors with p1=" << base2(pt[2]) << " p2=" << base2(pt[3])
                    << " a1=p2/p1=" << std::exp(pt[3] - pt[2]) << " a2=q(1|2)/q(2|1)=" << pt[4] << " p_accept=" << pt[5] << ". ";
        std::cerr << (o.sample_prob_after ? "sample(after add-back)" : go.expectation ? "sum-all-derivations" : go.mode ? "cheap(proposal)" : "cache-model")
                  << " prob=" << base2(v);
        if (n_sym) std::cerr << " per-point-ppx(N=" << n_sym << ")=" << base2(-v / n_sym);
        std::cerr << " per-block-ppx(N=" << nblocks << ")=" << base2(-v / nblocks) << "\n";
      }
    }
    if (o.pi_show) {  // gibbs.hpp:826-827
      std::cerr << "Final prior-scale=[";
      for (size_t k = 0; k < pcum.size(); ++k) std::cerr << (k ? " " : "") << pcum[k];
      std::cerr << "]\n";
    }
    if (go.restarts) std::cerr << "\nKept run " << best_run << " of " << go.restarts << " (gibbs_stats::better)\n";
    std::vector<double> pw(cascade ? params.logw.size() : logw.size());
    hip_check(carmel_hip_get_weights(t, pw.data()), "carmel_hip_get_weights");
    const double final_t = (double)go.iter - (double)(go.final_counts ? go.iter : std::min(go.burnin, go.iter));
    bool final_header = false;
    if (printing) {
      // gibbs_base::print_all -> carmel_gibbs::print_sample (gibbs.hpp:1066-1078; gibbs.cc:258-296): per block, for every
      // input transducer in [from, to) the arcs of the sampled path that belong to it, through WFST::path_print; an arc's
      // weight is its probability as trained (proposal_prob after the counts were finalised)
      const size_t n_members = cascade ? nw : 1;
      long a = o.print_from, b = o.print_to;
      if (!(b > a && a < (long)n_members)) {
        std::cerr << "--print-from,-to gibbs [" << a << "," << b << ") is out of range for " << n_members << " input transducers.\n";
      } else {
        if (b > (long)n_members) b = (long)n_members;
        std::cout << "\n# final best gibbs run (start #" << best_run << " t=" << final_t << "):\n";
        final_header = true;
        print_paths(final_sample, pw, a, b);
      }
    }
    if (want_counts || want_norms) {  // ... then the norm sums and the counts of the kept run (gibbs.hpp:1075-1076)
      if (!final_header) std::cout << "\n# final best gibbs run (start #" << best_run << " t=" << final_t << "):\n";
      std::vector<double> fprob(n_par);
      for (size_t pp = 0; pp < n_par; ++pp) fprob[pp] = std::exp(pw[pp]);  // final_prob: the weights (a locked arc's: its own)
      print_norms(go.iter + 1, final_t, final_x);
      print_counts(true, "", go.iter + 1, final_t, final_x, final_x, final_x, final_x, fprob, final_x);
    }
    const char* dir = std::getenv("CARMEL_TRAINED_DIR");
    for (size_t i = 0; i < nw; ++i) {  // cm.write_trained("trained") carmel.cc:1435-1437
      member[i].set_weights(pw.data() + (cascade ? params.member_base[i] : 0));
      std::string fn = std::string(o.files[i + 1]) + ".trained";
      if (dir) {
        std::string b = o.files[i + 1];
        size_t sl = b.rfind('/');
        fn = std::string(dir) + "/" + (sl == std::string::npos ? b : b.substr(sl + 1)) + ".trained";
      }
      std::cerr << "Writing trained " << o.files[i + 1] << " to " << fn << std::endl;
      std::ofstream of(fn.c_str());
      of << member[i].to_text(o.flags[(unsigned)'J'], o.flags[(unsigned)'H'], ws);
    }
    return 0;
  }
  train_em(o);
  if (rank > 0) return 0;  // the results are identical on every rank; rank 0 writes them
  // ---- forest-em side files (carmel.cc:818-831 fem_out; cascade.h:60-116, 167-178) ----
  if (!o.fem_norm.empty() || !o.fem_alpha.empty() || !o.fem_param.empty()) {
    std::vector<double> all_w(cascade ? params.logw.size() : logw.size());
    hip_check(carmel_hip_get_weights(t, all_w.data()), "carmel_hip_get_weights");
    std::vector<const Transducer*> mem;
    if (cascade)
      for (size_t i = 0; i < nw; ++i) mem.push_back(&member[i]);
    else
      mem.push_back(result);
    if (!o.fem_param.empty()) {
      log << "Writing cascade weights to --fem-param=" << o.fem_param << std::endl;
      std::ofstream of(o.fem_param.c_str());
      for (double w : all_w) of << format_weight(w, W_SOMETIMES_LOG) << "\n";
    }
    if (!o.fem_norm.empty()) {
      log << "Writing forest-em normgroups to --fem-norm=" << o.fem_norm << std::endl;
      std::ofstream of(o.fem_norm.c_str());
      of << "(";
      uint64_t id0 = 1;
      for (size_t i = 0; i < mem.size(); ++i) {
        of << "\n";
        const Transducer& m = *mem[i];
        // cascade.h:99-115 over NormGroupIter (fst.h:1362-1446): JOINT -- a group per state, arcs or not, arcs in list order;
        // CONDITIONAL -- per state the input symbols in the order the walk over State::index visits them (refhash.hpp), a
        // symbol's arcs in reversed list order (state.h:158-199 pushes each onto the front of its symbol's list)
        for (uint32_t s = 0; s < m.states.size(); ++s) {
          const auto& arcs = m.states[s];
          if (norms[i] == CARMEL_HIP_NORM_JOINT) {
            of << '(';
            for (size_t k = 0; k < arcs.size(); ++k) of << ' ' << id0 + k;
            of << " )\n";
          } else if (norms[i] == CARMEL_HIP_NORM_CONDITIONAL && !arcs.empty()) {
            std::vector<uint32_t> syms;
            for (auto& a : arcs) syms.push_back(a.in);
            for (uint32_t sym : carmel_host::conditional_group_order(syms)) {
              of << '(';
              for (size_t j = arcs.size(); j-- > 0;)
                if (arcs[j].in == sym) of << ' ' << id0 + j;
              of << " )\n";
            }
          }
          id0 += arcs.size();
        }
      }
      of << ")\n";
    }
    if (!o.fem_alpha.empty()) {
      log << "Writing forest-em alpha to --fem-alpha=" << o.fem_alpha << std::endl;
      std::ofstream of(o.fem_alpha.c_str());
      for (size_t i = 0; i < mem.size(); ++i) {
        const double prior = norms[i] == CARMEL_HIP_NORM_NONE ? -1.0 : addc[i];
        for (auto& st : mem[i]->states)
          for (auto& a : st) of << (a.group == kLocked ? -1.0 : prior) << '\n';
      }
    }
  }
  // ---- output (carmel.cc:1435-1437, 1485-1496; cascade.h:23-32) ----
  const bool full = o.flags[(unsigned)'J'], per_arc = o.flags[(unsigned)'H'];
  if (cascade) {
    std::vector<double> pw(params.logw.size());
    hip_check(carmel_hip_get_weights(t, pw.data()), "carmel_hip_get_weights");
    const char* dir = std::getenv("CARMEL_TRAINED_DIR");  // tests: write beside nothing read-only
    for (size_t i = 0; i < nw; ++i) {
      member[i].set_weights(pw.data() + params.member_base[i]);
      std::string fn = std::string(o.files[i + 1]) + ".trained";
      if (dir) {
        std::string b = o.files[i + 1];
        size_t sl = b.rfind('/');
        fn = std::string(dir) + "/" + (sl == std::string::npos ? b : b.substr(sl + 1)) + ".trained";
      }
      log << "Writing trained " << o.files[i + 1] << " to " << fn << std::endl;
      std::ofstream of(fn.c_str());
      of << member[i].to_text(full, per_arc, wstyle);
    }
  } else {
    std::vector<double> w(logw.size());
    hip_check(carmel_hip_get_weights(t, w.data()), "carmel_hip_get_weights");
    result->set_weights(w.data());
    std::string txt = result->to_text(full, per_arc, wstyle);
    if (!o.out_file.empty()) {
      std::ofstream of(o.out_file.c_str());
      if (!of) {
        std::cerr << "Could not create file " << o.out_file << ".\n";
        return -8;
      }
      of << txt;
    } else
      std::cout << txt;
  }
  return 0;
}

int main(int argc, char** argv) {
  int rc;
  carmel_host::import_env_options();  // (CARMEL_HIP_<KEY> / CARMEL_TIMING: the library takes them as options, not from the environment)
  try {
    rc = run(argc, argv);
  } catch (UsageError& e) {
    std::cerr << "carmel: " << e.what() << "\n";
    rc = -12;
  } catch (std::exception& e) {
    if (g_rank && g_err_fd >= 0)
      dprintf(g_err_fd, "[rank %d] ERROR: %s\n", g_rank, e.what());
    else
      std::cerr << "ERROR: " << e.what() << "\n";  // carmel.cc:1558-1561
    rc = -11;
  }
  // --gpus: rank 0 waits for the other ranks; if it failed itself they may be waiting in a collective -- end them
  g_waiting = 1;
  for (pid_t p : g_kids) {
    if (rc != 0) kill(p, SIGTERM);
    int st = 0;
    if (waitpid(p, &st, 0) > 0 && (!WIFEXITED(st) || WEXITSTATUS(st) != 0) && rc == 0) rc = -11;
  }
  return rc;
}
