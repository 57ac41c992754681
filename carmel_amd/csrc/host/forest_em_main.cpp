// forest-em front end: the reference's forest-em command line (forest-em/forest-em-params.hpp:70-170) over the
// carmel_hip_forests_* C-ABI.  Reads the forests / normalisation groups / initial parameter files the reference reads,
// runs EM (graehl/shared/em.hpp:107-216 for one start at learning rate 1; forest-em.hpp:561-655) or the Gibbs sampler
// (--crp, forest-em.hpp:694-766) on the GPU, and writes the parameter / count vectors the reference writes
// (forest-em.hpp:190-201).  Host code only parses, logs and decides when to stop.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/carmel_hip.h"
#include "env_options.hpp"
#include "forest_text.hpp"

using namespace carmel_host;

namespace {

struct Opts {
  // (defaults: ForestEmParams::set_defaults, forest-em-params.hpp:178-224 -- no parameter file is written unless -o names one)
  std::string forests_file, normgroups_file = "-0", initparam_file = "-0", outparam_file = "-0", outcounts_file = "-0";
  long max_iter = 1000;             // --max-iter (forest-em-params.hpp:195)
  long restarts = 0;                // -r / --random-restarts (forest-em-params.hpp:103, em.hpp:199-206)
  double converge_ratio = 1.0 / 65536;   // --converge, relative change of the average log prob (:197)
  double converge_delta = 0;             // --deltaparam-epsilon (:198): by default only a maximize that changes nothing ends the run
  double prior_counts = 0, add_k = 0;
  bool zero_zerocounts = false, normalize_initial = false, human_probs = false;
  bool initial_1 = false;    // -u / --initial-1-params: all parameters start at 1 (forest-em.hpp:307-308) instead of uniform per group
  bool random_set = false;   // --random-set: a random first parameter set (forest-em.hpp:313-316)
  // the outputs of the reference's "final iteration" (forest-em-params.cpp:125-132; forest-em.hpp:497-554)
  std::string out_inside_file = "-0";  // -S / --out-per-forest-inside-sum
  std::string out_pfc_file = "-0";     // -E / --out-per-forest-counts-file
  std::string outviterbi_file = "-0";  // -v / --outviterbi-file
  long crp = 0, burnin = 0;
  // the sampler's final tables (gibbs_opts.hpp:64-77; gibbs.hpp:970-1078 print_all at the end of run_gibbs, forest-em.hpp:719)
  unsigned long print_counts_from = 0, print_counts_to = 0, print_norms_from = 0, print_norms_to = 0;
  long width = 7;
  std::string print_file;           // --print-file (gibbs_opts.hpp:102-103): where the tables go (default stdout)
  long crp_restarts = 0;            // --crp-restarts (gibbs_opts.hpp; gibbs_base::run_starts, gibbs.hpp:880-914)
  bool argmax_final = false, argmax_sum = false;  // --crp-argmax-final / --crp-argmax-sum (gibbs_opts.hpp:270-316)
  double high_temp = 1, low_temp = 1;  // --high-temp / --low-temp (gibbs_opts.hpp:50-53)
  // --prior-inference-* (gibbs_opts.hpp:82-89; forest-em reads all of them through gibbs_opts' own option table)
  double pi_stddev = 0;
  bool pi_global = false, pi_local = false, pi_show = false;
  long pi_start = 0, pi_end = 0;
  bool exclude_prior = false;  // --crp-exclude-prior (gibbs_opts.hpp; gibbs.hpp:629-631)
  std::string outsample_file;  // --outsample-file (gibbs_opts.hpp:100-101; forest-em.hpp:768-787): the final sample, rule ids per forest
  double alpha = 0.1;             // --const-alpha (gibbs_opts.hpp:93)
  std::string alpha_file = "-0";  // --alpha: per-parameter alphas parallel to the weights, negative = locked (:98-99)
  bool final_counts = false, uniform_p0 = false, parallel = false;
  unsigned long long seed = 0;
  int gpu = 0;
  // checkpoints and reports on "watch iterations" (forest-em-params.hpp:138-168, forest-em.hpp:621-653): the first
  // watch_period M-steps of a (re)start and every watch_period-th after
  std::string checkpoint_prefix;      // -x / --checkpoint-prefix
  bool checkpoint_parameters = false; // -c: <prefix>.params.restart.R.iteration.I and <prefix>.counts. ...
  long watch_period = 10;             // -W
  double report_counts = std::numeric_limits<double>::infinity();  // -X / --report-counts-exceeding (ln)
  double report_probs = std::numeric_limits<double>::infinity();   // -Y / --report-probs-exceeding (ln)
  long viterbi_per = 0;              // -V / --checkpoint-viterbi-per-examples: on watch iterations the Viterbi derivation of every
                                     // n-th forest to <prefix>.viterbi.restart.R.iteration.I (forest-em.hpp:403-413, 546-550)
  long per_forest_counts_per = 0;    // -Z / --checkpoint-per-forest-counts: ... <prefix>.per_forest_counts. ... (:416-425, 535-545)
  // the watched rule (forest-em-params.hpp:134-137, 150-151; forest-em.hpp:120-131, 583-616): on watch iterations the top
  // watch_depth rules of the normalisation group that holds it, by weight
  long watch_rule = 0;               // -w / --watch-rule (0: none)
  long watch_depth = 20;             // -D / --watch-depth
  // rule files annotated by id (forest-em-params.hpp:158-165, forest-em-params.cpp:135-143; insert_byid, io.hpp:653-709): a copy of
  // -b with " <F>=weight <C>=count" behind every id=N that stands at a word boundary, under a $$$ header line
  std::string byid_rule_file = "-0", byid_output_file = "-0";  // -b / --byid-rule-file, -B / --byid-output-file
  std::string byid_prob_field = "emprob", byid_count_field = "emcount";  // -F / --byid-prob-field, -C / --byid-count-field
  std::string rules_file = "-0";     // -R / --rules-file: a description per rule, line i for rule i (without one: the number i - 1,
                                     // FileLines::getline of no file, filelines.hpp:80-83)
};

void usage() {
  std::cerr << "usage: forest-em -f forests [-n normgroups] [-I initparams] [-o outparams] [-O outcounts]\n"
               "                 [-i max-iter] [-e converge] [-d deltaparam-epsilon] [-p prior-counts-per] [-k add-k]\n"
               "                 [-z] [-N] [-H] [--crp=N --const-alpha=A --alpha=FILE --burnin=B --high-temp=T --low-temp=T --final-counts --uniform-p0 --crp-parallel\n"
               "                  --crp-restarts=R [--crp-argmax-final | --crp-argmax-sum]\n"
               "                  --prior-inference-stddev=S [--prior-inference-global|-local] [--prior-inference-start=I --prior-inference-end=J] [--prior-inference-show] [--outsample-file=F]]\n"
               "                 [-x checkpoint-prefix -c -V viterbi-per -Z per-forest-counts-per] [-W watch-period] [-X report-counts-exceeding] [-Y report-probs-exceeding]\n"
               "                 [-w watch-rule -D watch-depth -R rules-file] [-b byid-rule-file -B byid-output-file -F prob-field -C count-field]\n"
               "                 [--random-seed=S] [--gpu=D]\n"
               "file arguments: '-' = stdin/stdout, '-0' = none\n";
}

std::string slurp(const std::string& fn) {
  std::ostringstream ss;
  if (fn == "-") {
    ss << std::cin.rdbuf();
  } else {
    std::ifstream in(fn.c_str());
    if (!in) throw std::runtime_error("can't open " + fn);
    ss << in.rdbuf();
  }
  return ss.str();
}
void spit(const std::string& fn, const std::string& text) {
  if (fn == "-") {
    std::cout << text;
  } else {
    std::ofstream of(fn.c_str());
    if (!of) throw std::runtime_error("can't create " + fn);
    of << text;
  }
}
void check(int rc, const char* what) {
  if (rc != CARMEL_HIP_OK) throw std::runtime_error(std::string(what) + ": " + carmel_hip_last_error());
}

// print_width (graehl/shared/print_width.hpp:98-130): a number in at most `width` characters
void print_width(std::ostream& os, double d, int width0) {
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
}

Opts parse_args(int argc, char** argv) {
  Opts o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto value = [&](const std::string& inline_val) -> std::string {
      if (!inline_val.empty()) return inline_val;
      if (i + 1 >= argc) throw std::runtime_error("missing value after " + a);
      return argv[++i];
    };
    std::string key, val;
    if (a.compare(0, 2, "--") == 0) {
      size_t e = a.find('=');
      key = a.substr(2, e == std::string::npos ? std::string::npos : e - 2);
      if (e != std::string::npos) val = a.substr(e + 1);
    } else if (a.size() >= 2 && a[0] == '-') {
      key = std::string(1, a[1]);
      val = a.substr(2);
    } else {
      throw std::runtime_error("unexpected argument " + a);
    }
    if (key == "h" || key == "help") {
      usage();
      std::exit(0);
    } else if (key == "f" || key == "forests-file") o.forests_file = value(val);
    else if (key == "n" || key == "normgroups-file") o.normgroups_file = value(val);
    else if (key == "I" || key == "initparam-file") o.initparam_file = value(val);
    else if (key == "o" || key == "outparam-file") o.outparam_file = value(val);
    else if (key == "O" || key == "outcounts-file") o.outcounts_file = value(val);
    else if (key == "i" || key == "max-iter") o.max_iter = std::atol(value(val).c_str());
    else if (key == "r" || key == "random-restarts") o.restarts = std::atol(value(val).c_str());
    else if (key == "e" || key == "converge") o.converge_ratio = std::atof(value(val).c_str());
    else if (key == "d" || key == "deltaparam-epsilon") o.converge_delta = std::atof(value(val).c_str());
    else if (key == "p" || key == "prior-counts-per") o.prior_counts = std::atof(value(val).c_str());
    else if (key == "k" || key == "add-k-smoothing") o.add_k = std::atof(value(val).c_str());
    else if (key == "z" || key == "zero-zerocounts") o.zero_zerocounts = true;
    else if (key == "N" || key == "normalize-initial") o.normalize_initial = true;
    else if (key == "H" || key == "human-probs") o.human_probs = true;
    else if (key == "u" || key == "initial-1-params") o.initial_1 = true;
    else if (key == "random-set") o.random_set = true;
    else if (key == "S" || key == "out-per-forest-inside-sum") o.out_inside_file = value(val);
    else if (key == "E" || key == "out-per-forest-counts-file") o.out_pfc_file = value(val);
    else if (key == "v" || key == "outviterbi-file") o.outviterbi_file = value(val);
    else if (key == "U" || key == "use-double-precision") {}  // always double here
    else if (key == "crp") o.crp = std::atol(value(val).c_str());
    else if (key == "const-alpha") o.alpha = std::atof(value(val).c_str());
    else if (key == "alpha") o.alpha_file = value(val);
    else if (key == "burnin") o.burnin = std::atol(value(val).c_str());
    else if (key == "crp-restarts") o.crp_restarts = std::atol(value(val).c_str());
    else if (key == "print-counts-from") o.print_counts_from = std::strtoul(value(val).c_str(), 0, 10);
    else if (key == "print-counts-to") o.print_counts_to = std::strtoul(value(val).c_str(), 0, 10);
    else if (key == "print-norms-from") o.print_norms_from = std::strtoul(value(val).c_str(), 0, 10);
    else if (key == "print-norms-to") o.print_norms_to = std::strtoul(value(val).c_str(), 0, 10);
    else if (key == "print-file") o.print_file = value(val);
    else if (key == "width") {
      o.width = std::atol(value(val).c_str());
      if (o.width < 4) o.width = 20;  // gibbs_opts.hpp:255
    }
    else if (key == "crp-argmax-final") o.argmax_final = true;
    else if (key == "crp-argmax-sum") o.argmax_sum = true;
    else if (key == "high-temp") o.high_temp = std::atof(value(val).c_str());
    else if (key == "low-temp") o.low_temp = std::atof(value(val).c_str());
    else if (key == "final-counts") o.final_counts = true;
    else if (key == "uniform-p0") o.uniform_p0 = true;
    else if (key == "crp-parallel") o.parallel = true;
    else if (key == "outsample-file") o.outsample_file = value(val);
    else if (key == "crp-exclude-prior") o.exclude_prior = true;
    else if (key == "prior-inference-stddev") o.pi_stddev = std::atof(value(val).c_str());
    else if (key == "prior-inference-global") o.pi_global = true;
    else if (key == "prior-inference-local") o.pi_local = true;
    else if (key == "prior-inference-show") o.pi_show = true;
    else if (key == "prior-inference-start") o.pi_start = std::atol(value(val).c_str());
    else if (key == "prior-inference-end") o.pi_end = std::atol(value(val).c_str());
    else if (key == "prior-inference-restart-fresh") {}  // (restarts and prior inference do not go together here: the library refuses)
    else if (key == "x" || key == "checkpoint-prefix") o.checkpoint_prefix = value(val);
    else if (key == "c" || key == "checkpoint-parameters") o.checkpoint_parameters = true;
    else if (key == "W" || key == "watch-period") o.watch_period = std::atol(value(val).c_str());
    else if (key == "w" || key == "watch-rule") o.watch_rule = std::atol(value(val).c_str());
    else if (key == "D" || key == "watch-depth") o.watch_depth = std::atol(value(val).c_str());
    else if (key == "R" || key == "rules-file") o.rules_file = value(val);
    else if (key == "b" || key == "byid-rule-file") o.byid_rule_file = value(val);
    else if (key == "B" || key == "byid-output-file") o.byid_output_file = value(val);
    else if (key == "F" || key == "byid-prob-field") o.byid_prob_field = value(val);
    else if (key == "C" || key == "byid-count-field") o.byid_count_field = value(val);
    else if (key == "V" || key == "checkpoint-viterbi-per-examples") o.viterbi_per = std::atol(value(val).c_str());
    else if (key == "Z" || key == "checkpoint-per-forest-counts") o.per_forest_counts_per = std::atol(value(val).c_str());
    else if (key == "X" || key == "report-counts-exceeding") {
      if (!carmel_host::parse_weight_token(value(val), o.report_counts)) throw std::runtime_error("bad weight after " + a);
    } else if (key == "Y" || key == "report-probs-exceeding") {
      if (!carmel_host::parse_weight_token(value(val), o.report_probs)) throw std::runtime_error("bad weight after " + a);
    }
    else if (key == "random-seed") o.seed = std::strtoull(value(val).c_str(), 0, 10);
    else if (key == "gpu") o.gpu = std::atoi(value(val).c_str());
    else throw std::runtime_error("unknown option " + a);
  }
  if (o.forests_file.empty()) throw std::runtime_error("no forests file (-f)");
  if (o.checkpoint_prefix.empty()) {  // forest-em-params.cpp:43-50
    o.checkpoint_parameters = false;
    o.viterbi_per = o.per_forest_counts_per = 0;
  }
  return o;
}

// NormalizeGroups on the weights themselves (normalize.hpp:123-164 with source = destination): --normalize-initial
void normalize_weights(std::vector<double>& logw, const std::vector<uint64_t>& off, const std::vector<uint32_t>& rule,
                       bool zero_zerocounts) {
  for (size_t g = 0; g + 1 < off.size(); ++g) {
    double sum = 0;
    for (uint64_t j = off[g]; j < off[g + 1]; ++j) sum += std::exp(logw[rule[j]]);
    for (uint64_t j = off[g]; j < off[g + 1]; ++j) {
      if (sum > 0)
        logw[rule[j]] = logw[rule[j]] - std::log(sum);
      else
        logw[rule[j]] = zero_zerocounts ? -std::numeric_limits<double>::infinity() : -std::log((double)(off[g + 1] - off[g]));
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  carmel_host::import_env_options();
  try {
    Opts o = parse_args(argc, argv);
    std::ostream& log = std::cerr;
    ForestSet fs;
    {
      const std::string text = slurp(o.forests_file);
      ForestReader(text, fs).read_all();
    }
    if (fs.n_forests() == 0) throw std::runtime_error("no forests in " + o.forests_file);
    if (o.byid_output_file != "-0" && o.byid_rule_file == "-0") throw std::runtime_error("Must provide byid-rule-file.");  // forest-em-params.cpp:55-56
    if (o.normgroups_file == "-0" && (o.max_iter || o.normalize_initial))  // forest-em-params.cpp:59-60
      throw std::runtime_error("Missing normgroups-file.\n");
    std::vector<uint64_t> group_off(1, 0);
    std::vector<uint32_t> group_rule;
    uint32_t max_rule = fs.max_rule;
    if (o.normgroups_file != "-0") read_normgroups(slurp(o.normgroups_file), group_off, group_rule, max_rule);
    std::vector<double> init;
    if (o.initparam_file != "-0") {
      init = read_params(slurp(o.initparam_file));
      if (init.size() > max_rule) max_rule = (uint32_t)init.size();
    }
    const uint32_t n_rules = max_rule + 1;  // ids are 1-based; slot 0 is unused
    // FForests::init_rule_weights (forest-em.hpp:297-318): an initial parameter file must cover every rule of the forests
    // and the groups; without one every norm group starts uniform (NormalizeGroups::init_uniform, normalize.hpp:212-234:
    // each member 1, then divided by the group's sum) and a rule in no group keeps the default weight ZERO
    // (dynamic_array::reinit_nodestroy with T(), weight.h:339) -- or, with -u, every parameter starts at 1
    const double neg_inf = -std::numeric_limits<double>::infinity();
    std::vector<double> logw(n_rules, 0.0);
    if (o.initparam_file != "-0") {
      uint32_t need = fs.max_rule;
      for (uint32_t r : group_rule) need = std::max(need, r);
      if (init.size() < need) throw std::runtime_error("Initial params file wasn't large enough for forests/norms.");
      if (init.size() > need)
        log << "Warning: more initial rule weights were provided (" << init.size() + 1
            << ") than used in norms or forests: " << need + 1 << "\n";
      for (size_t r = 0; r < init.size(); ++r) logw[r + 1] = init[r];
      // forests.normalize() (forest-em-params.cpp:96-97 -> normalize.hpp:242-245): the default UNIFORM_ZEROCOUNTS, whatever -z says
      if (o.normalize_initial) normalize_weights(logw, group_off, group_rule, false);
    } else if (!o.initial_1) {
      std::fill(logw.begin(), logw.end(), neg_inf);
      for (size_t g = 0; g + 1 < group_off.size(); ++g)
        for (uint64_t j = group_off[g]; j < group_off[g + 1]; ++j)
          logw[group_rule[j]] = -std::log((double)(group_off[g + 1] - group_off[g]));
    }
    // FForests::randomize (forest-em.hpp:393-399) -> NormalizeGroups::init_random (normalize.hpp:235-238, visit :212-228):
    // every rule of a norm group gets a random positive fraction, then the group is divided by its sum.  The draws come
    // from this build's counter-based generator u(seed, restart, rule, 0) -- the reference's Boost stream is unpinned.
    auto randomize = [&](std::vector<double>& w, uint32_t restart) {
      for (size_t g = 0; g + 1 < group_off.size(); ++g) {
        double sum = 0;
        for (uint64_t j = group_off[g]; j < group_off[g + 1]; ++j) {
          const double v = 1.0 - carmel_hip_gibbs_uniform(o.seed, restart, group_rule[j], 0);
          w[group_rule[j]] = std::log(v);
          sum += v;
        }
        if (sum > 0)
          for (uint64_t j = group_off[g]; j < group_off[g + 1]; ++j) w[group_rule[j]] -= std::log(sum);
      }
    };
    if (o.random_set && o.initparam_file == "-0") randomize(logw, 0);
    log << fs.n_forests() << " forests, " << fs.label.size() << " nodes, " << max_rule << " parameters in "
        << group_off.size() - 1 << " normalization groups.\n";
    carmel_hip_forests* F = nullptr;
    check(carmel_hip_forests_create(&F, o.gpu, fs.n_forests(), fs.node_off.data(), fs.label.data(), fs.ref.data(),
                                    fs.next.data(), n_rules, logw.data(), group_off.size() - 1, group_off.data(),
                                    group_rule.data()),
          "carmel_hip_forests_create");
    const int style = o.human_probs ? W_NEVER_LOG : W_SOMETIMES_LOG;
    // write_viterbi (forest.hpp:581-632) for every `every`-th forest (1-based count, as forest_no % period == 0): best/sum=percent% tree
    auto viterbi_text = [&](const std::vector<double>& sum, uint64_t every) {
      const uint64_t nf = fs.n_forests();
      std::vector<double> best(nf);
      check(carmel_hip_forests_viterbi(F, best.data()), "carmel_hip_forests_viterbi");
      const uint32_t cap = std::max<uint32_t>(1, carmel_hip_forests_max_sample(F));
      std::vector<uint32_t> rules(cap), arity(cap);
      std::string out;
      for (uint64_t f = 0; f < nf; ++f) {
        if ((f + 1) % every) continue;
        uint32_t n = 0;
        check(carmel_hip_forests_get_viterbi(F, f, rules.data(), arity.data(), &n), "carmel_hip_forests_get_viterbi");
        char pct[64];
        std::snprintf(pct, sizeof pct, "%g", 100 * std::exp(best[f] - sum[f]));
        out += format_weight(best[f], style) + "/" + format_weight(sum[f], style) + "=" + pct + "% ";
        uint32_t k = 0;
        std::function<void()> tree = [&]() {
          if (k >= n) return;
          const uint32_t r = rules[k], a = arity[k];
          ++k;
          if (!a) {
            out += std::to_string(r);
            return;
          }
          out += "(" + std::to_string(r);
          for (uint32_t c = 0; c < a; ++c) {
            out += " ";
            tree();
          }
          out += ")";
        };
        tree();
        out += "\n";
      }
      return out;
    };

    if (o.crp > 0) {
      carmel_hip_gibbs_opts go;
      std::memset(&go, 0, sizeof go);
      go.iter = (uint32_t)o.crp;
      go.burnin = (uint32_t)o.burnin;
      go.seed = o.seed;
      go.mode = o.parallel ? 1 : 0;
      go.uniform_p0 = o.uniform_p0;
      go.final_counts = o.final_counts;
      go.exclude_prior = o.exclude_prior;
      go.high_temp = o.high_temp;
      go.low_temp = o.low_temp;
      go.restarts = (uint32_t)std::max(0L, o.crp_restarts);
      go.argmax_final = o.argmax_final;
      go.argmax_sum = o.argmax_sum;
      std::vector<double> alphas_file;
      if (o.alpha_file != "-0") {
        // "(a1 a2 ...)" indexed by rule id like forest-em.hpp:689-692 (alphas[i] for parameter i; entry 0 is the unused rule 0)
        const std::string txt = slurp(o.alpha_file);
        std::vector<double>& al = alphas_file;
        for (size_t p = 0; p < txt.size();) {
          if (std::isdigit((unsigned char)txt[p]) || txt[p] == '-' || txt[p] == '.' || txt[p] == '+') {
            char* e = nullptr;
            al.push_back(std::strtod(txt.c_str() + p, &e));
            p = (size_t)(e - txt.c_str());
          } else
            ++p;
        }
        check(carmel_hip_forests_set_alphas(F, al.data(), (uint32_t)al.size()), "carmel_hip_forests_set_alphas");
      }
      const size_t per_run = (size_t)o.crp + 1;
      std::vector<double> lp(per_run * ((size_t)go.restarts + 1));
      check(carmel_hip_forests_set_prior_inference(F, o.pi_stddev, o.pi_global, o.pi_local, (uint32_t)std::max(0L, o.pi_start),
                                                   (uint32_t)std::max(0L, o.pi_end)),
            "carmel_hip_forests_set_prior_inference");
      check(carmel_hip_forests_gibbs(F, &go, o.alpha, lp.data(), nullptr), "carmel_hip_forests_gibbs");
      std::vector<double> ptrace(lp.size() * 6, 0.0), pcum(group_off.size() + 1, 1.0);
      uint32_t n_scales = 0;
      check(carmel_hip_forests_prior_trace(F, ptrace.data(), (uint32_t)per_run, pcum.data(), (uint32_t)pcum.size(), &n_scales),
            "carmel_hip_forests_prior_trace");
      for (size_t i = 0; i < lp.size(); ++i) {
        if (go.restarts && i % per_run == 0)  // gibbs.hpp:897
          log << "(random restart " << i / per_run << " of " << go.restarts << "): \n";
        log << "i=" << i % per_run << " ";
        const double* pt = ptrace.data() + i * 6;
        if (pt[0] != 0)  // propose_new_priors' line (gibbs.hpp:539-547)
          log << (pt[1] != 0 ? "accepted" : "rejected") << " new priors with p1=2^" << pt[2] / std::log(2.0) << " p2=2^"
              << pt[3] / std::log(2.0) << " a1=p2/p1=" << std::exp(pt[3] - pt[2]) << " a2=q(1|2)/q(2|1)=" << pt[4]
              << " p_accept=" << pt[5] << ". ";
        log << "sample log-prob=" << lp[i] << " (2^" << lp[i] / std::log(2.0) << ")\n";
      }
      if (go.restarts) log << "\nKept run " << carmel_hip_forests_best_run(F) << " of " << go.restarts << " (gibbs_stats::better)\n";
      // gibbs_base::print_all (gibbs.hpp:1066-1078) at the end of FForests::run_gibbs (forest-em.hpp:719): the kept run's norm sums
      // and averaged counts.  Parameters are rule ids (define_param_id, forest-em.hpp:695-709), norm ids the groups' numbers from ONE
      // (visit_norm_param), a locked rule (negative --alpha entry) or one outside every group has none; the numbers through print_width.
      if (o.print_counts_to > o.print_counts_from || o.print_norms_to > o.print_norms_from) {
        std::ostringstream tab;
        std::vector<double> fx(n_rules), fw(n_rules);
        check(carmel_hip_forests_final_counts(F, fx.data()), "carmel_hip_forests_final_counts");
        check(carmel_hip_forests_get_weights(F, fw.data()), "carmel_hip_forests_get_weights");
        std::vector<int64_t> norm_of(n_rules, -1);
        uint32_t nnorm = 0;
        for (size_t g = 0; g + 1 < group_off.size(); ++g)
          for (uint64_t j = group_off[g]; j < group_off[g + 1]; ++j) {
            const uint32_t r = group_rule[j];
            const double a = r < alphas_file.size() ? alphas_file[r] : o.alpha;
            if (a < 0) continue;
            norm_of[r] = (int64_t)g + 1;
            nnorm = std::max<uint32_t>(nnorm, (uint32_t)g + 2);
          }
        const uint32_t burn = go.final_counts ? go.iter : std::min(go.burnin, go.iter);
        const double t_final = (double)go.iter - (double)burn, ta = t_final + 1;
        tab << "\n# final best gibbs run (start #" << carmel_hip_forests_best_run(F) << " t=" << t_final << "):\n";
        if (o.print_norms_to > o.print_norms_from) {
          const unsigned long to = std::min<unsigned long>(o.print_norms_to, nnorm);
          if (to > o.print_norms_from) {
            std::vector<double> ns(nnorm, 0.0);
            for (uint32_t r = 0; r < n_rules; ++r)
              if (norm_of[r] >= 0) ns[(size_t)norm_of[r]] += fx[r];
            tab << "\n# group\tnormalization group sums i=" << go.iter + 1 << " t=" << t_final << "\n(\n";
            for (unsigned long n = o.print_norms_from; n < to; ++n) tab << ' ' << ns[n] << "\n";
            tab << ")\n";
          }
        }
        if (o.print_counts_to > o.print_counts_from) {
          tab << "\n#id\tgroup\tcount\tprob\t\n";
          const unsigned long to = std::min<unsigned long>(o.print_counts_to, n_rules);
          for (unsigned long r = o.print_counts_from; r < to; ++r) {
            const bool has = norm_of[r] >= 0;
            const double avg = (has ? fx[r] : 0.0) / ta;
            tab << r << '\t';
            if (has)
              tab << norm_of[r];
            else
              tab << "LOCKED";
            tab << '\t';
            print_width(tab, avg, (int)o.width);
            tab << '\t';
            print_width(tab, std::exp(fw[r]), (int)o.width);
            tab << '\n';
          }
          tab << "\n";
        }
        if (o.print_file.empty() || o.print_file == "-")
          std::cout << tab.str();
        else
          spit(o.print_file, tab.str());
      }
      if (!o.outsample_file.empty()) {  // print_sample (forest-em.hpp:768-787): one line per forest, its rules in the order sampled
        std::ofstream of(o.outsample_file.c_str());
        std::vector<uint32_t> buf(std::max<uint32_t>(1, carmel_hip_forests_max_sample(F)));
        for (uint64_t f = 0; f < fs.n_forests(); ++f) {
          uint32_t n = 0;
          check(carmel_hip_forests_get_sample(F, f, buf.data(), &n), "carmel_hip_forests_get_sample");
          for (uint32_t k = 0; k < n; ++k) of << (k ? " " : "") << buf[k];
          of << "\n";
        }
      }
      if (o.pi_show) {  // gibbs.hpp:826-827
        log << "Final prior-scale=[";
        for (uint32_t k = 0; k < n_scales && k < pcum.size(); ++k) log << (k ? " " : "") << pcum[k];
        log << "]\n";
      }
    } else {
      // overrelaxed_em (em.hpp:107-216), learning rate 1, with its random restarts
      double best = -std::numeric_limits<double>::infinity();
      std::vector<double> best_w = logw;
      bool very_first = true;
      long restarts_left = o.restarts;
      const bool count_report = !(std::isinf(o.report_counts) && o.report_counts > 0 && std::isinf(o.report_probs) && o.report_probs > 0);
      // FForests::watch_report (forest-em.hpp:583-616): the watched group's members in the order the reports leave them (the
      // reference sorts the group's own member list in place; here the device's normalisation keeps the file's order -- the sums
      // of that one group may differ from the reference's in the last bit after the first report)
      std::vector<uint32_t> watch_members;
      std::vector<std::string> rule_names;
      if (o.watch_rule > 0) {
        for (size_t g = 0; g + 1 < group_off.size() && watch_members.empty(); ++g)  // find_group_holding (normalize.hpp:79-91)
          for (uint64_t j = group_off[g]; j < group_off[g + 1]; ++j)
            if (group_rule[j] == (uint32_t)o.watch_rule) {
              watch_members.assign(group_rule.begin() + group_off[g], group_rule.begin() + group_off[g + 1]);
              break;
            }
        if (watch_members.empty())
          throw std::runtime_error("Couldn't find rule " + std::to_string(o.watch_rule) + " in any normalization groups.\n");
        if (o.rules_file != "-0") {
          std::istringstream in(slurp(o.rules_file));
          for (std::string line; std::getline(in, line);) rule_names.push_back(line);
          if (rule_names.size() < n_rules - 1) {  // load_rule_names (forest-em.hpp:320-332)
            const std::string error = "Not enough lines in rule names file (" + std::to_string(n_rules - 1) + " expected, got " +
                                      std::to_string(rule_names.size()) + ")";
            log << error << std::endl;
            throw std::runtime_error(error);
          }
        }
      }
      auto watch_report = [&]() {
        if (watch_members.empty()) return;
        std::vector<double> cw(n_rules);
        check(carmel_hip_forests_get_weights(F, cw.data()), "carmel_hip_forests_get_weights");
        auto gt = [&](uint32_t a, uint32_t b) { return cw[a] > cw[b]; };  // indirect_gt over the rule weights
        const size_t size = watch_members.size();
        const size_t depth = std::min<size_t>((size_t)std::max<long>(o.watch_depth, 0), size);
        auto b = watch_members.begin(), mid = b + depth, end = watch_members.end();
        // (`firsttime` is already false when maximize gets here, forest-em.hpp:622-626: a group whose first ranking is the
        // file's order is reported as unchanged)
        if (std::is_sorted(b, mid, gt)) {
          log << " (no change in rank order of top " << depth << " rules)";
          return;
        }
        if (mid != end)
          std::partial_sort(b, mid, end, gt);
        else
          std::sort(b, end, gt);
        log << "\nNew top " << depth << " rules for normalization group:";
        for (; b != mid; ++b) {
          // boost::format("\n%1% %|15t|%2% (id = %3%)"): the weight (15 digits, the stream's default spelling), padded to column 15
          std::string line = "\n" + format_weight(cw[*b], W_SOMETIMES_LOG) + " ";
          if (line.size() < 15) line.append(15 - line.size(), ' ');
          line += (rule_names.empty() ? std::to_string(*b - 1) : rule_names[*b - 1]) + " (id = " + std::to_string(*b) + ")";
          log << line;
        }
        log << std::endl;
      };
      for (uint32_t restart = 0;; ++restart) {
      double last = -std::numeric_limits<double>::infinity();
      bool first = true;
      long m_steps = 0;  // FForests::iteration: M-steps since the (re)start (forest-em.hpp:376, 398, 653)
      for (long it = 1; it <= o.max_iter; ++it) {
        double alp = 0;
        uint64_t n_zero = 0;
        // watch_guard (forest-em.hpp:428-441): on a watch iteration the E-step also writes the Viterbi derivation of every -V-th
        // forest and the (empty: forest-em.hpp:383-389) per-forest counts of every -Z-th, under the parameters it runs with
        const bool watch_it = m_steps <= o.watch_period || (o.watch_period && m_steps % o.watch_period == 0);
        const bool ck_vit = watch_it && o.viterbi_per > 0, ck_pfc = watch_it && o.per_forest_counts_per > 0;
        std::vector<double> fsum(ck_vit ? fs.n_forests() : 0);
        check(carmel_hip_forests_estimate(F, o.prior_counts, &alp, &n_zero, ck_vit ? fsum.data() : nullptr), "carmel_hip_forests_estimate");
        if (ck_vit || ck_pfc) {
          const std::string suffix = ".restart." + std::to_string(restart + 1) + ".iteration." + std::to_string(m_steps + 1);
          if (ck_vit) spit(o.checkpoint_prefix + ".viterbi" + suffix, viterbi_text(fsum, (uint64_t)o.viterbi_per));
          if (ck_pfc) {
            std::string t2;
            for (uint64_t f = 0; f < fs.n_forests(); ++f)
              if ((f + 1) % (uint64_t)o.per_forest_counts_per == 0) t2 += "()\n";
            spit(o.checkpoint_prefix + ".per_forest_counts" + suffix, t2);
          }
        }
        log << "i=" << it << " average log-prob=" << alp << " (2^" << alp / std::log(2.0) << " per forest";
        if (n_zero) log << ", " << n_zero << " forests with zero probability ignored";
        log << ")";
        if (alp > best || very_first) {
          best = alp;
          check(carmel_hip_forests_get_weights(F, best_w.data()), "carmel_hip_forests_get_weights");
          log << " (new best)";
        }
        very_first = false;
        double rel = std::numeric_limits<double>::infinity();
        if (!first) {
          double la = std::fabs(last);
          if (la < 1e-5) la = 1e-5;  // LOGPROB_EPSILON (em.hpp)
          rel = (alp - last) / la;
          log << " relative change=" << rel;
        }
        log << "\n";
        first = false;
        if (rel < o.converge_ratio) {
          log << "Converged - relative change in average log-prob less than " << o.converge_ratio << " after " << it
              << " iterations.\n";
          break;
        }
        double delta = 0;
        check(carmel_hip_forests_maximize(F, o.prior_counts, o.add_k, o.zero_zerocounts ? 1 : 0, &delta),
              "carmel_hip_forests_maximize");
        // FForests::maximize's tail (forest-em.hpp:638-653): on a watch iteration the parameters and the counts they were
        // normalised from are dumped (dump_params :172-189) and the counts above the thresholds counted
        if (m_steps <= o.watch_period || (o.watch_period && m_steps % o.watch_period == 0)) {
          watch_report();
          if (o.checkpoint_parameters || count_report) {
            std::vector<double> cw(n_rules), cc(n_rules);
            check(carmel_hip_forests_get_counts(F, o.prior_counts, cc.data()), "carmel_hip_forests_get_counts");
            if (o.checkpoint_parameters) {
              check(carmel_hip_forests_get_weights(F, cw.data()), "carmel_hip_forests_get_weights");
              const std::string suffix = ".restart." + std::to_string(restart + 1) + ".iteration." + std::to_string(m_steps + 1);
              const std::string wf = o.checkpoint_prefix + ".params" + suffix, cf = o.checkpoint_prefix + ".counts" + suffix;
              std::vector<double> lc(n_rules);
              for (uint32_t r = 0; r < n_rules; ++r) lc[r] = cc[r] > 0 ? std::log(cc[r]) : -std::numeric_limits<double>::infinity();
              log << "\nWriting trained parameters to " << wf << "\n";
              spit(wf, write_params(cw.data() + 1, n_rules - 1, style));
              log << "Writing trained counts to " << cf << "\n";
              spit(cf, write_params(lc.data() + 1, n_rules - 1, style));
            }
            if (count_report) {
              // (both tallies run over the COUNTS, as forest-em.hpp:644-649 has them)
              uint64_t n_count = 0, n_prob = 0;
              for (uint32_t r = 1; r < n_rules; ++r) {
                const double lc = cc[r] > 0 ? std::log(cc[r]) : -std::numeric_limits<double>::infinity();
                if (lc >= o.report_counts) ++n_count;
                if (lc >= o.report_probs) ++n_prob;
              }
              log << " (out of " << n_rules - 1 << " parameters, " << n_count << " had count > " << format_weight(o.report_counts, style)
                  << ", and " << n_prob << " had prob > " << format_weight(o.report_probs, style) << ")";
            }
          }
        }
        ++m_steps;
        if (delta <= o.converge_delta) {
          log << "Converged - maximum parameter change " << delta << " after " << it << " iterations.\n";
          break;
        }
        last = alp;
      }
      if (restarts_left <= 0) break;
      --restarts_left;
      log << "\nRandom restart - " << restarts_left << " remaining.\n";
      std::vector<double> rw(n_rules);
      check(carmel_hip_forests_get_weights(F, rw.data()), "carmel_hip_forests_get_weights");
      randomize(rw, restart + 1);
      check(carmel_hip_forests_set_weights(F, rw.data()), "carmel_hip_forests_set_weights");
      }
      // FForests::save_best / restore_best act only when random restarts were asked for (save_best_enable = restarts,
      // forest-em.hpp:363, 660-671): otherwise the parameters stay as the last M-step left them
      if (o.restarts > 0) check(carmel_hip_forests_set_weights(F, best_w.data()), "carmel_hip_forests_set_weights");
      log << "Best average log-prob=" << best << "\n";
    }
    check(carmel_hip_forests_get_weights(F, logw.data()), "carmel_hip_forests_get_weights");
    if (o.outcounts_file != "-0") {
      std::vector<double> counts(n_rules);
      // FForests::write_counts prints the count table as the LAST estimate of the EM loop left it (forest-em-params.cpp:121-122)
      // -- the counts under the parameters of that estimate, one M-step behind the final ones when the loop ended after a
      // maximize; only a run without any estimate (the sampler, -i 0) collects them now
      if (o.crp > 0 || o.max_iter <= 0)
        check(carmel_hip_forests_estimate(F, o.prior_counts, nullptr, nullptr, nullptr), "carmel_hip_forests_estimate");
      check(carmel_hip_forests_get_counts(F, o.prior_counts, counts.data()), "carmel_hip_forests_get_counts");
      std::vector<double> lc(n_rules);
      for (uint32_t r = 0; r < n_rules; ++r) lc[r] = counts[r] > 0 ? std::log(counts[r]) : -std::numeric_limits<double>::infinity();
      log << "Writing trained counts to " << o.outcounts_file << "\n";
      spit(o.outcounts_file, write_params(lc.data() + 1, n_rules - 1, style));
    }
    if (o.outparam_file != "-0") {
      log << "Writing trained parameters to " << o.outparam_file << "\n";
      spit(o.outparam_file, write_params(logw.data() + 1, n_rules - 1, style));
    }
    // the reference's "final iteration" (forest-em-params.cpp:125-132; FForests::operator() forest-em.hpp:511-554): one more
    // pass over the forests with the final parameters, printing per forest the Viterbi derivation (-v), the per-forest
    // counts (-E) and the inside sum (-S)
    if (o.outviterbi_file != "-0" || o.out_pfc_file != "-0" || o.out_inside_file != "-0") {
      if (o.outviterbi_file != "-0") log << "Running final viterbi forests decoding.\n";
      if (o.out_pfc_file != "-0") log << "Running final per-forest counts collection.\n";
      if (o.out_inside_file != "-0") log << "Running final per-forest inside score printing.\n";
      log << "Repeating final iteration ...";
      const uint64_t nf = fs.n_forests();
      std::vector<double> sum(nf);
      check(carmel_hip_forests_estimate(F, o.prior_counts, nullptr, nullptr, sum.data()), "carmel_hip_forests_estimate");
      if (o.outviterbi_file != "-0") spit(o.outviterbi_file, viterbi_text(sum, 1));
      if (o.out_pfc_file != "-0") {
        // FForests::operator()(rule, inside, norm_outside) (forest-em.hpp:383-389) adds a forest's counts to the global table
        // and leaves per_forest_counts -- its accumulate is commented out -- empty: the reference prints "()" per forest
        std::string out;
        for (uint64_t f = 0; f < nf; ++f) out += "()\n";
        spit(o.out_pfc_file, out);
      }
      if (o.out_inside_file != "-0") {
        std::string out;
        for (uint64_t f = 0; f < nf; ++f) out += format_weight(sum[f], style) + "\n";
        spit(o.out_inside_file, out);
      }
      log << "\n";
    }
    if (o.byid_output_file != "-0") {
      // the counts are the table the EM loop's last estimate left (what -O prints); a run that never prepared EM has none
      // (forest-em.hpp:213: counts.size() == 0 -- the sampler, -i 0)
      std::vector<double> lc;
      if (o.crp <= 0 && o.max_iter > 0) {
        std::vector<double> counts(n_rules);
        check(carmel_hip_forests_get_counts(F, o.prior_counts, counts.data()), "carmel_hip_forests_get_counts");
        lc.resize(n_rules);
        for (uint32_t r = 0; r < n_rules; ++r) lc[r] = counts[r] > 0 ? std::log(counts[r]) : -std::numeric_limits<double>::infinity();
      }
      const std::string in = slurp(o.byid_rule_file);
      std::string out = "$$$";
      size_t p = 0;
      // copy_header (fileheader.hpp:65-83): an input that starts with $$$ keeps its header line and gets ours appended to it; one
      // that does not gets a new one, and whatever it matched of "$$$" -- consumed by then -- goes behind it
      while (p < in.size() && p < 3 && in[p] == '$') ++p;
      std::string partial;
      if (p == 3) {
        while (p < in.size() && in[p] != '\n') out += in[p++];
        if (p < in.size()) ++p;
      } else {
        out += " filetype=rule version=1.0";
        partial = in.substr(0, p);
      }
      // ForestEmParams::print (forest-em-params.hpp:313-315); the command line as get_command_line spells it (shell_escape.hpp)
      std::string cmd;
      for (int a = 0; a < argc; ++a) {
        const std::string arg = argv[a];
        const char* special = " \t\n\\><|&;\"'`~*?{}$!()";
        if (a) cmd += ' ';
        if (arg.empty())
          cmd += "\"\"";
        else if (arg.find_first_of(special) == std::string::npos)
          cmd += arg;
        else {
          cmd += '"';
          for (char c : arg) {
            if (c == '\\' || c == '"' || c == '$' || c == '`' || c == '!') cmd += '\\';
            cmd += c;
          }
          cmd += '"';
        }
      }
      out += " forest-em-version= {{ {v20}}} floating-point-precision= {{ {double}}} forest-em-cmdline= {{ {" + cmd + "}}}\n" + partial;
      auto fields = [&](unsigned N) {  // FForests::operator()(out, ruleid), forest-em.hpp:205-211
        if (N < n_rules && !o.byid_prob_field.empty()) out += " " + o.byid_prob_field + "=" + (N ? format_weight(logw[N], style) : std::string("0"));
        if (N < lc.size() && !o.byid_count_field.empty()) out += " " + o.byid_count_field + "=" + (N ? format_weight(lc[N], style) : std::string("0"));
      };
      // insert_byid's state machine (io.hpp:653-709), quirks and all: "id=" counts at the start of the input and after exactly one
      // blank; a blank in the `waiting for i` state, or a mismatch on a non-blank, sends it back to waiting for a blank
      enum { WAIT_SPACE, WAIT_I, SEEN_I, SEEN_ID, SCAN } st = WAIT_I;
      unsigned N = 0;
      for (; p < in.size(); ++p) {
        const char c = in[p];
        auto was_space = [&]() {
          if (c == ' ' || c == '\n' || c == '\t') st = WAIT_I;
        };
        switch (st) {
          case WAIT_SPACE: was_space(); break;
          case WAIT_I: st = c == 'i' ? SEEN_I : WAIT_SPACE; break;
          case SEEN_I: st = c == 'd' ? SEEN_ID : WAIT_SPACE; break;
          case SEEN_ID:
            if (c == '=') {
              N = 0;
              st = SCAN;
            } else
              st = WAIT_SPACE;
            break;
          case SCAN:
            if (c >= '0' && c <= '9')
              N = N * 10 + (unsigned)(c - '0');
            else {
              st = WAIT_SPACE;
              fields(N);
              was_space();
            }
            break;
        }
        out += c;
      }
      if (st == SCAN) fields(N);  // (the file ends without a newline)
      spit(o.byid_output_file, out);
    }
    carmel_hip_forests_destroy(F);
    return 0;
  } catch (std::exception& e) {
    std::cerr << "ERROR: " << e.what() << "\n";
    return 11;
  }
}
