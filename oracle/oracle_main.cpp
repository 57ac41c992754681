// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// oracle_carmel: a tiny command line over the restatement, accepting the training subset of carmel's flags
// (carmel.cc:929-1066) so the reference's recorded tutorial commands (carmel-tutorial/commands) can be replayed:
//   oracle_carmel [-t] [--train-cascade] [-M n] [-e d] [-X r] [-f w] [-U] [-u|-j] [-?] [-:] [-H] [-J] [-Z] [-D]
//                 [-d] [-q] [--normby=JCN..] [--priors=a,b,..] corpus wfst [wfst ...]
// EM log lines go to stderr, the trained transducer to stdout (or <file>.trained with --train-cascade).
#include "train.hpp"
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>
#include <pthread.h>

using namespace oracle;

static std::string slurp(const char* fn) {
  std::ifstream f(fn, std::ios::binary);
  if (!f) throw std::runtime_error(std::string("cannot open ") + fn);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static int real_main(int argc, char** argv) {
  bool flags[256] = {0};
  bool trainc = false, random_set = false;
  TrainOpts topt;
  LW converge = LW::from_real(1e-4), converge_pp = LW::from_real(.999), smoothFloor;
  NormalizeMethod nm;
  std::string normby, priors, digamma;
  bool have_digamma = false, plus_set = false;
  double plus_alpha = 0;
  std::vector<const char*> files;
  int idx_threshold = 32;
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
        trainc = true;
      else if (k == "random-set")
        random_set = true;
      else if (k == "normby")
        normby = v;
      else if (k == "priors")
        priors = v;
      else if (k == "digamma") {
        digamma = v;
        have_digamma = true;
      } else if (k == "restart-tolerance")  // carmel.cc:1426-1430
        topt.restart_tolerance = std::atof(v.c_str());
      else if (k == "final-restart-tolerance")
        topt.final_restart_tolerance = std::atof(v.c_str());
      else if (k == "final-restart")
        topt.final_restart = (unsigned)std::atoi(v.c_str());
      else
        std::cerr << "oracle_carmel: ignoring option --" << k << "\n";
    } else if (a.size() > 1 && a[0] == '-') {
      for (size_t j = 1; j < a.size(); ++j) {
        char c = a[j];
        flags[(unsigned char)c] = true;
        if (c == 'j') nm.group = NORM_JOINT;
        if (c == 'u') nm.group = NORM_NONE;
      }
      char last = a.back();
      auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
      if (last == 'M')
        topt.max_iter = (unsigned)std::atoi(next());
      else if (last == 'e')
        converge = LW::from_real(std::atof(next()));
      else if (last == 'X')
        converge_pp = LW::from_real(std::atof(next()));
      else if (last == 'f')
        smoothFloor = LW::from_real(std::atof(next()));
      else if (last == 'T')
        idx_threshold = std::atoi(next());
      else if (last == '!')  // carmel.cc:944-946
        topt.ran_restarts = (unsigned)std::atoi(next());
      else if (last == 'R')
        topt.restart_seed = std::strtoull(next(), 0, 10);
      else if (last == 'o')  // carmel.cc:940-943
        topt.learning_rate_growth_factor = std::max(1.0, std::atof(next()));
      else if (last == '+') {  // carmel.cc:1009-1013
        plus_alpha = std::atof(next());
        plus_set = true;
      }
    } else
      files.push_back(argv[i]);
  }
  if (trainc) flags['t'] = true;
  const bool scoring = !flags['t'] && flags['S'];
  if (!(flags['t'] || scoring) || files.size() < 2) {
    std::cerr << "usage: oracle_carmel -t [opts] corpus wfst [wfst...]\n";
    return 2;
  }
  topt.cache_derivations = flags['?'] || flags[':'];
  std::string corpus_text = slurp(files[0]);
  size_t nw = files.size() - 1;
  std::vector<Wfst> chain(nw);
  for (size_t i = 0; i < nw; ++i) {
    if (!chain[i].read_legible(slurp(files[i + 1]), !flags['K'] ? true : false) || !chain[i].valid) {
      // carmel passes alwaysNamed = !flags['K'] (carmel.cc:1197)
      std::cerr << "Bad format of transducer file: " << files[i + 1] << "\n";
      return 2;
    }
    if (!flags['m'] && nw > 1) chain[i].named_states = false;  // unNameStates carmel.cc:1200
  }
  // carmel_main::norms() carmel.cc:488-499
  size_t N = trainc ? nw : 1;
  std::vector<NormalizeMethod> nms(N, nm);
  for (size_t i = 0; i < normby.size() && i < N; ++i) {
    char c = normby[i];
    nms[i].group = (c == 'J' || c == 'j') ? NORM_JOINT : (c == 'N' || c == 'n') ? NORM_NONE : NORM_CONDITIONAL;
  }
  {
    std::stringstream ss(priors);
    std::string tok;
    size_t i = 0;
    while (std::getline(ss, tok, ',') && i < N) nms[i++].add_count = LW::from_real(std::atof(tok.c_str()));
  }
  if (plus_set)
    for (auto& m : nms) {
      m.scale.linear = false;
      m.scale.alpha = plus_alpha;
    }
  if (have_digamma) {  // carmel.cc:495: one component per transducer, empty = linear
    size_t i = 0, p0 = 0;
    while (i < N) {
      size_t c = digamma.find(',', p0);
      std::string tok = digamma.substr(p0, c == std::string::npos ? std::string::npos : c - p0);
      if (!tok.empty()) {
        nms[i].scale.linear = false;
        nms[i].scale.alpha = std::atof(tok.c_str());
      }
      ++i;
      if (c == std::string::npos) break;
      p0 = c + 1;
    }
  }
  // fem_in (carmel.cc:785-789) --random-set / -1 (fst.h:973-977): a new weight on (0..1] for (-1: a factor on) every unlocked arc
  // of the members not normalised by NONE, from the restart generator's stream 0 (train.hpp restart_uniform)
  if (random_set || flags['1']) {
    std::cerr << "Using random seed -R " << topt.restart_seed << std::endl;
    unsigned p = 0;
    for (size_t i = 0; i < nw; ++i)
      for (auto& st : chain[i].states)
        for (auto& a : st) {
          if (!a.locked() && nms[i < N ? i : 0].group != NORM_NONE) {
            const LW u = LW::from_real(1.0 - restart_uniform(topt.restart_seed, 0, p));
            a.weight = random_set ? u : a.weight * u;
          }
          ++p;
        }
  }
  // fem_in -> fem_normby (carmel.cc:778-783, 800): with --normby the inputs are normalised before composition
  if (!normby.empty()) {
    std::cerr << "Normalizing input transducers by --normby=" << normby << std::endl;
    for (size_t i = 0; i < nw; ++i) chain[i].normalize(nms[i < N ? i : 0]);
  }
  bool remember = trainc;
  Cascade cascade(remember);
  Wfst* result = &chain[0];
  if (!flags['d']) result->reduce();  // cm.minimize(result) carmel.cc:1292
  if (nw < 2 && !cascade.trivial) cascade.set_trivial();
  cascade.add(result);
  std::vector<Wfst*> owned;
  bool anycomposed = false;
  for (size_t i = 1; i < nw; ++i) {
    cascade.add(&chain[i]);
    if (i == 1)
      cascade.prepare_compose();
    else
      cascade.prepare_compose(false);
    Wfst* next = new Wfst();
    if (flags['a'])  // carmel.cc:1318
      compose_a(*next, cascade, *result, chain[i]);
    else
      compose(*next, cascade, *result, chain[i], (unsigned)idx_threshold);
    owned.push_back(next);
    result = next;
    if (!result->valid) {
      std::cerr << "Empty or invalid result of composition with transducer \"" << files[i + 1] << "\".\n";
      return 3;
    }
    size_t st0 = result->num_states(), ar0 = result->num_arcs();
    if (!flags['d']) result->reduce();
    if (!flags['q']) {
      std::cerr << "\n\t(" << st0 << " states / " << ar0 << " arcs";
      if (result->num_states() != st0 || result->num_arcs() != ar0)
        std::cerr << " reduce-> " << result->num_states() << "/" << result->num_arcs();
      std::cerr << ")";
    }
    cascade.done_composing(result);
    anycomposed = true;
  }
  if (!anycomposed) cascade.set_composed(result);
  if (!flags['q']) std::cerr << std::endl;
  int wmode = flags['Z'] ? LW_ALWAYS_LOG : LW_SOMETIMES_LOG;
  if (flags['D']) wmode = LW_NEVER_LOG;
  if (flags['B'])
    wmode |= LW_BASE_LOG10;
  else if (flags['2'])
    wmode |= LW_BASE_LN;
  if (scoring) {  // carmel -S (carmel.cc:1393-1410): WFST::sumOfAllPaths per pair of lines, no weight lines
    ArcTable arcs;
    arcs.build(*result, false, LW());
    IoIndex io;
    io.build(*result);
    size_t p = 0;
    auto next_line = [&](std::string& line) {
      if (p >= corpus_text.size()) return false;
      size_t e = corpus_text.find('\n', p);
      if (e == std::string::npos) e = corpus_text.size();
      line.assign(corpus_text, p, e - p);
      p = e + 1;
      return true;
    };
    std::string l1, l2;
    while (next_line(l1) && next_line(l2)) {
      Corpus one;
      std::string w2;
      read_training_corpus(*result, "1\n" + l1 + "\n" + l2 + "\n", one, &w2);
      Derivations d;
      LW prob;
      if (!one.examples.empty() && d.compute(*result, io, arcs, one.examples.front(), true, 0)) {
        std::vector<LW> f, b;
        prob = d.compute_fb(f, b, [&](const GArc& a) { return arcs.t[a.arcid].arc->weight; });
      }
      std::cout << lw_str(prob, wmode) << std::endl;
    }
    return 0;
  }
  Corpus corpus;
  std::string warn;
  read_training_corpus(*result, corpus_text, corpus, &warn);
  std::cerr << warn;
  std::vector<IterRecord> trace;
  train(*result, cascade, corpus, nms, flags['U'], smoothFloor, converge, converge_pp, topt, &std::cerr, &trace);
  if (trainc) {
    for (size_t i = 0; i < nw; ++i) {
      std::string fn = std::string(files[i + 1]) + ".trained";
      if (flags['O']) fn = std::string("/dev/null");  // not a carmel flag: lets tests avoid writing beside fixtures
      std::cerr << "Writing trained " << files[i + 1] << " to " << fn << std::endl;
      const char* od = std::getenv("ORACLE_TRAINED_DIR");
      if (od) {
        std::string base = files[i + 1];
        size_t sl = base.rfind('/');
        if (sl != std::string::npos) base = base.substr(sl + 1);
        fn = std::string(od) + "/" + base + ".trained";
      }
      std::ofstream of(fn.c_str());
      of << chain[i].write_legible(flags['J'], flags['H'], wmode);
    }
  } else {
    std::cout << result->write_legible(flags['J'], flags['H'], wmode);
  }
  for (Wfst* w : owned) delete w;
  return 0;
}

struct Args {
  int argc;
  char** argv;
  int rc;
};
static void* tramp(void* p) {
  Args* a = (Args*)p;
  try {
    a->rc = real_main(a->argc, a->argv);
  } catch (std::exception& e) {
    std::cerr << "ERROR: " << e.what() << "\n";
    a->rc = -11;
  }
  return 0;
}
int main(int argc, char** argv) {
  // the lattice builder recurses like the reference (derivations.h:640-704): give it a deep stack
  Args a{argc, argv, 0};
  pthread_attr_t attr;
  pthread_attr_init(&attr);
  pthread_attr_setstacksize(&attr, (size_t)2 << 30);
  pthread_t th;
  pthread_create(&th, &attr, tramp, &a);
  pthread_join(th, 0);
  return a.rc;
}
