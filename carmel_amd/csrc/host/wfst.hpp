// host/wfst.hpp — the host-side transducer model of the carmel-compatible front end: text reader / writer for
// carmel's WFST and corpus file formats, state pruning, flat-array export for the C-ABI.
//
// File formats follow carmel/doc/FORMATS and the behaviour of /root/reference/carmel/src/wfstio.cc
// (readLegible :341-506, getString :92-150, writeLegible :594-625, symbolList :631-651) and
// train.cc:985-1025 (corpus).  Arc order matters: arcs of a state are kept in file order (state.h:209-231), and
// arc ids handed to the GPU are state-major in that order (derivations.h:86-101).
#pragma once
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace carmel_host {

static const uint32_t kNoGroup = 0xFFFFFFFFu;  // FSTArc::no_group (arc.h:49)
static const uint32_t kLocked = 0u;            // FSTArc::locked_group
static const double kNegInf = -std::numeric_limits<double>::infinity();
static const double kLn10 = 2.30258509299404568402;

struct HArc {
  uint32_t in = 0, out = 0, dest = 0;
  double logw = 0;  // natural log of the weight (weight.h:132-135)
  uint32_t group = kNoGroup;
};

struct SymbolTable {  // *e* = 0, *w* = 1, then first seen (fst.h:58-59,171,409; strhash.h:182-200)
  std::vector<std::string> names;
  std::unordered_map<std::string, uint32_t> ids;
  SymbolTable() {
    add("*e*");
    add("*w*");
  }
  uint32_t add(const std::string& s) {
    auto it = ids.find(s);
    if (it != ids.end()) return it->second;
    uint32_t id = (uint32_t)names.size();
    names.push_back(s);
    ids.emplace(s, id);
    return id;
  }
  bool find(const std::string& s, uint32_t& id) const {
    auto it = ids.find(s);
    if (it == ids.end()) return false;
    id = it->second;
    return true;
  }
};

// ---- weights as text (weight.h:468-528): "x", "e^x", "xln", "10^x", "xlog" ----
inline bool parse_weight_token(const std::string& tok, double& logw) {
  const char* b = tok.c_str();
  const char* end = b + tok.size();
  char* e = 0;
  if (tok.size() >= 2 && b[0] == 'e' && b[1] == '^') {  // a bare "e^" reads as e^0, like the reference
    logw = std::strtod(b + 2, &e);
    return e == end;
  }
  if (tok.size() >= 3 && b[0] == '1' && b[1] == '0' && b[2] == '^') {
    logw = std::strtod(b + 3, &e) * kLn10;
    return e == end;
  }
  double d = std::strtod(b, &e);
  if (e[0] == 'l') {
    if (e[1] == 'n' && e + 2 == end) {
      logw = d;
      return true;
    }
    if (e[1] == 'o' && e[2] == 'g' && e + 3 == end) {
      logw = d * kLn10;
      return true;
    }
    return false;
  }
  if (e != end) return false;
  logw = d > 0 ? std::log(d) : kNegInf;
  return true;
}

// W_BASE_*: how a weight in log form is spelt -- e^x (default), `x ln` with carmel -2, `x log` (base 10) with -B
// (weight.h:468-489; carmel.cc:76-101 / WFST::output_format)
enum WeightStyle { W_SOMETIMES_LOG = 0, W_ALWAYS_LOG = 1, W_NEVER_LOG = 2, W_STYLE_MASK = 3, W_BASE_LN = 16, W_BASE_LOG10 = 32 };
inline std::string format_weight(double logw, int style) {  // weight.h:468-489, 15 significant digits
  char buf[64];
  if (!(logw > kNegInf)) return "0";
  bool fits = logw < 82.0 && logw > -82.0;
  const int st = style & W_STYLE_MASK;
  if ((st == W_SOMETIMES_LOG && fits) || st == W_NEVER_LOG)
    std::snprintf(buf, sizeof buf, "%.15g", std::exp(logw));
  else if (style & W_BASE_LN)
    std::snprintf(buf, sizeof buf, "%.15gln", logw);
  else if (style & W_BASE_LOG10)
    std::snprintf(buf, sizeof buf, "%.15glog", logw * (1. / 2.30258509299404568402));  // getLog10 = oo_ln10 * ln (weight.h:122-126, 265)
  else
    std::snprintf(buf, sizeof buf, "e^%.15g", logw);
  return buf;
}

class Transducer {
 public:
  std::vector<std::vector<HArc> > states;
  uint32_t final_state = 0;
  bool named = false;
  SymbolTable in_syms, out_syms;
  std::vector<std::string> state_names;

  size_t num_arcs() const {
    size_t n = 0;
    for (auto& s : states) n += s.size();
    return n;
  }
  std::string state_name(uint32_t s) const {
    if (named && s < state_names.size()) return state_names[s];
    return std::to_string(s);
  }
  void drop_state_names() {  // WFST::unNameStates (carmel.cc:1200)
    named = false;
    state_names.clear();
    name_ids_.clear();
  }

  // ---- reader ----
  void parse(const std::string& text, bool always_named) {
    Cursor c{text, 0};
    c.skip_comments();
    std::string fin;
    if (!c.token(fin)) throw std::runtime_error("transducer file: missing final state");
    named = always_named;
    if (!always_named)
      for (char ch : fin)
        if (!std::isdigit((unsigned char)ch)) named = true;
    if (!named) final_state = state_id(fin);
    while (c.skip_ws(), !c.eof()) {
      c.expect('(');
      c.skip_comments();
      std::string src;
      if (!c.token(src)) c.fail("state name expected");
      uint32_t s = state_id(src);
      for (;;) {
        c.skip_ws();
        if (c.peek() == ')') break;
        bool group_paren = c.accept('(');
        std::string dst;
        if (!c.token(dst)) c.fail("destination state expected");
        uint32_t d = state_id(dst);
        for (;;) {  // one or more arc specs for this destination
          c.skip_ws();
          bool spec_paren = c.accept('(');
          HArc a = arc_spec(c);
          a.dest = d;
          states[s].push_back(a);
          if (!spec_paren) break;
          c.skip_ws();
          c.expect(')');
          c.skip_ws();
          if (c.peek() == ')') break;
        }
        if (!group_paren) break;
        c.skip_ws();
        c.expect(')');
      }
      c.skip_ws();
      c.expect(')');
    }
    if (named) {
      auto it = name_ids_.find(fin);
      if (it == name_ids_.end()) throw std::runtime_error("Final state named " + fin + " not found.");
      final_state = it->second;
    } else if (final_state >= states.size())
      throw std::runtime_error("final state out of range");
  }

  // corpus line -> symbol ids (wfstio.cc:631-651: unseen symbols get fresh ids)
  void symbols_of_line(const std::string& line, bool output, std::vector<uint32_t>& ids) {
    Cursor c{line, 0};
    std::string tok;
    SymbolTable& t = output ? out_syms : in_syms;
    while (c.skip_ws(), !c.eof()) {
      if (!c.token(tok)) break;
      ids.push_back(t.add(tok));
    }
  }

  // drop states that are not on a start->final path, keeping order, then drop *e*:*e* self loops
  // (WFST::reduce fst.cc:468-524)
  bool prune_useless() {
    const uint32_t n = (uint32_t)states.size();
    if (!n) return false;
    std::vector<char> f(n, 0), b(n, 0);
    std::vector<std::vector<uint32_t> > rev(n);
    for (uint32_t s = 0; s < n; ++s)
      for (auto& a : states[s]) rev[a.dest].push_back(s);
    std::vector<uint32_t> work(1, 0u);
    f[0] = 1;
    while (!work.empty()) {
      uint32_t s = work.back();
      work.pop_back();
      for (auto& a : states[s])
        if (!f[a.dest]) {
          f[a.dest] = 1;
          work.push_back(a.dest);
        }
    }
    work.assign(1, final_state);
    b[final_state] = 1;
    while (!work.empty()) {
      uint32_t s = work.back();
      work.pop_back();
      for (uint32_t p : rev[s])
        if (!b[p]) {
          b[p] = 1;
          work.push_back(p);
        }
    }
    if (!(f[final_state] && b[0])) {
      states.clear();
      return false;
    }
    std::vector<uint32_t> remap(n, kNoGroup);
    uint32_t k = 0;
    for (uint32_t s = 0; s < n; ++s)
      if (f[s] && b[s]) remap[s] = k++;
    std::vector<std::vector<HArc> > ns(k);
    std::vector<std::string> nn;
    for (uint32_t s = 0; s < n; ++s) {
      if (remap[s] == kNoGroup) continue;
      for (auto a : states[s]) {
        if (remap[a.dest] == kNoGroup) continue;
        a.dest = remap[a.dest];
        if (a.in == 0 && a.out == 0 && a.dest == remap[s]) continue;  // state.h:280-289
        ns[remap[s]].push_back(a);
      }
      if (named && s < state_names.size()) nn.push_back(state_names[s]);
    }
    states.swap(ns);
    final_state = remap[final_state];
    if (named) {
      state_names.swap(nn);
      name_ids_.clear();
      for (uint32_t i = 0; i < state_names.size(); ++i) name_ids_[state_names[i]] = i;
    }
    return true;
  }

  // ---- writer (wfstio.cc:594-625) ----
  std::string to_text(bool full /*-J*/, bool arc_per_line /*-H*/, int wstyle, bool include_zero = false) const {
    std::string o = state_name(final_state);
    for (uint32_t s = 0; s < states.size(); ++s) {
      if (!arc_per_line) o += "\n(" + state_name(s);
      for (auto& a : states[s]) {
        if (!include_zero && !(a.logw > kNegInf)) continue;
        if (arc_per_line) o += "\n(" + state_name(s);
        o += " (" + state_name(a.dest);
        if (full || a.in || a.out) {
          const std::string& il = in_syms.names[a.in];
          const std::string& ol = out_syms.names[a.out];
          o += " " + il;
          if (full || il != ol) o += " " + ol;
        }
        if (full || a.group != kNoGroup || a.logw != 0.0) o += " " + format_weight(a.logw, wstyle);
        if (a.group != kNoGroup) {
          o += "!";
          if (a.group > 0) o += std::to_string(a.group);
        }
        o += ")";
        if (arc_per_line) o += ")";
      }
      if (!arc_per_line) o += ")";
    }
    o += "\n";
    return o;
  }

  // flat arrays in arc-id order
  void flatten(std::vector<uint32_t>& src, std::vector<uint32_t>& dst, std::vector<uint32_t>& in,
               std::vector<uint32_t>& out, std::vector<double>& logw, std::vector<uint32_t>& group) const {
    for (uint32_t s = 0; s < states.size(); ++s)
      for (auto& a : states[s]) {
        src.push_back(s);
        dst.push_back(a.dest);
        in.push_back(a.in);
        out.push_back(a.out);
        logw.push_back(a.logw);
        group.push_back(a.group);
      }
  }
  void set_weights(const double* logw) {
    size_t k = 0;
    for (auto& st : states)
      for (auto& a : st) a.logw = logw[k++];
  }
  // WFST::normalize (fst.cc:86-244) on the input transducers themselves: what carmel does to every input BEFORE composing
  // when --normby is given (fem_normby, carmel.cc:778-783, 800).  by: 0 = per (state, input), 1 = per state, 2 = none.
  // add_count is added to every weight first -- locked arcs keep it --, locked arcs reserve their weight, tied groups get
  // (sum of their arcs) / (sum of their states' unlocked mass, made room for the largest locked sum), the rest shares what
  // is left in proportion.  scale: the mean-field scale exp(digamma(x + alpha)) (mean_field_scale.hpp:40-52) or linear.
  // (The training-time M-step runs on the GPU, kernels.hip; this host pass only prepares inputs, like the parser.)
  static double ln_add(double a, double b) {
    if (a == kNegInf) return b;
    if (b == kNegInf) return a;
    return a > b ? a + std::log1p(std::exp(b - a)) : b + std::log1p(std::exp(a - b));
  }
  static double digamma_pos(double x) {  // recurrence up to 6, then the asymptotic series
    double r = 0;
    while (x < 6) {
      r -= 1 / x;
      x += 1;
    }
    const double f = 1 / (x * x);
    return r + std::log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
  }
  void normalize(int by, double add_count, bool dig, double dig_alpha) {
    if (by == 2) return;
    auto scale = [&](double lw) -> double {  // ln of the scaled weight
      if (!dig) return lw;
      const double xa = std::exp(lw) + dig_alpha, floor = .0002;
      if (xa < floor) return digamma_pos(floor) + std::log(xa / floor);
      return digamma_pos(xa);
    };
    const double ln_add_count = add_count > 0 ? std::log(add_count) : kNegInf;
    // norm groups: lists of (state, arc index)
    std::vector<std::vector<std::pair<uint32_t, uint32_t> > > groups;
    for (uint32_t st = 0; st < states.size(); ++st) {
      if (by == 1) {
        groups.emplace_back();
        for (uint32_t k = 0; k < states[st].size(); ++k) groups.back().push_back({st, k});
      } else {
        std::unordered_map<uint32_t, size_t> at;
        for (uint32_t k = 0; k < states[st].size(); ++k) {
          auto it = at.find(states[st][k].in);
          if (it == at.end()) {
            at.emplace(states[st][k].in, groups.size());
            groups.emplace_back();
            groups.back().push_back({st, k});
          } else
            groups[it->second].push_back({st, k});
        }
      }
    }
    std::unordered_map<uint32_t, double> tie_arc, tie_state, tie_maxlocked;  // ln; missing = 0
    auto get = [](std::unordered_map<uint32_t, double>& m, uint32_t k) -> double& {
      auto it = m.find(k);
      if (it == m.end()) it = m.emplace(k, kNegInf).first;
      return it->second;
    };
    for (auto& g : groups) {
      double sum = kNegInf, locked = kNegInf;
      for (auto& sk : g) {
        HArc& a = states[sk.first][sk.second];
        a.logw = ln_add(a.logw, ln_add_count);
        if (a.group == kLocked) locked = ln_add(locked, a.logw); else sum = ln_add(sum, a.logw);
      }
      for (auto& sk : g) {
        const HArc& a = states[sk.first][sk.second];
        if (a.group != kLocked && a.group != kNoGroup) {
          double& t = get(tie_arc, a.group);
          t = ln_add(t, a.logw);
          double& u = get(tie_state, a.group);
          u = ln_add(u, sum);
          double& m = get(tie_maxlocked, a.group);
          if (locked > m) m = locked;
        }
      }
    }
    for (auto& g : groups) {
      double normal = kNegInf, reserved = kNegInf;
      for (auto& sk : g) {
        HArc& a = states[sk.first][sk.second];
        if (a.group != kLocked && a.group != kNoGroup) {
          double gnorm = get(tie_state, a.group);
          const double gmax = get(tie_maxlocked, a.group);
          if (gmax > 0.0) {  // locked arcs of some state sum to more than 1
            a.logw = kNegInf;
          } else {
            if (gmax != kNegInf) gnorm -= std::log1p(-std::exp(gmax));
            const double gtot = get(tie_arc, a.group);
            if (gtot != kNegInf) {
              a.logw = scale(gtot) - scale(gnorm);
              reserved = ln_add(reserved, a.logw);
            } else
              a.logw = kNegInf;
          }
        } else if (a.group == kLocked)
          reserved = ln_add(reserved, a.logw);
        else
          normal = ln_add(normal, a.logw);
      }
      // 1 - reserved (weight.h's operator-=: a result below zero is zero)
      const double remain = reserved == kNegInf ? 0.0 : (reserved >= 0.0 ? kNegInf : std::log1p(-std::exp(reserved)));
      const bool left = remain != kNegInf && normal != kNegInf;
      const double scaled_sum = left ? scale(normal) : 0.0;
      for (auto& sk : g) {
        HArc& a = states[sk.first][sk.second];
        if (a.group == kNoGroup) a.logw = left ? remain + scale(a.logw) - scaled_sum : kNegInf;
      }
    }
  }
  // cascade_parameters::number_from (cascade.h:52-64) / WFST::numberArcsFrom (fst.cc:290-298): consecutive tie-group ids
  uint32_t number_arcs_from(uint32_t label) {
    for (auto& st : states)
      for (auto& a : st) a.group = label++;
    return label;
  }

 private:
  std::unordered_map<std::string, uint32_t> name_ids_;

  struct Cursor {
    const std::string& s;
    size_t p;
    bool eof() const { return p >= s.size(); }
    char peek() const { return p < s.size() ? s[p] : '\0'; }
    void skip_ws() {
      while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
    }
    void skip_comments() {  // '%' starts a comment that runs to the end of the line (wfstio.cc:338)
      for (;;) {
        skip_ws();
        if (peek() != '%') return;
        while (p < s.size() && s[p] != '\n') ++p;
      }
    }
    bool accept(char c) {
      skip_ws();
      if (peek() == c) {
        ++p;
        return true;
      }
      return false;
    }
    void expect(char c) {
      skip_ws();
      if (peek() != c) fail(std::string("expected '") + c + "'");
      ++p;
    }
    [[noreturn]] void fail(const std::string& what) const {
      size_t line = 1;
      for (size_t i = 0; i < p && i < s.size(); ++i)
        if (s[i] == '\n') ++line;
      throw std::runtime_error("transducer file, line " + std::to_string(line) + ": " + what);
    }
    // one symbol / state-name / weight token (wfstio.cc:92-150): "quoted with \\ escapes", *special* (lower-cased),
    // or bare text up to whitespace, '!' or ')'
    bool token(std::string& out) {
      out.clear();
      skip_ws();
      if (eof()) return false;
      char c = s[p];
      if (c == '(' || c == ')') return false;
      if (c == '"') {
        out.push_back(s[p++]);
        bool esc = false;
        while (p < s.size()) {
          char d = s[p++];
          out.push_back(d);
          if (d == '"' && !esc) return true;
          esc = (d == '\\') ? !esc : false;
        }
        return false;
      }
      if (c == '*') {
        out.push_back(s[p++]);
        while (p < s.size()) {
          char d = s[p++];
          if (d == '*') {
            out.push_back(d);
            return true;
          }
          out.push_back((char)std::tolower((unsigned char)d));
        }
        return false;
      }
      out.push_back(s[p++]);
      while (p < s.size()) {
        char d = s[p];
        if (d == ' ' || d == '\t' || d == '\n') {
          ++p;
          break;
        }
        if (d == '!' || d == ')') break;
        out.push_back(d);
        ++p;
      }
      if (!out.empty() && out.back() == '\r') out.pop_back();
      return true;
    }
  };

  uint32_t state_id(const std::string& name) {
    if (!named) {
      char* e = 0;
      long v = std::strtol(name.c_str(), &e, 10);
      if (*e != '\0' || v < 0)
        throw std::runtime_error("Since intial state was a number, expected an integer state index, but got: " + name);
      if ((size_t)v >= states.size()) states.resize((size_t)v + 1);
      return (uint32_t)v;
    }
    auto it = name_ids_.find(name);
    if (it != name_ids_.end()) return it->second;
    uint32_t id = (uint32_t)state_names.size();
    state_names.push_back(name);
    name_ids_.emplace(name, id);
    if (id >= states.size()) states.resize((size_t)id + 1);
    return id;
  }

  // [[input [output]] weight] [! [group]]   (carmel/doc/FORMATS; wfstio.cc:398-464)
  HArc arc_spec(Cursor& c) {
    HArc a;
    auto at_end = [&]() {
      c.skip_ws();
      return c.peek() == ')' || c.peek() == '!';
    };
    if (!at_end()) {
      std::string t1, t2, t3;
      if (!c.token(t1)) c.fail("symbol or weight expected");
      if (at_end()) {  // weight | symbol
        double w;
        if (parse_weight_token(t1, w))
          a.logw = w;
        else
          a.in = in_syms.add(t1), a.out = out_syms.add(t1);
      } else {
        a.in = in_syms.add(t1);
        if (!c.token(t2)) c.fail("symbol or weight expected");
        if (at_end()) {  // iosymbol weight | isymbol osymbol
          double w;
          if (parse_weight_token(t2, w)) {
            a.logw = w;
            a.out = out_syms.add(t1);
          } else
            a.out = out_syms.add(t2);
        } else {  // isymbol osymbol weight
          a.out = out_syms.add(t2);
          if (!c.token(t3)) c.fail("weight expected");
          double w;
          if (t3.size() > 3 && t3.compare(0, 3, "10^") == 0) {
            // the reference reads this position with operator>> (weight.h:536-587), whose "10^x" branch ends up
            // taking the exponent as a real number; kept so the same files load to the same weights
            double x = std::strtod(t3.c_str() + 3, 0);
            w = x > 0 ? std::log(x) : kNegInf;
          } else if (!parse_weight_token(t3, w))
            c.fail("bad weight: " + t3);
          a.logw = w;
          if (!at_end()) c.fail("')' or '!' expected after weight");
        }
      }
    }
    c.skip_ws();
    if (c.peek() == '!') {
      ++c.p;
      c.skip_ws();
      if (std::isdigit((unsigned char)c.peek())) {
        char* e = 0;
        a.group = (uint32_t)std::strtoul(c.s.c_str() + c.p, &e, 10);
        c.p = (size_t)(e - c.s.c_str());
      } else
        a.group = kLocked;
    }
    return a;
  }
};

// ---- training corpus (train.cc:985-1025; train.h:134-189) ----
struct HostPairs {
  std::vector<uint64_t> in_off{0}, out_off{0};
  std::vector<uint32_t> in_sym, out_sym;
  std::vector<double> weight;
  size_t size() const { return weight.size(); }
};

inline void parse_corpus(Transducer& x, const std::string& text, HostPairs& c, std::string* warnings = 0,
                         bool weight_lines = true) {  // carmel -S: plain alternating lines, no weights (carmel.cc:1393-1410)
  size_t p = 0;
  auto next_line = [&](std::string& line) {
    if (p >= text.size()) return false;
    size_t e = text.find('\n', p);
    if (e == std::string::npos) e = text.size();
    line.assign(text, p, e - p);
    p = e + 1;
    return true;
  };
  std::string line;
  while (next_line(line)) {
    double wt = 1.0;
    char f = line.empty() ? '\0' : line[0];
    if (weight_lines && (std::isdigit((unsigned char)f) || f == '-' || f == '.' || f == 'e')) {  // a weight line
      char* e = 0;
      wt = std::strtod(line.c_str(), &e);
      if (e == line.c_str()) {
        if (warnings) *warnings += "Bad training example weight: " + line + "\n";
        continue;
      }
      if (!next_line(line)) break;
    }
    std::vector<uint32_t> ins, outs;
    x.symbols_of_line(line, false, ins);
    if (!next_line(line)) {
      if (!ins.empty() && warnings) *warnings += "Incomplete input/output training pair\n";
      break;
    }
    x.symbols_of_line(line, true, outs);
    c.in_sym.insert(c.in_sym.end(), ins.begin(), ins.end());
    c.out_sym.insert(c.out_sym.end(), outs.begin(), outs.end());
    c.in_off.push_back(c.in_sym.size());
    c.out_off.push_back(c.out_sym.size());
    c.weight.push_back(wt);
  }
}

}  // namespace carmel_host
