// ORACLE — TEST INFRASTRUCTURE ONLY (see lw.hpp header).
//
// wfst.hpp: restatement of carmel's WFST data model, text reader/writer, reduce() and normalize().
// Follows /root/reference/carmel/src/:
//   arc.h:28-81        FSTArc{in,out,dest,weight,groupId}; no_group=-1, locked_group=0
//   state.h:51-105     State{arcs list}; arc_adder appends in file order (:209-231); addArc prepends (:234-246)
//   fst.h:58-59,171,409 alphabets start as {*e*=0, *w*=1}; strhash.h:182-200 first-seen ids
//   wfstio.cc:92-150   getString tokenizer; :301-330 getStateIndex; :341-506 readLegible; :594-625 writeLegible
//   wfstio.cc:631-651  symbolList (corpus line -> symbol ids, new symbols get fresh ids)
//   fst.cc:468-524     reduce() (+ state.h:280-289 remove_epsilons_to); fst.cc:528-543 removeMarkedStates
//   fst.cc:86-244      normalize() with locked / tied groups; fst.h:1362-1446 NormGroupIter
//   mean_field_scale.hpp:40-52 (linear, or exp(digamma(x + alpha)); boost::math::digamma is a third-party dependency
//                              absent from the tree: restated below from its published definition, to full f64
//                              precision -- the reference asks boost for 8 digits, digamma.hpp:24-30)
#pragma once
#include "lw.hpp"
#include "refhash.hpp"
#include <vector>
#include <string>
#include <unordered_map>
#include <map>
#include <stdexcept>
#include <cctype>
#include <algorithm>
#include <sstream>

namespace oracle {

static const unsigned NO_GROUP = (unsigned)-1;
static const unsigned LOCKED_GROUP = 0;
static const unsigned EPS = 0;

struct Arc {
  unsigned in, out, dest;
  LW weight;
  unsigned group;
  Arc() : in(0), out(0), dest(0), group(NO_GROUP) {}
  Arc(unsigned i, unsigned o, unsigned d, LW w, unsigned g = NO_GROUP) : in(i), out(o), dest(d), weight(w), group(g) {}
  bool locked() const { return group == LOCKED_GROUP; }
  bool normal() const { return group == NO_GROUP; }
  bool tied() const { return !locked() && !normal(); }
};

struct Alphabet {  // strhash.h:182-200 — index_of adds unseen symbols at the end
  std::vector<std::string> names;
  std::unordered_map<std::string, unsigned> ids;
  Alphabet() {}
  void init_special() {
    names.clear();
    ids.clear();
    index_of("*e*");
    index_of("*w*");
  }
  unsigned index_of(const std::string& s) {
    auto it = ids.find(s);
    if (it != ids.end()) return it->second;
    unsigned id = (unsigned)names.size();
    names.push_back(s);
    ids.emplace(s, id);
    return id;
  }
  const unsigned* find(const std::string& s) const {
    auto it = ids.find(s);
    return it == ids.end() ? 0 : &it->second;
  }
  unsigned size() const { return (unsigned)names.size(); }
};

// character stream over an in-memory buffer with the few istream behaviours readLegible relies on
struct CharStream {
  const std::string& s;
  size_t p;
  bool failed;
  CharStream(const std::string& s) : s(s), p(0), failed(false) {}
  bool good() const { return !failed; }
  void skipws() {
    while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
  }
  bool get_skipws(char& c) {  // istr >> c
    if (failed) return false;
    skipws();
    if (p >= s.size()) {
      failed = true;
      return false;
    }
    c = s[p++];
    return true;
  }
  bool get(char& c) {  // istr.get(c)
    if (failed) return false;
    if (p >= s.size()) {
      failed = true;
      return false;
    }
    c = s[p++];
    return true;
  }
  void unget() {
    if (p > 0) --p;
  }
  bool read_double(double& d) {  // istr >> d (strtod-based; see oracle/README for the edge cases)
    if (failed) return false;
    skipws();
    const char* b = s.c_str() + p;
    char* e;
    d = std::strtod(b, &e);
    if (e == b) {
      failed = true;
      return false;
    }
    p += (size_t)(e - b);
    return true;
  }
  bool read_unsigned(unsigned& u) {
    if (failed) return false;
    skipws();
    const char* b = s.c_str() + p;
    char* e;
    unsigned long v = std::strtoul(b, &e, 10);
    if (e == b) {
      failed = true;
      return false;
    }
    u = (unsigned)v;
    p += (size_t)(e - b);
    return true;
  }
  bool eof() const { return p >= s.size(); }
};

// io.hpp skip_comment: while next non-ws char is the comment char, discard through end of line
inline void skip_comment(CharStream& in, char cc) {
  for (;;) {
    char c;
    size_t save = in.p;
    bool f = in.failed;
    if (!in.get_skipws(c)) {
      in.p = save;
      in.failed = f;  // peeking at eof must not poison the stream for the caller's own read
      return;
    }
    if (c == cc) {
      while (in.p < in.s.size() && in.s[in.p] != '\n') ++in.p;
    } else {
      in.unget();
      return;
    }
  }
}

// wfstio.cc:92-150 getString. Returns false where the reference returns 0.
inline bool get_string(CharStream& in, std::string& tok) {
  tok.clear();
  char c;
  if (!in.get_skipws(c)) return false;
  switch (c) {
    case '"': {
      tok.push_back(c);
      bool l = false;  // backslash was last char
      for (;;) {
        char d;
        if (!in.get(d)) return false;
        tok.push_back(d);
        if (d == '"' && !l) break;
        if (d == '\\')
          l = !l;
        else
          l = false;
      }
      return true;
    }
    case '*': {
      tok.push_back(c);
      for (;;) {
        char d;
        if (!in.get(d)) return false;
        if (d == '*') {
          tok.push_back(d);
          break;
        }
        tok.push_back((char)std::tolower((unsigned char)d));
      }
      return true;
    }
    case '(':
    case ')':
      return false;
    default: {
      tok.push_back(c);
      char d;
      while (in.get(d)) {
        if (d == '\n' || d == '\t' || d == ' ') break;
        if (d == '!' || d == ')') {
          in.unget();
          break;
        }
        tok.push_back(d);
      }
      // reference: hitting EOF inside a bare token leaves the stream failed, but the token is still returned
      if (!tok.empty() && tok.back() == '\r') tok.pop_back();
      return true;
    }
  }
}

// weight.h:536-587 logweight::read (used for `istr >> weight` in the 3-field arc form)
inline bool lw_read(CharStream& in, LW& w) {
  static const double ln10 = 2.30258509299404568402;
  char c;
  double f = 0;
  if (!in.get_skipws(c)) return false;
  if (c != 'e')
    in.unget();
  else {
    if (!in.get_skipws(c) || c != '^') return false;
    if (!in.read_double(f)) return false;
    w = LW::from_ln(f);
    return true;
  }
  if (!in.read_double(f)) {
    w = LW();
    return false;
  }
  if (f == 10) {
    if (in.get_skipws(c)) {
      if (c == '^') {
        if (!in.read_double(f)) return false;
        w = LW::from_ln(f * ln10);
        // QUIRK kept (weight.h:553-586): control falls out of this block into the suffix check below, which
        // ends in setReal(f) with f = the exponent, so "a b 10^-2" in the 3-field form reads as real(-2) = 0.
      } else {
        in.unget();
        w = LW::from_real(f);
        return true;
      }
    } else {
      w = LW::from_real(f);
      return true;
    }
  }
  if (in.eof()) {
    w = LW::from_real(f);
    return true;
  }
  if (!in.get(c)) {
    w = LW::from_real(f);
    return true;
  }
  if (c == 'l') {
    char n;
    if (!in.get(n)) return false;
    if (n == 'n')
      w = LW::from_ln(f);
    else {
      char g;
      if (n == 'o' && in.get(g) && g == 'g')
        w = LW::from_ln(f * ln10);
      else {
        w = LW();
        return false;
      }
    }
  } else {
    in.unget();
    w = LW::from_real(f);
  }
  return true;
}

enum NormGroupBy { NORM_CONDITIONAL = 0, NORM_JOINT = 1, NORM_NONE = 2 };  // fst.h norm_group_by
// psi(x), x > 0 (Abramowitz & Stegun 6.3.5, 6.3.18): psi(x) = psi(x + 1) - 1/x until x >= 20, then
// ln x - 1/(2x) - 1/(12 x^2) + 1/(120 x^4) - 1/(252 x^6) + 1/(240 x^8) - 1/(132 x^10)
inline double digamma(double x) {
  double r = 0;
  while (x < 20) {
    r -= 1 / x;
    x += 1;
  }
  double z = 1 / (x * x);
  return r + std::log(x) - 0.5 / x - z * (1. / 12 - z * (1. / 120 - z * (1. / 252 - z * (1. / 240 - z / 132))));
}
struct MeanFieldScale {  // mean_field_scale.hpp:24-52
  bool linear = true;
  double alpha = 0;
  LW operator()(LW x) const {
    if (linear) return x;
    double xa = x.getReal() + alpha;
    const double floor = .0002;
    if (xa < floor) return LW::from_ln(digamma(floor)) * LW::from_real(xa / floor);
    return LW::from_ln(digamma(xa));
  }
};
struct NormalizeMethod {
  int group;
  LW add_count;  // --priors
  MeanFieldScale scale;  // --digamma / -+ (carmel.cc:495, 1009-1013)
  NormalizeMethod() : group(NORM_CONDITIONAL) {}
};

struct Wfst {
  std::vector<std::vector<Arc> > states;  // states[s] = arcs in reference *list order*
  unsigned final_state;
  bool named_states;
  bool valid;
  Alphabet in_alph, out_alph;
  std::vector<std::string> state_names;
  std::unordered_map<std::string, unsigned> state_ids;

  Wfst() : final_state(0), named_states(false), valid(true) {
    in_alph.init_special();
    out_alph.init_special();
  }

  unsigned num_states() const { return (unsigned)states.size(); }
  size_t num_arcs() const {
    size_t n = 0;
    for (auto& s : states) n += s.size();
    return n;
  }

  unsigned get_state_index(const std::string& buf) {  // wfstio.cc:301-330
    if (!named_states) {
      char* e;
      unsigned long st = std::strtol(buf.c_str(), &e, 10);
      if (!buf.empty() && *e != '\0') return (unsigned)-1;
      if (st >= states.size()) states.resize(st + 1);
      return (unsigned)st;
    } else {
      auto it = state_ids.find(buf);
      unsigned st;
      if (it == state_ids.end()) {
        st = (unsigned)state_names.size();
        state_names.push_back(buf);
        state_ids.emplace(buf, st);
      } else
        st = it->second;
      if (st >= states.size()) states.resize(st + 1);
      return st;
    }
  }
  std::string state_name(unsigned i) const {
    if (named_states && i < state_names.size()) return state_names[i];
    return std::to_string(i);
  }

  // wfstio.cc:341-506
  bool read_legible(const std::string& text, bool always_named = false) {
    CharStream istr(text);
    std::string buf, buf2, final_name;
    named_states = true;
    unsigned state_number, dest_state, inL = 0, outL = 0;
    LW weight;
    char c;
#define ORC_REQUIRE(x) \
  do {                 \
    if (!(x)) goto INVALID; \
  } while (0)
#define ORC_GETC ORC_REQUIRE(istr.get_skipws(c))
#define ORC_PEEKC                       \
  do {                                  \
    ORC_REQUIRE(istr.get_skipws(c));    \
    istr.unget();                       \
  } while (0)
#define ORC_ENDIOW (c == ')' || c == '!')
    skip_comment(istr, '%');
    ORC_REQUIRE(get_string(istr, buf));
    final_name = buf;
    if (!always_named) {
      named_states = false;
      for (char ch : final_name)
        if (!std::isdigit((unsigned char)ch)) {
          named_states = true;
          break;
        }
    }
    if (!named_states) final_state = get_state_index(buf);
    while (istr.get_skipws(c)) {
      skip_comment(istr, '%');
      ORC_REQUIRE(c == '(');
      ORC_REQUIRE(get_string(istr, buf));
      state_number = get_state_index(buf);
      if (!~state_number) goto INVALID;
      for (;;) {
        ORC_GETC;
        bool destparen = (c == '(');
        if (!destparen) istr.unget();
        if (c == ')') break;
        ORC_REQUIRE(get_string(istr, buf));
        dest_state = get_state_index(buf);
        if (!~dest_state) goto INVALID;
        for (;;) {
          ORC_GETC;
          bool iowparen = (c == '(');
          if (!iowparen)
            istr.unget();
          else
            ORC_PEEKC;
          if (ORC_ENDIOW) {
            inL = outL = EPS;
            weight = LW::one();
          } else {
            ORC_REQUIRE(get_string(istr, buf));
            ORC_PEEKC;
            if (ORC_ENDIOW) {
              if (lw_set_string(weight, buf.c_str())) {
                inL = outL = EPS;
              } else {
                inL = in_alph.index_of(buf);
                outL = out_alph.index_of(buf);
                weight = LW::one();
              }
            } else {
              inL = in_alph.index_of(buf);
              ORC_REQUIRE(get_string(istr, buf2));
              ORC_PEEKC;
              if (ORC_ENDIOW) {
                if (lw_set_string(weight, buf2.c_str())) {
                  outL = out_alph.index_of(buf);
                } else {
                  outL = out_alph.index_of(buf2);
                  weight = LW::one();
                }
              } else {
                outL = out_alph.index_of(buf2);
                ORC_REQUIRE(lw_read(istr, weight));
                ORC_PEEKC;
                ORC_REQUIRE(ORC_ENDIOW);
              }
            }
          }
          Arc to_add(inL, outL, dest_state, weight);
          ORC_GETC;
          if (c == '!') {
            ORC_PEEKC;
            if (std::isdigit((unsigned char)c)) {
              unsigned group;
              ORC_REQUIRE(istr.read_unsigned(group));
              to_add.group = group;
            } else
              to_add.group = LOCKED_GROUP;
          } else
            istr.unget();
          states[state_number].push_back(to_add);  // state.h:209-231 arc_adder: file order
          if (!iowparen) break;
          ORC_REQUIRE(istr.get_skipws(c) && c == ')');
          ORC_PEEKC;
          if (c == ')') break;
        }
        if (!destparen) break;
        ORC_REQUIRE(istr.get_skipws(c) && c == ')');
      }
      ORC_REQUIRE(istr.get_skipws(c) && c == ')');
    }
    if (!named_states) {
      if (!(final_state < states.size())) goto INVALID;
      return true;
    }
    {
      auto it = state_ids.find(final_name);
      if (it != state_ids.end()) {
        final_state = it->second;
        return true;
      }
      goto INVALID;
    }
  INVALID:
    valid = false;
    return false;
#undef ORC_REQUIRE
#undef ORC_GETC
#undef ORC_PEEKC
#undef ORC_ENDIOW
  }

  // wfstio.cc:631-651 symbolList: tokens of one corpus line -> ids (unseen symbols are added)
  void symbol_list(std::vector<unsigned>& ret, const std::string& line, bool output) {
    CharStream in(line);
    Alphabet& a = output ? out_alph : in_alph;
    std::string tok;
    while (in.good()) {
      if (!get_string(in, tok)) break;
      ret.push_back(a.index_of(tok));
    }
  }

  // fst.cc:468-524 reduce(): drop states not on a start->final path (stable compaction,
  // fst.cc:528-543), then drop *e*:*e* self loops (state.h:280-289)
  void reduce() {
    unsigned n = num_states();
    if (!valid || n == 0) return;
    std::vector<std::vector<unsigned> > rev(n);
    for (unsigned s = 0; s < n; ++s)
      for (auto& a : states[s]) rev[a.dest].push_back(s);
    std::vector<char> fwd(n, 0), bwd(n, 0);
    std::vector<unsigned> stack;
    stack.push_back(0);
    fwd[0] = 1;
    while (!stack.empty()) {
      unsigned s = stack.back();
      stack.pop_back();
      for (auto& a : states[s])
        if (!fwd[a.dest]) {
          fwd[a.dest] = 1;
          stack.push_back(a.dest);
        }
    }
    stack.push_back(final_state);
    bwd[final_state] = 1;
    while (!stack.empty()) {
      unsigned s = stack.back();
      stack.pop_back();
      for (unsigned p : rev[s])
        if (!bwd[p]) {
          bwd[p] = 1;
          stack.push_back(p);
        }
    }
    std::vector<unsigned> old2new(n, (unsigned)-1);
    unsigned k = 0;
    for (unsigned s = 0; s < n; ++s)
      if (fwd[s] && bwd[s]) old2new[s] = k++;
    if (k != n) {
      std::vector<std::vector<Arc> > ns(k);
      std::vector<std::string> nn;
      for (unsigned s = 0; s < n; ++s)
        if (~old2new[s]) {
          for (auto& a : states[s])
            if (~old2new[a.dest]) {
              Arc b = a;
              b.dest = old2new[a.dest];
              ns[old2new[s]].push_back(b);
            }
          if (named_states && s < state_names.size()) nn.push_back(state_names[s]);
        }
      states.swap(ns);
      if (named_states) {
        state_names.swap(nn);
        state_ids.clear();
        for (unsigned i = 0; i < state_names.size(); ++i) state_ids[state_names[i]] = i;
      }
      if (~old2new[final_state])
        final_state = old2new[final_state];
      else {
        valid = false;
        states.clear();
        return;
      }
    }
    for (unsigned s = 0; s < num_states(); ++s) {
      auto& v = states[s];
      v.erase(std::remove_if(v.begin(), v.end(), [s](const Arc& a) { return a.in == 0 && a.out == 0 && a.dest == s; }),
              v.end());
    }
  }

  // Norm groups in NormGroupIter order (fst.h:1362-1446). JOINT: one group per state (arcs or not), arcs in list order.
  // CONDITIONAL: per state that has arcs, one group per input symbol, in the order the reference's walk over State::index
  // visits the symbols -- a HashTable<UnsignedKey, List<HalfArc>> made for `size` entries and filled arc by arc in list order
  // (state.h:158-199; refhash.hpp restates the table's bucket order) -- and within a group the arcs in REVERSED list order
  // (every arc is pushed onto the front of its symbol's list).  The order matters for group NUMBERING only: the norm ids of the
  // Gibbs sampler (gibbs.cc:114-186: which prior-scale draw moves which group) and the lines of --fem-norm (cascade.h:85-116).
  template <class F>
  void for_each_norm_group(int group, F f) {
    for (unsigned s = 0; s < num_states(); ++s) {
      auto& arcs = states[s];
      if (group == NORM_JOINT) {
        std::vector<Arc*> g;
        for (auto& a : arcs) g.push_back(&a);
        f(s, g);
      } else if (group == NORM_CONDITIONAL) {
        if (arcs.empty()) continue;
        RefHashKeys index((unsigned)arcs.size());
        std::unordered_map<unsigned, std::vector<Arc*> > by_in;
        for (auto& a : arcs) {
          index.insert(a.in);
          by_in[a.in].push_back(&a);
        }
        for (unsigned k : index.keys()) {
          std::vector<Arc*>& g = by_in[k];
          std::reverse(g.begin(), g.end());  // push_front lists
          f(s, g);
        }
      }
    }
  }

  // fst.cc:86-244
  void normalize(const NormalizeMethod& method, bool uniform_zero_normgroups = false) {
    int group = method.group;
    if (group == NORM_NONE) return;
    LW addc = method.add_count;
    const MeanFieldScale& scale = method.scale;
    std::unordered_map<unsigned, LW> groupArcTotal, groupStateTotal, groupMaxLockedSum;
    // pass 1 (fst.cc:117-152)
    for_each_norm_group(group, [&](unsigned, std::vector<Arc*>& g) {
      LW sum, locked_sum;
      for (Arc* a : g) {
        a->weight += addc;
        if (a->locked())
          locked_sum += a->weight;
        else
          sum += a->weight;
      }
      for (Arc* a : g)
        if (a->tied()) {
          unsigned pg = a->group;
          groupArcTotal[pg] += a->weight;
          groupStateTotal[pg] += sum;
          LW& m = groupMaxLockedSum[pg];
          if (locked_sum > m) m = locked_sum;
        }
    });
    // pass 2 (fst.cc:156-230)
    for_each_norm_group(group, [&](unsigned, std::vector<Arc*>& g) {
      LW normal_sum, reserved;
      for (Arc* a : g) {
        unsigned pg = a->group;
        if (a->tied()) {
          LW groupNorm = groupStateTotal[pg];
          LW gmax = groupMaxLockedSum[pg];
          LW one = LW::one();
          if (gmax > one) {
            a->weight = LW();
          } else {
            if (!gmax.isZero()) div_eq(groupNorm, one - gmax);
            LW groupTotal = groupArcTotal[pg];
            if (!groupTotal.isZero()) {
              a->weight = scale(groupTotal) / scale(groupNorm);  // fst.cc:189
              reserved += a->weight;
            } else
              a->weight = LW();
          }
        } else if (a->locked()) {
          reserved += a->weight;
        } else {
          normal_sum += a->weight;
        }
      }
      LW fraction_remain = LW::one();
      fraction_remain = fraction_remain - reserved;
      bool something_left = !fraction_remain.isZero();
      if (something_left && (uniform_zero_normgroups || !normal_sum.isZero())) {
        LW scaled_sum = scale(normal_sum);  // fst.cc:217-221
        for (Arc* a : g)
          if (a->normal()) a->weight = fraction_remain * scale(a->weight) / scaled_sum;
      } else
        for (Arc* a : g)
          if (a->normal()) a->weight = LW();
    });
  }

  // wfstio.cc:594-625 writeLegible (brief/full, state-per-line/arc-per-line); OUTARCWEIGHT :74-83
  std::string write_legible(bool full /*-J*/, bool onearc /*-H*/, int wmode = LW_SOMETIMES_LOG,
                            bool include_zero = false) const {
    std::ostringstream os;
    bool brief = !full;
    os << state_name(final_state);
    for (unsigned i = 0; i < num_states(); ++i) {
      if (!onearc) os << "\n(" << state_name(i);
      for (auto& a : states[i]) {
        if (include_zero || a.weight.isPositive()) {
          if (onearc) os << "\n(" << state_name(i);
          os << " (" << state_name(a.dest);
          if (!brief || a.in || a.out) {
            const std::string& il = in_alph.names[a.in];
            const std::string& ol = out_alph.names[a.out];
            os << " " << il;
            if (!brief || il != ol) os << " " << ol;
          }
          if (!brief || ~a.group || a.weight.w != 0.0) os << " " << lw_str(a.weight, wmode);
          if (~a.group) {
            os << '!';
            if (a.group > 0) os << a.group;
          }
          os << ")";
          if (onearc) os << ")";
        }
      }
      if (!onearc) os << ")";
    }
    os << "\n";
    return os.str();
  }
};

}  // namespace oracle
