// forest_text.hpp: host-side text formats of forest-em (no GPU code): packed AND/OR forests, normalisation groups and
// parameter vectors.  Written from the format the reference reads (forest-em/forest.hpp:925-1034 reader,
// forest-em/README, graehl/shared/normalize.hpp group reader); the grammar, as a recursive descent:
//
//   forest := node
//   node   := RULE                       a leaf AND node using rule RULE (1-based id)
//           | '(' RULE node* ')'         an AND node: rule weight times the product of its children
//           | '(' 'OR' node+ ')'         an OR node: sum over alternatives
//           | '#'K '(' ... ')'           the same as the parenthesised node, remembered as shared sub-forest K
//           | '#'K                       a back-reference to shared sub-forest K (defined earlier in this forest)
//
// Nodes are stored in preorder exactly as include/carmel_hip.h expects them: label (0 = OR), next (one past the
// subtree, relative to the forest), ref (>= 0 for a back-reference: index of the shared node).
#pragma once
#include <cctype>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "wfst.hpp"

namespace carmel_host {

struct ForestSet {
  std::vector<uint64_t> node_off{0};
  std::vector<uint32_t> label, next;
  std::vector<int32_t> ref;
  uint32_t max_rule = 0;
  uint64_t n_forests() const { return node_off.size() - 1; }
};

class ForestReader {
 public:
  ForestReader(const std::string& text, ForestSet& out) : s_(text), out_(out) {}
  // reads every forest in the text; returns how many
  uint64_t read_all() {
    uint64_t n = 0;
    for (;;) {
      ws();
      if (p_ >= s_.size()) break;
      base_ = out_.label.size();
      shared_.clear();
      node();
      out_.node_off.push_back(out_.label.size());
      ++n;
    }
    return n;
  }

 private:
  const std::string& s_;
  ForestSet& out_;
  size_t p_ = 0;
  uint64_t base_ = 0;
  std::vector<int32_t> shared_;  // shared sub-forest id -> node index in this forest

  [[noreturn]] void fail(const std::string& what) const {
    throw std::runtime_error("forest " + std::to_string(out_.n_forests() + 1) + ", character " + std::to_string(p_) + ": " + what);
  }
  void ws() {
    while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) ++p_;
  }
  uint32_t number() {
    if (p_ >= s_.size() || !std::isdigit((unsigned char)s_[p_])) fail("number expected");
    uint64_t v = 0;
    while (p_ < s_.size() && std::isdigit((unsigned char)s_[p_])) {
      v = v * 10 + (uint64_t)(s_[p_++] - '0');
      if (v > 0xfffffffeull) fail("number too large");
    }
    return (uint32_t)v;
  }
  uint32_t push(uint32_t label, int32_t ref) {
    const uint32_t idx = (uint32_t)(out_.label.size() - base_);
    out_.label.push_back(label);
    out_.ref.push_back(ref);
    out_.next.push_back(idx + 1);
    return idx;
  }
  void paren_node() {  // after '('
    ws();
    uint32_t idx;
    if (s_.compare(p_, 2, "OR") == 0) {
      p_ += 2;
      idx = push(0u, -1);
    } else {
      const uint32_t rule = number();
      if (rule == 0) fail("rule ids start at 1");
      if (rule > out_.max_rule) out_.max_rule = rule;
      idx = push(rule, -1);
    }
    for (;;) {
      ws();
      if (p_ >= s_.size()) fail("')' expected before the end of the input");
      if (s_[p_] == ')') {
        ++p_;
        break;
      }
      node();
    }
    out_.next[base_ + idx] = (uint32_t)(out_.label.size() - base_);
  }
  void node() {
    ws();
    if (p_ >= s_.size()) fail("node expected");
    const char c = s_[p_];
    if (c == '#') {
      ++p_;
      const uint32_t id = number();
      if (p_ < s_.size() && s_[p_] == '(') {
        ++p_;
        if (shared_.size() <= id) shared_.resize((size_t)id + 1, -1);
        shared_[id] = (int32_t)(out_.label.size() - base_);
        paren_node();
      } else {
        if (id >= shared_.size() || shared_[id] < 0) fail("back-reference #" + std::to_string(id) + " to an undefined sub-forest");
        push(0u, shared_[id]);
      }
    } else if (c == '(') {
      ++p_;
      paren_node();
    } else if (std::isdigit((unsigned char)c)) {
      const uint32_t rule = number();
      if (rule == 0) fail("rule ids start at 1");
      if (rule > out_.max_rule) out_.max_rule = rule;
      push(rule, -1);
    } else {
      fail(std::string("unexpected character '") + c + "'");
    }
  }
};

// normalisation groups "((1 2 7) (3 4 5 6))" -> CSR over rule ids
inline void read_normgroups(const std::string& s, std::vector<uint64_t>& group_off, std::vector<uint32_t>& group_rule,
                            uint32_t& max_rule) {
  group_off.assign(1, 0);
  group_rule.clear();
  int depth = 0;
  for (size_t p = 0; p < s.size();) {
    const char c = s[p];
    if (c == '(') {
      ++depth;
      ++p;
    } else if (c == ')') {
      if (depth == 2) group_off.push_back(group_rule.size());
      --depth;
      ++p;
      if (depth == 0) return;  // one list is read (`in >> norm_groups`, forest-em.hpp read_norm_groups): whatever follows it --
                               // forests, in forest-em/sample/norm_and_forests -- is not the groups' business
    } else if (std::isdigit((unsigned char)c)) {
      uint64_t v = 0;
      while (p < s.size() && std::isdigit((unsigned char)s[p])) v = v * 10 + (uint64_t)(s[p++] - '0');
      if (depth != 2) throw std::runtime_error("normalisation groups: rule id outside a group");
      group_rule.push_back((uint32_t)v);
      if (v > max_rule) max_rule = (uint32_t)v;
    } else if (std::isspace((unsigned char)c)) {
      ++p;
    } else {
      throw std::runtime_error(std::string("normalisation groups: unexpected character '") + c + "'");
    }
  }
  throw std::runtime_error(depth ? "normalisation groups: unbalanced parentheses"
                                : "Expected normalization groups list e.g. ((1 2 3) (4 5) (6))");
}

// parameter vector "(1 .5 e^-3 0)", or -- what carmel --fem-param writes -- the bare weights up to the end of the
// file; an optional comma may follow a weight (graehl/shared/io.hpp:557-603 range_read).  The first weight belongs
// to rule 1 (forest-em.hpp:228-250).
inline std::vector<double> read_params(const std::string& s) {
  std::vector<double> logw;
  size_t p = 0;
  while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
  if (p >= s.size()) throw std::runtime_error("expected a vector of weights, e.g. (1 .5 0)");
  const bool parens = s[p] == '(';
  if (parens) ++p;
  for (;;) {
    while (p < s.size() && (std::isspace((unsigned char)s[p]) || s[p] == ',')) ++p;
    if (p >= s.size()) {
      if (parens) throw std::runtime_error("vector of weights: ')' expected");
      break;
    }
    if (parens && s[p] == ')') break;
    size_t e = p;
    while (e < s.size() && !std::isspace((unsigned char)s[e]) && s[e] != ')' && s[e] != ',') ++e;
    double lw;
    if (!parse_weight_token(s.substr(p, e - p), lw)) throw std::runtime_error("bad weight '" + s.substr(p, e - p) + "'");
    logw.push_back(lw);
    p = e;
  }
  return logw;
}

// FForests::write_params / write_counts (forest-em.hpp:190-201): print_range(out, begin + 1, end, multiline = true,
// parens = false) -- a space, the weight and a newline per parameter (graehl/shared/io.hpp:327-343) -- then std::endl: one
// weight per line and an empty line at the end, the format of forest-em/sample/best_weights and of what -I reads back
inline std::string write_params(const double* logw, size_t n, int style) {  // logw[0] belongs to rule 1
  std::string out;
  for (size_t i = 0; i < n; ++i) {
    out += ' ';
    out += format_weight(logw[i], style);
    out += '\n';
  }
  out += '\n';
  return out;
}

}  // namespace carmel_host
