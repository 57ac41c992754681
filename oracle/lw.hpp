// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is shipped, linked or called by the product
// path (carmel_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// lw.hpp: restatement of carmel's `logweight<double>` arithmetic.
// Follows /root/reference/graehl/shared/weight.h:
//   storage = natural log (weight.h:132-135), zero = -inf (:292), one = 0 (:294)
//   operator* / operator/  = add / subtract of logs (:737-738)
//   operator+ = max + log1p(exp(-|d|)), larger operand returned unchanged when |d| > 36 (:765-801,
//               MUCH_BIGGER_LN :103, GRAEHL_USE_LOG1P config.h:41, WEIGHT_CORRECT_ZERO config.h:149)
//   operator- clamped at zero (:803-830); absdiff (:837-856); pow/root (:431-447)
//   relative_perplexity_ratio (:247-249); ppxper (:311); setReal (:296-301)
#pragma once
#include <cmath>
#include <limits>
#include <string>
#include <cstdlib>
#include <cstring>
#include <cstdio>

namespace oracle {

static const double LW_MUCH_BIGGER_LN = 36.0;  // weight.h:103 (sizeof(Real)==8)
static const double LW_UNDERFLOW_LN = 82.0;    // weight.h:112

struct LW {
  double w;  // ln of the value
  LW() : w(-std::numeric_limits<double>::infinity()) {}  // weight.h:339 default = zero
  static LW from_ln(double l) {
    LW r;
    r.w = l;
    return r;
  }
  static LW from_real(double f) {  // weight.h:296-301 setReal
    LW r;
    if (f > 0) r.w = std::log(f);
    return r;
  }
  static LW one() { return from_ln(0); }
  static LW zero() { return LW(); }
  static LW inf() { return from_ln(std::numeric_limits<double>::infinity()); }
  bool isZero() const { return !(w > -std::numeric_limits<double>::infinity()); }  // weight.h:272-276
  bool isPositive() const { return w > -std::numeric_limits<double>::infinity(); }
  bool isOne() const { return w == 0; }
  bool isInfinity() const { return w == std::numeric_limits<double>::infinity(); }
  double getReal() const { return std::exp(w); }
  double getLn() const { return w; }
  bool fitsInReal() const { return isZero() || (w < LW_UNDERFLOW_LN && w > -LW_UNDERFLOW_LN); }  // :266-268
  LW pow(double n) const {  // weight.h:442-447 (WEIGHT_CORRECT_ZERO)
    if (isZero()) return *this;
    return from_ln(w * n);
  }
  LW root(double n) const {  // weight.h:435-440
    if (isZero()) return *this;
    return from_ln(w / n);
  }
  LW ppxper(double n = 1) const { return root(-n); }  // weight.h:311
};

inline LW operator*(LW a, LW b) { return LW::from_ln(a.w + b.w); }  // weight.h:737
inline LW operator/(LW a, LW b) { return LW::from_ln(a.w - b.w); }  // weight.h:738
inline LW& mul_eq(LW& a, LW b) {  // weight.h:389-395 operator*= (WEIGHT_CORRECT_ZERO)
  if (!a.isZero()) a.w += b.w;
  return a;
}
inline LW& div_eq(LW& a, LW b) {  // weight.h:396-407 operator/=
  if (!a.isZero()) a.w -= b.w;
  return a;
}

inline LW operator+(LW lhs, LW rhs) {  // weight.h:765-801
  if (lhs.isZero()) return rhs;
  if (rhs.isZero()) return lhs;
  double diff = lhs.w - rhs.w;
  if (diff > LW_MUCH_BIGGER_LN) return lhs;
  if (diff < -LW_MUCH_BIGGER_LN) return rhs;
  if (diff < 0) return LW::from_ln(rhs.w + log1p(std::exp(diff)));
  return LW::from_ln(lhs.w + log1p(std::exp(-diff)));
}
inline LW& operator+=(LW& a, LW b) {
  a = a + b;
  return a;
}

inline LW operator-(LW lhs, LW rhs) {  // weight.h:803-830
  if (rhs.isZero()) return lhs;
  LW result;
  double rdiff = rhs.w - lhs.w;
  if (rdiff >= 0) return result;  // clamp to zero
  if (rdiff < -LW_MUCH_BIGGER_LN) return lhs;
  result.w = lhs.w + log1p(-std::exp(rdiff));
  return result;
}

inline LW absdiff(LW a, LW b) {  // weight.h:851-855
  if (a.w > b.w) return a - b;
  return b - a;
}
inline bool operator<(LW a, LW b) { return a.w < b.w; }
inline bool operator>(LW a, LW b) { return a.w > b.w; }
inline bool operator<=(LW a, LW b) { return a.w <= b.w; }
inline bool operator>=(LW a, LW b) { return a.w >= b.w; }
inline bool operator==(LW a, LW b) { return a.w == b.w; }
inline bool operator!=(LW a, LW b) { return a.w != b.w; }

// weight.h:247-249 — (this/o).root(|ln this|)
inline LW relative_perplexity_ratio(LW self, LW o) { return (self / o).root(std::fabs(self.w)); }

// weight.h:503-528 setStringPartial: "e^x", "10^x", "<d>ln", "<d>log", "<d>"
// returns pointer one past the last char consumed, or null on error.
inline const char* lw_set_string_partial(LW& out, const char* b, const char* end) {
  char* e;
  static const double ln10 = 2.30258509299404568402;
  if (b + 1 < end && b[0] == 'e' && b[1] == '^') {
    out.w = std::strtod(b + 2, &e);
    return e;
  } else if (b + 2 < end && b[0] == '1' && b[1] == '0' && b[2] == '^') {
    out.w = std::strtod(b + 3, &e) * ln10;
    return e;
  } else {
    double d = std::strtod(b, &e);
    if (e[0] == 'l') {
      if (e[1] == 'n') {
        out.w = d;
        return e + 2;
      } else if (e[1] == 'o' && e[2] == 'g') {
        out.w = d * ln10;
        return e + 3;
      } else
        return 0;
    } else {
      out = LW::from_real(d);
      return e;
    }
  }
}
inline bool lw_set_string(LW& out, const char* s) {  // weight.h:492-495 setString: whole string must parse
  const char* end = s + std::strlen(s);
  LW tmp;
  const char* e = lw_set_string_partial(tmp, s, end);
  if (e == end) {  // quirk kept: the bare tokens "ln" / "log" parse as weight 1 (strtod eats nothing)
    out = tmp;
    return true;
  }
  return false;
}

// weight.h:468-489 print with precision 15; SOMETIMES_LOG: real if |ln|<82 else e^x (carmel default,
// wfstio.cc:47-50); always_log => e^x; never_log => real
// log base of the log form (weight.h:476-486): e^x, `x ln` (carmel -2), `x log` base 10 (carmel -B; getLog10 :265)
enum LwPrintMode { LW_SOMETIMES_LOG = 0, LW_ALWAYS_LOG = 1, LW_NEVER_LOG = 2, LW_MODE_MASK = 3, LW_BASE_LN = 16, LW_BASE_LOG10 = 32 };
inline std::string lw_str(LW x, int mode = LW_SOMETIMES_LOG, int precision = 15) {
  char buf[64];
  if (x.isZero()) return "0";
  const int m = mode & LW_MODE_MASK;
  if ((m == LW_SOMETIMES_LOG && x.fitsInReal()) || m == LW_NEVER_LOG) {
    std::snprintf(buf, sizeof buf, "%.*g", precision, x.getReal());
    return buf;
  }
  if (mode & LW_BASE_LN)
    std::snprintf(buf, sizeof buf, "%.*gln", precision, x.w);
  else if (mode & LW_BASE_LOG10)
    std::snprintf(buf, sizeof buf, "%.*glog", precision, (1. / 2.30258509299404568402) * x.w);
  else
    std::snprintf(buf, sizeof buf, "e^%.*g", precision, x.w);
  return buf;
}
// weight.h:529-532,603 as_base(2): "2^<log2>" at the stream's current precision (default 6)
inline std::string lw_base2(LW x, int precision = 6) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "2^%.*g", precision, x.w / std::log(2.0));
  return buf;
}

}  // namespace oracle
