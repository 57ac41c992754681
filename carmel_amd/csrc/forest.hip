// forest.hip — forest-em's packed AND/OR derivation forests on the GPU: inside, normalised outside, expected rule
// counts, the EM M-step over normalisation groups, and the Gibbs sampler (inside with proposal probabilities +
// top-down choice).
//
// Replaces /root/reference/forest-em/forest.hpp (inside_rec :636-697, compute_norm_outside :439-491,
// visit_inside_norm_outside :417-438, choose_random :725-758, compute_inside(W) :768-816) and
// forest-em.hpp (estimate :561-578, maximize :626-655, Gibbs glue :694-766); graehl/shared/normalize.hpp:123-164.
//
// Layout: one forest per LANE, 64 forests per wavefront (forests are small: config 5 has ~50 nodes each).  A forest
// is flattened on the host into two record streams over its non-reference nodes renumbered in POST-ORDER (children
// and shared sub-forests before the nodes that use them; a back-reference simply resolves to the shared node):
//   inside stream : per node a header {AND?, rule id} followed by one record per child {child index}
//   outside stream: the same per node, nodes in reverse post-order
// and the 64 streams of a group are interleaved record by record so every wave-wide load is one 512-byte row.
// inside[] (and outside[]) live in the lane's own LDS column(s).  Expected counts reuse the two-phase scheme of the
// lattice path: one posterior per AND node into post[], then count_reduce_kernel with rules in the role of arcs.
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <unordered_map>
#include "engine.hpp"
#include "options.hpp"
#include "forest_exact.hpp"
#include "gibbs_exact.hpp"  // launch_gibbs_broadcast
#include "rng.hpp"

namespace carmel_hip {

#define F_NEG_INF (-__builtin_huge_val())
static const uint32_t F_HEADER = 0x80000000u, F_VALID = 0x40000000u, F_LAST = 0x20000000u, F_AND = 0x10000000u;
static const uint32_t F_IDX = 0x0fffffffu;
#define F_NONORM 0xffffffffu
#define FOREST_GHASH 2048u

struct FGroup {  // 32 bytes, one wavefront of forests
  uint64_t stream_base;
  uint32_t maxlen, n_lanes, lane_base, max_nodes;
  uint64_t node_base;  // of the group's rows in node-indexed arrays: 64 * (sum of max_nodes over the groups before it)
};

struct FAnd {  // one per AND header record of the inside streams, in stream order (forest_proposal_kernel)
  uint64_t pos;    // position in the stream arrays
  uint32_t group;  // lane group
  uint32_t cls;    // rec_cls of the record
  uint32_t rule;
  uint32_t forest;
};

struct ForestArgs {
  const FAnd* and_list;          // the AND header records (n_and of them)
  uint64_t n_and;
  int p_only;                    // this sweep needs the proposal probabilities only, not their logarithms (temperature 1)
  const FGroup* groups;
  const uint2* ins_stream;
  const uint2* out_stream;
  const uint32_t* lane_forest;   // forest id per lane slot
  const uint32_t* lane_nodes;    // non-reference nodes per lane slot
  const double* rule_logw;
  double* post;                  // one slot per outside-stream record (only AND headers are used)
  double* forest_logprob;        // per forest: ln inside[root]
  double* scalars;               // {sum ln p over non-zero forests, n non-zero, n zero}
  // Gibbs
  const uint32_t* p_norm;        // per rule: norm group or F_NONORM
  const double* p_prior;
  const double* snap_x;          // counts / normsums the proposal is computed from
  const double* snap_norm;
  const uint32_t* hdr_pos;       // per (node, lane): position of the node's header in the inside stream
  const uint64_t* sample_off;    // per forest
  uint32_t* sample_len;
  uint32_t* sample_rules;
  const uint32_t* old_len;       // previous sample (counterfactual removal); may alias sample_* of the other buffer
  const uint32_t* old_rules;
  double* iter_out;
  // parallel sweep, second formulation (forest_proposal / forest_sample / forest_recount kernels)
  const uint32_t* rec_cls;       // per inside-stream record (AND headers): class of its rule | class of its norm group << 16,
                                 // both dense within the forest
  double* rec_logp;              // per inside-stream record (AND headers): ln proposal probability of the rule
  double* rec_p;                 //                                          the probability itself
  uint32_t* sample_cls;          // per sample entry: the rec_cls word of its record (0xffffffff: a rule outside every norm
                                 // group), written by the recount, scanned by the next sweep's proposal kernel
  uint32_t* sample_hdr;          // per sample entry: stream position of the AND header it came from
  const uint32_t* lane_of_forest;
  double* gcol;                  // forests too large for LDS: the inside (/ outside) columns of a group in global memory,
  uint64_t gcol_stride;          //   gcol + workgroup * gcol_stride (doubles)
  uint32_t* ghash;               // FOREST_GHASH slots per forest: own-sample table of lanes that overflow LDS (may be null)
  unsigned long long* trace;     // experiment (CARMEL_HIP_FOREST_TRACE): per block {start, after table, after inside, after walk, end}
  uint64_t seed;
  double power;                  // 1 / temperature of this sweep (annealing)
  uint32_t iter, first_group, serial_forest;  // serial_forest: exact mode processes exactly this forest (lane slot)
  int counterfactual;
};

__device__ __forceinline__ double f_lwadd(double a, double b) {
  if (a == F_NEG_INF) return b;
  if (b == F_NEG_INF) return a;
  double d = a - b;
  if (d > 36.0) return a;
  if (d < -36.0) return b;
  if (d < 0) return b + log1p(exp(d));
  return a + log1p(exp(-d));
}

// inside over the lane's stream with the rule weights.  The records run three chunks of four ahead of the fold and an
// AND header's weight -- a gather that depends on its record -- one chunk ahead: a wave is alone on its SIMD most of the
// time (two columns of LDS per forest bound the occupancy), nobody else hides the two round trips.
#define FE_CHUNK 4
__device__ __forceinline__ void f_inside(const ForestArgs& A, const FGroup& g, int lane, double* col) {
  const uint2* __restrict__ st = A.ins_stream + g.stream_base + lane;
  const double* __restrict__ lw = A.rule_logw;
  const uint32_t last = g.maxlen - 1;
  uint32_t d = 0;
  bool is_and = false;
  double acc = 0.0, m = F_NEG_INF, sum = 0.0;
  uint2 r[FE_CHUNK], r1[FE_CHUNK], r2[FE_CHUNK];
  double w[FE_CHUNK], w1[FE_CHUNK];
#define FE_LOAD(R, base) \
  _Pragma("unroll") for (int j = 0; j < FE_CHUNK; ++j) R[j] = st[(size_t)min((base) + j, last) * 64];
  // (only an AND header's second word is a rule id: every other record reads entry 0 and ignores it)
#define FE_GATHER(W, R)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < FE_CHUNK; ++j)                                                           \
      W[j] = lw[(R[j].x & (F_VALID | F_HEADER | F_AND)) == (F_VALID | F_HEADER | F_AND) ? R[j].y : 0u];
  FE_LOAD(r, 0u)
  FE_LOAD(r1, (uint32_t)FE_CHUNK)
  FE_GATHER(w, r)
  for (uint32_t k0 = 0; k0 < g.maxlen; k0 += FE_CHUNK) {
    if (k0) {
#pragma unroll
      for (int j = 0; j < FE_CHUNK; ++j) {
        r[j] = r1[j];
        r1[j] = r2[j];
        w[j] = w1[j];
      }
    }
    FE_LOAD(r2, k0 + 2u * FE_CHUNK)
    FE_GATHER(w1, r1)
#pragma unroll
    for (int j = 0; j < FE_CHUNK; ++j) {
      if (k0 + j > last || !(r[j].x & F_VALID)) continue;
      if (r[j].x & F_HEADER) {
        is_and = (r[j].x & F_AND) != 0;
        if (is_and)
          acc = w[j];
        else {
          m = F_NEG_INF;
          sum = 0.0;
        }
      } else {
        const double v = col[(size_t)(r[j].x & F_IDX) * 64];
        if (is_and)
          acc += v;
        else if (v != F_NEG_INF) {  // streaming logsumexp over the OR's children
          if (v <= m)
            sum += exp(v - m);
          else {
            sum = (m == F_NEG_INF) ? 1.0 : sum * exp(m - v) + 1.0;
            m = v;
          }
        }
      }
      if (r[j].x & F_LAST) {
        col[(size_t)d * 64] = is_and ? acc : (sum == 1.0 ? m : (sum > 0.0 ? m + log(sum) : F_NEG_INF));
        ++d;
      }
    }
  }
#undef FE_LOAD
#undef FE_GATHER
}

// EM E-step: inside, normalised outside, posteriors of AND nodes.  LDS: two columns per lane (inside, outside).
// GCOL: the columns live in global memory (A.gcol) -- forests with more nodes than LDS holds, e.g. the derivation
// lattices carmel --fem-forest exports (tens of thousands of nodes each); still one lane per forest.
template <bool GCOL>
__global__ __launch_bounds__(64) void forest_estimate_kernel(ForestArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const FGroup g = A.groups[A.first_group + blockIdx.x];
  const int lane = threadIdx.x;
  const bool active = (uint32_t)lane < g.n_lanes;
  const uint32_t n = active ? A.lane_nodes[g.lane_base + lane] : 0u;
  double* colbase = GCOL ? A.gcol + (size_t)blockIdx.x * A.gcol_stride : lds;
  double* ins = colbase + lane;
  double* out = colbase + (size_t)g.max_nodes * 64 + lane;
  f_inside(A, g, lane, ins);
  double lp = F_NEG_INF;
  if (active) {
    lp = ins[(size_t)(n - 1) * 64];
    A.forest_logprob[A.lane_forest[g.lane_base + lane]] = lp;
    for (uint32_t s = 0; s < n; ++s) out[(size_t)s * 64] = F_NEG_INF;
    if (lp != F_NEG_INF) out[(size_t)(n - 1) * 64] = -lp;  // norm_outside[root] = 1 / inside[root]
  }
  const uint2* __restrict__ st = A.out_stream + g.stream_base + lane;
  double* __restrict__ post = A.post + g.stream_base + lane;
  bool is_and = false;
  double op = F_NEG_INF, ip = F_NEG_INF;
  const uint32_t last = g.maxlen - 1;
  uint2 rr[FE_CHUNK], rr1[FE_CHUNK], rr2[FE_CHUNK];  // three chunks of records in flight, as in the inside pass
#define FE_LOAD(R, base) \
  _Pragma("unroll") for (int j = 0; j < FE_CHUNK; ++j) R[j] = st[(size_t)min((base) + j, last) * 64];
  FE_LOAD(rr, 0u)
  FE_LOAD(rr1, (uint32_t)FE_CHUNK)
  for (uint32_t k0 = 0; k0 < g.maxlen; k0 += FE_CHUNK) {
    if (k0) {
#pragma unroll
      for (int j = 0; j < FE_CHUNK; ++j) {
        rr[j] = rr1[j];
        rr1[j] = rr2[j];
      }
    }
    FE_LOAD(rr2, k0 + 2u * FE_CHUNK)
#pragma unroll
   for (int j = 0; j < FE_CHUNK; ++j) {
    const uint32_t k = k0 + j;
    const uint2 r = rr[j];
    if (k > last || !(r.x & F_VALID)) continue;
    if (r.x & F_HEADER) {
      const uint32_t p = r.x & F_IDX;
      is_and = (r.x & F_AND) != 0;
      op = out[(size_t)p * 64];
      ip = ins[(size_t)p * 64];
      if (is_and) post[(size_t)k * 64] = (lp != F_NEG_INF && op != F_NEG_INF) ? exp(ip + op) : 0.0;
    } else if (lp != F_NEG_INF && op != F_NEG_INF) {
      const uint32_t c = r.x & F_IDX;
      double contrib = op;
      if (is_and) {
        if (ip == F_NEG_INF) continue;  // 0/0 guard of forest.hpp:470
        contrib = op + ip - ins[(size_t)c * 64];
      }
      out[(size_t)c * 64] = f_lwadd(out[(size_t)c * 64], contrib);
    }
   }
  }
#undef FE_LOAD
  double s_lp = (active && lp != F_NEG_INF) ? lp : 0.0, s_n = (active && lp != F_NEG_INF) ? 1.0 : 0.0,
         s_z = (active && lp == F_NEG_INF) ? 1.0 : 0.0;
  for (int o = 32; o > 0; o >>= 1) {
    s_lp += __shfl_down(s_lp, o, 64);
    s_n += __shfl_down(s_n, o, 64);
    s_z += __shfl_down(s_z, o, 64);
  }
  if (lane == 0) {
    unsafeAtomicAdd(A.scalars + 0, s_lp);
    unsafeAtomicAdd(A.scalars + 1, s_n);
    unsafeAtomicAdd(A.scalars + 2, s_z);
  }
}

// The same E-step with inside / outside values as mantissa x 2^exponent pairs (as forest_sample_kernel<.., EXT>): a product is
// a multiply and an integer add, the OR fold / the outside accumulation an aligned add, an AND child's share a divide --
// where the log domain spends an exp and a log1p per child.  12 bytes per value instead of 8 (columns: inside mantissas,
// outside mantissas, then the two exponent columns); differences to the log-domain kernel are rounding (1e-16 relative).
struct FExt {
  double m;  // 0 (the value zero) or in [0.5, 1)
  int e;
};
__device__ __forceinline__ FExt fx_norm(double v, int e) {
  int t;
  FExt r;
  r.m = frexp(v, &t);
  r.e = v == 0.0 ? 0 : e + t;
  return r;
}
__device__ __forceinline__ FExt fx_add(FExt a, FExt b) {  // a + b, either may be zero
  if (a.m == 0.0) return b;
  if (b.m == 0.0) return a;
  const int dd = b.e - a.e;
  return dd <= 0 ? fx_norm(a.m + ldexp(b.m, dd), a.e) : fx_norm(ldexp(a.m, -dd) + b.m, b.e);
}
// a rule's weight from its natural log without leaving the range of a double: 2^(w log2 e) = 2^k x 2^f
__device__ __forceinline__ FExt fx_from_ln(double lnw) {
  FExt r;
  if (lnw == F_NEG_INF) {
    r.m = 0.0;
    r.e = 0;
    return r;
  }
  const double l2 = lnw * 1.4426950408889634074, k = floor(l2);
  return fx_norm(exp2(l2 - k), (int)k);
}
__global__ __launch_bounds__(64) void forest_estimate_ext_kernel(ForestArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const FGroup g = A.groups[A.first_group + blockIdx.x];
  const int lane = threadIdx.x;
  const bool active = (uint32_t)lane < g.n_lanes;
  const uint32_t n = active ? A.lane_nodes[g.lane_base + lane] : 0u;
  const uint32_t rows = g.max_nodes;
  double* im = lds + lane;
  double* om = lds + (size_t)rows * 64 + lane;
  int* ie = (int*)(lds + (size_t)2 * rows * 64) + lane;
  int* oe = ie + (size_t)rows * 64;
  const uint32_t last = g.maxlen - 1;
  {  // inside (f_inside's pipeline: records three chunks ahead, AND headers' weights one chunk ahead)
    const uint2* __restrict__ st = A.ins_stream + g.stream_base + lane;
    const double* __restrict__ lw = A.rule_logw;
    uint32_t d = 0;
    bool is_and = false;
    FExt acc = {0.0, 0}, sum = {0.0, 0};
    uint2 r[FE_CHUNK], r1[FE_CHUNK], r2[FE_CHUNK];
    double w[FE_CHUNK], w1[FE_CHUNK];
#define FE_LOAD(R, base) \
  _Pragma("unroll") for (int j = 0; j < FE_CHUNK; ++j) R[j] = st[(size_t)min((base) + j, last) * 64];
#define FE_GATHER(W, R)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < FE_CHUNK; ++j)                                                           \
      W[j] = lw[(R[j].x & (F_VALID | F_HEADER | F_AND)) == (F_VALID | F_HEADER | F_AND) ? R[j].y : 0u];
    FE_LOAD(r, 0u)
    FE_LOAD(r1, (uint32_t)FE_CHUNK)
    FE_GATHER(w, r)
    for (uint32_t k0 = 0; k0 < g.maxlen; k0 += FE_CHUNK) {
      if (k0) {
#pragma unroll
        for (int j = 0; j < FE_CHUNK; ++j) {
          r[j] = r1[j];
          r1[j] = r2[j];
          w[j] = w1[j];
        }
      }
      FE_LOAD(r2, k0 + 2u * FE_CHUNK)
      FE_GATHER(w1, r1)
#pragma unroll
      for (int j = 0; j < FE_CHUNK; ++j) {
        // without branches, as the sampler's fold: every candidate is computed, selects keep the one that applies, the
        // node's running value is stored at every record (the last store stays)
        const uint32_t rx = r[j].x;
        if (k0 + j <= last && (rx & F_VALID)) {
          const bool hdr = (rx & F_HEADER) != 0, child = !hdr;
          const size_t c = (size_t)min(rx & F_IDX, rows - 1) * 64;  // (a header's low bits are not a row: read and ignored)
          const double vm = im[c];
          const int ve = ie[c];
          const FExt h = fx_from_ln(w[j]);
          int pe, se;
          const double pm = frexp(acc.m * vm, &pe);
          const int dd = ve - sum.e;
          const bool le = dd <= 0;
          const double lo = le ? vm : sum.m, hi = le ? sum.m : vm;
          const double sm = frexp(hi + ldexp(lo, le ? dd : -dd), &se);
          const bool fold_and = child && is_and, fold_or = child && !is_and && vm != 0.0, sum0 = sum.m == 0.0;
          is_and = hdr ? (rx & F_AND) != 0 : is_and;
          const bool take_h = hdr && is_and;
          acc.e = take_h ? h.e : fold_and ? (acc.m * vm == 0.0 ? 0 : acc.e + ve + pe) : acc.e;
          acc.m = take_h ? h.m : fold_and ? pm : acc.m;
          const int nsum_e = sum0 ? ve : (le ? sum.e : ve) + se;
          const double nsum = sum0 ? vm : sm;
          sum.e = hdr ? 0 : fold_or ? nsum_e : sum.e;
          sum.m = hdr ? 0.0 : fold_or ? nsum : sum.m;
          im[(size_t)d * 64] = is_and ? acc.m : sum.m;
          ie[(size_t)d * 64] = is_and ? acc.e : sum.e;
          d += (rx & F_LAST) ? 1u : 0u;
        }
      }
    }
#undef FE_LOAD
#undef FE_GATHER
  }
  double lp = F_NEG_INF;
  if (active) {
    const double rm = im[(size_t)(n - 1) * 64];
    const int re = ie[(size_t)(n - 1) * 64];
    if (rm != 0.0) lp = log(rm) + (double)re * 0.69314718055994530942;
    A.forest_logprob[A.lane_forest[g.lane_base + lane]] = lp;
    for (uint32_t q = 0; q < n; ++q) {
      om[(size_t)q * 64] = 0.0;
      oe[(size_t)q * 64] = 0;
    }
    if (rm != 0.0) {  // norm_outside[root] = 1 / inside[root]
      const FExt o = fx_norm(1.0 / rm, -re);
      om[(size_t)(n - 1) * 64] = o.m;
      oe[(size_t)(n - 1) * 64] = o.e;
    }
  }
  const uint2* __restrict__ st = A.out_stream + g.stream_base + lane;
  double* __restrict__ post = A.post + g.stream_base + lane;
  bool is_and = false;
  FExt op = {0.0, 0}, ip = {0.0, 0};
  uint2 rr[FE_CHUNK], rr1[FE_CHUNK], rr2[FE_CHUNK];
#define FE_LOAD(R, base) \
  _Pragma("unroll") for (int j = 0; j < FE_CHUNK; ++j) R[j] = st[(size_t)min((base) + j, last) * 64];
  FE_LOAD(rr, 0u)
  FE_LOAD(rr1, (uint32_t)FE_CHUNK)
  for (uint32_t k0 = 0; k0 < g.maxlen; k0 += FE_CHUNK) {
    if (k0) {
#pragma unroll
      for (int j = 0; j < FE_CHUNK; ++j) {
        rr[j] = rr1[j];
        rr1[j] = rr2[j];
      }
    }
    FE_LOAD(rr2, k0 + 2u * FE_CHUNK)
#pragma unroll
    for (int j = 0; j < FE_CHUNK; ++j) {
      const uint32_t k = k0 + j;
      const uint2 r = rr[j];
      if (k > last || !(r.x & F_VALID)) continue;
      // one straight line for headers and children alike (the lanes of a wave are at both): the row's four values are
      // read, a header keeps them as its node's, a child adds its share to them and stores them back
      const bool hdr = (r.x & F_HEADER) != 0;
      const size_t c = (size_t)min(r.x & F_IDX, rows - 1) * 64;
      const double cm_o = om[c], cm_i = im[c];
      const int ce_o = oe[c], ce_i = ie[c];
      if (hdr) {
        is_and = (r.x & F_AND) != 0;
        op.m = cm_o;
        op.e = ce_o;
        ip.m = cm_i;
        ip.e = ce_i;
        if (is_and) post[(size_t)k * 64] = (lp != F_NEG_INF && op.m != 0.0) ? ldexp(ip.m * op.m, ip.e + op.e) : 0.0;
      }
      // a child's share: the parent's outside value (OR), or outside x inside of the parent / inside of the child (AND;
      // 0/0 guard of forest.hpp:470: an AND parent of inside zero passes nothing on)
      const bool share = !hdr && lp != F_NEG_INF && op.m != 0.0 && !(is_and && ip.m == 0.0);
      int qe, se;
      const double qm = frexp(is_and ? op.m * ip.m / cm_i : op.m, &qe);
      const int q_e = (is_and ? op.e + ip.e - ce_i : op.e) + qe;
      const int dd = q_e - ce_o;
      const bool le = dd <= 0;
      const double lo = le ? qm : cm_o, hi = le ? cm_o : qm;
      const double sm = frexp(hi + ldexp(lo, le ? dd : -dd), &se);
      const bool cur0 = cm_o == 0.0;
      if (share) {
        om[c] = cur0 ? qm : sm;
        oe[c] = cur0 ? q_e : (le ? ce_o : q_e) + se;
      }
    }
  }
#undef FE_LOAD
  double s_lp = (active && lp != F_NEG_INF) ? lp : 0.0, s_n = (active && lp != F_NEG_INF) ? 1.0 : 0.0,
         s_z = (active && lp == F_NEG_INF) ? 1.0 : 0.0;
  for (int o = 32; o > 0; o >>= 1) {
    s_lp += __shfl_down(s_lp, o, 64);
    s_n += __shfl_down(s_n, o, 64);
    s_z += __shfl_down(s_z, o, 64);
  }
  if (lane == 0) {
    unsafeAtomicAdd(A.scalars + 0, s_lp);
    unsafeAtomicAdd(A.scalars + 1, s_n);
    unsafeAtomicAdd(A.scalars + 2, s_z);
  }
}

// Gibbs: resample every forest of the group (or, exact mode, the single forest A.serial_forest) against snap_x /
// snap_norm.  LDS per lane: the inside column.
template <bool GCOL>
__global__ __launch_bounds__(64) void forest_gibbs_kernel(ForestArgs A, uint32_t max_sample, uint32_t ins_rows,
                                                           uint32_t own_cap, uint32_t stack_lds) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  double* colbase = GCOL ? A.gcol + (size_t)blockIdx.x * A.gcol_stride : lds_all;
  double* aux = GCOL ? lds_all : lds_all + (size_t)ins_rows * 64;  // tables and stack follow the column when it is in LDS
  const FGroup g = A.groups[A.first_group + blockIdx.x];
  const int lane = threadIdx.x;
  bool active = (uint32_t)lane < g.n_lanes;
  if (A.serial_forest != 0xffffffffu) active = active && (g.lane_base + lane == A.serial_forest);
  const uint32_t n = active ? A.lane_nodes[g.lane_base + lane] : 0u;
  double* ins = colbase + lane;
  const uint32_t forest = active ? A.lane_forest[g.lane_base + lane] : 0u;
  // the lane's previous sample (counterfactual removal) is read straight from global memory; the traversal stack
  // lives at the tail of the forest's own sample buffer (recorded rules grow from the front, pending nodes from the
  // back: every pending node still owes at least one rule, so the two never meet)
  unsigned long long tr0 = A.trace ? __builtin_readcyclecounter() : 0, tr1 = 0, tr2 = 0, tr3 = 0;
  uint32_t own_len = 0;
  const uint32_t* own = A.old_rules + (active ? A.sample_off[forest] : 0);
  if (active && A.counterfactual) own_len = A.old_len[forest];
  // ... counted into a small open-addressing table in the lane's LDS column ({rule or norm-group id, uses}): every
  // proposal probability needs "how often does my previous sample use this rule / this group" (counterfactual CRP
  // counts), and scanning the sample for each AND node (two dependent global gathers per scanned rule) made this
  // kernel 15x slower than the estimate kernel on the same forests.  own_cap = table slots (a power of two, 0 = scan).
  // a lane whose sample is too long for its LDS column uses a table in global memory instead (FOREST_GHASH slots
  // per forest): the rare long derivation must not fall back to scanning -- one such lane held its wave 30x longer.
  // The two tables are handled by separate code (LDS / global address spaces), never through one generic pointer.
  uint32_t* ht_l = (uint32_t*)aux + lane;  // stride 64
  uint32_t* ht_g = A.ghash ? A.ghash + (size_t)forest * FOREST_GHASH : nullptr;  // stride 1
  const bool hashed_l = own_cap != 0 && own_len * 20 <= own_cap * 9;  // <= 2 keys per rule, load factor <= 0.9
  const bool hashed_g = !hashed_l && active && ht_g != nullptr && own_len * 20 <= FOREST_GHASH * 9;
  uint32_t gcap = 64;  // slots of the global table this lane uses: the smallest power of two with load factor <= 0.9
  while (gcap * 9 < own_len * 20) gcap <<= 1;
  const bool hashed = hashed_l || hashed_g;
#define FH_SLOT(key, mask) ((((key) * 2654435761u) >> 7) & (mask))
#define FH_ADD(tab, stride, mask, key_)                              \
  {                                                                  \
    const uint32_t key = (key_);                                     \
    for (uint32_t h = FH_SLOT(key, mask);; h = (h + 1) & (mask)) {   \
      const uint32_t cur = (tab)[(size_t)h * (stride)];              \
      if (cur == 0xffffffffu) {                                      \
        (tab)[(size_t)h * (stride)] = (key << 8) | 1u;               \
        break;                                                       \
      }                                                              \
      if ((cur >> 8) == key) {                                       \
        (tab)[(size_t)h * (stride)] = cur + 1u;                      \
        break;                                                       \
      }                                                              \
    }                                                                \
  }
  if (hashed_l) {
    const uint32_t mask = own_cap - 1;
    for (uint32_t i = 0; i < own_cap; ++i) ht_l[(size_t)i * 64] = 0xffffffffu;
    for (uint32_t q = 0; q < own_len; ++q) {
      const uint32_t rr = own[q], nr = A.p_norm[rr];
      FH_ADD(ht_l, 64, mask, rr)
      if (nr != F_NONORM) FH_ADD(ht_l, 64, mask, nr | 0x800000u)
    }
  } else if (hashed_g) {
    const uint32_t mask = gcap - 1;
    for (uint32_t i = 0; i < gcap; ++i) ht_g[i] = 0xffffffffu;
    for (uint32_t q = 0; q < own_len; ++q) {
      const uint32_t rr = own[q], nr = A.p_norm[rr];
      FH_ADD(ht_g, 1, mask, rr)
      if (nr != F_NONORM) FH_ADD(ht_g, 1, mask, nr | 0x800000u)
    }
  }
#undef FH_ADD
  auto own_uses = [&](uint32_t key) -> double {  // uses of a rule (key = id) or of a norm group (key = id | 1 << 23)
    if (hashed_l) {
      const uint32_t mask = own_cap - 1;
      for (uint32_t h = FH_SLOT(key, mask);; h = (h + 1) & mask) {
        const uint32_t cur = ht_l[(size_t)h * 64];
        if (cur == 0xffffffffu) return 0.0;
        if ((cur >> 8) == key) return (double)(cur & 0xffu);
      }
    } else {
      const uint32_t mask = gcap - 1;
      for (uint32_t h = FH_SLOT(key, mask);; h = (h + 1) & mask) {
        const uint32_t cur = ht_g[h];
        if (cur == 0xffffffffu) return 0.0;
        if ((cur >> 8) == key) return (double)(cur & 0xffu);
      }
    }
  };
  if (A.trace) tr1 = __builtin_readcyclecounter();
  // inside with proposal probabilities (forest.hpp:768-816)
  {
    const uint2* __restrict__ st = A.ins_stream + g.stream_base + lane;
    uint32_t d = 0;
    bool is_and = false;
    double acc = 0.0, sum = F_NEG_INF;
    for (uint32_t k = 0; k < g.maxlen; ++k) {
      const uint2 r = st[(size_t)k * 64];
      if (!active || !(r.x & F_VALID)) continue;
      if (r.x & F_HEADER) {
        is_and = (r.x & F_AND) != 0;
        if (is_and) {
          const uint32_t rule = r.y, nn = A.p_norm[rule];
          double pr;
          if (nn == F_NONORM)
            pr = A.p_prior[rule];
          else {
            double x = A.snap_x[rule], ns = A.snap_norm[nn];
            if (hashed) {
              x -= own_uses(rule);
              ns -= own_uses(nn | 0x800000u);
            } else {
              for (uint32_t q = 0; q < own_len; ++q) {
                const uint32_t rr = own[q];
                if (rr == rule) x -= 1.0;
                if (A.p_norm[rr] == nn) ns -= 1.0;
              }
            }
            pr = x / ns;
          }
          acc = log(pr);
        } else
          sum = F_NEG_INF;
      } else {
        const double v = ins[(size_t)(r.x & F_IDX) * 64];
        if (is_and)
          acc += v;
        else
          sum = f_lwadd(sum, v);  // the reference's pairwise OR fold (forest.hpp:790-797)
      }
      if (r.x & F_LAST) {
        ins[(size_t)d * 64] = is_and ? acc : sum;
        ++d;
      }
    }
  }
  if (A.trace) tr2 = __builtin_readcyclecounter();
  // top-down choice (forest.hpp:725-758) with an explicit stack; children are pushed in reverse so they pop in order
  double cheap = 0.0;
  if (active) {
    const uint2* __restrict__ st = A.ins_stream + g.stream_base + lane;
    const uint32_t* __restrict__ hp = A.hdr_pos + g.stream_base + lane;  // indexed [node * 64]
    uint32_t* outr = A.sample_rules + A.sample_off[forest];
    const uint32_t cap = (uint32_t)(A.sample_off[forest + 1] - A.sample_off[forest]);
    uint32_t* stack = outr + cap;  // deep part of the stack: stack[-1 - i]
    uint32_t* stk_sh = (uint32_t*)aux + (size_t)own_cap * 64 + lane;  // first stack_lds entries
#define FSTACK_PUSH(v)                                       \
  {                                                          \
    if (sp < stack_lds)                                      \
      stk_sh[(size_t)sp * 64] = (v);                         \
    else                                                     \
      stack[-(int)(sp - stack_lds) - 1] = (v);               \
    ++sp;                                                    \
  }
    uint32_t sp = 0, ns = 0, step = 0;
    FSTACK_PUSH(n - 1)
    while (sp) {
      --sp;
      // bit 31 of a stack entry: the node was reached through a back-reference, below which the reference chooses at
      // temperature 1 (forest.hpp:731-732 drops `power`)
      const uint32_t entry = sp < stack_lds ? stk_sh[(size_t)sp * 64] : stack[-(int)(sp - stack_lds) - 1];
      const uint32_t node = entry & F_IDX, cold = entry & 0x80000000u;
      const uint32_t h = hp[(size_t)node * 64];
      const uint2 hr = st[(size_t)h * 64];
      // children occupy records h+1 .. h+nch; the count rides in the header (no dependent scan of the records)
      uint32_t nch = (hr.x >> 20) & 0xffu;
      if (nch == 255u) {
        nch = 0;
        for (uint32_t k = h + 1;; ++k) {
          ++nch;
          if (st[(size_t)k * 64].x & F_LAST) break;
        }
      }
      if (hr.x & F_AND) {
        if (ns < max_sample) outr[ns] = hr.y;
        ++ns;
        for (uint32_t k = nch; k-- > 0;) {
          const uint2 cr = st[(size_t)(h + 1 + k) * 64];
          FSTACK_PUSH((cr.x & F_IDX) | cold | (cr.y & 0x80000000u))
        }
      } else {
        const double power = cold ? 1.0 : A.power;
        double norm = F_NEG_INF;
        for (uint32_t k = 0; k < nch; ++k) norm = f_lwadd(norm, ins[(size_t)(st[(size_t)(h + 1 + k) * 64].x & F_IDX) * 64] * power);
        double choice = gibbs_uniform(A.seed, A.iter, forest, step++);
        uint32_t pick = 0;
        for (uint32_t k = 0;; ++k) {
          pick = k;
          choice -= exp(ins[(size_t)(st[(size_t)(h + 1 + k) * 64].x & F_IDX) * 64] * power - norm);
          if (choice < 0 || k + 1 == nch) break;
        }
        const uint2 cr = st[(size_t)(h + 1 + pick) * 64];
        FSTACK_PUSH((cr.x & F_IDX) | cold | (cr.y & 0x80000000u))
      }
    }
#undef FSTACK_PUSH
    if (A.trace) tr3 = __builtin_readcyclecounter();
    A.sample_len[forest] = ns < max_sample ? ns : max_sample;
    for (uint32_t k = 0; k < ns && k < max_sample; ++k) {
      const uint32_t rule = outr[k], nn = A.p_norm[rule];
      double pr;
      if (nn == F_NONORM)
        pr = A.p_prior[rule];
      else {
        double x = A.snap_x[rule], nsum = A.snap_norm[nn];
        if (hashed) {
          x -= own_uses(rule);
          nsum -= own_uses(nn | 0x800000u);
        } else {
          for (uint32_t q = 0; q < own_len; ++q) {
            const uint32_t rr = own[q];
            if (rr == rule) x -= 1.0;
            if (A.p_norm[rr] == nn) nsum -= 1.0;
          }
        }
        pr = x / nsum;
      }
      cheap += log(pr);
    }
  }
  for (int o = 32; o > 0; o >>= 1) cheap += __shfl_down(cheap, o, 64);
  if (lane == 0) unsafeAtomicAdd(A.iter_out + 1, cheap);
  if (A.trace) {
    unsigned long long t3 = tr3;
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_down(t3, o, 64);
      t3 = other > t3 ? other : t3;
    }
    if (lane == 0) {
      unsigned long long* o = A.trace + (size_t)(A.first_group + blockIdx.x) * 8;
      o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = t3; o[4] = __builtin_readcyclecounter(); o[5] = g.maxlen; o[6] = g.n_lanes;
    }
  }
}

// ---------------- parallel sweep, second formulation ----------------
// The sweep above spends its time in chains of dependent gathers made by one lane per forest (count tables, rule ->
// group -> counts per AND node, one node per round trip in the walk): at 1563 waves of 64 forests the chip idles.
// Here everything that does not depend on the recursion runs with one thread per record or per sample entry:
//   forest_proposal_kernel  one thread per AND header (a static list of them, forest after forest): proposal probability
//                           of its rule ((count - own uses) / (group sum - own uses), gibbs.hpp:589-592 with the block's
//                           own sample taken out).  Equal rules / equal norm groups within a forest form CLASSES
//                           (rec_cls, static); "own uses" = how many class words of the forest's previous sample
//                           (sample_cls, left by the recount) match -- a short scan, shared by the threads of a wave.
//   forest_sample_kernel    one lane per forest: inside pass as a pure stream (record + its p, three chunks in flight),
//                           top-down walk over 16-bit tables the inside pass leaves in LDS (LW; otherwise over the
//                           global stream, one round trip per visited node).
//   forest_recount_kernel   one thread per sample entry: rule ids and class words of the new sample, counts, proposal
//                           probability of the sample.
__global__ __launch_bounds__(256) void forest_proposal_kernel(ForestArgs A) {
  // one thread per AND header, from the static list of them ({position, group, classes, rule}: the records themselves are
  // not read, and the 55 % of the stream that is not an AND header is not visited)
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n_and) return;
  const FAnd e = A.and_list[i];
  const uint32_t rule = e.rule, nn = A.p_norm[rule];
  double pr;
  if (nn == F_NONORM)
    pr = A.p_prior[rule];
  else {
    // how often the forest's previous sample uses this rule / this norm group: a scan of the sample's class words.  The
    // list is ordered by forest, so the threads of a wave scan the same few samples together (the reads are broadcasts)
    const uint32_t c = e.cls, cr = c & 0xffffu, cn = c >> 16;
    const uint32_t* __restrict__ sc = A.sample_cls + A.sample_off[e.forest];
    const uint32_t len = A.old_len[e.forest];
    uint32_t own_r = 0, own_n = 0, j = 0;
    for (; j + 4 <= len; j += 4) {  // four words in flight
      const uint32_t w0 = sc[j], w1 = sc[j + 1], w2 = sc[j + 2], w3 = sc[j + 3];
      own_r += ((w0 & 0xffffu) == cr) + ((w1 & 0xffffu) == cr) + ((w2 & 0xffffu) == cr) + ((w3 & 0xffffu) == cr);
      own_n += ((w0 >> 16) == cn) + ((w1 >> 16) == cn) + ((w2 >> 16) == cn) + ((w3 >> 16) == cn);
    }
    for (; j < len; ++j) {
      const uint32_t w = sc[j];
      own_r += (w & 0xffffu) == cr ? 1u : 0u;
      own_n += (w >> 16) == cn ? 1u : 0u;
    }
    const double x = A.snap_x[rule] - (double)own_r;
    const double ns = A.snap_norm[nn] - (double)own_n;
    pr = x / ns;
  }
  A.rec_p[e.pos] = pr;
  if (!A.p_only) A.rec_logp[e.pos] = log(pr);
}

#define FS_CHUNK 4
// EXT (temperature 1): inside values as mantissa x 2^exponent (frexp / ldexp are single instructions) instead of
// logarithms -- a product is a multiply and an integer add, the OR fold an aligned add, a choice probability a
// multiply by the reciprocal of the node's own inside value: a few instructions where the log domain spends an exp
// and a log1p per child.  Differences to the log-domain fold are rounding (1e-16 relative).
//
// LW (walk tables in LDS): the top-down walk is a chain of dependent reads -- a node's record names its children -- and
// from the global stream each visited node costs a memory round trip (~2 500 cycles; the walk was 2/3 of a wave's time).
// The inside pass sees every record anyway, so it leaves the walk's view of the forest in LDS as 16-bit words: per node
// the slot of its first child (| 0x8000 = AND) and the stream position of its header, per child entry the child's node
// (| 0x8000 = reached through a back-reference).  The walk then reads LDS only (same order, same uniforms, same
// arithmetic: the same sample) and leaves the header positions of the chosen rules; forest_recount_kernel, which reads
// the records at those positions anyway, writes the rule ids into the sample.
template <bool GCOL, bool EXT, bool LW>
__global__ __launch_bounds__(64) void forest_sample_kernel(ForestArgs A, uint32_t max_sample, uint32_t ins_rows,
                                                            uint32_t stack_lds, uint32_t kid_rows) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* colbase = GCOL ? A.gcol + (size_t)blockIdx.x * A.gcol_stride : lds;
  // the stack follows the column(s) when they are in LDS (EXT: mantissas, then the exponents at half the size)
  double* aux = GCOL ? lds : lds + (size_t)ins_rows * 64 + (EXT ? (size_t)ins_rows * 32 : 0);
  const FGroup g = A.groups[A.first_group + blockIdx.x];
  const int lane = threadIdx.x;
  const bool active = (uint32_t)lane < g.n_lanes;
  const uint32_t n = active ? A.lane_nodes[g.lane_base + lane] : 0u;
  const uint32_t forest = active ? A.lane_forest[g.lane_base + lane] : 0u;
  double* ins = colbase + lane;                                           // ln inside, or its mantissa (EXT)
  int* ine = (int*)(colbase + (size_t)ins_rows * 64) + lane;              // EXT: its exponent
  // LW: 16-bit rows after the columns: first-child slot per node (+ one closing row), header position per node, child
  // entries, the walk's stack
  unsigned short* wt = (unsigned short*)aux + lane;
  unsigned short* ht = wt + (size_t)(ins_rows + 1) * 64;
  unsigned short* ct = ht + (size_t)ins_rows * 64;
  unsigned short* stk16 = ct + (size_t)kid_rows * 64;
  uint32_t slot = 0, hpos = 0;
  const uint2* __restrict__ st = A.ins_stream + g.stream_base + lane;
  const double* __restrict__ lp = (EXT ? A.rec_p : A.rec_logp) + g.stream_base + lane;
  const uint32_t last = g.maxlen - 1;
  unsigned long long tr0 = A.trace ? __builtin_readcyclecounter() : 0, tr2 = 0, tr3 = 0;
  // inside with the proposal probabilities (forest.hpp:768-816)
  {
    uint32_t d = 0;
    bool is_and = false;
    double acc = 0.0, sum = F_NEG_INF;
    int acc_e = 0, sum_e = 0;  // EXT: acc / sum are mantissas
    // three chunks of records in flight: the loads of chunk k + 2 are issued before chunk k is folded (a wave is alone
    // on its SIMD most of the time -- LDS bounds the occupancy --, so nobody else hides the stream's latency)
    uint2 r[FS_CHUNK], r1[FS_CHUNK], r2[FS_CHUNK];
    double p[FS_CHUNK], p1[FS_CHUNK], p2[FS_CHUNK];
#define FS_LOAD(R, P, base)                                  \
  _Pragma("unroll") for (int j = 0; j < FS_CHUNK; ++j) {     \
    const uint32_t k = min((base) + j, last);                \
    R[j] = st[(size_t)k * 64];                               \
    P[j] = lp[(size_t)k * 64];                               \
  }
    FS_LOAD(r, p, 0u)
    FS_LOAD(r1, p1, (uint32_t)FS_CHUNK)
    for (uint32_t k0 = 0; k0 < g.maxlen; k0 += FS_CHUNK) {
      if (k0) {
#pragma unroll
        for (int j = 0; j < FS_CHUNK; ++j) {
          r[j] = r1[j];
          p[j] = p1[j];
          r1[j] = r2[j];
          p1[j] = p2[j];
        }
      }
      FS_LOAD(r2, p2, k0 + 2u * FS_CHUNK)
#pragma unroll
      for (int j = 0; j < FS_CHUNK; ++j) {
        if constexpr (EXT && LW) {
          // the same fold without branches: a wave's lanes sit at headers, AND children and OR children alike, every
          // branch was taken by somebody and the jumps around them cost as much as the arithmetic.  All candidates
          // are computed (the same operations on the same operands as below), selects keep the one that applies; the
          // node's running value and header position are stored at every record, the last store stays.
          const uint32_t rx = r[j].x, ry = r[j].y;
          if (k0 + j <= last && active && (rx & F_VALID)) {
            const bool hdr = (rx & F_HEADER) != 0, child = !hdr;
            const uint32_t row = min(rx & F_IDX, ins_rows - 1);  // (a header's low bits are not a row: read and ignored)
            const double vm = ins[(size_t)row * 64];
            const int ve = ine[(size_t)row * 64];
            hpos = hdr ? k0 + j : hpos;
            unsigned short* tp = hdr ? wt + (size_t)d * 64 : ct + (size_t)slot * 64;
            *tp = (unsigned short)(hdr ? (slot | ((rx & F_AND) ? 0x8000u : 0u)) : ((rx & 0x7fffu) | ((ry >> 31) << 15)));
            ht[(size_t)d * 64] = (unsigned short)hpos;
            slot += child ? 1u : 0u;
            int he, pe, se;
            const double hm = frexp(p[j], &he);
            const double pm = frexp(acc * vm, &pe);
            const int dd = ve - sum_e;
            const bool le = dd <= 0;
            const double lo = le ? vm : sum, hi = le ? sum : vm;
            const double sm = frexp(hi + ldexp(lo, le ? dd : -dd), &se);
            const bool fold_and = child && is_and, fold_or = child && !is_and && vm != 0.0, sum0 = sum == 0.0;
            acc_e = hdr ? he : fold_and ? acc_e + ve + pe : acc_e;
            acc = hdr ? hm : fold_and ? pm : acc;
            const int nsum_e = sum0 ? ve : (le ? sum_e : ve) + se;
            const double nsum = sum0 ? vm : sm;
            sum_e = hdr ? 0 : fold_or ? nsum_e : sum_e;
            sum = hdr ? 0.0 : fold_or ? nsum : sum;
            is_and = hdr ? (rx & F_AND) != 0 : is_and;
            ins[(size_t)d * 64] = is_and ? acc : sum;
            ine[(size_t)d * 64] = is_and ? acc_e : sum_e;
            d += (rx & F_LAST) ? 1u : 0u;
          }
          continue;
        }
        if (k0 + j > last || !active || !(r[j].x & F_VALID)) continue;
        if (LW) {
          if (r[j].x & F_HEADER) {
            wt[(size_t)d * 64] = (unsigned short)(slot | ((r[j].x & F_AND) ? 0x8000u : 0u));
            ht[(size_t)d * 64] = (unsigned short)(k0 + j);
          } else {
            ct[(size_t)slot * 64] = (unsigned short)((r[j].x & 0x7fffu) | ((r[j].y >> 31) << 15));
            ++slot;
          }
        }
        if (EXT) {
          if (r[j].x & F_HEADER) {
            is_and = (r[j].x & F_AND) != 0;
            acc = frexp(p[j], &acc_e);
            sum = 0.0;
            sum_e = 0;
          } else {
            const double vm = ins[(size_t)(r[j].x & F_IDX) * 64];
            const int ve = ine[(size_t)(r[j].x & F_IDX) * 64];
            if (is_and) {
              int t;
              acc = frexp(acc * vm, &t);
              acc_e += ve + t;
            } else if (vm != 0.0) {
              if (sum == 0.0) {
                sum = vm;
                sum_e = ve;
              } else {
                const int dd = ve - sum_e;
                int t;
                if (dd <= 0)
                  sum = frexp(sum + ldexp(vm, dd), &t);
                else {
                  sum = frexp(ldexp(sum, -dd) + vm, &t);
                  sum_e = ve;
                }
                sum_e += t;
              }
            }
          }
          if (r[j].x & F_LAST) {
            ins[(size_t)d * 64] = is_and ? acc : sum;
            ine[(size_t)d * 64] = is_and ? acc_e : sum_e;
            ++d;
          }
          continue;
        }
        if (r[j].x & F_HEADER) {
          is_and = (r[j].x & F_AND) != 0;
          acc = p[j];
          sum = F_NEG_INF;
        } else {
          const double v = ins[(size_t)(r[j].x & F_IDX) * 64];
          if (is_and)
            acc += v;
          else
            sum = f_lwadd(sum, v);  // the reference's pairwise OR fold (forest.hpp:790-797)
        }
        if (r[j].x & F_LAST) {
          ins[(size_t)d * 64] = is_and ? acc : sum;
          ++d;
        }
      }
    }
  }
  if (A.trace) tr2 = __builtin_readcyclecounter();
  if (LW && active) {
    // the same walk over the LDS tables; stack entries: node | 0x8000 = below a back-reference
    wt[(size_t)n * 64] = (unsigned short)slot;
    uint32_t* outh = A.sample_hdr + A.sample_off[forest];
    const uint32_t cap = (uint32_t)(A.sample_off[forest + 1] - A.sample_off[forest]);
    uint32_t* stack = A.sample_rules + A.sample_off[forest] + cap;  // deep part of the stack: stack[-1 - i]
#define FSTACK_PUSH(v)                                       \
  {                                                          \
    if (sp < stack_lds)                                      \
      stk16[(size_t)sp * 64] = (unsigned short)(v);          \
    else                                                     \
      stack[-(int)(sp - stack_lds) - 1] = (v);               \
    ++sp;                                                    \
  }
    // `entry` = the node to visit next, kept in a register when it is the child just chosen or an AND node's first
    // child (what the stack would hand back at once); F_NONE = take it from the stack
    const uint32_t F_NONE = 0xffffffffu;
    uint32_t sp = 0, ns = 0, step = 0, entry = n - 1;
    // One turn of the loop = an OR node's choice AND the chosen AND node's expansion: the lanes of a wave sit at OR and at
    // AND nodes alike, so both halves are executed every turn anyway -- visiting the pair in one turn halves the turns.
    for (;;) {
      if (entry == F_NONE) {
        if (!sp) break;
        --sp;
        entry = sp < stack_lds ? (uint32_t)stk16[(size_t)sp * 64] : stack[-(int)(sp - stack_lds) - 1];
      }
      uint32_t me = entry & 0x7fffu, cold = entry & 0x8000u;
      uint32_t w0 = wt[(size_t)me * 64], w1 = wt[(size_t)(me + 1) * 64];
      uint32_t first = w0 & 0x7fffu, nch = (w1 & 0x7fffu) - first;
      if (!(w0 & 0x8000u)) {
        uint32_t pick = 0;
        if (EXT) {
          // the first four children's shares at once (rows past the node's children are read and ignored: every row
          // below kid_rows exists), then the reference's serial subtraction without branches: the same differences in
          // the same order, the first one below zero (or the last child) is the choice
          uint32_t ci4[4];
          double t4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) ci4[q] = ct[(size_t)min(first + q, kid_rows - 1) * 64] & 0x7fffu;
          // (the shares are taken against u x the node's own value instead of dividing each by it: choice, children and
          // the node's value all in units of 2^ne -- the reference's comparison times a positive constant)
          const double zm = ins[(size_t)me * 64];
          const int ne = ine[(size_t)me * 64];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t row = min(ci4[q], ins_rows - 1);
            t4[q] = ldexp(ins[(size_t)row * 64], ine[(size_t)row * 64] - ne);
          }
          double choice = gibbs_uniform(A.seed, A.iter, forest, step++) * zm;
          const double c0 = choice - t4[0], c1 = c0 - t4[1], c2 = c1 - t4[2], c3 = c2 - t4[3];
          const bool s0 = c0 < 0 || nch == 1, s1 = c1 < 0 || nch == 2, s2 = c2 < 0 || nch == 3, s3 = c3 < 0 || nch == 4;
          pick = s0 ? 0u : s1 ? 1u : s2 ? 2u : 3u;
          if (!(s0 || s1 || s2 || s3)) {
            choice = c3;
            for (uint32_t k = 4;; ++k) {
              pick = k;
              const uint32_t ci = ct[(size_t)(first + k) * 64] & 0x7fffu;
              choice -= ldexp(ins[(size_t)ci * 64], ine[(size_t)ci * 64] - ne);
              if (choice < 0 || k + 1 == nch) break;
            }
          }
        } else {
          const double power = cold ? 1.0 : A.power;
          double norm = ins[(size_t)me * 64];
          if (power != 1.0) {
            norm = F_NEG_INF;
            for (uint32_t k = 0; k < nch; ++k)
              norm = f_lwadd(norm, ins[(size_t)(ct[(size_t)(first + k) * 64] & 0x7fffu) * 64] * power);
          }
          double choice = gibbs_uniform(A.seed, A.iter, forest, step++);
          for (uint32_t k = 0;; ++k) {
            pick = k;
            choice -= exp(ins[(size_t)(ct[(size_t)(first + k) * 64] & 0x7fffu) * 64] * power - norm);
            if (choice < 0 || k + 1 == nch) break;
          }
        }
        entry = (uint32_t)ct[(size_t)(first + pick) * 64] | cold;
        me = entry & 0x7fffu;
        cold = entry & 0x8000u;
        w0 = wt[(size_t)me * 64];
        w1 = wt[(size_t)(me + 1) * 64];
        first = w0 & 0x7fffu;
        nch = (w1 & 0x7fffu) - first;
      }
      if (w0 & 0x8000u) {
        if (ns < max_sample) outh[ns] = ht[(size_t)me * 64];
        ++ns;
        for (uint32_t k = nch; k-- > 1;) FSTACK_PUSH((uint32_t)ct[(size_t)(first + k) * 64] | cold)
        entry = nch ? ((uint32_t)ct[(size_t)first * 64] | cold) : F_NONE;
      }
    }
#undef FSTACK_PUSH
    A.sample_len[forest] = ns < max_sample ? ns : max_sample;
  }
  // top-down choice (forest.hpp:725-758); stack entries: header position | bit 31 = below a back-reference
  if (!LW && active) {
    uint32_t* outr = A.sample_rules + A.sample_off[forest];
    uint32_t* outh = A.sample_hdr + A.sample_off[forest];
    const uint32_t cap = (uint32_t)(A.sample_off[forest + 1] - A.sample_off[forest]);
    uint32_t* stack = outr + cap;  // deep part of the stack: stack[-1 - i]
    uint32_t* stk_sh = (uint32_t*)aux + lane;
#define FSTACK_PUSH(v)                                       \
  {                                                          \
    if (sp < stack_lds)                                      \
      stk_sh[(size_t)sp * 64] = (v);                         \
    else                                                     \
      stack[-(int)(sp - stack_lds) - 1] = (v);               \
    ++sp;                                                    \
  }
    uint32_t sp = 0, ns = 0, step = 0;
    FSTACK_PUSH(A.hdr_pos[g.stream_base + (size_t)(n - 1) * 64 + lane])
    while (sp) {
      --sp;
      const uint32_t entry = sp < stack_lds ? stk_sh[(size_t)sp * 64] : stack[-(int)(sp - stack_lds) - 1];
      const uint32_t h = entry & 0x7fffffffu, cold = entry & 0x80000000u;
      const uint2 hr = st[(size_t)h * 64];
      uint2 c[4];  // the first children ride along with the header: one round trip per node
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = st[(size_t)min(h + 1 + j, last) * 64];
      uint32_t nch = (hr.x >> 20) & 0xffu;
      if (nch == 255u) {
        nch = 0;
        for (uint32_t k = h + 1;; ++k) {
          ++nch;
          if (st[(size_t)k * 64].x & F_LAST) break;
        }
      }
      if (hr.x & F_AND) {
        if (ns < max_sample) {
          outr[ns] = hr.y;
          outh[ns] = h;
        }
        ++ns;
        for (uint32_t k = nch; k-- > 0;) {
          const uint2 cr = k < 4 ? (k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3]) : st[(size_t)(h + 1 + k) * 64];
          FSTACK_PUSH(cr.y | cold)
        }
      } else {
        uint32_t pick = 0;
        if (EXT) {  // temperature 1 throughout (the host picks this instantiation only then)
          const uint32_t me = hr.x & 0xfffffu;
          const double inv = 1.0 / ins[(size_t)me * 64];
          const int ne = ine[(size_t)me * 64];
          double choice = gibbs_uniform(A.seed, A.iter, forest, step++);
          for (uint32_t k = 0;; ++k) {
            pick = k;
            const uint2 cr = k < 4 ? (k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3]) : st[(size_t)(h + 1 + k) * 64];
            const uint32_t ci = cr.x & F_IDX;
            choice -= ldexp(ins[(size_t)ci * 64] * inv, ine[(size_t)ci * 64] - ne);
            if (choice < 0 || k + 1 == nch) break;
          }
        } else {
        const double power = cold ? 1.0 : A.power;
        // at temperature 1 the normaliser is the node's own inside value: the same fold over the same children
        double norm = ins[(size_t)(hr.x & 0xfffffu) * 64];
        if (power != 1.0) {
          norm = F_NEG_INF;
          for (uint32_t k = 0; k < nch; ++k) {
            const uint2 cr = k < 4 ? (k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3]) : st[(size_t)(h + 1 + k) * 64];
            norm = f_lwadd(norm, ins[(size_t)(cr.x & F_IDX) * 64] * power);
          }
        }
        double choice = gibbs_uniform(A.seed, A.iter, forest, step++);
        for (uint32_t k = 0;; ++k) {
          pick = k;
          const uint2 cr = k < 4 ? (k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3]) : st[(size_t)(h + 1 + k) * 64];
          choice -= exp(ins[(size_t)(cr.x & F_IDX) * 64] * power - norm);
          if (choice < 0 || k + 1 == nch) break;
        }
        }
        const uint2 cr = pick < 4 ? (pick == 0 ? c[0] : pick == 1 ? c[1] : pick == 2 ? c[2] : c[3]) : st[(size_t)(h + 1 + pick) * 64];
        FSTACK_PUSH(cr.y | cold)
      }
    }
#undef FSTACK_PUSH
    A.sample_len[forest] = ns < max_sample ? ns : max_sample;
  }
  if (A.trace) {
    tr3 = __builtin_readcyclecounter();
    unsigned long long t3 = tr3;
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_down(t3, o, 64);
      t3 = other > t3 ? other : t3;
    }
    if (lane == 0) {
      unsigned long long* o = A.trace + (size_t)(A.first_group + blockIdx.x) * 8;
      o[0] = tr0; o[1] = tr0; o[2] = tr2; o[3] = t3; o[4] = __builtin_readcyclecounter(); o[5] = g.maxlen; o[6] = g.n_lanes;
    }
  }
}

// ---- several lanes per forest (parallel sweep, temperature 1) ----
// forest_sample_kernel is one forest per lane: a forest's inside values take a column of LDS per lane (40-110 KB per wave on
// config 5: two or three waves per CU, nobody to hide a wave's dependent LDS round trips behind) and a sweep costs what its
// slowest lane costs.  Here a forest gets FM_G lanes (FM_FPW = 64 / FM_G forests per wavefront) and a few hundred bytes of
// LDS: its tables (children lists, node order by HEIGHT) are copied in once, the inside pass goes height by height with the
// lanes over the nodes of a height (a node's children are all of lower height), and the walk is breadth-first: the lanes
// take the entries of the current frontier, an OR entry chooses one child, an AND entry records its rule and hands on all of
// its children; slots in the next frontier and in the sample come from a prefix sum over the lanes (no atomics: the order is
// the same in every run).  Arithmetic is forest_sample_kernel's EXT form (mantissa x 2^exponent).  ONE difference in the
// chain: the sequential walk draws its uniforms by the ORDER OF VISITS of a depth-first walk, which a breadth-first walk
// does not have -- the uniform of a visit is keyed by its position in the breadth-first order instead.  Every visit still has
// a uniform of its own (a shared sub-forest expanded twice chooses twice): the same kind of chain -- the stale-count sweep of
// forest-em.hpp:750-766 with other random numbers -- validated against the sweep's enumerated stationary distribution
// (tests/test_bench_workloads_gpu.py) instead of draw for draw.
#ifndef FM_G
#define FM_G 8
#endif
#define FM_FPW (64 / FM_G)
struct FMultiArgs {
  const uint16_t* tab;      // per forest, its nodes numbered by height: {n, H, n_kids, -}, lvl_off[H + 1], kid_off[n + 1],
                            // kids[n_kids] (| 0x8000: back-reference)
  const uint32_t* hdr;      // per forest, per node: {row of its header in the lane's inside stream | bit 31 = AND, rule id,
                            // class word (ForestArgs::rec_cls), norm group}: four words per node
  const uint4* slots;       // per lane slot, two words of 16 bytes: {tab offset (u16 words, a multiple of 8: the table is copied
                            // 16 bytes at a time), hdr offset (u32 words)} as two 64-bit numbers, {sample offset (64 bit),
                            // forest (0xffffffff: none), nodes | table words << 15}: everything the staging needs to address
                            // its loads, in one round trip
  uint32_t lane_lo, lane_hi;            // the lane slots of this launch (a launch class)
  uint32_t max_tab, max_n, max_front;   // LDS per forest: table words, nodes, frontier entries
  int own_proposal;                     // the kernel computes the rules' proposal probabilities itself (forest_proposal_kernel
                                        // folded in: each AND node scans the forest's previous sample for its own uses)
  uint16_t* node_cnt;                   // own_proposal, non-null: per node (in the order of hdr) how often this sweep's sample
                                        // records it -- what the counts are gathered from afterwards (forest_rule_gather_kernel);
                                        // counted in the low half of the node's header word in LDS, which the walk does not use
  double* prob;                         // own_proposal: per node (in the order of hdr, four words a node) its rule's proposal
                                        // probability, and the sample is written as NODE numbers: what the recount needs of a
                                        // sampled rule -- id, class word, norm group, probability -- then lies in the forest's
                                        // own few lines of hdr / prob instead of three interleaved record streams (round 6)
};
__device__ __forceinline__ uint32_t fm_prefix(uint32_t v, uint32_t li, uint32_t& total) {
  // exclusive prefix sum over the FM_G lanes of a forest; total = the sum
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < FM_G; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, FM_G);
    if (li >= (uint32_t)d) x += y;
  }
  total = __shfl(x, FM_G - 1, FM_G);
  return x - v;
}
#define FM_EBIAS 2048
#if FM_G == 8
#define FM_TQ 4  // 16-byte pieces of the forest's table per lane in the first round of loads (x FM_G lanes x 8 words)
#define FM_SC 4  // class words of the previous sample per lane ...
#define FM_HR 8  // header rows per lane ...
#else
#define FM_TQ 8
#define FM_SC 8
#define FM_HR 12
#endif
#define FM_KP 4  // children of a node whose values the inside pass requests before it folds them
__global__ __launch_bounds__(64) void forest_sample_multi_kernel(ForestArgs A, FMultiArgs M, uint32_t max_sample) {
  extern __shared__ __attribute__((aligned(16))) double fm_lds[];
  const uint32_t sub = threadIdx.x / FM_G, li = threadIdx.x % FM_G;
  const unsigned long long tr0 = A.trace ? __builtin_readcyclecounter() : 0;
  unsigned long long tr1 = 0, tr2 = 0, tr3 = 0;
  const uint32_t slot = M.lane_lo + blockIdx.x * FM_FPW + sub;
  const uint32_t forest = slot < M.lane_hi ? A.lane_forest[slot] : 0xffffffffu;  // (rides along with the descriptor's loads)
  const bool active = forest != 0xffffffffu;
  // this forest's stretch of LDS: mantissas (f64), exponents (i32), header words (u32: row of the node's header record | the
  // exponent of an AND node's proposal probability + FM_EBIAS, bits 16..30 | bit 31 = AND), table + two frontiers (u16)
  const size_t per = (size_t)M.max_n * 16 + (((size_t)M.max_tab + 2 * (size_t)M.max_front) * 2 + 15) / 16 * 16;
  char* mine = (char*)fm_lds + per * sub;
  double* vm = (double*)mine;
  int* ve = (int*)(vm + M.max_n);
  uint32_t* hd = (uint32_t*)(ve + M.max_n);
  unsigned short* tb = (unsigned short*)(hd + M.max_n);
  unsigned short* fr0 = tb + M.max_tab;
  unsigned short* fr1 = fr0 + M.max_front;
  // ---- staging: three rounds of loads.  (1) the slot's descriptor; (2) the forest's tables (16 bytes a load), the length
  // and the first FM_SC x FM_G class words of its previous sample, the header rows of its first FM_HR x FM_G nodes; (3) the
  // snapshot counts of those rows' rules.  Forests with more nodes / longer samples continue in loops afterwards.
  const FGroup g = A.groups[active ? slot / 64 : M.lane_lo / 64];
  const uint32_t lane = slot % 64;
  uint4 d0 = make_uint4(0, 0, 0, 0), d1 = make_uint4(0, 0, 0, 0);
  if (active) {
    d0 = M.slots[2 * (size_t)slot];
    d1 = M.slots[2 * (size_t)slot + 1];
  }
  const uint32_t n = d1.w & 0x7fffu, words = d1.w >> 15;
  const unsigned short* __restrict__ src = M.tab + (((uint64_t)d0.y << 32) | d0.x);
  const uint4* __restrict__ hs = (const uint4*)(M.hdr + (((uint64_t)d0.w << 32) | d0.z));
  const uint32_t* __restrict__ sc = A.sample_cls + (((uint64_t)d1.y << 32) | d1.x);
  double* __restrict__ recp0 = A.rec_p + g.stream_base + lane;
  const bool own = M.own_proposal != 0;
  double* __restrict__ probp = M.prob + ((((uint64_t)d0.w << 32) | d0.z) >> 2);
  uint32_t plen = 0;
  uint32_t scq[FM_SC];
  uint4 hq[FM_HR];
  double sx[FM_HR], sn[FM_HR];
  {
    const uint4* __restrict__ s4 = (const uint4*)src;
    uint4* t4 = (uint4*)tb;
    const uint32_t w4 = (words + 7) / 8;
    uint4 tq[FM_TQ];
#pragma unroll
    for (int q = 0; q < FM_TQ; ++q) tq[q] = li + q * FM_G < w4 ? s4[li + q * FM_G] : make_uint4(0, 0, 0, 0);
    if (active && own) plen = A.old_len[forest];
#pragma unroll
    for (int q = 0; q < FM_SC; ++q) scq[q] = (active && own) ? sc[li + q * FM_G] : 0xffffffffu;  // (read past the sample: its capacity, or the padding)
#pragma unroll
    for (int q = 0; q < FM_HR; ++q) hq[q] = li + q * FM_G < n ? hs[li + q * FM_G] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < FM_TQ; ++q)
      if (li + q * FM_G < w4) t4[li + q * FM_G] = tq[q];
    for (uint32_t k = li + FM_TQ * FM_G; k < w4; k += FM_G) t4[k] = s4[k];  // (tables beyond FM_TQ x FM_G x 8 words)
    if (own)
      for (uint32_t k = li; k < n; k += FM_G) ve[k] = 0;
    // round three: what the rows' probabilities need from the snapshot (or the probabilities themselves)
#pragma unroll
    for (int q = 0; q < FM_HR; ++q) {
      const uint4 h = hq[q];
      sx[q] = 0.0;
      sn[q] = 1.0;
      if (h.x & 0x80000000u) {
        if (!own)
          sx[q] = recp0[(size_t)(h.x & 0x7fffffffu) * 64];
        else if (h.w == F_NONORM)
          sx[q] = A.p_prior[h.y];
        else {
          sx[q] = A.snap_x[h.y];
          sn[q] = A.snap_norm[h.w];
        }
      }
    }
  }
  __syncthreads();
  if (active && own) {
    // how often the forest's previous sample uses each of its rule classes (low half) and norm-group classes (high half):
    // one pass over the sample's class words (forest_proposal_kernel scans them per rule); the exponents' rows hold the
    // counts until the inside pass writes them
#define FM_HIST(wd)                                                                                          \
  if ((wd) != 0xffffffffu) { /* (0xffffffff: a rule outside the normalisation groups) */                    \
    __hip_atomic_fetch_add(&ve[(wd) & 0xffffu], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);          \
    __hip_atomic_fetch_add(&ve[(wd) >> 16], 0x10000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        \
  }
#pragma unroll
    for (int q = 0; q < FM_SC; ++q)
      if (li + q * FM_G < plen) FM_HIST(scq[q])
    for (uint32_t j = li + FM_SC * FM_G; j < plen; j += FM_G) {
      const uint32_t wd = sc[j];
      FM_HIST(wd)
    }
#undef FM_HIST
  }
  __syncthreads();
  const uint32_t H = active ? tb[1] : 0u;
  if (active) {
    // an AND node's row: its proposal probability -- forest_proposal_kernel's (count - own uses) / (norm sum - own uses of the
    // group) -- as the mantissa where the node's value will be and the exponent in the header word (the exponents' rows
    // still hold the histogram); the recount reads the sampled rules' probabilities from rec_p
#define FM_ROW(k, hh, px, pn)                                                                                 \
  {                                                                                                          \
    uint32_t hw = (hh).x;                                                                                     \
    if (own) hw &= 0x80000000u; /* (the row of the header record: the one-per-lane kernels' business; here: a counter) */ \
    if (hw & 0x80000000u) {                                                                                  \
      double pr = (px);                                                                                       \
      if (own) {                                                                                             \
        if ((hh).w != F_NONORM) {                                                                             \
          const uint32_t own_r = (uint32_t)ve[(hh).z & 0xffffu] & 0xffffu, own_n = (uint32_t)ve[(hh).z >> 16] >> 16; \
          pr = ((px) - (double)own_r) / ((pn) - (double)own_n);                                              \
        }                                                                                                    \
        probp[k] = pr;                                                                                       \
      }                                                                                                      \
      int e;                                                                                                 \
      vm[k] = frexp(pr, &e);                                                                                 \
      hw |= (uint32_t)(e + FM_EBIAS) << 16;                                                                  \
    }                                                                                                        \
    hd[k] = hw;                                                                                              \
  }
#pragma unroll
    for (int q = 0; q < FM_HR; ++q)
      if (li + q * FM_G < n) FM_ROW(li + q * FM_G, hq[q], sx[q], sn[q])
    for (uint32_t k = li + FM_HR * FM_G; k < n; k += FM_G) {  // (forests beyond FM_HR x FM_G nodes: two more rounds)
      const uint4 h = hs[k];
      double x = 0.0, nrm = 1.0;
      if (h.x & 0x80000000u) {
        if (!own)
          x = recp0[(size_t)(h.x & 0x7fffffffu) * 64];
        else if (h.w == F_NONORM)
          x = A.p_prior[h.y];
        else {
          x = A.snap_x[h.y];
          nrm = A.snap_norm[h.w];
        }
      }
      FM_ROW(k, h, x, nrm)
    }
#undef FM_ROW
  }
  __syncthreads();
  if (A.trace) tr1 = __builtin_readcyclecounter();
  const unsigned short* lvl = tb + 4;
  const unsigned short* koff = lvl + H + 1;
  const unsigned short* kids = koff + n + 1;
  // ---- inside, height by height (forest.hpp:768-816 with the proposal probabilities) ----
  uint32_t Hmax = H;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) Hmax = max(Hmax, (uint32_t)__shfl_xor((int)Hmax, o, 64));
  for (uint32_t h = 0; h < Hmax; ++h) {
    if (h < H)
      for (uint32_t node = lvl[h] + li; node < lvl[h + 1]; node += FM_G) {  // (the nodes of a height: a range of ids)
        const uint32_t hw = hd[node], k0 = koff[node], k1 = koff[node + 1];
        // the first FM_KP children's values are requested together, before the fold (which is a chain): one LDS round trip
        // for the ids, one for the values, instead of two per child
        uint32_t cq[FM_KP];
        double mq[FM_KP];
        int eq[FM_KP];
#pragma unroll
        for (int i = 0; i < FM_KP; ++i) cq[i] = k0 + i < k1 ? (uint32_t)(kids[k0 + i] & 0x7fffu) : node;
#pragma unroll
        for (int i = 0; i < FM_KP; ++i) {
          mq[i] = vm[cq[i]];
          eq[i] = ve[cq[i]];
        }
        double m;
        int e;
        if (hw & 0x80000000u) {  // AND: its rule's proposal probability times its children
          m = vm[node];
          e = (int)((hw >> 16) & 0x7fffu) - FM_EBIAS;
#define FM_AND_FOLD(cm, ce)   \
  {                           \
    int t;                    \
    m = frexp(m * (cm), &t);  \
    e += (ce) + t;            \
  }
#pragma unroll
          for (int i = 0; i < FM_KP; ++i)
            if (k0 + i < k1) FM_AND_FOLD(mq[i], eq[i])
          for (uint32_t k = k0 + FM_KP; k < k1; ++k) {
            const uint32_t c = kids[k] & 0x7fffu;
            FM_AND_FOLD(vm[c], ve[c])
          }
#undef FM_AND_FOLD
        } else {  // OR: the sum of its children, aligned to the larger exponent
          m = 0.0;
          e = 0;
#define FM_OR_FOLD(cm_, ce_)                      \
  {                                               \
    const double cm = (cm_);                      \
    const int ce = (ce_);                         \
    if (cm != 0.0) {                              \
      if (m == 0.0) {                             \
        m = cm;                                   \
        e = ce;                                   \
      } else {                                    \
        const int dd = ce - e;                    \
        int t;                                    \
        if (dd <= 0)                              \
          m = frexp(m + ldexp(cm, dd), &t);       \
        else {                                    \
          m = frexp(ldexp(m, -dd) + cm, &t);      \
          e = ce;                                 \
        }                                         \
        e += t;                                   \
      }                                           \
    }                                             \
  }
#pragma unroll
          for (int i = 0; i < FM_KP; ++i)
            if (k0 + i < k1) FM_OR_FOLD(mq[i], eq[i])
          for (uint32_t k = k0 + FM_KP; k < k1; ++k) {
            const uint32_t c = kids[k] & 0x7fffu;
            FM_OR_FOLD(vm[c], ve[c])
          }
#undef FM_OR_FOLD
        }
        vm[node] = m;
        ve[node] = e;
      }
    __syncthreads();
  }
  if (A.trace) tr2 = __builtin_readcyclecounter();
  // ---- the walk, breadth first (forest.hpp:725-758) ----
  uint32_t nfr = active ? 1u : 0u, ns = 0, visited = 0;
  if (active && li == 0) fr0[0] = (unsigned short)(n - 1);
  __syncthreads();
  uint32_t* outh = active ? A.sample_hdr + A.sample_off[forest] : nullptr;
  unsigned short* cur = fr0;
  unsigned short* nxt = fr1;
  for (;;) {
    if (!__any(nfr != 0)) break;
    uint32_t nn = 0;  // entries of the next frontier so far
    for (uint32_t base = 0; __any(base < nfr); base += FM_G) {
      const uint32_t idx = base + li;
      const bool have = idx < nfr;
      uint32_t push = 0, rec = 0, node = 0, k0 = 0, pick = 0;
      bool is_and = false;
      if (have) {
        node = cur[idx] & 0x7fffu;
        const uint32_t hw = hd[node];
        k0 = koff[node];
        const uint32_t nch = koff[node + 1] - k0;
        is_and = (hw & 0x80000000u) != 0;
        if (is_and) {
          rec = 1;
          push = nch;
        } else if (nch) {
          // the reference's serial subtraction: the first child whose share takes the choice below zero, or the last
          // (the first FM_KP children's values requested together, as in the inside pass; the same differences in the same order)
          uint32_t cq[FM_KP];
          double mq[FM_KP];
          int eq[FM_KP];
#pragma unroll
          for (int i = 0; i < FM_KP; ++i) cq[i] = (uint32_t)i < nch ? (uint32_t)(kids[k0 + i] & 0x7fffu) : node;
#pragma unroll
          for (int i = 0; i < FM_KP; ++i) {
            mq[i] = vm[cq[i]];
            eq[i] = ve[cq[i]];
          }
          const int ne = ve[node];
          double choice = gibbs_uniform(A.seed, A.iter, forest, visited + idx) * vm[node];
          bool done = false;
#pragma unroll
          for (int i = 0; i < FM_KP; ++i)
            if (!done && (uint32_t)i < nch) {
              pick = (uint32_t)i;
              choice -= ldexp(mq[i], eq[i] - ne);
              done = choice < 0 || (uint32_t)i + 1 == nch;
            }
          for (uint32_t k = FM_KP; !done; ++k) {
            pick = k;
            const uint32_t c = kids[k0 + k] & 0x7fffu;
            choice -= ldexp(vm[c], ve[c] - ne);
            done = choice < 0 || k + 1 == nch;
          }
          push = 1;
        }
      }
      uint32_t tot_push, tot_rec;
      const uint32_t at = fm_prefix(push, li, tot_push), ar = fm_prefix(rec, li, tot_rec);
      if (have) {
        if (is_and) {
          if (ns + ar < max_sample) {
            outh[ns + ar] = own ? node : (hd[node] & 0xffffu);
            if (own && M.node_cnt) __hip_atomic_fetch_add(&hd[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          for (uint32_t k = 0; k < push; ++k)
            if (nn + at + k < M.max_front) nxt[nn + at + k] = kids[k0 + k];
        } else if (push) {
          if (nn + at < M.max_front) nxt[nn + at] = kids[k0 + pick];
        }
      }
      nn += tot_push;
      ns += tot_rec;
    }
    visited += nfr;
    __syncthreads();
    nfr = nn < M.max_front ? nn : M.max_front;
    unsigned short* t = cur;
    cur = nxt;
    nxt = t;
  }
  if (active && li == 0) A.sample_len[forest] = ns < max_sample ? ns : max_sample;
  if (own && M.node_cnt && active) {  // every node of the forest says how often it was recorded (most: not at all)
    uint16_t* __restrict__ nc = M.node_cnt + ((((uint64_t)d0.w << 32) | d0.z) >> 2);
    for (uint32_t k = li; k < n; k += FM_G) nc[k] = (uint16_t)(hd[k] & 0xffffu);
  }
  if (A.trace) {
    tr3 = __builtin_readcyclecounter();
    uint32_t nmax = n, hmax = H, vmax = visited;
    for (int o = 32; o > 0; o >>= 1) {
      nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, o, 64));
      hmax = max(hmax, (uint32_t)__shfl_xor((int)hmax, o, 64));
      vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, o, 64));
    }
    if (threadIdx.x == 0) {
      unsigned long long* o = A.trace + ((size_t)M.lane_lo / FM_FPW + blockIdx.x) * 8;
      o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = tr3; o[4] = tr3; o[5] = nmax; o[6] = hmax; o[7] = vmax;
    }
  }
}

// Viterbi (forest.hpp:507-632): max-product inside -- an AND node is its rule's weight times its children, an OR node
// keeps its FIRST best child (a later child must be strictly better, forest.hpp:547) -- and the best derivation walked
// from the root, recorded in pre-order as {rule, number of children} per AND node (what write_viterbi_rec prints).  One
// lane per forest over the same record streams as the E-step; GCOL: the column in global memory (forests beyond LDS).
template <bool GCOL>
__global__ __launch_bounds__(64) void forest_viterbi_kernel(ForestArgs A, uint32_t max_sample, uint32_t ins_rows,
                                                             uint32_t stack_lds, double* best_logprob) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  double* colbase = GCOL ? A.gcol + (size_t)blockIdx.x * A.gcol_stride : lds_all;
  double* aux = GCOL ? lds_all : lds_all + (size_t)ins_rows * 64;
  const FGroup g = A.groups[A.first_group + blockIdx.x];
  const int lane = threadIdx.x;
  if ((uint32_t)lane >= g.n_lanes) return;
  const uint32_t n = A.lane_nodes[g.lane_base + lane];
  const uint32_t forest = A.lane_forest[g.lane_base + lane];
  double* ins = colbase + lane;
  const uint2* __restrict__ st = A.ins_stream + g.stream_base + lane;
  {
    uint32_t d = 0;
    bool is_and = false, first = true;
    double acc = 0.0, best = F_NEG_INF;
    for (uint32_t k = 0; k < g.maxlen; ++k) {
      const uint2 r = st[(size_t)k * 64];
      if (!(r.x & F_VALID)) continue;
      if (r.x & F_HEADER) {
        is_and = (r.x & F_AND) != 0;
        acc = is_and ? A.rule_logw[r.y] : 0.0;
        best = F_NEG_INF;
        first = true;
      } else {
        const double v = ins[(size_t)(r.x & F_IDX) * 64];
        if (is_and)
          acc += v;
        else if (first || best < v)
          best = v;
        first = false;
      }
      if (r.x & F_LAST) {
        ins[(size_t)d * 64] = is_and ? acc : best;
        ++d;
      }
    }
  }
  best_logprob[forest] = ins[(size_t)(n - 1) * 64];
  const uint32_t* __restrict__ hp = A.hdr_pos + g.stream_base + lane;
  uint32_t* outr = A.sample_rules + A.sample_off[forest];
  uint32_t* outa = A.sample_hdr + A.sample_off[forest];
  const uint32_t cap = (uint32_t)(A.sample_off[forest + 1] - A.sample_off[forest]);
  uint32_t* stack = outr + cap;  // deep part of the stack: stack[-1 - i]
  uint32_t* stk_sh = (uint32_t*)aux + lane;
#define FSTACK_PUSH(v)                                       \
  {                                                          \
    if (sp < stack_lds)                                      \
      stk_sh[(size_t)sp * 64] = (v);                         \
    else                                                     \
      stack[-(int)(sp - stack_lds) - 1] = (v);               \
    ++sp;                                                    \
  }
  uint32_t sp = 0, ns = 0;
  FSTACK_PUSH(n - 1)
  while (sp) {
    --sp;
    const uint32_t node = sp < stack_lds ? stk_sh[(size_t)sp * 64] : stack[-(int)(sp - stack_lds) - 1];
    const uint32_t h = hp[(size_t)node * 64];
    const uint2 hr = st[(size_t)h * 64];
    uint32_t nch = (hr.x >> 20) & 0xffu;
    if (nch == 255u) {
      nch = 0;
      for (uint32_t k = h + 1;; ++k) {
        ++nch;
        if (st[(size_t)k * 64].x & F_LAST) break;
      }
    }
    if (hr.x & F_AND) {
      if (ns < max_sample) {
        outr[ns] = hr.y;
        outa[ns] = nch;
      }
      ++ns;
      for (uint32_t k = nch; k-- > 0;) FSTACK_PUSH(st[(size_t)(h + 1 + k) * 64].x & F_IDX)
    } else {
      uint32_t pick = 0;
      double best = ins[(size_t)(st[(size_t)(h + 1) * 64].x & F_IDX) * 64];
      for (uint32_t k = 1; k < nch; ++k) {
        const double v = ins[(size_t)(st[(size_t)(h + 1 + k) * 64].x & F_IDX) * 64];
        if (best < v) {
          best = v;
          pick = k;
        }
      }
      FSTACK_PUSH(st[(size_t)(h + 1 + pick) * 64].x & F_IDX)
    }
  }
#undef FSTACK_PUSH
  A.sample_len[forest] = ns < max_sample ? ns : max_sample;
}

// counts of a sweep's samples: x[rule] += 1, normsum[group] += 1 per use (the caller starts from the priors).
// A popular rule is used by a large share of the forests (the rule ids of real grammars, and of config 5, are Zipf
// distributed) and adds to one address serialise (~9 ns each: 10^5 uses of one rule = 1 ms), so a workgroup first
// counts in two LDS tables (slot = hash of the id, claimed by the first id that arrives; an id that finds its
// slot taken by another goes straight to global memory) and adds each claimed slot to global memory once.
// First formulation: 16 lanes per forest, a workgroup covers 64 forests at a time.  Second formulation (sweep2): the
// entries of 256 forests dealt out evenly; it also writes the sample's rule ids and class words (from the records at
// the positions the sampler left) and adds up the sample's ln proposal probability.
#define FRC_FORESTS 256u
__global__ __launch_bounds__(1024) void forest_recount_kernel(const uint64_t* sample_off, const uint32_t* sample_len,
                                                              uint32_t* rules, const uint32_t* p_norm, double* x,
                                                              double* normsum, uint32_t n_forests, ForestArgs A, int sweep2,
                                                              const uint32_t* slot_forest, uint32_t slot0, uint32_t slot1,
                                                              uint32_t n_slots0, uint32_t n_slots1, const uint32_t* node_hdr,
                                                              const double* node_prob, const uint4* node_slots) {
  // the two tables in dynamic LDS: n_slots0 {key, count} pairs for rules, n_slots1 for norm groups (powers of two)
  extern __shared__ uint32_t frc_lds[];
  uint32_t* const key0 = frc_lds;
  uint32_t* const cnt0 = key0 + n_slots0;
  uint32_t* const key1 = cnt0 + n_slots0;
  uint32_t* const cnt1 = key1 + n_slots1;
  __shared__ double cheap_sh[16];
  const bool count_here = !(sweep2 & 2);  // (bit 1: the counts are gathered from the nodes' use counts, forest_rule_gather_kernel)
  for (uint32_t i = threadIdx.x; i < n_slots0 && count_here; i += 1024) {
    key0[i] = 0xffffffffu;
    cnt0[i] = 0u;
  }
  for (uint32_t i = threadIdx.x; i < n_slots1 && count_here; i += 1024) {
    key1[i] = 0xffffffffu;
    cnt1[i] = 0u;
  }
  __syncthreads();
  auto add = [&](int t, uint32_t id, double* g) {
    if (!count_here) return;
    uint32_t* const key = t ? key1 : key0;
    uint32_t* const cnt = t ? cnt1 : cnt0;
    const uint32_t slot = (id * 2654435761u >> 9) & ((t ? n_slots1 : n_slots0) - 1);
    const uint32_t old = atomicCAS(&key[slot], 0xffffffffu, id);
    if (old == 0xffffffffu || old == id)
      atomicAdd(&cnt[slot], 1u);
    else
      unsafeAtomicAdd(g + id, 1.0);
  };
  double cheap = 0.0;
  // slot_forest: the forests of the lane slots [slot0, slot1) -- one launch class, recounted as soon as its sample kernel is
  // done, beside the other classes still sampling; otherwise all forests in corpus order
  const uint32_t i_begin = slot_forest ? slot0 : 0u, i_end = slot_forest ? slot1 : n_forests;
  if (sweep2) {
    // The entries of FRC_FORESTS forests at a time, dealt out evenly: a prefix sum of the sample lengths in LDS, entry e
    // belongs to the forest whose range holds it (binary search).  (Sixteen lanes per forest, forest after forest, left
    // a workgroup waiting for its largest sample four times in a row: 75-110 us for the classes of large forests.)
    __shared__ uint32_t pre[FRC_FORESTS + 1];
    __shared__ uint32_t fid[FRC_FORESTS];
    __shared__ uint32_t nbase[FRC_FORESTS];  // node_hdr: the forest's first node in node_hdr / node_prob
    for (uint32_t f0 = i_begin + blockIdx.x * FRC_FORESTS; f0 < i_end; f0 += gridDim.x * FRC_FORESTS) {
      __syncthreads();
      if (threadIdx.x < FRC_FORESTS) {
        const uint32_t i = f0 + threadIdx.x;
        const uint32_t f = i < i_end ? (slot_forest ? slot_forest[i] : i) : 0xffffffffu;
        fid[threadIdx.x] = f;
        pre[threadIdx.x + 1] = f == 0xffffffffu ? 0u : sample_len[f];
        if (node_hdr && f != 0xffffffffu) {
          const uint4 d0 = node_slots[2 * (size_t)(slot_forest ? i : A.lane_of_forest[f])];
          nbase[threadIdx.x] = (uint32_t)(((((uint64_t)d0.w) << 32) | d0.z) >> 2);
        }
      }
      if (threadIdx.x == 0) pre[0] = 0;
      __syncthreads();
      for (uint32_t o = 1; o < FRC_FORESTS; o <<= 1) {  // inclusive scan of pre[1 ..]
        uint32_t v = 0;
        if (threadIdx.x < FRC_FORESTS && threadIdx.x >= o) v = pre[threadIdx.x + 1 - o];
        __syncthreads();
        if (threadIdx.x < FRC_FORESTS) pre[threadIdx.x + 1] += v;
        __syncthreads();
      }
      const uint32_t total = pre[FRC_FORESTS];
      if (node_hdr) {
        // the sampler wrote NODE numbers: a sampled rule's id, class word and norm group are one 16-byte record of the forest's
        // header table, its probability one double beside it -- the forest's own lines, fetched once for all its entries
        for (uint32_t e0 = threadIdx.x; e0 < total; e0 += 4 * 1024) {
          size_t so[4];
          uint32_t nd[4];
          uint4 h[4];
          double pv[4];
          bool ok[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t e = e0 + 1024u * q;
            ok[q] = e < total;
            uint32_t lo = 0, hi = FRC_FORESTS;
            while (hi - lo > 1) {
              const uint32_t mid = (lo + hi) >> 1;
              if (pre[mid] <= (ok[q] ? e : 0u))
                lo = mid;
              else
                hi = mid;
            }
            so[q] = ok[q] ? sample_off[fid[lo]] + (e - pre[lo]) : 0u;
            nd[q] = ok[q] ? nbase[lo] : 0u;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) nd[q] += ok[q] ? A.sample_hdr[so[q]] : 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            h[q] = ((const uint4*)node_hdr)[nd[q]];
            pv[q] = node_prob[nd[q]];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            rules[so[q]] = h[q].y;
            A.sample_cls[so[q]] = h[q].w == F_NONORM ? 0xffffffffu : h[q].z;
            cheap += log(pv[q]);
            if (h[q].w == F_NONORM) continue;
            add(0, h[q].y, x);
            add(1, h[q].w, normsum);
          }
        }
      } else
      for (uint32_t e0 = threadIdx.x; e0 < total; e0 += 4 * 1024) {  // four entries per thread in flight
        size_t pos[4], so[4];
        uint32_t rule[4], nn[4], c[4];
        double lpv[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t e = e0 + 1024u * q;
          ok[q] = e < total;
          uint32_t lo = 0, hi = FRC_FORESTS;  // the last forest slot whose range starts at or before e
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pre[mid] <= (ok[q] ? e : 0u))
              lo = mid;
            else
              hi = mid;
          }
          const uint32_t f = ok[q] ? fid[lo] : 0u;  // (a range that holds an entry belongs to a real forest)
          const uint32_t slot = ok[q] ? (slot_forest ? f0 + lo : A.lane_of_forest[f]) : 0u;
          const FGroup g = A.groups[slot >> 6];
          so[q] = sample_off[f] + (ok[q] ? e - pre[lo] : 0u);
          pos[q] = g.stream_base + (slot & 63u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) pos[q] += (size_t)(ok[q] ? A.sample_hdr[so[q]] : 0u) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rule[q] = A.ins_stream[pos[q]].y;
          lpv[q] = A.p_only ? A.rec_p[pos[q]] : A.rec_logp[pos[q]];
          c[q] = A.rec_cls[pos[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) nn[q] = ok[q] ? p_norm[rule[q]] : F_NONORM;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!ok[q]) continue;
          rules[so[q]] = rule[q];
          A.sample_cls[so[q]] = nn[q] == F_NONORM ? 0xffffffffu : c[q];
          cheap += A.p_only ? log(lpv[q]) : lpv[q];
          if (nn[q] == F_NONORM) continue;
          add(0, rule[q], x);
          add(1, nn[q], normsum);
        }
      }
    }
  } else
  for (uint32_t f0 = i_begin + blockIdx.x * 64; f0 < i_end; f0 += gridDim.x * 64) {
    const uint32_t i = f0 + (threadIdx.x >> 4);
    if (i >= i_end) continue;
    const uint32_t f = slot_forest ? slot_forest[i] : i;
    if (f == 0xffffffffu) continue;  // an empty lane slot
    const uint64_t so = sample_off[f];
    const uint32_t* r = rules + so;
    const uint32_t len = sample_len[f];
    for (uint32_t k = threadIdx.x & 15u; k < len; k += 16) {
      const uint32_t rule = r[k], nn = p_norm[rule];
      if (nn == F_NONORM) continue;
      add(0, rule, x);
      add(1, nn, normsum);
    }
  }
  if (sweep2) {
    for (int o = 32; o > 0; o >>= 1) cheap += __shfl_down(cheap, o, 64);
    if ((threadIdx.x & 63) == 0) cheap_sh[threadIdx.x >> 6] = cheap;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_slots0 && count_here; i += 1024)
    if (cnt0[i]) unsafeAtomicAdd(x + key0[i], (double)cnt0[i]);
  for (uint32_t i = threadIdx.x; i < n_slots1 && count_here; i += 1024)
    if (cnt1[i]) unsafeAtomicAdd(normsum + key1[i], (double)cnt1[i]);
  if (sweep2 && threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += cheap_sh[i];
    unsafeAtomicAdd(A.iter_out + 1, t);
  }
}
// ---- the parallel sweep's counts WITHOUT atomics (round 6).  The recount above ends in one device-scope atomic per distinct
// rule and norm group of every 256 forests -- ~4 M a sweep on config 5, most of them rules of the Zipf tail that no LDS table
// folds -- and those atomics, not its reads, are what it takes 85 us for.  But which nodes carry a rule never changes: the
// sampler leaves per node how often it recorded it (FMultiArgs::node_cnt, two bytes a node, every node written every sweep),
// and a rule's uses are the sum over ITS nodes (inv_off / inv_node, built with the forests): a gather through a static index,
// one thread a rule; rules on more than FRG_COLD nodes in pieces of FRG_PIECE, a workgroup each, added with integer atomics
// (a few hundred a sweep).  The norm groups' sums are sums over their rules' integers; both meet their priors in ONE rounding
// (prior + uses), whatever the order the samples came in -- the atomics' sums depended on it.
#define FRG_COLD 8u      // a rule on at most so many nodes: one thread, its loads side by side
#define FRG_PIECE 512u   // other rules: pieces of so many nodes, a wavefront each (eight loads a lane, side by side)
__global__ __launch_bounds__(256) void forest_rule_gather_kernel(const uint32_t* __restrict__ inv_off, const uint32_t* __restrict__ inv_node,
                                                                 const uint16_t* __restrict__ node_cnt, uint32_t* __restrict__ rule_cnt,
                                                                 uint32_t n_rules, const uint32_t* __restrict__ pieces, uint32_t n_pieces,
                                                                 uint32_t cold_blocks, uint32_t n_inv) {
  if (blockIdx.x >= cold_blocks) {  // four pieces a workgroup: {rule, first, end} each
    const uint32_t pc = (blockIdx.x - cold_blocks) * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (pc >= n_pieces) return;
    const uint32_t r = pieces[3 * pc], j0 = pieces[3 * pc + 1], j1 = pieces[3 * pc + 2];
    uint32_t nd[FRG_PIECE / 64], c = 0;
#pragma unroll
    for (int q = 0; q < (int)(FRG_PIECE / 64); ++q) nd[q] = j0 + lane + 64u * q < j1 ? inv_node[j0 + lane + 64u * q] : 0xffffffffu;
#pragma unroll
    for (int q = 0; q < (int)(FRG_PIECE / 64); ++q)
      if (nd[q] != 0xffffffffu) c += node_cnt[nd[q]];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (lane == 0 && c) atomicAdd(rule_cnt + r, c);
    return;
  }
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rules) return;
  const uint32_t j0 = inv_off[r], j1 = inv_off[r + 1];
  if (j1 - j0 > FRG_COLD) return;  // (its pieces add into rule_cnt[r], zero since the last sweep's apply)
  if (j1 == j0) return;  // (a rule on no node: its count stays zero)
  uint32_t nd[FRG_COLD], c = 0;
#pragma unroll
  for (int q = 0; q < (int)FRG_COLD; ++q) nd[q] = j0 + (uint32_t)q < j1 ? inv_node[j0 + (uint32_t)q] : 0xffffffffu;
#pragma unroll
  for (int q = 0; q < (int)FRG_COLD; ++q)
    if (nd[q] != 0xffffffffu) c += node_cnt[nd[q]];
  rule_cnt[r] = c;
}
// the norm groups' sums from their rules' use counts (the rules' own new counts: forest_commit_kernel, in rule order).  Eight
// lanes a group (config 5: eight rules a group on average).
__global__ __launch_bounds__(256) void forest_group_sum_kernel(const uint64_t* __restrict__ group_off, const uint32_t* __restrict__ group_rule,
                                                               uint64_t n_groups, const uint32_t* __restrict__ rule_cnt,
                                                               const double* __restrict__ prior_norm, double* __restrict__ normsum) {
  const uint64_t g = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const uint32_t li = threadIdx.x & 7u;
  const bool on = g < n_groups;
  const uint64_t j0 = on ? group_off[g] : 0, j1 = on ? group_off[g + 1] : 0;
  uint32_t sum = 0;
  for (uint64_t j = j0 + li; j < j1; j += 8) sum += rule_cnt[group_rule[j]];
  sum += __shfl_xor(sum, 1, 8);
  sum += __shfl_xor(sum, 2, 8);
  sum += __shfl_xor(sum, 4, 8);
  if (on && li == 0) normsum[g] = prior_norm[g] + (double)sum;
}
// reset_x / reset_norm (may be null): the count buffers of the NEXT sweep start from the priors -- set here, by the thread
// that has just read the slot, instead of two copies behind the kernel
__global__ void forest_commit_kernel(double* new_x, double* p_x, double* p_s, double* p_tmax, const uint32_t* p_norm,
                                     double time, uint64_t n, const double* reset_x, double* next_norm,
                                     const double* reset_norm, uint64_t n_norm, uint32_t* rule_cnt = nullptr,
                                     const double* prior = nullptr) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_norm && next_norm;
       p += (uint64_t)gridDim.x * blockDim.x)
    next_norm[p] = reset_norm[p];
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
    if (p_norm[p] == F_NONORM) continue;
    double nx;
    if (rule_cnt) {  // the sweep's use counts, gathered (forest_rule_gather_kernel): the new count in one rounding; cleared for the next sweep
      nx = prior[p] + (double)rule_cnt[p];
      rule_cnt[p] = 0u;
    } else {
      nx = new_x[p];
      if (reset_x) new_x[p] = reset_x[p];
    }
    const double d = nx - p_x[p];
    const double moret = time - p_tmax[p];
    if (moret > 0) {
      p_tmax[p] = time;
      p_s[p] += moret * p_x[p];
    } else if (moret < 0)
      p_s[p] += d * (-moret);
    p_x[p] += d;
  }
}

// M-step (normalize.hpp:123-164): per group w = (count) / (sum + add_k); zero-count group -> uniform or zero
__global__ void forest_mstep_kernel(double* rule_logw, const double* counts, double prior, const uint64_t* group_off,
                                    const uint32_t* group_rule, uint64_t n_groups, double add_k, int zero_zero,
                                    unsigned long long* max_bits) {
  double mx = 0.0;
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t j0 = group_off[g], j1 = group_off[g + 1];
    double sum = 0.0;
    for (uint64_t j = j0; j < j1; ++j) sum += counts[group_rule[j]] + prior;
    for (uint64_t j = j0; j < j1; ++j) {
      const uint32_t r = group_rule[j];
      double nw;
      if (sum > 0.0) {
        const double c = counts[r] + prior;
        nw = c > 0.0 ? log(c / (sum + add_k)) : F_NEG_INF;
      } else
        nw = zero_zero ? F_NEG_INF : -log((double)(j1 - j0));
      mx = fmax(mx, fabs(exp(nw) - exp(rule_logw[r])));
      rule_logw[r] = nw;
    }
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(max_bits, (unsigned long long)__double_as_longlong(mx));
}

}  // namespace carmel_hip

using namespace carmel_hip;

static const size_t F_LDS_LIMIT = 150 * 1024;  // dynamic LDS a forest kernel may ask for

struct carmel_hip_forests {
  uint32_t best_run = 0;  // --crp-restarts: the run that was kept (carmel_hip_forests_best_run)
  std::vector<double> h_final_x;  // ... its counts as finalize_cumulative_counts left them (carmel_hip_forests_final_counts)
  // --prior-inference-* (gibbs_opts.hpp:82-89): carmel_hip_forests_set_prior_inference / _prior_trace
  double pi_stddev = 0;
  bool pi_global = false, pi_local = false;
  uint32_t pi_start = 0, pi_end = 0;
  std::vector<double> pi_trace, pi_cumulative;
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t n_forests = 0, n_groups = 0;
  uint32_t n_rules = 0, max_nodes = 0, max_sample = 0;
  uint64_t node_total = 0, stream_total = 0;
  static const int N_SIDE = 3;  // + the caller's stream (carmel_hip_forests_create: how they come by hardware queues of their own)
  std::vector<int> class_side;  // per launch class: -1 = the caller's stream, k = side[k] (dealt by load, largest class first)
  std::vector<int> sweep_side;     // ... for the several-lanes sampler's sweep (sweep_stream): dealt by the classes' LONGEST forest
  std::vector<size_t> sweep_order;  // ... and the order they are launched in (the class of the largest forests first)
  hipStream_t side[N_SIDE] = {};  // launch classes of one sweep run side by side
  hipEvent_t ev_fork = nullptr, ev_side[N_SIDE] = {}, ev_samp[N_SIDE] = {};  // (ev_samp: a side stream's samplers are done)
  bool sweep2_ok = false;  // the second formulation of the parallel sweep applies (class ids fit 16 bits)
  // several lanes per forest (forest_sample_multi_kernel): per-forest tables, per lane slot
  bool multi_ok = false;
  DevBuf<uint16_t> mt_tab;
  DevBuf<uint32_t> mt_hdr;
  DevBuf<uint32_t> mt_slots;  // FMultiArgs::slots
  DevBuf<double> mt_prob;     // FMultiArgs::prob
  DevBuf<uint16_t> mt_node_cnt;                         // FMultiArgs::node_cnt
  DevBuf<uint32_t> inv_off, inv_node, inv_pieces, rule_cnt;  // forest_rule_gather_kernel: rule -> its AND nodes (indices into mt_hdr / 4)
  uint32_t n_inv_pieces = 0;
  DevBuf<uint32_t> x_desc, x_rec;  // forest_exact_kernel's per-forest descriptors and per-node records (forest_exact.hpp)
  std::vector<FGroup> h_groups;
  struct Cls {
    uint32_t first, count, max_nodes;
    uint32_t max_kids = 0, maxlen = 0;  // child entries / records of the class's largest lane (LDS walk tables)
    uint32_t m_tab = 0, m_n = 0, m_front = 0;  // forest_sample_multi_kernel: table words / nodes / frontier entries of its largest forest
  };
  std::vector<Cls> classes;
  std::vector<uint32_t> h_norm, lane_of_forest;
  std::vector<double> h_alphas;  // --alpha=FILE: per-rule prior strength, negative = locked (empty: the scalar alpha)
  std::vector<uint64_t> h_group_off;
  std::vector<uint32_t> h_group_rule;
  std::vector<uint64_t> h_sample_off;
  DevBuf<FGroup> groups;
  DevBuf<uint2_t> ins_stream, out_stream;
  DevBuf<uint32_t> lane_forest, lane_nodes, hdr_pos, group_rule, p_norm, sample_len[2], sample_rules[2];
  DevBuf<uint32_t> rec_cls, sample_cls, sample_hdr, lane_of_forest_d;
  DevBuf<FAnd> and_list;
  uint64_t n_and = 0;
  DevBuf<double> gcol;               // columns of the launch classes whose forests do not fit LDS
  std::vector<uint64_t> gcol_off;    // per class: offset into gcol (doubles), room for two columns per group
  DevBuf<double> rec_logp, rec_p;
  DevBuf<uint64_t> group_off, arc_off, slot_pos, hot_chunks, sample_off;
  DevBuf<double> normsum2;  // the norm sums being recounted while a sweep still reads the current ones (carmel_hip_forests_gibbs)
  DevBuf<double> rule_logw, counts, post, forest_logprob, scalars, p_prior, p_x, p_s, p_tmax, normsum, prior_norm, new_x,
      iter_out;
  DevBuf<unsigned long long> maxbits;
};

extern "C" {

int carmel_hip_forests_create(carmel_hip_forests** out, int device, uint64_t n_forests, const uint64_t* node_off,
                              const uint32_t* label, const int32_t* ref, const uint32_t* next, uint32_t n_rules,
                              const double* rule_logw, uint64_t n_groups, const uint64_t* group_off,
                              const uint32_t* group_rule) {
  if (!out || !node_off || !label || !ref || !next || !rule_logw) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(CARMEL_HIP_ERR_HIP, "no HIP device: forest-em has no CPU fallback here");
  if (device < 0 || device >= ndev) return fail(CARMEL_HIP_ERR_ARG, "bad device index");
  HIPCHK(hipSetDevice(device));
  std::unique_ptr<carmel_hip_forests> F(new carmel_hip_forests());
  F->device = device;
  F->n_forests = n_forests;
  F->n_rules = n_rules;
  F->n_groups = n_groups;
  HIPCHK(hipStreamCreateWithFlags(&F->stream, hipStreamNonBlocking));
  // (the side streams right behind it: four streams created in a row land on four different hardware queues)
  HIPCHK(hipEventCreateWithFlags(&F->ev_fork, hipEventDisableTiming));
  // The side streams belong to the HIGH priority class -- not for the priority: the runtime keeps a pool of hardware queues per
  // priority class (four each), and streams of the default class share theirs with every other stream of the process (torch's,
  // a trainer's): in bench.py's full run two launch classes landed on one queue and ran one after the other, 0.41 ms per sweep
  // against 0.33 on its own.  In a class of their own the three get a queue each: 0.33 in both places.
  {
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for (int k = 0; k < carmel_hip_forests::N_SIDE; ++k) {
      HIPCHK(hipStreamCreateWithPriority(&F->side[k], hipStreamNonBlocking, prio_hi));
      HIPCHK(hipEventCreateWithFlags(&F->ev_side[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&F->ev_samp[k], hipEventDisableTiming));
    }
  }
  hipStream_t s = F->stream;
  // ---- per forest: post-order over non-reference nodes, streams ----
  struct Flat {
    std::vector<uint2_t> ins, outs;
    std::vector<uint32_t> hdr;  // per post-order node: header position in ins
    uint32_t n = 0;
    uint64_t max_deriv = 0;     // rules in the largest derivation (shared sub-forests count once per use)
    std::vector<uint16_t> mt;   // forest_sample_multi_kernel's table block (FMultiArgs::tab); empty: the forest does not fit it
    std::vector<uint32_t> mh;   // ... header row | AND per node
    uint32_t m_front = 0;       // ... entries of its widest breadth-first frontier (bounded by the largest derivation)
    std::vector<uint16_t> m_ord;  // ... its nodes by height: sampler's node id -> node
  };
  std::vector<Flat> flat(n_forests);
  for (uint64_t f = 0; f < n_forests; ++f) {
    const uint64_t b = node_off[f], e = node_off[f + 1];
    const uint32_t N = (uint32_t)(e - b);
    if (!N) return fail(CARMEL_HIP_ERR_ARG, "empty forest");
    std::vector<uint32_t> pi(N, 0xffffffffu), order;
    // post-order = nodes sorted by (end of subtree ascending, start descending); references are skipped
    std::vector<uint32_t> idx;
    for (uint32_t i = 0; i < N; ++i) {
      if (next[b + i] <= i || next[b + i] > N) return fail(CARMEL_HIP_ERR_ARG, "bad forest node extent");
      if (ref[b + i] >= 0) {
        if ((uint32_t)ref[b + i] >= i) return fail(CARMEL_HIP_ERR_ARG, "forest back-reference must point backwards");
        continue;
      }
      if (label[b + i] >= n_rules && label[b + i] != 0) return fail(CARMEL_HIP_ERR_ARG, "rule id out of range");
      idx.push_back(i);
    }
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
      if (next[b + x] != next[b + y]) return next[b + x] < next[b + y];
      return x > y;
    });
    for (uint32_t k = 0; k < idx.size(); ++k) pi[idx[k]] = k;
    Flat& fl = flat[f];
    fl.n = (uint32_t)idx.size();
    if (fl.n > F_IDX) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "forest too large");
    auto resolve = [&](uint32_t c) {
      while (ref[b + c] >= 0) c = (uint32_t)ref[b + c];
      return pi[c];
    };
    std::vector<std::vector<uint32_t> > kids(fl.n), kid_ref(fl.n);  // kid_ref: 1 = reached through a back-reference
    for (uint32_t k = 0; k < fl.n; ++k) {
      uint32_t i = idx[k];
      for (uint32_t c = i + 1; c < next[b + i]; c = next[b + c]) {
        kids[k].push_back(resolve(c));
        kid_ref[k].push_back(ref[b + c] >= 0 ? 1u : 0u);
      }
    }
    fl.hdr.resize(fl.n);
    for (uint32_t k = 0; k < fl.n; ++k) {
      uint32_t i = idx[k];
      bool is_and = label[b + i] != 0;
      fl.hdr[k] = (uint32_t)fl.ins.size();
      // header word: flags | child count in bits 20..27 (255 = "255 or more: count the records") | node order k
      uint32_t hx = F_HEADER | F_VALID | (is_and ? F_AND : 0u) | (kids[k].empty() ? F_LAST : 0u) |
                    ((uint32_t)std::min<size_t>(kids[k].size(), 255) << 20) | (k & 0xfffffu);
      fl.ins.push_back(uint2_t{hx, label[b + i]});
      for (size_t c = 0; c < kids[k].size(); ++c)
        fl.ins.push_back(uint2_t{F_VALID | (c + 1 == kids[k].size() ? F_LAST : 0u) | kids[k][c], kid_ref[k][c] << 31});
    }
    // second word of a child record: stream position of the child's header | bit 31 = reached through a back-reference
    for (uint32_t k = 0; k < fl.n; ++k)
      for (size_t c = 0; c < kids[k].size(); ++c) fl.ins[fl.hdr[k] + 1 + c].y |= fl.hdr[kids[k][c]];
    for (uint32_t k = fl.n; k-- > 0;) {
      uint32_t i = idx[k];
      bool is_and = label[b + i] != 0;
      fl.outs.push_back(uint2_t{F_HEADER | F_VALID | (is_and ? F_AND : 0u) | k, label[b + i]});
      for (uint32_t c : kids[k]) fl.outs.push_back(uint2_t{F_VALID | c, 0u});
    }
    if (!(label[b + 0] == 0 || ref[b + 0] < 0)) return fail(CARMEL_HIP_ERR_ARG, "forest root cannot be a reference");
    std::vector<uint64_t> dsz(fl.n, 0);
    for (uint32_t k = 0; k < fl.n; ++k) {
      uint64_t v = 0;
      if (label[b + idx[k]] != 0) {
        v = 1;
        for (uint32_t c : kids[k]) v += dsz[c];
      } else
        for (uint32_t c : kids[k]) v = std::max(v, dsz[c]);
      dsz[k] = std::min<uint64_t>(v, 1u << 20);
    }
    fl.max_deriv = std::max<uint64_t>(1, dsz[fl.n - 1]);
    if (fl.max_deriv >= (1u << 20)) return fail(CARMEL_HIP_ERR_ARG, "forest derivation larger than 2^20 rules");
    {
      // tables of the several-lanes-per-forest sampler: nodes by height (leaves 0; a node is above all of its children,
      // reached directly or through a back-reference), children lists, header rows
      size_t nk = 0;
      for (uint32_t k = 0; k < fl.n; ++k) nk += kids[k].size();
      // the widest frontier a breadth-first walk can reach: w[d][k] = most entries d levels below node k (an AND node hands on
      // all of its children, an OR node the widest of them); depth by depth until nothing is left
      uint64_t front = 1;
      {
        std::vector<uint64_t> w(fl.n, 1), w2(fl.n);
        for (uint32_t depth = 0; depth < 4096; ++depth) {
          bool any = false;
          for (uint32_t k = 0; k < fl.n; ++k) {  // (children have smaller ids: w of the previous depth is complete)
            uint64_t v = 0;
            if (label[b + idx[k]] != 0)
              for (uint32_t c : kids[k]) v += w[c];
            else
              for (uint32_t c : kids[k]) v = std::max(v, w[c]);
            w2[k] = std::min<uint64_t>(v, 1u << 20);
            any = any || v;
          }
          w.swap(w2);
          front = std::max(front, w[fl.n - 1]);
          if (!any) break;
        }
        front += 1;
      }
      if (fl.n < 0x7fffu && nk < 0x7fffu && front < 4096) {
        std::vector<uint32_t> height(fl.n, 0);
        uint32_t Hh = 0;
        for (uint32_t k = 0; k < fl.n; ++k) {  // post-order: children first
          uint32_t hh = 0;
          for (uint32_t c : kids[k]) hh = std::max(hh, height[c] + 1);
          height[k] = hh;
          Hh = std::max(Hh, hh + 1);
        }
        std::vector<uint16_t>& mt = fl.mt;
        mt.assign(4, 0);
        mt[0] = (uint16_t)fl.n;
        mt[1] = (uint16_t)Hh;
        mt[2] = (uint16_t)nk;
        std::vector<uint32_t> cnt(Hh + 1, 0);
        for (uint32_t k = 0; k < fl.n; ++k) cnt[height[k] + 1]++;
        for (uint32_t h = 0; h < Hh; ++h) cnt[h + 1] += cnt[h];
        for (uint32_t h = 0; h <= Hh; ++h) mt.push_back((uint16_t)cnt[h]);
        // the sampler numbers the nodes BY HEIGHT (stable: the root, alone at the top, stays last): the nodes of a height are a
        // range of ids, a node's children have smaller ids
        std::vector<uint16_t>& ordv = fl.m_ord;
        ordv.assign(fl.n, 0);
        std::vector<uint16_t> newid(fl.n);
        {
          std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
          for (uint32_t k = 0; k < fl.n; ++k) {
            newid[k] = (uint16_t)cur[height[k]];
            ordv[cur[height[k]]++] = (uint16_t)k;
          }
        }
        uint32_t off = 0;
        for (uint32_t q = 0; q < fl.n; ++q) {
          mt.push_back((uint16_t)off);
          off += (uint32_t)kids[ordv[q]].size();
        }
        mt.push_back((uint16_t)off);
        for (uint32_t q = 0; q < fl.n; ++q) {
          const uint32_t k = ordv[q];
          for (size_t c = 0; c < kids[k].size(); ++c) mt.push_back((uint16_t)(newid[kids[k][c]] | (kid_ref[k][c] ? 0x8000u : 0u)));
        }
        fl.mh.assign((size_t)4 * fl.n, 0u);  // (class words and norm groups follow once they are known)
        for (uint32_t q = 0; q < fl.n; ++q) fl.mh[4 * q] = fl.hdr[ordv[q]] | (label[b + idx[ordv[q]]] != 0 ? 0x80000000u : 0u);
        fl.m_front = (uint32_t)front;
      }
    }
  }
  // ---- groups of 64, sorted by stream length ----
  std::vector<uint32_t> ord(n_forests);
  std::iota(ord.begin(), ord.end(), 0u);
  std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b2) { return flat[a].ins.size() > flat[b2].ins.size(); });
  const size_t ng = (n_forests + 63) / 64;
  F->h_groups.resize(ng);
  std::vector<uint32_t> lane_forest(ng * 64, 0xffffffffu), lane_nodes(ng * 64, 0);
  F->lane_of_forest.assign(n_forests, 0);
  uint64_t base = 0, node_total = 0;
  for (size_t gidx = 0; gidx < ng; ++gidx) {
    FGroup& G = F->h_groups[gidx];
    std::memset(&G, 0, sizeof G);
    size_t l0 = gidx * 64, l1 = std::min<size_t>(n_forests, l0 + 64);
    G.stream_base = base;
    G.n_lanes = (uint32_t)(l1 - l0);
    G.lane_base = (uint32_t)l0;
    for (size_t l = l0; l < l1; ++l) {
      const Flat& fl = flat[ord[l]];
      G.maxlen = std::max<uint32_t>(G.maxlen, (uint32_t)fl.ins.size());
      G.max_nodes = std::max(G.max_nodes, fl.n);
      lane_forest[l] = ord[l];
      lane_nodes[l] = fl.n;
      F->lane_of_forest[ord[l]] = (uint32_t)l;
    }
    F->max_nodes = std::max(F->max_nodes, G.max_nodes);
    G.node_base = node_total;
    node_total += (uint64_t)G.max_nodes * 64;
    base += (uint64_t)G.maxlen * 64;
  }
  F->node_total = node_total;
  F->stream_total = base;
  std::vector<uint2_t> si(base, uint2_t{0, 0}), so(base, uint2_t{0, 0});
  std::vector<uint32_t> hp(base, 0);
  for (size_t gidx = 0; gidx < ng; ++gidx) {
    const FGroup& G = F->h_groups[gidx];
    for (uint32_t l = 0; l < G.n_lanes; ++l) {
      const Flat& fl = flat[ord[G.lane_base + l]];
      for (size_t k = 0; k < fl.ins.size(); ++k) si[G.stream_base + k * 64 + l] = fl.ins[k];
      for (size_t k = 0; k < fl.outs.size(); ++k) so[G.stream_base + k * 64 + l] = fl.outs[k];
      for (uint32_t k = 0; k < fl.n; ++k) hp[G.stream_base + (size_t)k * 64 + l] = fl.hdr[k];
    }
  }
  {  // launch classes by LDS need: a class ends where the groups have shrunk to 2/3 of its largest, 256 groups at least.
     // Finer classes (4/5, 64 groups: eleven for config 5) pad less LDS but were slower, 0.96 against 0.79 ms per sweep: only
     // four or five kernels run side by side, the rest queue behind them
    const unsigned cls_num = 2, cls_den = 3, cls_min = 256;
    size_t i = 0;
    while (i < ng) {
      uint32_t mx = F->h_groups[i].max_nodes;
      size_t j = i + 1;
      while (j < ng) {
        uint32_t m = F->h_groups[j].max_nodes;
        if (m > mx) mx = m;
        if (j - i >= cls_min && (uint64_t)m * cls_den <= (uint64_t)mx * cls_num) break;
        ++j;
      }
      carmel_hip_forests::Cls c{(uint32_t)i, (uint32_t)(j - i), mx};
      for (size_t q = i; q < j; ++q) {
        const FGroup& G = F->h_groups[q];
        c.maxlen = std::max(c.maxlen, G.maxlen);
        for (uint32_t l = 0; l < G.n_lanes; ++l) {
          const Flat& fl = flat[ord[G.lane_base + l]];
          c.max_kids = std::max<uint32_t>(c.max_kids, (uint32_t)(fl.ins.size() - fl.n));
          c.m_tab = std::max<uint32_t>(c.m_tab, (uint32_t)((fl.mt.size() + 7) / 8 * 8));  // (copied 16 bytes at a time)
          c.m_n = std::max(c.m_n, fl.n);
          c.m_front = std::max(c.m_front, fl.m_front);
        }
      }
      F->classes.push_back(c);
      i = j;
    }
  }
  {  // classes too large for LDS keep their columns in global memory
    uint64_t tot = 0;
    for (auto& c : F->classes) {
      F->gcol_off.push_back(tot);
      if ((size_t)c.max_nodes * 64 * 8 * 2 > F_LDS_LIMIT || lib_opt("forest_gcol")) tot += (uint64_t)c.count * 2 * c.max_nodes * 64;
    }
    if (tot) HIPCHK(F->gcol.alloc(tot));
  }
  // ---- posterior slots grouped by rule (AND headers of the outside stream) ----
  std::vector<uint64_t> cnt((size_t)n_rules + 1, 0);
  for (uint64_t k = 0; k < base; ++k)
    if ((so[k].x & (F_VALID | F_HEADER | F_AND)) == (F_VALID | F_HEADER | F_AND)) cnt[so[k].y + 1]++;
  for (uint32_t r = 0; r < n_rules; ++r) cnt[r + 1] += cnt[r];
  std::vector<uint64_t> arc_off = cnt, slot_pos(cnt[n_rules]), hot;
  for (uint64_t k = 0; k < base; ++k)
    if ((so[k].x & (F_VALID | F_HEADER | F_AND)) == (F_VALID | F_HEADER | F_AND)) slot_pos[cnt[so[k].y]++] = k;
  for (uint32_t r = 0; r < n_rules; ++r)
    if (arc_off[r + 1] - arc_off[r] > 64)
      for (uint64_t j = arc_off[r]; j < arc_off[r + 1]; j += 4096) {
        hot.push_back(r);
        hot.push_back(j);
        hot.push_back(std::min(arc_off[r + 1], j + 4096));
      }
  // ---- normalisation groups ----
  F->h_norm.assign(n_rules, F_NONORM);
  F->h_group_off.assign(group_off, group_off + n_groups + 1);
  F->h_group_rule.assign(group_rule, group_rule + group_off[n_groups]);
  for (uint64_t gi = 0; gi < n_groups; ++gi)
    for (uint64_t j = group_off[gi]; j < group_off[gi + 1]; ++j) {
      if (group_rule[j] >= n_rules) return fail(CARMEL_HIP_ERR_ARG, "normalization group rule id out of range");
      if (F->h_norm[group_rule[j]] != F_NONORM)
        return fail(CARMEL_HIP_ERR_ARG, "a rule occurs in more than one normalization group");
      F->h_norm[group_rule[j]] = (uint32_t)gi;
    }
  // classes of equal rules / equal norm groups within a forest, per AND header record (the parallel sweep's
  // counterfactual counts are kept per class: forest_proposal_kernel)
  std::vector<uint32_t> rc_all(base, 0);
  {
    std::vector<uint32_t>& rc = rc_all;
    F->sweep2_ok = F->max_nodes <= 0xffffu;
    std::unordered_map<uint32_t, uint32_t> rid, gid;
    for (size_t gidx = 0; gidx < ng && F->sweep2_ok; ++gidx) {
      const FGroup& G = F->h_groups[gidx];
      for (uint32_t l = 0; l < G.n_lanes; ++l) {
        const Flat& fl = flat[ord[G.lane_base + l]];
        rid.clear();
        gid.clear();
        for (uint32_t k = 0; k < fl.n; ++k) {
          const uint2_t hr = fl.ins[fl.hdr[k]];
          if (!(hr.x & F_AND)) continue;
          const uint32_t a = rid.emplace(hr.y, (uint32_t)rid.size()).first->second;
          const uint32_t nn = F->h_norm[hr.y];
          const uint32_t b2 = nn == F_NONORM ? 0u : gid.emplace(nn, (uint32_t)gid.size()).first->second;
          rc[G.stream_base + (size_t)fl.hdr[k] * 64 + l] = a | (b2 << 16);
        }
      }
    }
    if (F->sweep2_ok) {
      HIPCHK(F->rec_cls.upload(rc, s));
      std::vector<FAnd> al;  // forest after forest: the threads of a wave scan the same forest's previous sample
      for (size_t gidx = 0; gidx < ng; ++gidx) {
        const FGroup& G = F->h_groups[gidx];
        for (uint32_t l = 0; l < G.n_lanes; ++l)
          for (uint32_t k = 0; k < G.maxlen; ++k) {
            const uint64_t q = G.stream_base + (uint64_t)k * 64 + l;
            const uint2_t r = si[q];
            if ((r.x & (F_VALID | F_HEADER | F_AND)) == (F_VALID | F_HEADER | F_AND))
              al.push_back(FAnd{q, (uint32_t)gidx, rc[q], r.y, lane_forest[G.lane_base + l]});
          }
      }
      F->n_and = al.size();
      HIPCHK(F->and_list.upload(al, s));
    }
    HIPCHK(F->lane_of_forest_d.upload(F->lane_of_forest, s));
  }
  {  // the several-lanes-per-forest tables, in lane-slot order (a wavefront's forests are neighbours)
    F->multi_ok = F->sweep2_ok;
    for (uint64_t f = 0; f < n_forests; ++f)
      if (flat[f].mt.empty()) F->multi_ok = false;
    if (F->multi_ok) {
      std::vector<uint64_t> toff(ng * 64, 0), hoff(ng * 64, 0);
      std::vector<uint32_t> slots(ng * 64 * 8, 0u);
      for (size_t l = 0; l < ng * 64; ++l) slots[8 * l + 6] = 0xffffffffu;
      // (the samples' offsets: capacity = size of the largest derivation of the forest, as below)
      std::vector<uint64_t> so_all(n_forests + 1, 0);
      for (uint64_t f = 0; f < n_forests; ++f) so_all[f + 1] = so_all[f] + flat[f].max_deriv + 2;
      auto F_sample_off_of = [&](uint32_t f) { return so_all[f]; };
      std::vector<uint16_t> tab;
      std::vector<uint32_t> hdrs;
      for (size_t l = 0; l < ng * 64; ++l) {
        if (lane_forest[l] == 0xffffffffu) continue;
        const Flat& fl = flat[lane_forest[l]];
        toff[l] = tab.size();
        hoff[l] = hdrs.size();
        tab.insert(tab.end(), fl.mt.begin(), fl.mt.end());
        tab.resize((tab.size() + 7) / 8 * 8, 0);  // (copied 16 bytes at a time)
        {
          const uint64_t so = F_sample_off_of(lane_forest[l]);
          uint32_t* d = &slots[8 * l];
          d[0] = (uint32_t)toff[l];
          d[1] = (uint32_t)(toff[l] >> 32);
          d[2] = (uint32_t)hoff[l];
          d[3] = (uint32_t)(hoff[l] >> 32);
          d[4] = (uint32_t)so;
          d[5] = (uint32_t)(so >> 32);
          d[6] = lane_forest[l];
          d[7] = fl.n | ((uint32_t)fl.mt.size() << 15);
        }
        hdrs.insert(hdrs.end(), fl.mh.begin(), fl.mh.end());
        const FGroup& G = F->h_groups[l / 64];
        for (uint32_t q = 0; q < fl.n; ++q) {  // rule, class word (rec_cls) and norm group of the node's header record
          const uint32_t k = fl.m_ord[q];
          const uint2_t hr = fl.ins[fl.hdr[k]];
          if (!(hr.x & F_AND)) continue;
          uint32_t* w = &hdrs[hoff[l] + 4 * (size_t)q];
          w[1] = hr.y;
          w[2] = rc_all[G.stream_base + (size_t)fl.hdr[k] * 64 + (l % 64)];
          w[3] = F->h_norm[hr.y];
        }
      }
      HIPCHK(F->mt_tab.upload(tab, s));
      HIPCHK(F->mt_hdr.upload(hdrs, s));
      HIPCHK(F->mt_slots.upload(slots, s));
      if (hdrs.size() / 4 < 0xffffffffull) {  // rule -> the AND nodes that carry it (rules inside a normalisation group: the counted ones)
        const size_t nn_ = hdrs.size() / 4;
        std::vector<uint32_t> ioff((size_t)n_rules + 1, 0u);
        for (size_t q = 0; q < nn_; ++q)
          if ((hdrs[4 * q] & 0x80000000u) && hdrs[4 * q + 3] != F_NONORM) ioff[(size_t)hdrs[4 * q + 1] + 1]++;
        for (uint32_t r = 0; r < n_rules; ++r) ioff[r + 1] += ioff[r];
        std::vector<uint32_t> inode(ioff[n_rules]), fill(ioff.begin(), ioff.end() - 1), pieces;
        for (size_t q = 0; q < nn_; ++q)
          if ((hdrs[4 * q] & 0x80000000u) && hdrs[4 * q + 3] != F_NONORM) inode[fill[hdrs[4 * q + 1]]++] = (uint32_t)q;
        for (uint32_t r = 0; r < n_rules; ++r)
          if (ioff[r + 1] - ioff[r] > FRG_COLD)
            for (uint32_t j = ioff[r]; j < ioff[r + 1]; j += FRG_PIECE) {
              pieces.push_back(r);
              pieces.push_back(j);
              pieces.push_back(std::min(ioff[r + 1], j + FRG_PIECE));
            }
        F->n_inv_pieces = (uint32_t)(pieces.size() / 3);
        if (pieces.empty()) pieces.assign(3, 0u);
        if (inode.empty()) inode.assign(1, 0u);
        HIPCHK(F->inv_off.upload(ioff, s));
        HIPCHK(F->inv_node.upload(inode, s));
        HIPCHK(F->inv_pieces.upload(pieces, s));
        HIPCHK(F->mt_node_cnt.alloc(nn_ + 8));
        HIPCHK(hipMemsetAsync(F->mt_node_cnt.p, 0, (nn_ + 8) * sizeof(uint16_t), s));
        HIPCHK(F->rule_cnt.alloc((size_t)n_rules + 1));
        HIPCHK(hipMemsetAsync(F->rule_cnt.p, 0, ((size_t)n_rules + 1) * sizeof(uint32_t), s));
      }
      // the exact chain's records (forest_exact.hip), forest after forest in the order of the chain: per node its children,
      // rule, norm group, height; per forest where they start, how many, how high, where its sample lives, and whether it
      // fits the register path (FX_NODES nodes, FX_KIDS children a node, FX_STACK pending nodes, FX_NODES rules a derivation)
      {
        std::vector<uint32_t> xd(4 * (size_t)n_forests), xr, xm;
        std::vector<uint32_t> need;
        for (uint64_t f = 0; f < n_forests; ++f) {
          const Flat& fl = flat[f];
          const uint16_t* mt = fl.mt.data();
          const uint32_t n = mt[0], H = mt[1];
          const uint16_t* lvl = mt + 4;
          const uint16_t* koff = lvl + H + 1;
          const uint16_t* kids = koff + n + 1;
          bool slow = n > FX_NODES || fl.max_deriv > FX_NODES;
          const uint64_t first = xm.size();
          need.assign(n, 0);
          for (uint32_t h = 0; h < H; ++h)
            for (uint32_t q = lvl[h]; q < lvl[h + 1]; ++q) {
              const uint32_t nch = (uint32_t)koff[q + 1] - koff[q];
              const bool is_and = (fl.mh[4 * (size_t)q] & 0x80000000u) != 0;
              if (nch > FX_KIDS) slow = true;
              uint32_t kid[4] = {0xffu, 0xffu, 0xffu, 0xffu}, nd = 0;
              for (uint32_t c = 0; c < nch; ++c) {
                const uint32_t id = kids[koff[q] + c] & 0x7fffu;
                if (c < 4) kid[c] = id & 0xffu;  // (ids beyond a byte: a forest of the LDS path, which reads other tables)
                nd = std::max(nd, is_and ? (nch - 1 - c) + need[id] : need[id]);
              }
              need[q] = std::min(nd, 1u << 20);
              const uint2_t hrw = fl.ins[fl.hdr[fl.m_ord[q]]];
              const uint32_t rule = is_and ? hrw.y : 0u;
              xr.push_back(kid[0] | (std::min(nch, 255u) << 8) | (std::min(h, 0x7fffu) << 16) | (is_and ? 0x80000000u : 0u));
              xr.push_back(kid[1] | (kid[2] << 8) | (kid[3] << 16));
              xr.push_back(rule);
              xr.push_back(is_and ? F->h_norm[rule] : F_NONORM);
              xm.push_back(0);
            }
          if (need[n - 1] > FX_STACK) slow = true;
          if (first > 0xffffffffull || (so_all[f] >> 48)) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "forests too large for the exact sampler's tables");
          uint32_t* d = &xd[4 * (size_t)f];
          d[0] = (uint32_t)first;
          d[1] = n | (H << 16);
          d[2] = (uint32_t)so_all[f];
          // (bit 17: a derivation of more than 64 rules -- shared sub-forests count once per use, so 64 NODES can yield more --:
          // the register path keeps a sample entry per lane and register, such a forest takes two registers whatever its size)
          d[3] = (uint32_t)(so_all[f] >> 32) | (slow ? 0x10000u : 0u) | (fl.max_deriv > 64 ? 0x20000u : 0u);
        }
        HIPCHK(F->x_desc.upload(xd, s));
        HIPCHK(F->x_rec.upload(xr, s));
      }
      HIPCHK(hipStreamSynchronize(s));
    }
  }
  // samples: capacity = size of the largest derivation of the forest
  F->h_sample_off.assign(n_forests + 1, 0);
  for (uint64_t f = 0; f < n_forests; ++f) {
    F->h_sample_off[f + 1] = F->h_sample_off[f] + flat[f].max_deriv + 2;
    F->max_sample = std::max<uint32_t>(F->max_sample, (uint32_t)flat[f].max_deriv);
  }
  HIPCHK(F->groups.upload(F->h_groups, s));
  HIPCHK(F->ins_stream.upload(si, s));
  HIPCHK(F->out_stream.upload(so, s));
  HIPCHK(F->hdr_pos.upload(hp, s));
  HIPCHK(F->lane_forest.upload(lane_forest, s));
  HIPCHK(F->lane_nodes.upload(lane_nodes, s));
  HIPCHK(F->rule_logw.upload(std::vector<double>(rule_logw, rule_logw + n_rules), s));
  HIPCHK(F->counts.alloc(n_rules));
  HIPCHK(F->post.alloc(base));
  HIPCHK(F->forest_logprob.alloc(n_forests));
  HIPCHK(F->scalars.alloc(4));
  HIPCHK(F->arc_off.upload(arc_off, s));
  HIPCHK(F->slot_pos.upload(slot_pos, s));
  HIPCHK(F->hot_chunks.upload(hot, s));
  HIPCHK(F->group_off.upload(F->h_group_off, s));
  HIPCHK(F->group_rule.upload(F->h_group_rule, s));
  HIPCHK(F->p_norm.upload(F->h_norm, s));
  HIPCHK(F->sample_off.upload(F->h_sample_off, s));
  HIPCHK(F->maxbits.alloc(1));
  HIPCHK(F->iter_out.alloc(2));
  HIPCHK(hipStreamSynchronize(s));
  *out = F.release();
  return CARMEL_HIP_OK;
}

int carmel_hip_forests_destroy(carmel_hip_forests* F) {
  if (F) {
    (void)hipSetDevice(F->device);
    (void)hipDeviceSynchronize();
    hipStream_t s = F->stream;
    for (int k = 0; k < carmel_hip_forests::N_SIDE; ++k) {
      if (F->side[k]) (void)hipStreamDestroy(F->side[k]);
      if (F->ev_side[k]) (void)hipEventDestroy(F->ev_side[k]);
      if (F->ev_samp[k]) (void)hipEventDestroy(F->ev_samp[k]);
    }
    if (F->ev_fork) (void)hipEventDestroy(F->ev_fork);
    delete F;
    if (s) (void)hipStreamDestroy(s);
  }
  return CARMEL_HIP_OK;
}

// side streams: the launch classes of one pass run side by side (each ends with a few slow waves; no class fills the
// chip).  with_side(F, s, i) = the stream for class i after forking from s; join_side(F, s) folds them back.
static hipError_t ensure_side(carmel_hip_forests* F) {
  return F->ev_fork ? hipSuccess : hipErrorNotInitialized;  // (created with the forests, carmel_hip_forests_create)
}
static int n_side_for(const carmel_hip_forests* F) {
  return F->classes.size() < 2 ? 0 : (int)std::min<size_t>(carmel_hip_forests::N_SIDE, F->classes.size() - 1);
}
static hipError_t fork_side(carmel_hip_forests* F, hipStream_t s) {
  if (!n_side_for(F)) return hipSuccess;
  hipError_t e = ensure_side(F);
  if (e == hipSuccess) e = hipEventRecord(F->ev_fork, s);
  for (int k = 0; k < n_side_for(F) && e == hipSuccess; ++k) e = hipStreamWaitEvent(F->side[k], F->ev_fork, 0);
  return e;
}
static hipStream_t class_stream(carmel_hip_forests* F, hipStream_t s, size_t ci) {
  const int n = n_side_for(F);
  if (!n) return s;
  if (F->class_side.size() != F->classes.size()) {
    // longest class first onto the least loaded stream (load = lane groups x rows of the longest lane: the records a class reads)
    std::vector<size_t> order(F->classes.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    auto cost = [&](size_t i) { return (double)F->classes[i].count * (double)F->classes[i].maxlen; };
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost(a) > cost(b); });
    std::vector<double> load((size_t)n + 1, 0.0);
    F->class_side.assign(F->classes.size(), -1);
    for (size_t i : order) {
      const size_t k = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
      F->class_side[i] = (int)k - 1;
      load[k] += cost(i);
    }
  }
  const int k = F->class_side[ci];
  return k < 0 ? s : F->side[k];
}
// The several-lanes sampler's sweep: every wavefront of every class is resident at once, so a class takes what its LONGEST
// forests take (a wavefront's chain of dependent steps grows with the nodes of its forests), not what its records add up to
// (round 6, tools/c5_timeline.sh: by records the class of the largest forests -- 204 us -- shared a stream with the smallest and
// started 19 us after the first).  Classes by their largest forest, the largest on the caller's stream (it need not wait for
// the fork) and launched first; a stream takes a second class only after every stream has one.
static hipStream_t sweep_stream(carmel_hip_forests* F, hipStream_t s, size_t ci) {
  const int n = n_side_for(F);
  if (!n) return s;
  const int k = F->sweep_side[ci];
  return k < 0 ? s : F->side[k];
}
static void sweep_schedule(carmel_hip_forests* F) {
  if (F->sweep_order.size() == F->classes.size()) return;
  const int n = n_side_for(F);
  F->sweep_order.resize(F->classes.size());
  for (size_t i = 0; i < F->sweep_order.size(); ++i) F->sweep_order[i] = i;
  auto cost = [&](size_t i) { return (double)std::max(F->classes[i].m_n, F->classes[i].max_nodes); };
  std::stable_sort(F->sweep_order.begin(), F->sweep_order.end(), [&](size_t a, size_t b) { return cost(a) > cost(b); });
  std::vector<double> load((size_t)n + 1, 0.0);
  F->sweep_side.assign(F->classes.size(), -1);
  for (size_t i : F->sweep_order) {
    const size_t k = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
    F->sweep_side[i] = (int)k - 1;
    load[k] += cost(i);
  }
}
static hipError_t join_side(carmel_hip_forests* F, hipStream_t s) {
  hipError_t e = hipSuccess;
  for (int k = 0; k < n_side_for(F) && e == hipSuccess; ++k) {
    e = hipEventRecord(F->ev_side[k], F->side[k]);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, F->ev_side[k], 0);
  }
  return e;
}

static void fill_args(carmel_hip_forests* F, ForestArgs& A) {
  std::memset(&A, 0, sizeof A);
  A.groups = F->groups.p;
  A.ins_stream = (const uint2*)F->ins_stream.p;
  A.out_stream = (const uint2*)F->out_stream.p;
  A.lane_forest = F->lane_forest.p;
  A.lane_nodes = F->lane_nodes.p;
  A.rule_logw = F->rule_logw.p;
  A.post = F->post.p;
  A.forest_logprob = F->forest_logprob.p;
  A.scalars = F->scalars.p;
  A.p_norm = F->p_norm.p;
  A.hdr_pos = F->hdr_pos.p;
  A.sample_off = F->sample_off.p;
  A.iter_out = F->iter_out.p;
  A.trace = nullptr;
  A.ghash = nullptr;
  A.serial_forest = 0xffffffffu;
}

// FForests::estimate (forest-em.hpp:561-578): counts = prior_count * n_forests + expected rule counts;
// returns the average log probability over the forests with non-zero probability
int carmel_hip_forests_estimate(carmel_hip_forests* F, double prior_count, double* avg_logprob, uint64_t* n_zero,
                                double* per_forest_logprob) {
  if (!F) return fail(CARMEL_HIP_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(F->device));
  hipStream_t s = F->stream;
  ForestArgs A;
  fill_args(F, A);
  HIPCHK(hipMemsetAsync(F->scalars.p, 0, 4 * sizeof(double), s));
  HIPCHK(fork_side(F, s));
  // mantissa / exponent arithmetic where the columns fit LDS at 12 bytes per value (otherwise: the log domain)
  static const bool em_ext = true;
  for (size_t ci = 0; ci < F->classes.size(); ++ci) {
    const auto& c = F->classes[ci];
    A.first_group = c.first;
    size_t lds = (size_t)c.max_nodes * 64 * sizeof(double) * 2;
    if (lds > F_LDS_LIMIT) {
      A.gcol = F->gcol.p + F->gcol_off[ci];
      A.gcol_stride = (uint64_t)2 * c.max_nodes * 64;
      hipLaunchKernelGGL(forest_estimate_kernel<true>, dim3(c.count), dim3(64), 0, class_stream(F, s, ci), A);
      continue;
    }
    const size_t lds_ext = (size_t)c.max_nodes * 64 * 24;
    if (em_ext && lds_ext <= F_LDS_LIMIT) {
      if (lds_ext > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)forest_estimate_ext_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ext);
      hipLaunchKernelGGL(forest_estimate_ext_kernel, dim3(c.count), dim3(64), lds_ext, class_stream(F, s, ci), A);
      continue;
    }
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)forest_estimate_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(forest_estimate_kernel<false>, dim3(c.count), dim3(64), lds, class_stream(F, s, ci), A);
  }
  HIPCHK(join_side(F, s));
  HIPCHK(hipGetLastError());
  ReduceArgs R;
  R.arc_off = F->arc_off.p;
  R.slot_pos = F->slot_pos.p;
  R.hot_chunks = F->hot_chunks.p;
  R.post = F->post.p;
  R.counts = F->counts.p;
  R.n_arcs = F->n_rules;
  R.n_hot_chunks = F->hot_chunks.n / 3;
  HIPCHK(launch_count_reduce(R, s));
  double sc[4];
  HIPCHK(hipMemcpyAsync(sc, F->scalars.p, sizeof sc, hipMemcpyDeviceToHost, s));
  if (per_forest_logprob)
    HIPCHK(hipMemcpyAsync(per_forest_logprob, F->forest_logprob.p, F->n_forests * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  (void)prior_count;  // folded in at maximize / get_counts (it is a constant added to every count)
  if (avg_logprob) *avg_logprob = sc[1] > 0 ? sc[0] / sc[1] : -std::numeric_limits<double>::infinity();
  if (n_zero) *n_zero = (uint64_t)(sc[2] + 0.5);
  return CARMEL_HIP_OK;
}

int carmel_hip_forests_get_counts(carmel_hip_forests* F, double prior_count, double* counts) {
  if (!F || !counts) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(F->device));
  HIPCHK(hipMemcpyAsync(counts, F->counts.p, F->n_rules * sizeof(double), hipMemcpyDeviceToHost, F->stream));
  HIPCHK(hipStreamSynchronize(F->stream));
  const double wp = prior_count * (double)F->n_forests;
  for (uint32_t r = 0; r < F->n_rules; ++r) counts[r] += wp;
  return CARMEL_HIP_OK;
}

// FForests::maximize (forest-em.hpp:626-655) -> NormalizeGroups (normalize.hpp:123-164)
int carmel_hip_forests_maximize(carmel_hip_forests* F, double prior_count, double add_k, int zero_zerocounts,
                                double* max_delta) {
  if (!F) return fail(CARMEL_HIP_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(F->device));
  hipStream_t s = F->stream;
  HIPCHK(hipMemsetAsync(F->maxbits.p, 0, sizeof(unsigned long long), s));
  if (F->n_groups) {
    unsigned grid = (unsigned)std::min<uint64_t>((F->n_groups + 255) / 256, 4096);
    hipLaunchKernelGGL(forest_mstep_kernel, dim3(grid), dim3(256), 0, s, F->rule_logw.p, F->counts.p,
                       prior_count * (double)F->n_forests, F->group_off.p, F->group_rule.p, F->n_groups, add_k,
                       zero_zerocounts, F->maxbits.p);
    HIPCHK(hipGetLastError());
  }
  unsigned long long bits = 0;
  HIPCHK(hipMemcpyAsync(&bits, F->maxbits.p, sizeof bits, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  double d;
  std::memcpy(&d, &bits, sizeof d);
  if (max_delta) *max_delta = d;
  return CARMEL_HIP_OK;
}

int carmel_hip_forests_get_weights(carmel_hip_forests* F, double* rule_logw) {
  if (!F || !rule_logw) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(F->device));
  HIPCHK(hipMemcpyAsync(rule_logw, F->rule_logw.p, F->n_rules * sizeof(double), hipMemcpyDeviceToHost, F->stream));
  HIPCHK(hipStreamSynchronize(F->stream));
  return CARMEL_HIP_OK;
}
int carmel_hip_forests_set_weights(carmel_hip_forests* F, const double* rule_logw) {
  if (!F || !rule_logw) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(F->device));
  HIPCHK(hipMemcpyAsync(F->rule_logw.p, rule_logw, F->n_rules * sizeof(double), hipMemcpyHostToDevice, F->stream));
  HIPCHK(hipStreamSynchronize(F->stream));
  return CARMEL_HIP_OK;
}

uint32_t carmel_hip_forests_best_run(carmel_hip_forests* F) { return F ? F->best_run : 0; }
int carmel_hip_forests_final_counts(carmel_hip_forests* F, double* x) {
  if (!F || !x) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (F->h_final_x.size() != F->n_rules) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_forests_final_counts: run the sampler first");
  std::memcpy(x, F->h_final_x.data(), F->h_final_x.size() * sizeof(double));
  return CARMEL_HIP_OK;
}
int carmel_hip_forests_set_prior_inference(carmel_hip_forests* F, double stddev, int global, int local, uint32_t start,
                                           uint32_t end) {
  if (!F) return fail(CARMEL_HIP_ERR_ARG, "null handle");
  F->pi_stddev = stddev;
  F->pi_global = global != 0;
  F->pi_local = local != 0;
  F->pi_start = start;
  F->pi_end = end;
  return CARMEL_HIP_OK;
}
int carmel_hip_forests_prior_trace(carmel_hip_forests* F, double* out6, uint32_t n_sweeps, double* cumulative, uint32_t n_cumulative,
                                   uint32_t* n_scales) {
  if (!F) return fail(CARMEL_HIP_ERR_ARG, "null handle");
  if (out6)
    for (size_t k = 0; k < (size_t)n_sweeps * 6; ++k) out6[k] = k < F->pi_trace.size() ? F->pi_trace[k] : 0.0;
  if (cumulative)
    for (uint32_t k = 0; k < n_cumulative; ++k) cumulative[k] = k < F->pi_cumulative.size() ? F->pi_cumulative[k] : 1.0;
  if (n_scales) *n_scales = (uint32_t)F->pi_cumulative.size();
  return CARMEL_HIP_OK;
}

// FForests::run_gibbs (forest-em.hpp:714-734): to_gibbs (normalise, prior = alpha * p * |group|), gibbs_base::run,
// from_gibbs (rule weights = time-averaged probabilities).  opts->mode 0: forests strictly in order (the
// reference's chain); 1: all forests of a sweep in parallel against the previous sweep's counts with each forest's
// own previous sample taken out.
int carmel_hip_forests_gibbs(carmel_hip_forests* F, const carmel_hip_gibbs_opts* o, double alpha, double* iter_logprob,
                             double* iter_cheap_logprob) {
  if (!F || !o) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (F->pi_stddev > 0 && o->mode != 0)
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "prior inference works with the cache-model probability of the exact blocked sampler only (gibbs.hpp:528-529)");
  if (o->include_self || o->random_start || o->expectation)
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "--include-self / --random-start / --expectation are carmel's (carmel_hip_gibbs_create), not the forest sampler's");
  HIPCHK(hipSetDevice(F->device));
  hipStream_t s = F->stream;
  const uint32_t nr = F->n_rules;
  const uint64_t ng = F->n_groups, nf = F->n_forests;
  // define_gibbs(true): normalise the current weights (counts := weights), then priors
  std::vector<double> lw(nr);
  HIPCHK(hipMemcpyAsync(lw.data(), F->rule_logw.p, nr * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  // per-parameter alphas (forest-em.hpp:689-709): a locked parameter (alpha < 0) is defined without a norm group, i.e.
  // with the fixed probability it has after normalisation; the host and device norm tables are switched for this run
  struct NormGuard {
    carmel_hip_forests* F;
    std::vector<uint32_t> saved;
    bool changed = false;
    ~NormGuard() {
      if (!changed) return;
      F->h_norm = saved;
      (void)hipMemcpy(F->p_norm.p, F->h_norm.data(), F->h_norm.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
  } guard{F, {}};
  auto alpha_of = [&](uint32_t r) { return r < F->h_alphas.size() ? F->h_alphas[r] : alpha; };
  std::vector<double> prior(nr), pn(ng, 0.0);
  for (uint64_t gi = 0; gi < ng; ++gi) {
    double sum = 0;
    const uint64_t j0 = F->h_group_off[gi], j1 = F->h_group_off[gi + 1];
    for (uint64_t j = j0; j < j1; ++j) sum += std::exp(lw[F->h_group_rule[j]]);
    for (uint64_t j = j0; j < j1; ++j) {
      uint32_t r = F->h_group_rule[j];
      double p = sum > 0 ? std::exp(lw[r]) / sum : 1.0 / (double)(j1 - j0);
      lw[r] = p > 0 ? std::log(p) : -std::numeric_limits<double>::infinity();
      const double a = alpha_of(r);
      if (a < 0) {
        if (!guard.changed) {
          guard.saved = F->h_norm;
          guard.changed = true;
        }
        F->h_norm[r] = F_NONORM;
        continue;
      }
      prior[r] = o->uniform_p0 ? a : a * p * (double)(j1 - j0);
      pn[gi] += prior[r];
    }
  }
  // prior-scale groups as forest-em builds them (forest-em.hpp:723-734 to_gibbs; normalize.hpp:194-210; gibbs.hpp:572-579):
  // the norm ids given to define_param_id start at ONE while to_gibbs registers scale groups for ids 0 .. G-1, so norm group
  // g is scaled by scale index g + 2, the first factor is drawn for nobody, and finish_params' resize(nnorm) leaves the LAST
  // norm group with the never-scaled index 0; every drawn factor enters q(old|new)/q(new|old) all the same.
  std::vector<uint32_t> meta;  // by reference norm id (= group index + 1)
  uint32_t nexti = 1;
  F->pi_trace.assign((size_t)(o->iter + 1) * 6, 0.0);
  F->pi_cumulative.clear();
  if (F->pi_stddev > 0) {
    uint32_t nnorm = 0;
    for (uint32_t r = 0; r < nr; ++r)
      if (F->h_norm[r] != F_NONORM) nnorm = std::max(nnorm, F->h_norm[r] + 2);
    meta.assign(nnorm, 0u);
    for (uint32_t i = 0; i < nnorm && i < ng; ++i) meta[i] = i + 1;
    nexti = (uint32_t)ng + 1;
    if (F->pi_global) {
      nexti = 2;
      std::fill(meta.begin(), meta.end(), 1u);
    }
    if (F->pi_local) {
      nexti = nnorm + 1;
      for (uint32_t i = 0; i < nnorm; ++i) meta[i] = i + 1;
    }
    F->pi_cumulative.assign(nexti - 1, 1.0);
  }
  if (guard.changed) HIPCHK(hipMemcpyAsync(F->p_norm.p, F->h_norm.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  for (uint32_t r = 0; r < nr; ++r)
    if (F->h_norm[r] == F_NONORM) prior[r] = std::exp(lw[r]);
  HIPCHK(F->p_prior.upload(prior, s));
  HIPCHK(F->prior_norm.upload(pn, s));
  HIPCHK(F->p_x.upload(prior, s));
  HIPCHK(F->normsum.upload(pn, s));
  HIPCHK(F->p_s.alloc(nr));
  HIPCHK(F->p_tmax.alloc(nr));
  HIPCHK(F->new_x.alloc(nr));
  HIPCHK(hipMemsetAsync(F->p_s.p, 0, nr * sizeof(double), s));
  HIPCHK(hipMemsetAsync(F->p_tmax.p, 0, nr * sizeof(double), s));
  for (int k = 0; k < 2; ++k) {
    HIPCHK(F->sample_len[k].alloc(nf));
    HIPCHK(F->sample_rules[k].alloc(F->h_sample_off.back() + 128));  // (+ forest_exact_kernel's staging reads a fixed number of words ahead)
    HIPCHK(hipMemsetAsync(F->sample_len[k].p, 0, nf * sizeof(uint32_t), s));
  }
  ForestArgs A;
  fill_args(F, A);
  A.p_prior = F->p_prior.p;
  A.seed = o->seed;
  A.counterfactual = 1;
  // parallel mode, second formulation (CARMEL_HIP_FOREST_SWEEP=1 selects the first, kept as the A/B reference)
  bool split_recount = false;  // set below
  const bool sweep2 = o->mode == 1 && F->sweep2_ok && !(lib_opt("forest_sweep") && atoi(lib_opt("forest_sweep")) == 1);
  if (sweep2) {
    // the classes' recounts beside the classes still sampling (false: one recount after all, the earlier form)
    split_recount = true;
    if (split_recount) HIPCHK(F->normsum2.alloc(ng));
    HIPCHK(F->sample_cls.alloc(F->h_sample_off.back() + 64));  // (+ the sampler's staging reads a fixed number of words ahead)
    HIPCHK(F->rec_logp.alloc(F->stream_total));
    HIPCHK(F->rec_p.alloc(F->stream_total));
    HIPCHK(hipMemsetAsync(F->rec_p.p, 0, F->rec_p.bytes(), s));  // the sample kernel reads every slot of its chunks
    HIPCHK(hipMemsetAsync(F->rec_logp.p, 0, F->rec_logp.bytes(), s));
    HIPCHK(F->sample_hdr.alloc(F->h_sample_off.back()));
    A.rec_cls = F->rec_cls.p;
    A.rec_logp = F->rec_logp.p;
    A.rec_p = F->rec_p.p;
    A.sample_hdr = F->sample_hdr.p;
    A.lane_of_forest = F->lane_of_forest_d.p;
  }
  DevBuf<double> gcol_exact;  // exact mode: the inside column of one forest too large for LDS
  DevBuf<uint32_t> ghash;  // parallel mode: global own-sample tables, only when some derivation can overflow the LDS table
  const uint32_t own_cap_max = 256u;
  // the walk of forest_sample_kernel over tables in LDS (CARMEL_HIP_FOREST_LDSWALK=0: over the global stream, the A/B reference)
  // groups per workgroup of a class's recount (fewer groups per workgroup = more workgroups, each adding its share of
  // the popular rules to the same addresses: 1 and 2 measured slower than 4, 122 and 84 against 77 us for the last class)
  const uint32_t recount_div = 4;
  // the recount's LDS tables: rules / norm groups (powers of two)
  const uint32_t frc_slots0 = 8192, frc_slots1 = 4096;
  const size_t frc_bytes = (size_t)(frc_slots0 + frc_slots1) * 8;
  (void)hipFuncSetAttribute((const void*)forest_recount_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)frc_bytes);
  const bool lds_walk = !(lib_opt("forest_ldswalk") && atoi(lib_opt("forest_ldswalk")) == 0);
  // several lanes per forest in the parallel sweep (CARMEL_HIP_FOREST_MULTI=0: one forest per lane, the A/B reference -- and the
  // chain whose uniforms are keyed like the sequential walk's)
  const bool multi = sweep2 && F->multi_ok && !(lib_opt("forest_multi") && atoi(lib_opt("forest_multi")) == 0);
  const uint64_t nf_slots = F->h_groups.size() * 64;
  auto fm_bytes = [](const carmel_hip_forests::Cls& c) {  // LDS of one forest in forest_sample_multi_kernel
    return (size_t)c.m_n * 16 + (((size_t)c.m_tab + 2 * (size_t)c.m_front) * 2 + 15) / 16 * 16;
  };
  const uint32_t stack_lds = 32u;
  if (o->mode == 1 && !sweep2 && (uint64_t)F->max_sample * 20 > 32 * 9 && !lib_opt("forest_nohash")) {
    HIPCHK(ghash.alloc((size_t)nf * FOREST_GHASH));
    A.ghash = ghash.p;
  }
  DevBuf<unsigned long long> trace_buf;  // experiment: per-wave phase stamps of the last parallel sweep
  const char* trace_path = lib_opt("forest_trace");
  if (trace_path && o->mode == 1) {
    HIPCHK(trace_buf.alloc(F->h_groups.size() * 8 * 8));  // (the several-lanes sampler: eight workgroups per lane group)
    HIPCHK(hipMemset(trace_buf.p, 0, trace_buf.bytes()));
    A.trace = trace_buf.p;
  }
  // exact mode on the device (forest_exact.hip): one persistent wavefront per sweep, every count in device memory.  It needs
  // the per-forest height tables of the several-lanes sampler and runs at temperature 1; annealed runs and locked parameters keep
  // the host-driven loop below (prior-scale inference: the proposals between sweeps are made on the host either way).
  FExactArgs XA;
  std::memset(&XA, 0, sizeof XA);
  DevBuf<double> x_ccount, x_csum;
  DevBuf<unsigned long long> x_clk;
  DevBuf<double> x_idle;
  bool exact_dev = o->mode == 0 && F->multi_ok && !guard.changed && (o->high_temp == 0 || o->high_temp == 1) &&
                   (o->low_temp == 0 || o->low_temp == 1) && nf > 0 && !lib_opt("forest_exact_host");
  if (exact_dev) {
    for (auto& c : F->classes) {
      XA.max_n = std::max(XA.max_n, c.m_n);
      XA.max_tab = std::max(XA.max_tab, c.m_tab);
      XA.max_stack = std::max(XA.max_stack, c.max_kids + 2);
    }
    XA.max_sample = F->max_sample + 1;
    if (XA.max_stack > 0xffffu || XA.max_sample > 0xffffu ||
        forest_exact_lds_bytes(XA.max_n, XA.max_tab, XA.max_stack, XA.max_sample) > F_LDS_LIMIT)
      exact_dev = false;
  }
  if (exact_dev) {
    HIPCHK(x_ccount.alloc(nr));
    HIPCHK(x_csum.alloc(std::max<uint64_t>(ng, 1)));
    HIPCHK(F->sample_cls.alloc(F->h_sample_off.back() + 128));  // here: the norm group of every sample entry
    XA.xdesc = (const uint4*)F->x_desc.p;
    XA.xrec = (const uint4*)F->x_rec.p;
    XA.tab = F->mt_tab.p;
    XA.hdr = F->mt_hdr.p;
    XA.slots = (const uint4*)F->mt_slots.p;
    XA.lane_of_forest = F->lane_of_forest_d.p;
    XA.sample_len = F->sample_len[0].p;
    XA.sample_rules = F->sample_rules[0].p;
    XA.sample_nn = F->sample_cls.p;
    XA.p_x = F->p_x.p;
    XA.normsum = F->normsum.p;
    XA.p_prior = F->p_prior.p;
    XA.ccount = x_ccount.p;
    XA.csum = x_csum.p;
    XA.iter_out = F->iter_out.p;
    XA.seed = o->seed;
    XA.n_forests = (uint32_t)nf;
    HIPCHK(x_idle.alloc(256));
    HIPCHK(hipMemsetAsync(x_idle.p, 0, 256 * sizeof(double), s));
    XA.idle = x_idle.p;
    if (lib_opt("forest_exact_clk")) {
      HIPCHK(x_clk.alloc(8));
      HIPCHK(hipMemsetAsync(x_clk.p, 0, 64, s));
      XA.phase_clk = x_clk.p;
    }
  }
  // gibbs_opts::validate (gibbs_opts.hpp:253-266): --final-counts makes every sweep but the last burn-in; burnin <= iter
  const uint32_t Ni = o->iter, burnin = o->final_counts ? o->iter : std::min(o->burnin, o->iter);
  F->best_run = 0;
  // finalize_cumulative_counts + from_gibbs of one finished run (gibbs.hpp:629-640, forest-em.hpp:736-741): ln weights from its
  // counts, their time-weighted sums and stamps
  auto final_weights = [&](std::vector<double>& x, std::vector<double>& sacc, const std::vector<double>& tm, std::vector<double>& out) {
    if (!(o->final_counts && !o->exclude_prior)) {
      const double tmax1 = ((double)Ni - (double)burnin) + 1.0;
      if (o->exclude_prior)  // --crp-exclude-prior (gibbs.hpp:629-631): addbase(-prior) before the counts are extended
        for (uint32_t r = 0; r < nr; ++r)
          if (F->h_norm[r] != F_NONORM) {
            sacc[r] += -prior[r] * tm[r];
            x[r] += -prior[r];
          }
      if (!o->final_counts)
        for (uint32_t r = 0; r < nr; ++r)
          if (F->h_norm[r] != F_NONORM) {
            sacc[r] += x[r] * (tmax1 - tm[r]);
            x[r] = sacc[r];
          }
    }
    std::vector<double> ns(ng, 0.0);
    for (uint32_t r = 0; r < nr; ++r)
      if (F->h_norm[r] != F_NONORM) ns[F->h_norm[r]] += x[r];
    for (uint32_t r = 0; r < nr; ++r) {
      double pr = F->h_norm[r] == F_NONORM ? prior[r] : (x[r] > 0 ? x[r] / ns[F->h_norm[r]] : 0.0);
      out[r] = pr > 0 ? std::log(pr) : -std::numeric_limits<double>::infinity();
    }
  };
  // ---- --crp-restarts (gibbs_base::run_starts, gibbs.hpp:880-914, which forest-em's sampler runs through like carmel's): every
  // run starts from the priors and draws the uniforms of its own sweeps (run r, sweep i: those of sweep r * (iter + 1) + i), the
  // run that is better by gibbs_stats::better gives the weights and the sample.  Independent chains: they run SIDE BY SIDE, chain c
  // = workgroup c of forest_exact_kernel (one wavefront each; FExactArgs::n_chains), one launch per sweep for all of them, in
  // batches of at most 64 chains / 8 GB of state.  The device chain only: temperature 1, no locked parameter, no prior inference.
  if (o->restarts > 0) {
    if (o->mode != 0 || !exact_dev || F->pi_stddev > 0)
      return fail(CARMEL_HIP_ERR_UNSUPPORTED, "--crp-restarts runs the exact chain on the device: no --crp-parallel, annealing, locked parameters (negative --alpha entries) or prior inference");
    const uint32_t n_runs = o->restarts + 1;
    const uint64_t S = F->sample_rules[0].n, ngs = std::max<uint64_t>(ng, 1);
    uint32_t cap = 64;
    if (const char* e = lib_opt("gibbs_chains")) cap = (uint32_t)std::max(1, atoi(e));  // 1: one run after the other (A/B)
    const uint64_t chain_bytes = ((uint64_t)nr * 4 + ngs * 2) * 8 + S * 8 + nf * 4 + 64;
    cap = (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(1, (8ull << 30) / chain_bytes));
    DevBuf<double> mx, ms, mt, mn, mcc, mcs, mio;
    DevBuf<uint32_t> mlen, mrules, mnn;
    double best_all = 0, best_final = 0, best_sum = 0;
    bool ran_any = false;
    std::vector<double> best_lw(nr), clw(nr), x(nr), sacc(nr), tm(nr);
    std::vector<uint32_t> best_rules, best_len;
    for (uint32_t b0 = 0; b0 < n_runs; b0 += cap) {
      const uint32_t R = std::min(cap, n_runs - b0);
      if (mx.n < (size_t)R * nr) {
        HIPCHK(mx.alloc((size_t)R * nr));
        HIPCHK(ms.alloc((size_t)R * nr));
        HIPCHK(mt.alloc((size_t)R * nr));
        HIPCHK(mn.alloc((size_t)R * ngs));
        HIPCHK(mcc.alloc((size_t)R * nr));
        HIPCHK(mcs.alloc((size_t)R * ngs));
        HIPCHK(mio.alloc((size_t)R * 2));
        HIPCHK(mlen.alloc((size_t)R * nf));
        HIPCHK(mrules.alloc((size_t)R * S));
        HIPCHK(mnn.alloc((size_t)R * S));
      }
      // init_run for every chain: counts = priors, norm sums = their sums, no sample, time 0
      HIPCHK(launch_gibbs_broadcast(mx.p, F->p_prior.p, nr, R, s));
      if (ng) HIPCHK(launch_gibbs_broadcast(mn.p, F->prior_norm.p, ng, R, s));
      HIPCHK(hipMemsetAsync(ms.p, 0, (size_t)R * nr * sizeof(double), s));
      HIPCHK(hipMemsetAsync(mt.p, 0, (size_t)R * nr * sizeof(double), s));
      HIPCHK(hipMemsetAsync(mlen.p, 0, (size_t)R * nf * sizeof(uint32_t), s));
      FExactArgs XC = XA;
      XC.sample_len = mlen.p;
      XC.sample_rules = mrules.p;
      XC.sample_nn = mnn.p;
      XC.p_x = mx.p;
      XC.normsum = mn.p;
      XC.ccount = mcc.p;
      XC.csum = mcs.p;
      XC.iter_out = mio.p;
      XC.phase_clk = nullptr;
      XC.n_chains = R;
      XC.iter_stride = Ni + 1;
      XC.ch_rules = nr;
      XC.ch_norms = ng;
      XC.ch_sample = S;
      XC.ch_forests = nf;
      if (R == 1) {  // (a lone chain is the kernel's plain form: the strides do not apply, the base sweep does)
        XC.n_chains = 0;
      }
      std::vector<double> st_all(R, 0.0), st_final(R, 0.0), st_sum(R, -std::numeric_limits<double>::infinity()), io((size_t)R * 2);
      for (uint32_t iter = 0; iter <= Ni; ++iter) {
        const double time = iter == 0 ? 0.0 : std::max(0.0, (double)iter - (double)burnin);
        HIPCHK(hipMemsetAsync(mio.p, 0, (size_t)R * 2 * sizeof(double), s));
        HIPCHK(launch_gibbs_broadcast(mcc.p, F->p_prior.p, nr, R, s));
        if (ng) HIPCHK(launch_gibbs_broadcast(mcs.p, F->prior_norm.p, ng, R, s));
        HIPCHK(launch_forest_fold(ms.p, mt.p, mx.p, time, (uint64_t)R * nr, s));
        XC.iter = b0 * (Ni + 1) + iter;
        HIPCHK(launch_forest_exact(XC, s));
        HIPCHK(hipMemcpyAsync(io.data(), mio.p, io.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (uint32_t c = 0; c < R; ++c) {
          const uint32_t run = b0 + c;
          const double plog = io[(size_t)c * 2];
          if (iter_logprob) iter_logprob[(size_t)run * (Ni + 1) + iter] = plog;
          if (iter_cheap_logprob) iter_cheap_logprob[(size_t)run * (Ni + 1) + iter] = io[(size_t)c * 2 + 1];
          if (iter >= burnin) {  // gibbs.hpp:942-943: the statistics runs are compared by
            st_all[c] += plog;
            st_final[c] = plog;
            const double hi = std::max(st_sum[c], plog), lo = std::min(st_sum[c], plog);
            st_sum[c] = hi + (lo == -std::numeric_limits<double>::infinity() ? 0.0 : std::log1p(std::exp(lo - hi)));
          }
        }
      }
      for (uint32_t c = 0; c < R; ++c) {  // the better run by gibbs_stats::better (gibbs_opts.hpp:313-316), in run order
        const bool better = !ran_any || (o->argmax_final ? st_final[c] > best_final : o->argmax_sum ? st_sum[c] > best_sum : st_all[c] > best_all);
        ran_any = true;
        if (!better) continue;
        HIPCHK(hipMemcpyAsync(x.data(), mx.p + (size_t)c * nr, nr * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sacc.data(), ms.p + (size_t)c * nr, nr * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(tm.data(), mt.p + (size_t)c * nr, nr * sizeof(double), hipMemcpyDeviceToHost, s));
        best_rules.resize(S);
        best_len.resize(nf);
        HIPCHK(hipMemcpyAsync(best_rules.data(), mrules.p + (size_t)c * S, S * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(best_len.data(), mlen.p + (size_t)c * nf, nf * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        final_weights(x, sacc, tm, clw);
        F->h_final_x = x;
        best_lw = clw;
        F->best_run = b0 + c;
        best_all = st_all[c];
        best_final = st_final[c];
        best_sum = st_sum[c];
      }
    }
    // the kept run's sample is the sampler's sample (carmel_hip_forests_get_sample, --outsample-file)
    HIPCHK(hipMemcpyAsync(F->sample_rules[0].p, best_rules.data(), S * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(F->sample_len[0].p, best_len.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return carmel_hip_forests_set_weights(F, best_lw.data());
  }
  // host mirror of counts for the exact schedule (one forest at a time: the counts move between forests)
  std::vector<double> hx, hs, ht, hn;
  std::vector<std::vector<uint32_t> > hsample;
  std::vector<double> ccount, csum;
  if (o->mode == 0 && !exact_dev) {
    hx = prior;
    hs.assign(nr, 0.0);
    ht.assign(nr, 0.0);
    hn = pn;
    hsample.assign(nf, {});
  }
  int cur = 0;
  DevBuf<double> iter_all;  // parallel mode: {-, ln proposal probability of the sweep's samples} per sweep
  std::vector<double> iter_host;
  uint32_t io_done = 0;
  if (o->mode == 1) {
    HIPCHK(iter_all.alloc(2 * ((size_t)Ni + 1)));
    HIPCHK(hipMemsetAsync(iter_all.p, 0, iter_all.bytes(), s));
    iter_host.assign(2 * ((size_t)Ni + 1), 0.0);
  }
  bool side_pending = false;  // recounts of the parallel sweep still on the side streams (joined before anything reads what they write)
  for (uint32_t iter = 0; iter <= Ni; ++iter) {
    const double time = iter == 0 ? 0.0 : std::max(0.0, (double)iter - (double)burnin);
    A.iter = iter;
    A.power = gibbs_anneal_power(o->high_temp, o->low_temp, Ni, iter);
    double cache_lp = 0.0, cheap_lp = 0.0;
    std::vector<std::function<void()>> late_recounts;  // (parallel sweep with gathered counts: the recounts, launched behind the commit)
    bool gathered = false;  // this sweep's counts are in rule_cnt (forest_rule_gather_kernel), not in new_x
    if (o->mode == 1)  // a slot per sweep, read back in batches: the host runs ahead of the device, no round trip per sweep
      A.iter_out = iter_all.p + 2 * (size_t)iter;
    else
      HIPCHK(hipMemsetAsync(F->iter_out.p, 0, 2 * sizeof(double), s));
    if (o->mode == 1) {
      // all forests against the counts of the previous sweep, own previous sample taken out in-kernel
      A.snap_x = F->p_x.p;
      A.snap_norm = F->normsum.p;
      A.old_len = F->sample_len[cur].p;
      A.old_rules = F->sample_rules[cur].p;
      A.sample_len = F->sample_len[cur ^ 1].p;
      A.sample_rules = F->sample_rules[cur ^ 1].p;
      if (sweep2) {
        A.sample_cls = F->sample_cls.p;
        // (the previous sample's class words, written by its recount: what the proposal kernel scans for the forest's own uses)
        A.and_list = F->and_list.p;
        A.n_and = F->n_and;
        A.p_only = (A.power == 1.0 && !lib_opt("forest_logdomain")) ? 1 : 0;
        // every launch class on the several-lanes sampler: it computes the proposal probabilities itself (no kernel in front
        // of the classes, no rec_p round trip)
        const bool ext_now = A.p_only != 0;
        bool fold_proposal = multi && ext_now;
        for (auto& c : F->classes)
          if (fm_bytes(c) * FM_FPW > 64 * 1024) fold_proposal = false;
        if (lib_opt("forest_gcol")) fold_proposal = false;
        // the counts gathered from the nodes' use counts instead of added up by the recounts' atomics (forest_rule_gather_kernel)
        // (forest_gather = 1; measured on config 5: 348 us a sweep against 316 -- the gather is 2.5 M scattered two-byte reads, 47 us,
        // as many requests as the atomics it replaces, and the sweep gains two cross-stream waits; what it buys is counts that
        // are the same bits run after run)
        const bool gather_counts = fold_proposal && split_recount && F->inv_off.n && F->mt_node_cnt.n && lib_opt("forest_gather") &&
                                   atoi(lib_opt("forest_gather")) == 1;
        if (F->n_and && !fold_proposal)
          hipLaunchKernelGGL(forest_proposal_kernel, dim3((unsigned)((F->n_and + 255) / 256)), dim3(256), 0, s, A);
        if (split_recount && iter == 0) {  // the new counts start from the priors; the norm sums go to the other buffer (this
                                           // sweep reads the current one).  Later sweeps: prepared at the end of the previous one
          HIPCHK(hipMemcpyAsync(F->new_x.p, F->p_prior.p, nr * sizeof(double), hipMemcpyDeviceToDevice, s));
          HIPCHK(hipMemcpyAsync(F->normsum2.p, F->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        HIPCHK(fork_side(F, s));
        sweep_schedule(F);
        for (size_t ci : F->sweep_order) {
          const auto& c = F->classes[ci];
          A.first_group = c.first;
          // temperature 1: mantissa / exponent arithmetic (12 bytes per node); annealing: the log domain
          const bool ext = A.power == 1.0 && !lib_opt("forest_logdomain");
          const size_t lds = (size_t)c.max_nodes * 64 * (ext ? 12 : 8) + (size_t)stack_lds * 64 * 4;
          // with the walk's tables in LDS: 16-bit rows (2 per node + 1, the child entries, the stack)
          const uint32_t kid_rows = std::max(c.max_kids, 1u);
          const size_t lds_lw = (size_t)c.max_nodes * 64 * (ext ? 12 : 8) +
                                ((size_t)2 * c.max_nodes + 1 + kid_rows + stack_lds) * 64 * 2;
          const bool lw = lds_walk && c.max_nodes < 0x8000u && c.max_kids < 0x8000u && c.maxlen <= 0x10000u &&
                          (size_t)c.max_nodes * 64 * 8 * 2 <= F_LDS_LIMIT && lds_lw <= F_LDS_LIMIT;
          auto launch = [&](auto kernel, size_t bytes) {
            if (bytes > 64 * 1024)
              (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            hipLaunchKernelGGL(kernel, dim3(c.count), dim3(64), bytes, sweep_stream(F, s, ci), A, F->max_sample, c.max_nodes,
                               stack_lds, kid_rows);
          };
          // several lanes per forest (temperature 1, tables within LDS): forest_sample_multi_kernel
          const size_t fm_per = fm_bytes(c);
          bool class_nodes = false;  // the class's sample is written as node numbers (FMultiArgs::prob)
          const bool force_gcol = lib_opt("forest_gcol") != nullptr;  // experiment: every class one forest per lane, columns in global memory
          if (multi && ext && fm_per * FM_FPW <= 64 * 1024 && !force_gcol) {
            FMultiArgs MA;
            MA.tab = F->mt_tab.p;
            MA.hdr = F->mt_hdr.p;
            MA.slots = (const uint4*)F->mt_slots.p;
            MA.lane_lo = c.first * 64u;
            MA.lane_hi = (uint32_t)std::min<uint64_t>((uint64_t)(c.first + c.count) * 64u, nf_slots);
            MA.max_tab = c.m_tab;
            MA.max_n = c.m_n;
            MA.max_front = c.m_front;
            MA.own_proposal = fold_proposal ? 1 : 0;
            if (fold_proposal && !F->mt_prob.n) HIPCHK(F->mt_prob.alloc(F->mt_hdr.n / 4 + 8));
            MA.prob = F->mt_prob.p;
            MA.node_cnt = gather_counts ? F->mt_node_cnt.p : nullptr;
            class_nodes = fold_proposal;
            const uint32_t nwg = (MA.lane_hi - MA.lane_lo + FM_FPW - 1) / FM_FPW;
            if (iter == 0 && lib_opt("timing"))
              fprintf(stderr, "timing: forest sweep class %zu: %u wavefronts of %d forests, nodes <= %u, table <= %u words, frontier <= %u: %zu bytes of LDS a wavefront\n",
                      ci, nwg, (int)FM_FPW, c.m_n, c.m_tab, c.m_front, fm_per * FM_FPW);
            hipLaunchKernelGGL(forest_sample_multi_kernel, dim3(nwg), dim3(64), fm_per * FM_FPW, sweep_stream(F, s, ci), A, MA, F->max_sample);
          } else if ((size_t)c.max_nodes * 64 * 8 * 2 > F_LDS_LIMIT || force_gcol) {
            A.gcol = F->gcol.p + F->gcol_off[ci];
            A.gcol_stride = (uint64_t)2 * c.max_nodes * 64;
            if (ext)
              launch(forest_sample_kernel<true, true, false>, (size_t)stack_lds * 64 * 4);
            else
              launch(forest_sample_kernel<true, false, false>, (size_t)stack_lds * 64 * 4);
          } else if (lw) {
            if (ext)
              launch(forest_sample_kernel<false, true, true>, lds_lw);
            else
              launch(forest_sample_kernel<false, false, true>, lds_lw);
          } else if (ext)
            launch(forest_sample_kernel<false, true, false>, lds);
          else
            launch(forest_sample_kernel<false, false, false>, lds);
          if (split_recount) {
            // this class's new samples: rule ids, class words, ln proposal probability -- and, unless the counts are gathered, the
            // counts -- on its own stream, while the other classes still sample.  (With the counts in it, all samplers first and
            // the recounts behind them was 335 us against 316: bound by their atomics they take as long side by side.)
            const ForestArgs Ac = A;
            const int cur_new = cur ^ 1;
            auto rc = [=]() {
              const auto& cc = F->classes[ci];
              hipLaunchKernelGGL(forest_recount_kernel, dim3(std::min<uint32_t>(std::max<uint32_t>(cc.count / recount_div, 1u), 2048u)), dim3(1024), frc_bytes,
                                 sweep_stream(F, s, ci), F->sample_off.p, F->sample_len[cur_new].p, F->sample_rules[cur_new].p,
                                 F->p_norm.p, F->new_x.p, F->normsum2.p, (uint32_t)nf, Ac, gather_counts ? 3 : 1, F->lane_forest.p, cc.first * 64u,
                                 (cc.first + cc.count) * 64u, frc_slots0, frc_slots1, class_nodes ? (const uint32_t*)F->mt_hdr.p : nullptr,
                                 (const double*)F->mt_prob.p, (const uint4*)F->mt_slots.p);
            };
            if (gather_counts)
              late_recounts.push_back(rc);  // (behind the commit: nothing the next sweep's counts need waits for them)
            else
              rc();
          }
        }
        if (gather_counts) {
          // the caller's stream waits for the SAMPLERS of the side streams only, gathers the counts and commits them; the recounts
          // follow on their streams and run into the next sweep (a class's next sampler is behind its recount on its own stream)
          for (int k = 0; k < n_side_for(F); ++k) {
            HIPCHK(hipEventRecord(F->ev_samp[k], F->side[k]));
            HIPCHK(hipStreamWaitEvent(s, F->ev_samp[k], 0));
          }
          const uint32_t cold_blocks = (uint32_t)((nr + 255) / 256);
          hipLaunchKernelGGL(forest_rule_gather_kernel, dim3(cold_blocks + (F->n_inv_pieces + 3) / 4), dim3(256), 0, s, F->inv_off.p, F->inv_node.p,
                             (const uint16_t*)F->mt_node_cnt.p, F->rule_cnt.p, (uint32_t)nr, (const uint32_t*)F->inv_pieces.p, F->n_inv_pieces,
                             cold_blocks, (uint32_t)F->inv_node.n);
          hipLaunchKernelGGL(forest_group_sum_kernel, dim3((unsigned)((ng * 8 + 255) / 256)), dim3(256), 0, s, F->group_off.p, F->group_rule.p,
                             (uint64_t)ng, (const uint32_t*)F->rule_cnt.p, (const double*)F->prior_norm.p, F->normsum2.p);
          side_pending = true;
          gathered = true;
        } else
          HIPCHK(join_side(F, s));
      } else
      for (auto& c : F->classes) {
        A.first_group = c.first;
        // LDS: the inside column + up to own_cap {rule, norm group} pairs of the previous sample per lane
        uint32_t own_cap = own_cap_max;  // hash slots per lane, fewer when the inside column is large
        while (own_cap && (size_t)c.max_nodes * 512 + (size_t)own_cap * 256 + stack_lds * 256 > 156 * 1024) own_cap >>= 1;
        if (own_cap < 32) own_cap = 0;
        const bool nohash = lib_opt("forest_nohash") != nullptr;  // A/B: scan the previous sample instead
        if (nohash) own_cap = 0;
        if ((size_t)c.max_nodes * 64 * 8 * 2 > F_LDS_LIMIT) {
          own_cap = nohash ? 0 : own_cap_max;
          A.gcol = F->gcol.p + F->gcol_off[&c - &F->classes[0]];
          A.gcol_stride = (uint64_t)2 * c.max_nodes * 64;
          const size_t l2 = (size_t)own_cap * 64 * 4 + (size_t)stack_lds * 64 * 4;
          if (l2 > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)forest_gibbs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
          hipLaunchKernelGGL(forest_gibbs_kernel<true>, dim3(c.count), dim3(64), l2, s, A, F->max_sample, c.max_nodes, own_cap, stack_lds);
          continue;
        }
        size_t lds = (size_t)c.max_nodes * 64 * 8 + (size_t)own_cap * 64 * 4 + (size_t)stack_lds * 64 * 4;
        if (lds > 64 * 1024)
          (void)hipFuncSetAttribute((const void*)forest_gibbs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(forest_gibbs_kernel<false>, dim3(c.count), dim3(64), lds, s, A, F->max_sample, c.max_nodes, own_cap, stack_lds);
      }
      HIPCHK(hipGetLastError());
      cur ^= 1;
      if (split_recount)
        std::swap(F->normsum.p, F->normsum2.p);  // the sums the classes just recounted become the current ones
      else {
      HIPCHK(hipMemcpyAsync(F->new_x.p, F->p_prior.p, nr * sizeof(double), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(F->normsum.p, F->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(forest_recount_kernel, dim3((unsigned)std::min<uint64_t>((nf + 255) / 256, 2048)), dim3(1024), frc_bytes, s,
                         F->sample_off.p, F->sample_len[cur].p, F->sample_rules[cur].p, F->p_norm.p, F->new_x.p,
                         F->normsum.p, (uint32_t)nf, A, sweep2 ? 1 : 0, (const uint32_t*)nullptr, 0u, 0u, frc_slots0, frc_slots1,
                         (const uint32_t*)nullptr, (const double*)nullptr, (const uint4*)nullptr);
      }
      {  // (split recount: the next sweep's count buffers start from the priors, reset by the commit itself)
        const bool reset = split_recount && iter < Ni;
        hipLaunchKernelGGL(forest_commit_kernel, dim3((nr + 255) / 256), dim3(256), 0, s, F->new_x.p, F->p_x.p, F->p_s.p,
                           F->p_tmax.p, F->p_norm.p, time, (uint64_t)nr, reset ? (const double*)F->p_prior.p : nullptr,
                           reset ? F->normsum2.p : nullptr, (const double*)F->prior_norm.p, (uint64_t)ng,
                           gathered ? F->rule_cnt.p : nullptr, (const double*)F->p_prior.p);
      }
      for (auto& r : late_recounts) r();
      HIPCHK(hipGetLastError());
      if (iter == Ni || (iter & 63u) == 63u) {  // the sweeps' probabilities, 64 sweeps at a time
        if (side_pending) {  // (the side streams' recounts add to the sweeps' probabilities)
          HIPCHK(join_side(F, s));
          side_pending = false;
        }
        HIPCHK(hipMemcpyAsync(iter_host.data() + 2 * (size_t)io_done, iter_all.p + 2 * (size_t)io_done,
                              2 * (size_t)(iter + 1 - io_done) * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (uint32_t q = io_done; q <= iter; ++q) {
          if (iter_logprob) iter_logprob[q] = iter_host[2 * (size_t)q + 1];
          if (iter_cheap_logprob) iter_cheap_logprob[q] = iter_host[2 * (size_t)q + 1];
        }
        io_done = iter + 1;
      }
      continue;
    } else if (exact_dev) {
      // exact, on the device: the whole sweep is one launch (forest_exact.hip)
      HIPCHK(hipMemcpyAsync(x_ccount.p, F->p_prior.p, nr * sizeof(double), hipMemcpyDeviceToDevice, s));
      if (ng) HIPCHK(hipMemcpyAsync(x_csum.p, F->prior_norm.p, ng * sizeof(double), hipMemcpyDeviceToDevice, s));
      XA.iter = iter;
      // delta_sum's fold for every parameter at once: at the start of a sweep every count is what the previous sweep left,
      // which is what the reference folds at a parameter's first touch in this sweep (delta_sum.hpp:74-84)
      HIPCHK(launch_forest_fold(F->p_s.p, F->p_tmax.p, F->p_x.p, time, nr, s));
      HIPCHK(launch_forest_exact(XA, s));
      double io[2] = {0, 0};
      HIPCHK(hipMemcpyAsync(io, F->iter_out.p, sizeof io, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      cache_lp = io[0];
      cheap_lp = io[1];
    } else {
      // exact: forest after forest; each launch resamples ONE forest on the GPU against the current counts
      ccount = prior;
      csum = pn;
      A.counterfactual = 0;
      A.snap_x = F->p_x.p;
      A.snap_norm = F->normsum.p;
      A.sample_len = F->sample_len[0].p;
      A.sample_rules = F->sample_rules[0].p;
      A.old_len = F->sample_len[0].p;
      A.old_rules = F->sample_rules[0].p;
      auto addc = [&](const std::vector<uint32_t>& b, double d) {  // gibbs.hpp:769-792 + delta_sum.hpp:74-84
        for (uint32_t r : b) {
          uint32_t n = F->h_norm[r];
          if (n == F_NONORM) continue;
          hn[n] += d;
          double moret = time - ht[r];
          if (moret > 0) {
            ht[r] = time;
            hs[r] += moret * hx[r];
          } else if (moret < 0)
            hs[r] += d * (-moret);
          hx[r] += d;
        }
      };
      std::vector<uint32_t> buf(F->max_sample);
      for (uint64_t f = 0; f < nf; ++f) {
        addc(hsample[f], -1.0);
        // push the (few) changed counts: upload only what the removal touched
        for (uint32_t r : hsample[f]) {
          uint32_t n = F->h_norm[r];
          if (n == F_NONORM) continue;
          HIPCHK(hipMemcpyAsync(F->p_x.p + r, &hx[r], sizeof(double), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(F->normsum.p + n, &hn[n], sizeof(double), hipMemcpyHostToDevice, s));
        }
        const uint32_t slot = F->lane_of_forest[f];
        A.serial_forest = slot;
        const uint32_t gidx = slot / 64;
        A.first_group = gidx;
        const FGroup& G = F->h_groups[gidx];
        size_t lds = (size_t)G.max_nodes * 64 * 8;
        if (lds > F_LDS_LIMIT) {
          if (!gcol_exact.n) HIPCHK(gcol_exact.alloc((size_t)F->max_nodes * 64));
          A.gcol = gcol_exact.p;
          A.gcol_stride = 0;
          hipLaunchKernelGGL(forest_gibbs_kernel<true>, dim3(1), dim3(64), 0, s, A, F->max_sample, G.max_nodes, 0u, 0u);
        } else {
          if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)forest_gibbs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL(forest_gibbs_kernel<false>, dim3(1), dim3(64), lds, s, A, F->max_sample, G.max_nodes, 0u, 0u);
        }
        uint32_t len = 0;
        HIPCHK(hipMemcpyAsync(&len, F->sample_len[0].p + f, sizeof len, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (len) HIPCHK(hipMemcpyAsync(buf.data(), F->sample_rules[0].p + F->h_sample_off[f], len * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        hsample[f].assign(buf.begin(), buf.begin() + len);
        for (uint32_t r : hsample[f]) {  // cheap prob before re-adding; cache model (gibbs.hpp:712-742)
          uint32_t n = F->h_norm[r];
          cheap_lp += std::log(n == F_NONORM ? prior[r] : hx[r] / hn[n]);
          double q = prior[r];
          if (n != F_NONORM) {
            q = ccount[r] / csum[n];
            ccount[r] += 1.0;
            csum[n] += 1.0;
          }
          cache_lp += std::log(q);
        }
        addc(hsample[f], 1.0);
        for (uint32_t r : hsample[f]) {
          uint32_t n = F->h_norm[r];
          if (n == F_NONORM) continue;
          HIPCHK(hipMemcpyAsync(F->p_x.p + r, &hx[r], sizeof(double), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(F->normsum.p + n, &hn[n], sizeof(double), hipMemcpyHostToDevice, s));
        }
        HIPCHK(hipStreamSynchronize(s));
      }
    }
    // propose_new_priors (gibbs.hpp:525-553) on the sweeps that infer (gibbs.hpp:559-563), on the host: the proposal rescales
    // every prior, count, norm sum and time-weighted sum and scores the whole sample twice.  The host-driven schedule keeps its
    // counts there anyway; the device chain hands its state over for the proposal and takes it back (a few tens of MB per
    // inferring sweep against a 0.5 s sweep).
    const uint32_t pstart = F->pi_start ? F->pi_start : burnin;
    if (o->mode == 0 && F->pi_stddev > 0 && nexti > 1 && iter > 0 && pstart <= iter && (!F->pi_end || iter < F->pi_end)) {
      if (exact_dev) {
        hx.resize(nr);
        hs.resize(nr);
        ht.resize(nr);
        hn.resize(ng);
        std::vector<uint32_t> sl(nf), sr(F->h_sample_off.back());
        HIPCHK(hipMemcpyAsync(hx.data(), F->p_x.p, nr * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hs.data(), F->p_s.p, nr * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(ht.data(), F->p_tmax.p, nr * sizeof(double), hipMemcpyDeviceToHost, s));
        if (ng) HIPCHK(hipMemcpyAsync(hn.data(), F->normsum.p, ng * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sl.data(), F->sample_len[0].p, nf * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sr.data(), F->sample_rules[0].p, sr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        hsample.resize(nf);
        for (uint64_t f = 0; f < nf; ++f) hsample[f].assign(sr.begin() + F->h_sample_off[f], sr.begin() + F->h_sample_off[f] + sl[f]);
      }
      const double sdev = F->pi_stddev;
      const double q0 = gibbs_norm_cdf((0.0 - 1.0) / sdev), qrem = 1.0 - q0;
      std::vector<double> sc(nexti, 1.0);
      double ln_a2 = 0.0;
      for (uint32_t k = 1; k < nexti; ++k) {
        sc[k] = 1.0 + sdev * gibbs_norm_quantile(q0 + gibbs_uniform(o->seed, iter, 0xfffffffeu, k) * qrem);
        const double d_old = 1.0 / sc[k] - 1.0, d_new = sc[k] - 1.0;
        ln_a2 += (d_new * d_new - d_old * d_old) / (2.0 * sdev * sdev);
      }
      auto cache_prob_all = [&]() {
        std::vector<double> cc = prior, cs = pn;
        double lp = 0.0;
        for (uint64_t f = 0; f < nf; ++f)
          for (uint32_t r : hsample[f]) {
            const uint32_t n = F->h_norm[r];
            double q = prior[r];
            if (n != F_NONORM) {
              q = cc[r] / cs[n];
              cc[r] += 1.0;
              cs[n] += 1.0;
            }
            lp += std::log(q);
          }
        return lp;
      };
      auto scale = [&](bool invert) {
        std::fill(pn.begin(), pn.end(), 0.0);
        for (uint32_t r = 0; r < nr; ++r) {
          const uint32_t n = F->h_norm[r];
          if (n == F_NONORM) continue;
          const uint32_t i = meta[n + 1];
          if (i > 0) {
            double fct = sc[i];
            if (invert) fct = 1.0 / fct;
            const double s2 = fct * prior[r], d = s2 - prior[r];
            hs[r] += d * ht[r];
            hx[r] += d;
            hn[n] += d;
            prior[r] = s2;
          }
          pn[n] += prior[r];
        }
      };
      const double p1 = cache_prob_all();
      scale(false);
      const double p2 = cache_prob_all();
      const double a = std::exp((p2 - p1) + ln_a2);
      const bool accept = gibbs_uniform(o->seed, iter, 0xffffffffu, 0) < a;
      if (!accept)
        scale(true);
      else
        for (uint32_t k = 1; k < nexti; ++k) F->pi_cumulative[k - 1] *= sc[k];
      HIPCHK(hipMemcpyAsync(F->p_prior.p, prior.data(), nr * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(F->prior_norm.p, pn.data(), ng * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(F->p_x.p, hx.data(), nr * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(F->normsum.p, hn.data(), ng * sizeof(double), hipMemcpyHostToDevice, s));
      if (exact_dev) HIPCHK(hipMemcpyAsync(F->p_s.p, hs.data(), nr * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
      double* tr = F->pi_trace.data() + (size_t)iter * 6;
      tr[0] = 1;
      tr[1] = accept ? 1 : 0;
      tr[2] = p1;
      tr[3] = p2;
      tr[4] = std::exp(ln_a2);
      tr[5] = a;
    }
    if (iter_logprob) iter_logprob[iter] = cache_lp;
    if (iter_cheap_logprob) iter_cheap_logprob[iter] = cheap_lp;
  }
  // finalize_cumulative_counts + from_gibbs
  std::vector<double> x(nr), sacc(nr), tm(nr);
  if (x_clk.n) {
    unsigned long long c[8];
    HIPCHK(hipMemcpy(c, x_clk.p, sizeof c, hipMemcpyDeviceToHost));
    fprintf(stderr, "[carmel_hip] forest_exact cycles per forest: wait+proposal %.0f, inside %.0f, walk %.0f, entries+counts %.0f (register path: %llu forests x sweeps, LDS path: %llu)\n",
            c[0] / (double)c[4], c[1] / (double)c[4], c[2] / (double)c[4], c[3] / (double)c[4], c[4], c[5]);
  }
  if (o->mode == 0 && !exact_dev) {
    x = hx;
    sacc = hs;
    tm = ht;
  } else {
    HIPCHK(hipMemcpyAsync(x.data(), F->p_x.p, nr * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(sacc.data(), F->p_s.p, nr * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tm.data(), F->p_tmax.p, nr * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (cur != 0) {  // keep the final samples in buffer 0 for carmel_hip_forests_get_sample
      std::swap(F->sample_len[0].p, F->sample_len[1].p);
      std::swap(F->sample_rules[0].p, F->sample_rules[1].p);
    }
  }
  final_weights(x, sacc, tm, lw);
  F->h_final_x = x;
  if (trace_buf.n) {
    std::vector<unsigned long long> h(trace_buf.n);
    HIPCHK(hipMemcpy(h.data(), trace_buf.p, trace_buf.bytes(), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "wb")) {
      fwrite(h.data(), 8, h.size(), f);
      fclose(f);
    }
  }
  return carmel_hip_forests_set_weights(F, lw.data());
}

int carmel_hip_forests_set_alphas(carmel_hip_forests* F, const double* alpha_per_rule, uint32_t n) {
  if (!F) return fail(CARMEL_HIP_ERR_ARG, "null handle");
  if (alpha_per_rule && n)
    F->h_alphas.assign(alpha_per_rule, alpha_per_rule + n);
  else
    F->h_alphas.clear();
  return CARMEL_HIP_OK;
}

int carmel_hip_forests_get_sample(carmel_hip_forests* F, uint64_t forest, uint32_t* rules, uint32_t* n) {
  if (!F || !n || forest >= F->n_forests) return fail(CARMEL_HIP_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(F->device));
  uint32_t len = 0;
  HIPCHK(hipMemcpyAsync(&len, F->sample_len[0].p + forest, sizeof len, hipMemcpyDeviceToHost, F->stream));
  HIPCHK(hipStreamSynchronize(F->stream));
  if (rules && len)
    HIPCHK(hipMemcpyAsync(rules, F->sample_rules[0].p + F->h_sample_off[forest], len * 4, hipMemcpyDeviceToHost, F->stream));
  HIPCHK(hipStreamSynchronize(F->stream));
  *n = len;
  return CARMEL_HIP_OK;
}
uint32_t carmel_hip_forests_max_sample(carmel_hip_forests* F) { return F ? F->max_sample : 0; }

// Replaces FForest::compute_viterbi + write_viterbi's walk (forest.hpp:507-632) for every forest, with the current weights.
int carmel_hip_forests_viterbi(carmel_hip_forests* F, double* best_logprob) {
  if (!F || !best_logprob) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(F->device));
  hipStream_t s = F->stream;
  const uint64_t nf = F->n_forests;
  if (!F->sample_len[0].n) HIPCHK(F->sample_len[0].alloc(nf));
  if (!F->sample_rules[0].n) HIPCHK(F->sample_rules[0].alloc(F->h_sample_off.back()));
  if (!F->sample_hdr.n) HIPCHK(F->sample_hdr.alloc(F->h_sample_off.back()));
  DevBuf<double> best;
  HIPCHK(best.alloc(nf));
  ForestArgs A;
  fill_args(F, A);
  A.sample_len = F->sample_len[0].p;
  A.sample_rules = F->sample_rules[0].p;
  A.sample_hdr = F->sample_hdr.p;
  const uint32_t stack_lds = 32u;
  HIPCHK(fork_side(F, s));
  for (size_t ci = 0; ci < F->classes.size(); ++ci) {
    const auto& c = F->classes[ci];
    A.first_group = c.first;
    const size_t col = (size_t)c.max_nodes * 64 * sizeof(double), stk = (size_t)stack_lds * 64 * 4;
    if (col * 2 > F_LDS_LIMIT) {  // (the class has room for two columns per group in gcol: the E-step's)
      A.gcol = F->gcol.p + F->gcol_off[ci];
      A.gcol_stride = (uint64_t)2 * c.max_nodes * 64;
      hipLaunchKernelGGL(forest_viterbi_kernel<true>, dim3(c.count), dim3(64), stk, class_stream(F, s, ci), A, F->max_sample,
                         c.max_nodes, stack_lds, best.p);
      continue;
    }
    if (col + stk > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)forest_viterbi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(col + stk));
    hipLaunchKernelGGL(forest_viterbi_kernel<false>, dim3(c.count), dim3(64), col + stk, class_stream(F, s, ci), A,
                       F->max_sample, c.max_nodes, stack_lds, best.p);
  }
  HIPCHK(join_side(F, s));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(best_logprob, best.p, nf * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}

int carmel_hip_forests_get_viterbi(carmel_hip_forests* F, uint64_t forest, uint32_t* rules, uint32_t* arity, uint32_t* n) {
  if (!F || !n || forest >= F->n_forests || !F->sample_hdr.n || !F->sample_len[0].n)
    return fail(CARMEL_HIP_ERR_ARG, "bad argument (carmel_hip_forests_viterbi first)");
  HIPCHK(hipSetDevice(F->device));
  uint32_t len = 0;
  HIPCHK(hipMemcpyAsync(&len, F->sample_len[0].p + forest, sizeof len, hipMemcpyDeviceToHost, F->stream));
  HIPCHK(hipStreamSynchronize(F->stream));
  if (rules && arity && len) {
    HIPCHK(hipMemcpyAsync(rules, F->sample_rules[0].p + F->h_sample_off[forest], len * 4, hipMemcpyDeviceToHost, F->stream));
    HIPCHK(hipMemcpyAsync(arity, F->sample_hdr.p + F->h_sample_off[forest], len * 4, hipMemcpyDeviceToHost, F->stream));
  }
  HIPCHK(hipStreamSynchronize(F->stream));
  *n = len;
  return CARMEL_HIP_OK;
}


}  // extern "C"
