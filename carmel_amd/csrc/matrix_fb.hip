// matrix_fb.hip -- `carmel --matrix-fb`: the E-step over the dense (input position x output position x state) matrix
// instead of derivation lattices.
//
// Replaces forward_backward::matrix_compute / matrix_fb / estimate_matrix / matrix_count
// (/root/reference/carmel/src/train.cc:698-745, 747-759, 776-860, 288-296) and matrix_io_index (train.cc:80-100).
// The reference fills f[i][o][s] cell by cell: inside a cell the *e*:*e* arcs in a topological order of the epsilon
// graph (train.cc:716-722), then every state's arcs that consume the next input symbol, the next output symbol or both
// (train.cc:723-741); the backward matrix is the same walk over the reversed strings and arcs (train.cc:254-261), and an
// arc's count is the sum over all cells of f[i][o][src] * w * b[i + di][o + do][dst], scaled by weight / prob
// (train.cc:826-853).  No lattice is built and nothing is pruned: dead cells simply hold zero.
//
// Here: one workgroup per training pair, the two matrices in a slab of HBM owned by the workgroup (the legacy mode is
// for small transducers: (|in| + 1)(|out| + 1) * states doubles each).  The cells of an anti-diagonal i + o = d depend
// only on diagonals d - 1 and d - 2, so the threads take (cell of the diagonal, state) and PULL: a state's value is the
// streaming log-sum over its in-arcs whose labels match the cell (arcs grouped by destination; by source for the
// backward matrix), then the *e*:*e* arcs level by level of the epsilon graph (longest-path levels from the host; an
// epsilon cycle is refused -- the reference walks such a graph in a depth-first order that drops its back edges).  Counts:
// one thread per arc sums its cells in row-major order into the workgroup's own count row; the rows are added up in a
// fixed order afterwards, so the result is reproducible run to run.  Log-semiring arithmetic as in kernels.hip (running
// maximum + scaled sum per state; counts linear f64).
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>
#include "engine.hpp"

namespace carmel_hip {

#define MNEG_INF (-__builtin_huge_val())

struct MLse {
  double m, acc;
  __device__ __forceinline__ void init() {
    m = MNEG_INF;
    acc = 0.0;
  }
  __device__ __forceinline__ void add(double x) {
    if (x == MNEG_INF) return;
    if (x <= m) {
      acc += exp(x - m);
    } else {
      acc = (m == MNEG_INF) ? 1.0 : acc * exp(m - x) + 1.0;
      m = x;
    }
  }
  __device__ __forceinline__ double value() const { return acc == 1.0 ? m : (acc > 0.0 ? m + log(acc) : MNEG_INF); }
};

struct MatrixArgs {
  const uint32_t *a_src, *a_dst, *a_in, *a_out;
  const double* logw;
  const uint32_t *in_off, *in_arc;      // arcs other than *e*:*e*, grouped by destination
  const uint32_t *out_off, *out_arc;    // ... by source
  const uint32_t *ein_off, *ein_arc;    // *e*:*e* arcs by destination
  const uint32_t *eout_off, *eout_arc;  // ... by source
  const uint32_t *flev_off, *flev_state;  // states with *e*:*e* in-arcs, by level of the epsilon graph (longest path to them)
  const uint32_t *blev_off, *blev_state;  // states with *e*:*e* out-arcs, by longest epsilon path from them
  uint32_t n_flev, n_blev;
  const uint64_t *cin_off, *cout_off;
  const uint32_t *cin_sym, *cout_sym;
  const double* pair_w;  // < 0: the pair was dropped (no derivation)
  double* slab;
  uint64_t slab_stride;  // doubles per workgroup
  double* wg_counts;     // gridDim.x rows of n_arcs
  double* pair_logprob;
  uint32_t n_states, n_arcs, final_state;
  uint64_t n_pairs;
};

__global__ __launch_bounds__(256) void matrix_fb_kernel(MatrixArgs A) {
  const uint32_t S = A.n_states;
  double* const f = A.slab + (size_t)blockIdx.x * A.slab_stride;
  double* const mycounts = A.wg_counts + (size_t)blockIdx.x * A.n_arcs;
  for (uint32_t a = threadIdx.x; a < A.n_arcs; a += 256) mycounts[a] = 0.0;
  for (uint64_t p = blockIdx.x; p < A.n_pairs; p += gridDim.x) {
    const double pw = A.pair_w[p];
    if (pw < 0.0) continue;
    const uint32_t* __restrict__ inS = A.cin_sym + A.cin_off[p];
    const uint32_t* __restrict__ outS = A.cout_sym + A.cout_off[p];
    const uint32_t nI = (uint32_t)(A.cin_off[p + 1] - A.cin_off[p]), nO = (uint32_t)(A.cout_off[p + 1] - A.cout_off[p]);
    const uint32_t W = nO + 1, cells = (nI + 1) * W;
    double* const b = f + (size_t)cells * S;
    __syncthreads();  // the previous pair's count loop still reads the slab
    // ---------- forward: diagonals ascending ----------
    for (uint32_t d = 0; d <= nI + nO; ++d) {
      const uint32_t ilo = d > nO ? d - nO : 0u, ihi = d < nI ? d : nI, nc = ihi - ilo + 1;
      for (uint32_t idx = threadIdx.x; idx < nc * S; idx += 256) {
        const uint32_t i = ilo + idx / S, t = idx % S, o = d - i;
        MLse acc;
        acc.init();
        if (d == 0 && t == 0) acc.add(0.0);  // the start state is state 0 (train.cc:260)
        for (uint32_t k = A.in_off[t]; k < A.in_off[t + 1]; ++k) {
          const uint32_t a = A.in_arc[k], ai = A.a_in[a], ao = A.a_out[a];
          const uint32_t di = ai != 0u, dn = ao != 0u;
          if (i < di || o < dn) continue;
          if (di && ai != inS[i - 1]) continue;
          if (dn && ao != outS[o - 1]) continue;
          acc.add(f[(size_t)((i - di) * W + (o - dn)) * S + A.a_src[a]] + A.logw[a]);
        }
        f[(size_t)(i * W + o) * S + t] = acc.value();
      }
      __syncthreads();
      for (uint32_t L = 0; L < A.n_flev; ++L) {
        const uint32_t s0 = A.flev_off[L], ns = A.flev_off[L + 1] - s0;
        for (uint32_t idx = threadIdx.x; idx < nc * ns; idx += 256) {
          const uint32_t i = ilo + idx / ns, t = A.flev_state[s0 + idx % ns], o = d - i;
          double* cell = f + (size_t)(i * W + o) * S;
          MLse acc;
          acc.init();
          acc.add(cell[t]);
          for (uint32_t k = A.ein_off[t]; k < A.ein_off[t + 1]; ++k) {
            const uint32_t a = A.ein_arc[k];
            acc.add(cell[A.a_src[a]] + A.logw[a]);
          }
          cell[t] = acc.value();
        }
        __syncthreads();
      }
    }
    const double fin = f[(size_t)(nI * W + nO) * S + A.final_state];
    if (threadIdx.x == 0) A.pair_logprob[p] = fin;
    if (fin == MNEG_INF) continue;  // (uniform: every thread read the same value)
    // ---------- backward: diagonals descending ----------
    for (uint32_t dd = 0; dd <= nI + nO; ++dd) {
      const uint32_t d = nI + nO - dd;
      const uint32_t ilo = d > nO ? d - nO : 0u, ihi = d < nI ? d : nI, nc = ihi - ilo + 1;
      for (uint32_t idx = threadIdx.x; idx < nc * S; idx += 256) {
        const uint32_t i = ilo + idx / S, s = idx % S, o = d - i;
        MLse acc;
        acc.init();
        if (dd == 0 && s == A.final_state) acc.add(0.0);
        for (uint32_t k = A.out_off[s]; k < A.out_off[s + 1]; ++k) {
          const uint32_t a = A.out_arc[k], ai = A.a_in[a], ao = A.a_out[a];
          const uint32_t di = ai != 0u, dn = ao != 0u;
          if (i + di > nI || o + dn > nO) continue;
          if (di && ai != inS[i]) continue;
          if (dn && ao != outS[o]) continue;
          acc.add(A.logw[a] + b[(size_t)((i + di) * W + (o + dn)) * S + A.a_dst[a]]);
        }
        b[(size_t)(i * W + o) * S + s] = acc.value();
      }
      __syncthreads();
      for (uint32_t L = 0; L < A.n_blev; ++L) {
        const uint32_t s0 = A.blev_off[L], ns = A.blev_off[L + 1] - s0;
        for (uint32_t idx = threadIdx.x; idx < nc * ns; idx += 256) {
          const uint32_t i = ilo + idx / ns, s = A.blev_state[s0 + idx % ns], o = d - i;
          double* cell = b + (size_t)(i * W + o) * S;
          MLse acc;
          acc.init();
          acc.add(cell[s]);
          for (uint32_t k = A.eout_off[s]; k < A.eout_off[s + 1]; ++k) {
            const uint32_t a = A.eout_arc[k];
            acc.add(A.logw[a] + cell[A.a_dst[a]]);
          }
          cell[s] = acc.value();
        }
        __syncthreads();
      }
    }
    // ---------- counts (train.cc:826-853): scratch = sum over cells, counts += weight / prob * scratch ----------
    for (uint32_t a = threadIdx.x; a < A.n_arcs; a += 256) {
      const uint32_t ai = A.a_in[a], ao = A.a_out[a], di = ai != 0u, dn = ao != 0u;
      const uint32_t src = A.a_src[a], dst = A.a_dst[a];
      const double w = A.logw[a] - fin;
      double sum = 0.0;
      if (w > MNEG_INF)
        for (uint32_t i = 0; i + di <= nI; ++i) {
          if (di && ai != inS[i]) continue;
          for (uint32_t o = 0; o + dn <= nO; ++o) {
            if (dn && ao != outS[o]) continue;
            const double fv = f[(size_t)(i * W + o) * S + src];
            const double bv = b[(size_t)((i + di) * W + (o + dn)) * S + dst];
            if (fv > MNEG_INF && bv > MNEG_INF) sum += exp(fv + w + bv);
          }
        }
      mycounts[a] += pw * sum;
    }
  }
}

__global__ __launch_bounds__(256) void matrix_counts_kernel(const double* wg_counts, uint32_t n_wg, uint32_t n_arcs, double* counts) {
  const uint32_t a = blockIdx.x * 256 + threadIdx.x;
  if (a >= n_arcs) return;
  double v = 0.0;
  for (uint32_t g = 0; g < n_wg; ++g) v += wg_counts[(size_t)g * n_arcs + a];
  counts[a] = v;
}

// device tables of the matrix E-step (owned by the trainer through an opaque pointer)
struct MatrixState {
  DevBuf<uint32_t> a_src, a_dst, a_in, a_out, in_off, in_arc, out_off, out_arc, ein_off, ein_arc, eout_off, eout_arc, flev_off,
      flev_state, blev_off, blev_state, cin_sym, cout_sym;
  DevBuf<uint64_t> cin_off, cout_off;
  DevBuf<double> slab, wg_counts;
  uint32_t n_flev = 0, n_blev = 0, n_wg = 0;
  uint64_t slab_stride = 0;
};

static void group_by(const std::vector<uint32_t>& key, const std::vector<uint32_t>& arcs, uint32_t n_states, std::vector<uint32_t>& off,
                     std::vector<uint32_t>& list) {
  off.assign(n_states + 1, 0);
  for (uint32_t a : arcs) ++off[key[a] + 1];
  for (uint32_t s = 0; s < n_states; ++s) off[s + 1] += off[s];
  list.resize(arcs.size());
  std::vector<uint32_t> at(off.begin(), off.end() - 1);
  for (uint32_t a : arcs) list[at[key[a]]++] = a;  // arc-id order inside a state
}

// longest-path levels of the *e*:*e* graph; false if it has a cycle.  level_of[s] = 0 for states no epsilon arc enters.
static bool eps_levels(uint32_t n, const std::vector<uint32_t>& from, const std::vector<uint32_t>& to, const std::vector<uint32_t>& eps,
                       std::vector<uint32_t>& level_of) {
  std::vector<uint32_t> indeg(n, 0), out_off, out_list;
  group_by(from, eps, n, out_off, out_list);
  for (uint32_t a : eps) ++indeg[to[a]];
  level_of.assign(n, 0);
  std::vector<uint32_t> queue;
  for (uint32_t s = 0; s < n; ++s)
    if (!indeg[s]) queue.push_back(s);
  size_t done = 0;
  while (done < queue.size()) {
    const uint32_t s = queue[done++];
    for (uint32_t k = out_off[s]; k < out_off[s + 1]; ++k) {
      const uint32_t t = to[out_list[k]];
      level_of[t] = std::max(level_of[t], level_of[s] + 1);
      if (!--indeg[t]) queue.push_back(t);
    }
  }
  return queue.size() == n;
}

static void levels_to_lists(const std::vector<uint32_t>& level_of, std::vector<uint32_t>& off, std::vector<uint32_t>& states) {
  uint32_t mx = 0;
  for (uint32_t l : level_of) mx = std::max(mx, l);
  off.assign(mx + 1, 0);  // levels 1 .. mx -> slots 0 .. mx - 1
  for (uint32_t l : level_of)
    if (l) ++off[l];
  for (uint32_t l = 1; l <= mx; ++l) off[l] += off[l - 1];
  states.resize(off[mx]);
  std::vector<uint32_t> at(off.begin(), off.end());
  for (uint32_t s = 0; s < level_of.size(); ++s)
    if (level_of[s]) states[at[level_of[s] - 1]++] = s;
}

void matrix_release(void* p) { delete (MatrixState*)p; }

int matrix_setup(carmel_hip_trainer* t, void** out) {
  const HostWfst& w = t->w;
  if (w.n_arcs >= (1ull << 31)) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "--matrix-fb: too many arcs");
  std::vector<uint32_t> eps, other;
  for (uint32_t a = 0; a < (uint32_t)w.n_arcs; ++a) (w.in[a] == 0 && w.out[a] == 0 ? eps : other).push_back(a);
  std::vector<uint32_t> flev, blev;
  if (!eps_levels(w.n_states, w.src, w.dst, eps, flev) || !eps_levels(w.n_states, w.dst, w.src, eps, blev))
    return fail(CARMEL_HIP_ERR_UNSUPPORTED,
                "--matrix-fb: the *e*:*e* arcs form a cycle (the matrix walk needs a topological order of the epsilon graph, "
                "train.cc:341-356); train on derivation lattices instead");
  MatrixState* M = new MatrixState;
  hipStream_t s = t->stream;
  std::vector<uint32_t> off, list;
#define MUP(buf, vec)                                       \
  do {                                                      \
    hipError_t e_ = M->buf.upload(vec, s);                  \
    if (e_ != hipSuccess) {                                 \
      delete M;                                             \
      return fail(CARMEL_HIP_ERR_HIP, hipGetErrorString(e_)); \
    }                                                       \
  } while (0)
  MUP(a_src, w.src);
  MUP(a_dst, w.dst);
  MUP(a_in, w.in);
  MUP(a_out, w.out);
  group_by(w.dst, other, w.n_states, off, list);
  MUP(in_off, off);
  MUP(in_arc, list);
  group_by(w.src, other, w.n_states, off, list);
  MUP(out_off, off);
  MUP(out_arc, list);
  group_by(w.dst, eps, w.n_states, off, list);
  MUP(ein_off, off);
  MUP(ein_arc, list);
  group_by(w.src, eps, w.n_states, off, list);
  MUP(eout_off, off);
  MUP(eout_arc, list);
  levels_to_lists(flev, off, list);
  M->n_flev = (uint32_t)off.size() - 1;
  MUP(flev_off, off);
  MUP(flev_state, list);
  levels_to_lists(blev, off, list);
  M->n_blev = (uint32_t)off.size() - 1;
  MUP(blev_off, off);
  MUP(blev_state, list);
  const HostCorpus& c = t->corpus;
  MUP(cin_off, c.in_off);
  MUP(cout_off, c.out_off);
  MUP(cin_sym, c.in_sym);
  MUP(cout_sym, c.out_sym);
  uint64_t max_cells = 1, max_diag = 1;
  for (uint64_t p = 0; p < c.n_pairs; ++p) {
    const uint64_t ni = c.in_off[p + 1] - c.in_off[p], no = c.out_off[p + 1] - c.out_off[p];
    max_cells = std::max(max_cells, (ni + 1) * (no + 1));
    max_diag = std::max(max_diag, std::min(ni, no) + 1);
  }
  if (max_diag * w.n_states >= (1ull << 31) || max_cells >= (1ull << 31)) {
    delete M;
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "--matrix-fb: (cells of a diagonal) x states exceeds the kernel's 32-bit thread indices; train on derivation lattices");
  }
  M->slab_stride = 2 * max_cells * w.n_states;
  // (|in| + 1)(|out| + 1) * states doubles, twice, per resident pair: at most 1024 pairs at a time, inside a quarter of the
  // free memory
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  uint64_t n_wg = std::min<uint64_t>(std::max<uint64_t>(c.n_pairs, 1), 1024);
  while (n_wg > 1 && n_wg * M->slab_stride * 8 > free_b / 4) n_wg /= 2;
  if (n_wg * M->slab_stride * 8 > free_b / 2) {
    delete M;
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "--matrix-fb: the (input, output, state) matrices of the longest pair do not fit in device memory");
  }
  M->n_wg = (uint32_t)n_wg;
  if (M->slab.alloc(n_wg * M->slab_stride) != hipSuccess || M->wg_counts.alloc(n_wg * w.n_arcs) != hipSuccess) {
    delete M;
    return fail(CARMEL_HIP_ERR_HIP, "--matrix-fb: hipMalloc of the matrices failed");
  }
#undef MUP
  if (hipStreamSynchronize(s) != hipSuccess) {
    delete M;
    return fail(CARMEL_HIP_ERR_HIP, "--matrix-fb: upload failed");
  }
  *out = M;
  return CARMEL_HIP_OK;
}

int matrix_estimate(carmel_hip_trainer* t, void* state, hipStream_t s) {
  MatrixState* M = (MatrixState*)state;
  MatrixArgs A;
  A.a_src = M->a_src.p;
  A.a_dst = M->a_dst.p;
  A.a_in = M->a_in.p;
  A.a_out = M->a_out.p;
  A.logw = t->arc_logw.p;
  A.in_off = M->in_off.p;
  A.in_arc = M->in_arc.p;
  A.out_off = M->out_off.p;
  A.out_arc = M->out_arc.p;
  A.ein_off = M->ein_off.p;
  A.ein_arc = M->ein_arc.p;
  A.eout_off = M->eout_off.p;
  A.eout_arc = M->eout_arc.p;
  A.flev_off = M->flev_off.p;
  A.flev_state = M->flev_state.p;
  A.blev_off = M->blev_off.p;
  A.blev_state = M->blev_state.p;
  A.n_flev = M->n_flev;
  A.n_blev = M->n_blev;
  A.cin_off = M->cin_off.p;
  A.cout_off = M->cout_off.p;
  A.cin_sym = M->cin_sym.p;
  A.cout_sym = M->cout_sym.p;
  A.pair_w = t->pair_w.p;
  A.slab = M->slab.p;
  A.slab_stride = M->slab_stride;
  A.wg_counts = M->wg_counts.p;
  A.pair_logprob = t->pair_logprob.p;
  A.n_states = t->w.n_states;
  A.n_arcs = (uint32_t)t->w.n_arcs;
  A.final_state = t->w.final_state;
  A.n_pairs = t->corpus.n_pairs;
  hipLaunchKernelGGL(matrix_fb_kernel, dim3(M->n_wg), dim3(256), 0, s, A);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(matrix_counts_kernel, dim3((A.n_arcs + 255) / 256), dim3(256), 0, s, M->wg_counts.p, M->n_wg, A.n_arcs,
                     t->counts_ptr());
  HIPCHK(hipGetLastError());
  return CARMEL_HIP_OK;
}

}  // namespace carmel_hip
