// gibbs_lane.hip — `carmel --crp --crp-parallel` with one block per LANE (see gibbs_lane.hpp).
//
// Replaces, for the stale-count sweep over trellis lattices: /root/reference/carmel/src/gibbs.cc:306-371 (resample_block: proposal
// weights from the counts with the block's own sample out), carmel/src/derivations.h:318-375 (random_path: backward sweep, walk),
// graehl/shared/random.ipp:111-127 (choose_p).  Same uniforms (one per level of a block), same order of every state's
// subtractions (the reference's list order) as gibbs_exact.hip's kernels: the same sample.
//
//   sweep  a lane streams its lattice's arcs, levels from the goal's down, a state's arcs in list order.  Per arc: two 16-byte
//          records (global ids for the count gathers; places in the level windows, local ids, flags), four gathers from the
//          snapshot counts, four byte reads from the lane's own-sample tables (how often the block's previous path uses this
//          parameter / norm group: indexed by the block's LOCAL numbering, filled from the previous path before the sweep), two
//          divisions, one read of the destination's value in the next level's window; the share and the weight go to a scratch row
//          (16 bytes, coalesced).  A finished state's total goes into this level's window; a finished level's largest total sets
//          the power of two the next level's terms are divided by (only ratios within a state matter: no underflow however long
//          the sentence, and no second pass).  Records are requested four rows ahead and the counts two, in rings indexed by
//          constants (lesson 3).
//   walk   a state per level: its rows' shares (and records) in one round of loads, summed in list order (the total the sweep
//          wrote into the window, bit for bit), subtracted from u x total in the same order; the chosen row's record names the
//          next state's rows.  The path is written as {row, local ids, place in the block's sample}: what the next sweep's
//          tables and the recount need.
#include <algorithm>
#include <numeric>
#include <unordered_map>
#include "gibbs_lane.hpp"
#include "rng.hpp"

namespace carmel_hip {

#define GL_SYNC()                                        \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_wave_barrier();                     \
  } while (0)
#define GL_CH 2    // rows per stage of the pipeline
#define GL_R 4     // stages in flight: records are requested GL_R - 1 stages ahead, the counts one
#define GL_SLACK 8 // zero rows behind a group's last (the pipeline's look-ahead reads them)

struct GlProd {  // a running product as mantissa x 2^exponent
  double m;
  long long e;
  __device__ __forceinline__ void mul(double x) {
    int t;
    m = frexp(m * x, &t);
    e += t;
  }
  __device__ __forceinline__ double ln() const { return log(m) + (double)e * 0.69314718055994530942; }
};

template <bool INIT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gibbs_lane_kernel(GlArgs A) {
  extern __shared__ __attribute__((aligned(16))) double gl_lds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t gi = A.first_group + blockIdx.x;
  const GlGroup g = A.groups[gi];
  const GlLane L = A.lanes[(size_t)gi * 64 + lane];
  const bool active = L.block != GL_NONE;
  const uint32_t W = A.W, LP = A.LP;
  double* win = gl_lds;                                            // [2][W][64]: backward values of this level and the next
  unsigned char* cp = (unsigned char*)(win + (size_t)2 * W * 64);  // [LP][64]: uses of a local parameter by the previous path
  unsigned char* cn = cp + (size_t)LP * 64;                        // [LN][64]: ... of a local norm group
  unsigned long long t0 = A.phase_clk ? __builtin_readcyclecounter() : 0;
  {
    uint32_t* tz = (uint32_t*)cp;
    for (uint32_t i = lane; i < (LP + A.LN) * 16u; i += 64) tz[i] = 0u;
  }
  GL_SYNC();
  const uint32_t path = active ? L.path : 0u;
  if (A.have_old) {
    const uint4* __restrict__ so = A.samp_old + 2 * (g.samp_base + lane);
    for (uint32_t k = 0; k < g.path; k += 4) {
      uint4 e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = so[(size_t)(k + q) * 128];  // (two records an entry; the buffer is padded by a few entries)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (k + q < path) {
          const uint32_t l0 = e[q].y & 0xffffu, l1 = e[q].y >> 16, n0 = e[q].z & 0xffffu, n1 = e[q].z >> 16;
          if (l0 != 0xffffu) cp[l0 * 64u + lane] += 1;
          if (l1 != 0xffffu) cp[l1 * 64u + lane] += 1;
          if (n0 != 0xffffu) cn[n0 * 64u + lane] += 1;
          if (n1 != 0xffffu) cn[n1 * 64u + lane] += 1;
        }
    }
  }
  // the goal: the one state of the last level, value 1, in the window the first level's arcs read
  win[lane] = 1.0;
  unsigned long long t1 = A.phase_clk ? __builtin_readcyclecounter() : 0;
  // ---- backward sweep.  Every load and store of the loop is unconditional (a row without an arc, or without a second parameter,
  // reads entry 0 and drops it): no branch between the stages, so the waits the compiler places count exactly the loads in flight ----
  const uint4* __restrict__ pa = A.recA + g.rec_base + lane;
  const uint4* __restrict__ pb = A.recB + g.rec_base + lane;
  double2* __restrict__ psw = A.sw + g.rec_base + lane;
  double* __restrict__ pwq = A.wq + g.rec_base + lane;
  const double wt = L.wt;
  double acc = 0.0, mx = 0.0;
  int e_next = 0;
  uint32_t par = 0;   // the window the arcs READ (the next level's); they write the other
  uint4 ra[GL_R][GL_CH], rb[GL_R][GL_CH];
  double gx0[2][GL_CH], gs0[2][GL_CH], gx1[2][GL_CH], gs1[2][GL_CH], gin[2][GL_CH];
#define GL_LOAD(c, S)                                                    \
  _Pragma("unroll") for (int q = 0; q < GL_CH; ++q) {                    \
    ra[S][q] = pa[(size_t)((c) * GL_CH + q) * 64];                       \
    rb[S][q] = pb[(size_t)((c) * GL_CH + q) * 64];                       \
  }
#define GL_GATHER(c, S, T)                                                                            \
  _Pragma("unroll") for (int q = 0; q < GL_CH; ++q) {                                                 \
    const uint32_t c_ = rb[S][q].x;                                                                   \
    const double* s0_ = (c_ & GL_FIX0) ? A.p_prior : A.p_x;                                           \
    const double* s1_ = (c_ & GL_FIX1) ? A.p_prior : A.p_x;                                           \
    gx0[T][q] = s0_[(c_ & GL_HAS0) ? ra[S][q].x : 0u];                                                \
    gs0[T][q] = A.normsum[((c_ & GL_HAS0) && !(c_ & GL_FIX0)) ? ra[S][q].z : 0u];                     \
    gx1[T][q] = s1_[(c_ & GL_HAS1) ? ra[S][q].y : 0u];                                                \
    gs1[T][q] = A.normsum[((c_ & GL_HAS1) && !(c_ & GL_FIX1)) ? ra[S][q].w : 0u];                     \
    if (INIT) gin[T][q] = A.init_logw[A.arc_id[g.rec_base + (size_t)((c) * GL_CH + q) * 64 + lane]]; \
  }
#define GL_COMPUTE(c, S, T)                                                                            \
  _Pragma("unroll") for (int q = 0; q < GL_CH; ++q) {                                                  \
    const uint4 b_ = rb[S][q];                                                                         \
    const uint32_t c_ = b_.x;                                                                          \
    const uint32_t l0 = b_.y & 0xffffu, l1 = b_.y >> 16, n0 = b_.z & 0xffffu, n1 = b_.z >> 16;         \
    const uint32_t u0 = cp[(l0 == 0xffffu ? 0u : l0) * 64u + lane], u1 = cp[(l1 == 0xffffu ? 0u : l1) * 64u + lane]; \
    const uint32_t m0 = cn[(n0 == 0xffffu ? 0u : n0) * 64u + lane], m1 = cn[(n1 == 0xffffu ? 0u : n1) * 64u + lane]; \
    const double bnext = win[((size_t)par * W + GL_DST(c_)) * 64 + lane];                              \
    const double x0_ = (c_ & GL_HAS0) ? gx0[T][q] : 1.0, s0_ = ((c_ & GL_HAS0) && !(c_ & GL_FIX0)) ? gs0[T][q] : 1.0; \
    const double x1_ = (c_ & GL_HAS1) ? gx1[T][q] : 1.0, s1_ = ((c_ & GL_HAS1) && !(c_ & GL_FIX1)) ? gs1[T][q] : 1.0; \
    const double a0 = x0_ - (double)(l0 == 0xffffu ? 0u : u0) * wt, d0 = s0_ - (double)(n0 == 0xffffu ? 0u : m0) * wt; \
    const double a1 = x1_ - (double)(l1 == 0xffffu ? 0u : u1) * wt, d1 = s1_ - (double)(n1 == 0xffffu ? 0u : m1) * wt; \
    const double wgt = (a0 * a1) / (d0 * d1); /* (one division: ~30 of a row's ~200 instructions) */   \
    const double gw = INIT ? exp(gin[T][q]) : wgt;                                                     \
    const double term = (c_ & GL_VALID) ? gw * ldexp(bnext, -e_next) : 0.0;                            \
    acc += term;                                                                                       \
    /* (a state's total rides with its LAST row's share: known there, and written with the row's own coalesced store) */ \
    psw[(size_t)((c) * GL_CH + q) * 64] = make_double2(term, (c_ & GL_STATE_LAST) ? acc : 0.0);        \
    pwq[(size_t)((c) * GL_CH + q) * 64] = wgt;                                                         \
    if (c_ & GL_STATE_LAST) {                                                                          \
      win[((size_t)(par ^ 1u) * W + GL_SRC(c_)) * 64 + lane] = acc;                                    \
      mx = fmax(mx, acc);                                                                              \
      acc = 0.0;                                                                                       \
    }                                                                                                  \
    if (c_ & GL_LEVEL_LAST) {                                                                          \
      int t_;                                                                                          \
      (void)frexp(mx, &t_);                                                                            \
      e_next = mx > 0.0 ? t_ : 0;                                                                      \
      mx = 0.0;                                                                                        \
      par ^= 1u;                                                                                       \
    }                                                                                                  \
  }
  const uint32_t nchunks = g.rows / GL_CH;  // (rows: a multiple of GL_R GL_CH; GL_SLACK zero rows behind them)
  GL_LOAD(0, 0)
  GL_LOAD(1, 1)
  GL_LOAD(2, 2)
  GL_GATHER(0, 0, 0)
  for (uint32_t i = 0; i < nchunks; i += GL_R) {
    GL_LOAD(i + 3, 3)
    GL_GATHER(i + 1, 1, 1)
    GL_COMPUTE(i, 0, 0)
    GL_LOAD(i + 4, 0)
    GL_GATHER(i + 2, 2, 0)
    GL_COMPUTE(i + 1, 1, 1)
    GL_LOAD(i + 5, 1)
    GL_GATHER(i + 3, 3, 1)
    GL_COMPUTE(i + 2, 2, 0)
    GL_LOAD(i + 6, 2)
    GL_GATHER(i + 4, 0, 0)
    GL_COMPUTE(i + 3, 3, 1)
  }
#undef GL_LOAD
#undef GL_GATHER
#undef GL_COMPUTE
  unsigned long long t2 = A.phase_clk ? __builtin_readcyclecounter() : 0;
  // ---- the walk (derivations.h:361-374; random.ipp:111-127): a state per level.  One round of loads per step: the state's
  // first four rows -- shares and records -- and its last row, whose entry carries the state's total (four rows or fewer: the
  // same line again); further rows only while some lane's choice is still above zero.  What the path needs of the chosen arc
  // beyond its destination -- its weight, its parameters -- is requested when the arc is chosen and written down a step later,
  // behind the next step's round ----
  const double2* __restrict__ rsw = A.sw + g.rec_base + lane;
  const double* __restrict__ rwq = A.wq + g.rec_base + lane;
  uint4* __restrict__ sn = A.samp_new + 2 * (g.samp_base + lane);
  GlProd cheap{1.0, 0};
  uint32_t cur = active ? L.start : 0u, pos = 0;
  bool pend = false;  // the previous step's arc is still to be written down
  uint32_t prow = 0;
  double pw = 1.0;
  uint4 pb4 = make_uint4(0, 0, 0, 0), pa4 = make_uint4(0, 0, 0, 0);
  for (uint32_t k = 0; k <= g.path; ++k) {
    const bool on = k < path;
    const uint32_t fr = cur >> 8, deg = on ? (cur & 0xffu) : 0u;
    const uint32_t lastr = fr + (deg ? deg - 1u : 0u);
    const double u = gibbs_uniform(A.seed, A.iter, L.block, k);
    double2 s4[4];
    uint4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t r = min(fr + (uint32_t)j, lastr);
      s4[j] = rsw[(size_t)r * 64];
      b4[j] = pb[(size_t)r * 64];
    }
    const double tot = rsw[(size_t)lastr * 64].y;
    if (pend) {
      sn[(size_t)(k - 1) * 128] = make_uint4(prow, pb4.y, pb4.z, pos);
      sn[(size_t)(k - 1) * 128 + 1] = pa4;  // (the chosen arc's parameters and norm groups: what the recount adds up)
      pos += GL_NPAR(pb4.x);
      cheap.mul(pw);
    }
    double choice = u * tot;
    bool done = deg == 0;
    uint32_t pick = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!done && (uint32_t)j < deg) {
        choice -= s4[j].x;
        pick = (uint32_t)j;
        pb4 = b4[j];
        done = choice < 0;
      }
    for (uint32_t c = 4; __any(!done && c < deg); c += 4) {
      double2 sx[4];
      uint4 bx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t r = min(fr + c + (uint32_t)j, lastr);
        sx[j] = rsw[(size_t)r * 64];
        bx[j] = pb[(size_t)r * 64];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!done && c + j < deg) {
          choice -= sx[j].x;
          pick = c + j;
          pb4 = bx[j];
          done = choice < 0;
        }
    }
    pend = on && deg;
    prow = fr + pick;
    pw = rwq[(size_t)prow * 64];
    pa4 = pa[(size_t)prow * 64];
    if (pend) cur = pb4.w;
  }
  double cl = cheap.ln();
  for (int o = 32; o > 0; o >>= 1) cl += __shfl_down(cl, o, 64);
  if (lane == 0) {
    unsafeAtomicAdd(A.iter_out + 1, cl);
    if (A.phase_clk) {
      const unsigned long long t3 = __builtin_readcyclecounter();
      atomicAdd(A.phase_clk + 0, t1 - t0);
      atomicAdd(A.phase_clk + 1, t2 - t1);
      atomicAdd(A.phase_clk + 2, t3 - t2);
      atomicAdd(A.phase_clk + 4, 64ull);
    }
  }
}

size_t gibbs_lane_lds_bytes(uint32_t W, uint32_t LP, uint32_t LN) { return (size_t)2 * W * 64 * 8 + ((size_t)LP + LN) * 64; }
hipError_t launch_gibbs_lane(const GlArgs& A, uint32_t n_groups, hipStream_t s) {
  if (!n_groups) return hipSuccess;
  const size_t lds = gibbs_lane_lds_bytes(A.W, A.LP, A.LN);
  if (lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)gibbs_lane_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gibbs_lane_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (A.init_logw)
    hipLaunchKernelGGL(gibbs_lane_kernel<true>, dim3(n_groups), dim3(64), lds, s, A);
  else
    hipLaunchKernelGGL(gibbs_lane_kernel<false>, dim3(n_groups), dim3(64), lds, s, A);
  return hipGetLastError();
}

// ---- counts of the new paths (the caller has set them to the priors): as gibbs_recount_tables_kernel, a workgroup's uses meet in
// two LDS tables first -- a tagger's frequent parameters are used by most blocks ----
__device__ __forceinline__ void gl_tab_add(uint32_t* keys, double* vals, uint32_t mask, uint32_t key, double v, double* global) {
  uint32_t h = (key * 2654435761u) >> 10;
  for (int probe = 0; probe < 8; ++probe, ++h) {
    const uint32_t at = h & mask;
    const uint32_t old = atomicCAS(keys + at, 0xffffffffu, key);
    if (old == 0xffffffffu || old == key) {
      __hip_atomic_fetch_add(vals + at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
  }
  unsafeAtomicAdd(global + key, v);
}
__global__ __launch_bounds__(1024) void gibbs_lane_recount_kernel(const GlGroup* groups, const GlLane* lanes, const uint4* recA, const uint4* samp,
                                                                  uint32_t n_groups, double* new_x, double* new_norm, uint32_t p_slots,
                                                                  uint32_t n_slots) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_lds[];
  double* pv = (double*)rc_lds;
  double* nv = pv + p_slots;
  uint32_t* pk = (uint32_t*)(nv + n_slots);
  uint32_t* nk = pk + p_slots;
  for (uint32_t i = threadIdx.x; i < p_slots; i += blockDim.x) {
    pk[i] = 0xffffffffu;
    pv[i] = 0.0;
  }
  for (uint32_t i = threadIdx.x; i < n_slots; i += blockDim.x) {
    nk[i] = 0xffffffffu;
    nv[i] = 0.0;
  }
  __syncthreads();
  // four wavefronts to a group, each taking every fourth stretch of four path entries: the entries' two records are one
  // coalesced 32-byte piece per lane
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u, nw = blockDim.x >> 6, sub = wv & 3u;
  for (uint32_t gi = blockIdx.x * (nw >> 2) + (wv >> 2); gi < n_groups; gi += gridDim.x * (nw >> 2)) {
    const GlGroup g = groups[gi];
    const GlLane L = lanes[(size_t)gi * 64 + lane];
    const uint32_t path = L.block != GL_NONE ? L.path : 0u;
    const uint4* __restrict__ sp = samp + 2 * (g.samp_base + lane);
    for (uint32_t k = 4 * sub; k < g.path; k += 16) {
      uint4 a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = sp[(size_t)(k + q) * 128 + 1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (k + q >= path) continue;
        if (a[q].x != GL_NONE && a[q].z != GL_NONE) {
          gl_tab_add(pk, pv, p_slots - 1, a[q].x, L.wt, new_x);
          if (new_norm) gl_tab_add(nk, nv, n_slots - 1, a[q].z, L.wt, new_norm);
        }
        if (a[q].y != GL_NONE && a[q].w != GL_NONE) {
          gl_tab_add(pk, pv, p_slots - 1, a[q].y, L.wt, new_x);
          if (new_norm) gl_tab_add(nk, nv, n_slots - 1, a[q].w, L.wt, new_norm);
        }
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < p_slots; i += blockDim.x)
    if (pk[i] != 0xffffffffu) unsafeAtomicAdd(new_x + pk[i], pv[i]);
  for (uint32_t i = threadIdx.x; i < n_slots && new_norm; i += blockDim.x)
    if (nk[i] != 0xffffffffu) unsafeAtomicAdd(new_norm + nk[i], nv[i]);
}
hipError_t launch_gibbs_lane_recount(const GlGroup* groups, const GlLane* lanes, const uint4* recA, const uint4* samp, uint32_t n_groups,
                                     double* new_x, double* new_norm, hipStream_t s) {
  if (!n_groups) return hipSuccess;
  const uint32_t p_slots = 8192, n_slots = 1024;
  const size_t lds = (size_t)(p_slots + n_slots) * 12;
  (void)hipFuncSetAttribute((const void*)gibbs_lane_recount_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t grid = std::min<uint32_t>(256u, (n_groups + 3) / 4);
  hipLaunchKernelGGL(gibbs_lane_recount_kernel, dim3(grid ? grid : 1u), dim3(1024), lds, s, groups, lanes, recA, samp, n_groups, new_x, new_norm,
                     p_slots, n_slots);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void gibbs_lane_materialize_kernel(const GlGroup* groups, const GlLane* lanes, const uint4* recA, const uint4* samp,
                                                                     uint32_t n_groups, uint32_t* ids, uint32_t* nrm, uint32_t* len) {
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
  for (uint32_t gi = blockIdx.x * nw + wv; gi < n_groups; gi += gridDim.x * nw) {
    const GlGroup g = groups[gi];
    const GlLane L = lanes[(size_t)gi * 64 + lane];
    if (L.block == GL_NONE) continue;
    uint32_t total = 0;
    for (uint32_t k = 0; k < L.path; ++k) {
      const uint4 e = samp[2 * (g.samp_base + (size_t)k * 64 + lane)];
      const uint4 a = samp[2 * (g.samp_base + (size_t)k * 64 + lane) + 1];
      uint32_t at = e.w;
      if (a.x != GL_NONE) {
        ids[L.sample_off + at] = a.x;
        nrm[L.sample_off + at] = a.z;
        ++at;
      }
      if (a.y != GL_NONE) {
        ids[L.sample_off + at] = a.y;
        nrm[L.sample_off + at] = a.w;
        ++at;
      }
      total = at;
    }
    len[L.block] = total;
  }
}
hipError_t launch_gibbs_lane_materialize(const GlGroup* groups, const GlLane* lanes, const uint4* recA, const uint4* samp, uint32_t n_groups,
                                         uint32_t* ids, uint32_t* nrm, uint32_t* len, hipStream_t s) {
  if (!n_groups) return hipSuccess;
  hipLaunchKernelGGL(gibbs_lane_materialize_kernel, dim3(std::min<uint32_t>((n_groups + 3) / 4, 4096u)), dim3(256), 0, s, groups, lanes, recA, samp,
                     n_groups, ids, nrm, len);
  return hipGetLastError();
}

// ---- the layout (host) ----
void gibbs_lane_build(const LatticeSet& L, const std::vector<uint32_t>& bb, const std::vector<GxBlock>& gb, const std::vector<uint64_t>& coff,
                      const std::vector<uint32_t>& cpar, const std::vector<uint32_t>& p_norm, GlHost& out) {
  const size_t nb = bb.size();
  out.taken.assign(nb, 0);
  // per block: eligibility and LDS need (widest level, local parameters and norm groups with a count)
  struct Need {
    uint32_t W = 0, LP = 0, LN = 0;
  };
  std::vector<Need> need(nb);
  std::vector<uint32_t> elig;
  const size_t lds_cap = 640;  // bytes of LDS per lane: 40 KB a wavefront
  std::vector<uint32_t> ps, ns;
  for (size_t b = 0; b < nb; ++b) {
    const BundleDesc& d = L.bundles[bb[b]];
    const GxBlock& B = gb[b];
    if (!(B.n_levels & 0x80000000u) || d.n_levels < 2 || d.n_levels > 128 || d.n_arcs >= (1u << 16) || d.n_states >= (1u << 16)) continue;
    const uint32_t* lo = L.level_off.data() + d.level_base;
    const uint32_t* ooff = L.out_off.data() + d.off_base;
    uint32_t W = 0, degmax = 0;
    for (uint32_t l = 0; l < d.n_levels; ++l) W = std::max(W, lo[l + 1] - lo[l]);
    for (uint32_t st = 0; st < d.n_states; ++st) degmax = std::max(degmax, ooff[st + 1] - ooff[st]);
    // the goal alone on the last level (a pruned lattice: every other state has a way on)
    if (lo[d.n_levels] - lo[d.n_levels - 1] != 1 || lo[d.n_levels - 1] != B.fin || W > 255 || degmax > 255) continue;
    ps.clear();
    ns.clear();
    for (uint64_t a = 0; a < d.n_arcs; ++a) {
      const uint32_t arc = L.out_arcs[d.out_base + a].y;
      for (uint64_t j = coff[arc]; j < coff[arc + 1]; ++j)
        if (p_norm[cpar[j]] != 0xffffffffu) {
          ps.push_back(cpar[j]);
          ns.push_back(p_norm[cpar[j]]);
        }
    }
    std::sort(ps.begin(), ps.end());
    ps.erase(std::unique(ps.begin(), ps.end()), ps.end());
    std::sort(ns.begin(), ns.end());
    ns.erase(std::unique(ns.begin(), ns.end()), ns.end());
    Need n;
    n.W = W;
    n.LP = ((uint32_t)std::max<size_t>(ps.size(), 1) + 3) / 4 * 4;
    n.LN = ((uint32_t)std::max<size_t>(ns.size(), 1) + 3) / 4 * 4;
    if (n.LP >= 0xffffu || (size_t)2 * n.W * 8 + n.LP + n.LN > lds_cap) continue;
    need[b] = n;
    elig.push_back((uint32_t)b);
  }
  if (elig.empty()) return;
  // groups of 64 blocks of similar length (the longest first: a launch drains through its short groups)
  std::stable_sort(elig.begin(), elig.end(), [&](uint32_t x, uint32_t y) { return L.bundles[bb[x]].n_arcs > L.bundles[bb[y]].n_arcs; });
  const size_t ng = (elig.size() + 63) / 64;
  out.groups.assign(ng, GlGroup());
  out.lanes.assign(ng * 64, GlLane());
  for (auto& l : out.lanes) l.block = GL_NONE;
  std::vector<Need> gneed(ng);
  uint64_t rec = 0, samp = 0;
  for (size_t gi = 0; gi < ng; ++gi) {
    GlGroup& G = out.groups[gi];
    uint32_t rows = 0, path = 0;
    for (size_t i = gi * 64; i < std::min(elig.size(), gi * 64 + 64); ++i) {
      const uint32_t b = elig[i];
      const BundleDesc& d = L.bundles[bb[b]];
      rows = std::max(rows, (uint32_t)d.n_arcs);
      path = std::max(path, d.n_levels - 1);
      gneed[gi].W = std::max(gneed[gi].W, need[b].W);
      gneed[gi].LP = std::max(gneed[gi].LP, need[b].LP);
      gneed[gi].LN = std::max(gneed[gi].LN, need[b].LN);
      out.taken[b] = 1;
    }
    G.rows = (rows + GL_R * GL_CH - 1) / (GL_R * GL_CH) * (GL_R * GL_CH);
    G.path = path;
    G.rec_base = rec;
    G.samp_base = samp;
    rec += (uint64_t)(G.rows + GL_SLACK) * 64;
    samp += (uint64_t)(path + 8) * 64;  // (+ the entries the table fill and the recount read ahead)
  }
  out.n_rec = rec;
  out.n_samp = samp;
  out.recA.assign(4 * rec, 0u);
  out.recB.assign(4 * rec, 0u);
  out.arc_id.assign(rec, 0u);
  // ONE launch: the groups in order of length, the LDS of a wavefront sized by the neediest group (at most 40 KB: four wavefronts a
  // CU at worst).  A launch per LDS class ran the classes one after the other on the sampler's stream with a third of the chip's
  // SIMDs busy in each (profiles/r6_v2_crp_kernel_stats.csv: five launches of 177 us, each as long as its longest wavefront); in
  // one launch the short groups fill in as the long ones finish.
  {
    GlClass c;
    c.first = 0;
    c.count = (uint32_t)ng;
    for (size_t gi = 0; gi < ng; ++gi) {
      c.W = std::max(c.W, gneed[gi].W);
      c.LP = std::max(c.LP, gneed[gi].LP);
      c.LN = std::max(c.LN, gneed[gi].LN);
    }
    out.classes.push_back(c);
  }
  // the streams
  std::vector<uint32_t> srow, lp_of, ln_of;
  for (size_t i = 0; i < elig.size(); ++i) {
    const uint32_t b = elig[i];
    const size_t gi = i / 64, lane = i % 64;
    const BundleDesc& d = L.bundles[bb[b]];
    const GxBlock& B = gb[b];
    const GlGroup& G = out.groups[gi];
    const uint32_t* lo = L.level_off.data() + d.level_base;
    const uint32_t* ooff = L.out_off.data() + d.off_base;
    // local numbering of the parameters and norm groups that carry counts
    ps.clear();
    ns.clear();
    for (uint64_t a = 0; a < d.n_arcs; ++a) {
      const uint32_t arc = L.out_arcs[d.out_base + a].y;
      for (uint64_t j = coff[arc]; j < coff[arc + 1]; ++j)
        if (p_norm[cpar[j]] != 0xffffffffu) {
          ps.push_back(cpar[j]);
          ns.push_back(p_norm[cpar[j]]);
        }
    }
    std::sort(ps.begin(), ps.end());
    ps.erase(std::unique(ps.begin(), ps.end()), ps.end());
    std::sort(ns.begin(), ns.end());
    ns.erase(std::unique(ns.begin(), ns.end()), ns.end());
    auto lp = [&](uint32_t p) { return (uint32_t)(std::lower_bound(ps.begin(), ps.end(), p) - ps.begin()); };
    auto ln = [&](uint32_t n) { return (uint32_t)(std::lower_bound(ns.begin(), ns.end(), n) - ns.begin()); };
    srow.assign(d.n_states, 0u);  // per state: first row << 8 | out-degree
    uint32_t row = 0;
    for (uint32_t l = d.n_levels - 1; l-- > 0;) {
      uint32_t last_row_of_level = GL_NONE;
      for (uint32_t st = lo[l]; st < lo[l + 1]; ++st) {
        const uint32_t a0 = ooff[st], a1 = ooff[st + 1];
        srow[st] = (row << 8) | (a1 - a0);
        for (uint32_t a = a1; a-- > a0;) {  // list order: newest first
          const uint2_t oa = L.out_arcs[d.out_base + a];
          const size_t at = G.rec_base + (size_t)row * 64 + lane;
          uint32_t* A4 = &out.recA[4 * at];
          uint32_t* B4 = &out.recB[4 * at];
          uint32_t ctrl = (oa.x - lo[l + 1]) | ((st - lo[l]) << 8) | GL_VALID;
          if (a == a0) ctrl |= GL_STATE_LAST;
          uint32_t par[2] = {GL_NONE, GL_NONE}, nrm[2] = {GL_NONE, GL_NONE}, lpi[2] = {0xffffu, 0xffffu}, lni[2] = {0xffffu, 0xffffu};
          const uint64_t c0 = coff[oa.y], c1 = coff[oa.y + 1];
          for (uint64_t j = c0; j < c1 && j < c0 + 2; ++j) {
            par[j - c0] = cpar[j];
            nrm[j - c0] = p_norm[cpar[j]];
            if (nrm[j - c0] != 0xffffffffu) {
              lpi[j - c0] = lp(cpar[j]);
              lni[j - c0] = ln(nrm[j - c0]);
            }
          }
          if (par[0] != GL_NONE) ctrl |= GL_HAS0 | (nrm[0] == GL_NONE ? GL_FIX0 : 0u);
          if (par[1] != GL_NONE) ctrl |= GL_HAS1 | (nrm[1] == GL_NONE ? GL_FIX1 : 0u);
          A4[0] = par[0];
          A4[1] = par[1];
          A4[2] = nrm[0];
          A4[3] = nrm[1];
          B4[0] = ctrl;
          B4[1] = lpi[0] | (lpi[1] << 16);
          B4[2] = lni[0] | (lni[1] << 16);
          B4[3] = srow[oa.x];  // (a destination lies a level on: its rows are already laid)
          out.arc_id[at] = oa.y;
          last_row_of_level = row;
          ++row;
        }
      }
      if (last_row_of_level != GL_NONE) out.recB[4 * (G.rec_base + (size_t)last_row_of_level * 64 + lane)] |= GL_LEVEL_LAST;
    }
    GlLane& Ln = out.lanes[gi * 64 + lane];
    Ln.block = b;
    Ln.n_arcs = (uint32_t)d.n_arcs;
    Ln.start = srow[B.start];
    Ln.path = d.n_levels - 1;
    Ln.wt = B.wt;
    Ln.sample_off = B.sample_off;
  }
}

}  // namespace carmel_hip
