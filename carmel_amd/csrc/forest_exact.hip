// forest_exact.hip — the reference's own chain of forest-em's Gibbs sampler, forests strictly one after another, as
// one persistent wavefront per sweep with every count on the device.
//
// Replaces, for `forest-em --crp` without --crp-parallel: /root/reference/forest-em/forest-em.hpp:750-766 (resample_block),
// forest.hpp:768-816 (compute_inside with the proposal probabilities), forest.hpp:725-758 (choose_random),
// graehl/shared/gibbs.hpp:835-877 (iteration: per block remove old sample, resample, probabilities, add new sample),
// gibbs.hpp:769-792 + delta_sum.hpp:74-84 (addc), gibbs.hpp:712-742 (cache-model probability).
//
// The chain is sequential by definition: forest f's proposal reads the counts forest f - 1 just changed.  What can run side
// by side is everything INSIDE one forest, so the 64 lanes of one wavefront work on one forest, and what the chain does not
// wait for is taken off its critical path:
//   counts     one round of device atomics per forest: +1 for the sample just drawn, -1 for the NEXT forest's previous
//              sample (prefetched), issued together; the cache-model probability comes from the values those atomics
//              RETURN (a rule used k times in a sample sees {c, c+1, .., c+k-1} in some order: the same product)
//   fold       delta_sum's time-weighted sum is folded for every parameter once per sweep (forest_fold_kernel: at a sweep's
//              start every count is what it was when the previous sweep ended, which is what the first touch would fold)
//   proposal   lane per node: count / norm sum of an AND node's rule (loads that bypass the non-coherent L1)
//   inside     height by height, lane per node (children are of lower height); register path: plain doubles, every node's
//              record and its children's values stay in the lane's registers
//   walk       depth first, uniforms keyed by the order of visits -- the reference's order, so that the samples are the
//              oracle's draw for draw.  Register path: node records are read across lanes (v_readlane), the stack and the
//              sample live in a register indexed by lane, an OR node's choice is computed by its own lane:
//              no memory round trip per step.
// Forests beyond the register path (FX_NODES nodes, FX_KIDS children per node, FX_STACK pending nodes, a root value below
// 1e-150) take the LDS path: tables staged in LDS, mantissa x 2^exponent arithmetic, one lane walking.  The next forest's
// records, descriptor and previous sample are requested one forest ahead and wait in registers.
#include "forest_exact.hpp"
#include "rng.hpp"

namespace carmel_hip {

#define FX_NONORM 0xffffffffu
#define FX_NS (FX_NODES / 64)  // node records (and sample entries) per lane
#define FX_VALS (FX_NODES + 4)

size_t forest_exact_lds_bytes(uint32_t max_n, uint32_t max_tab, uint32_t max_stack, uint32_t max_sample) {
  const size_t n4 = ((size_t)max_n + 3) / 4 * 4, s4 = ((size_t)max_sample + 3) / 4 * 4;
  return n4 * (8 + 8 + 4 + 4 + 4) + s4 * (8 + 4 + 4) + (((size_t)max_tab + 7) / 8 * 8 + (size_t)max_stack + s4 + 8) * 2;
}

// The counts are read and changed by this one wavefront only, so WORKGROUP scope is all the coherence the chain needs: the
// atomics execute in this XCD's L2, a load after them (ordered by fx_order's wait) may be served by L2 or by the CU's own L1,
// which an atomic through it invalidates.  Agent scope would send every load past the L2 (eight XCDs, eight L2s) and make
// every fence an L2 write-back + invalidate: ~1 us per round trip on a chain that makes two per forest.
__device__ __forceinline__ double fx_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void fx_order() {  // earlier atomics are performed before later loads are issued
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
__device__ __forceinline__ double fx_add(double* p, double v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t fx_rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// lane `at` of a register takes the (wavefront-uniform) value v: a compare and a select in every lane (v_writelane_b32 wants
// its lane select in M0 next to an SGPR value, and this compiler has no builtin for it)
__device__ __forceinline__ uint32_t fx_wl(uint32_t v, uint32_t at, uint32_t old) { return threadIdx.x == at ? v : old; }
__device__ __forceinline__ double fx_rl(double v, uint32_t lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (int)lane), __builtin_amdgcn_readlane(__double2loint(v), (int)lane));
}

struct FxStage {  // one forest's records and previous sample, on their way
  uint4 rec[FX_NS];
  uint32_t sr[FX_NS], sn[FX_NS];
  uint32_t plen;
};

__device__ __forceinline__ void fx_request(const uint4* __restrict__ xrec, const uint32_t* s_rules, const uint32_t* s_nn, const uint4 d,
                                           uint32_t lane, FxStage& S) {
  const uint32_t n = d.y & 0xffffu;
  const uint64_t so = ((uint64_t)(d.w & 0xffffu) << 32) | d.z;
#pragma unroll
  for (int q = 0; q < FX_NS; ++q) {
    const bool have = lane + q * 64 < n;
    S.rec[q] = have ? xrec[(size_t)d.x + lane + q * 64] : make_uint4(0xffu, 0xffffffu, 0, FX_NONORM);  // (an OR node without children)
  }
#pragma unroll
  for (int q = 0; q < FX_NS; ++q) {  // (read past the sample: within its capacity or the buffer's padding)
    S.sr[q] = s_rules[so + lane + q * 64];
    S.sn[q] = s_nn[so + lane + q * 64];
  }
}
// the previous sample of the forest about to be resampled leaves the counts (gibbs.hpp:851-852)
__device__ __forceinline__ void fx_take_out(const FExactArgs& A, const uint4 d, uint32_t lane, const FxStage& S, uint32_t plen) {
  const uint64_t so = ((uint64_t)(d.w & 0xffffu) << 32) | d.z;
#pragma unroll
  for (int q = 0; q < FX_NS; ++q)
    if (lane + q * 64 < plen && S.sn[q] != FX_NONORM) {
      fx_add(A.p_x + S.sr[q], -1.0);
      fx_add(A.normsum + S.sn[q], -1.0);
    }
  for (uint32_t i = lane + FX_NS * 64; i < plen; i += 64) {
    const uint32_t nn = A.sample_nn[so + i];
    if (nn == FX_NONORM) continue;
    fx_add(A.p_x + A.sample_rules[so + i], -1.0);
    fx_add(A.normsum + nn, -1.0);
  }
}


// a running product as mantissa x 2^exponent: what the sweep's log-probabilities are made of (one logarithm per lane and
// sweep instead of one per sample entry: a double-precision log is ~100 instructions, and a lone wavefront issues one
// instruction every few cycles)
struct FxProd {
  double m;
  long long e;
  __device__ __forceinline__ void mul(double x) {
    int t;
    m = frexp(m * x, &t);
    e += t;
  }
  __device__ __forceinline__ double ln() const { return log(m) + (double)e * 0.69314718055994530942; }
};

// LDS written by some lanes of a wavefront is read by others of the SAME wavefront: LDS operations of a wavefront complete in
// order, so all it takes is to wait for them and to keep the compiler from moving memory operations across.  (A workgroup
// barrier would also wait for the global loads in flight: the next forest's prefetch.)
#define FX_WAVE_SYNC()                                   \
  do {                                                   \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0), as the builtin: the compiler's own wait counts know of it */ \
    asm volatile("" ::: "memory");                       \
    __builtin_amdgcn_wave_barrier();                     \
  } while (0)

// ... and between the passes of the inside sweep not even the wait: a pass's reads are issued behind the pass before's writes
// and see them (in order); only the compiler has to be kept from reordering
#define FX_LDS_ORDER()                   \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_wave_barrier();     \
  } while (0)

// What a forest's walk reads, per node slot of a lane: the node's value, the running sums of its children's shares in the
// same units (+inf where there is no further child: never passed), its children.
template <int NS>
struct FxWalkTab {
  double val[NS], t0[NS], t1[NS], t2[NS];
  uint32_t kp[NS];  // the children, a byte each, first lowest (one register: an array indexed by the choice would live in scratch memory)
};

// ---- inside (forest.hpp:768-816) for a forest of at most NS * 64 nodes, a node per lane slot.  Every node recomputes its
// value from its children's once per height: after pass h the nodes of height <= h are final (a node of height h has children
// below h) and nobody branches on where it stands; pass h reads buffer (h - 1) & 1 and writes buffer h & 1.  A missing child
// reads a slot that holds the operation's neutral element.
//   EXT = false: plain doubles.  Every value is a sum of products of probabilities <= 1, so what underflows next to a root
//                >= 1e-150 could not have been chosen anyway; a smaller root returns false and the caller comes back with
//   EXT = true:  mantissa x 2^exponent per node (vals_e holds the exponents), exact at any depth.
// A node's record (xrec): x = first child | children << 8 | height << 16 | AND << 31, y = the other children (a byte each,
// first to visit lowest), z = rule, w = norm group; node ids are bytes (< FX_NODES), 0xff = none.
template <int NS, bool EXT>
__device__ __forceinline__ bool fx_inside(const FxStage& S, const double (&p)[FX_NS], const uint32_t n, const uint32_t H,
                                          const uint32_t lane, double* vals, int* vals_e, FxWalkTab<NS>& W) {
  double val[NS], c0[NS], c1[NS], c2[NS], tot[NS];
  int ve[NS];
  uint32_t ka[NS][FX_KIDS];  // LDS slots of the children
  bool is_and[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const uint32_t w0 = S.rec[q].x, w1 = S.rec[q].y, nch = (w0 >> 8) & 0xffu;
    is_and[q] = (w0 & 0x80000000u) != 0;
    const uint32_t none = is_and[q] ? FX_NODES : FX_NODES + 1;  // 1.0 / 0.0
    const uint32_t kid[FX_KIDS] = {w0 & 0xffu, w1 & 0xffu, (w1 >> 8) & 0xffu, (w1 >> 16) & 0xffu};
#pragma unroll
    for (int j = 0; j < FX_KIDS; ++j) {
      ka[q][j] = (uint32_t)j < nch ? kid[j] & (FX_NODES - 1) : none;
    }
    W.kp[q] = (w0 & 0xffu) | (w1 << 8);
    val[q] = is_and[q] ? p[q] : 0.0;  // height 0: an AND leaf is its rule's probability
    ve[q] = 0;
    if (EXT) val[q] = frexp(val[q], &ve[q]);
    const uint32_t at = lane + q * 64 < n ? lane + q * 64 : FX_NODES + 2;
    vals[at] = val[q];
    if (EXT) vals_e[at] = ve[q];
    c0[q] = c1[q] = c2[q] = 0.0;
    tot[q] = val[q];
  }
  FX_LDS_ORDER();
  for (uint32_t h = 1; h < H; ++h) {
    const double* rd = vals + ((h - 1) & 1u) * FX_VALS;
    double* wr = vals + (h & 1u) * FX_VALS;
    const int* rde = vals_e + ((h - 1) & 1u) * FX_VALS;
    int* wre = vals_e + (h & 1u) * FX_VALS;
    double a[NS][FX_KIDS];
    int ae[NS][FX_KIDS];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int j = 0; j < FX_KIDS; ++j) {
        a[q][j] = rd[ka[q][j]];
        if (EXT) ae[q][j] = rde[ka[q][j]];
      }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (!EXT) {
        const double prod = (((p[q] * a[q][0]) * a[q][1]) * a[q][2]) * a[q][3];
        c0[q] = a[q][0];  // running sums of the children's shares: what the walk compares u * value with
        c1[q] = a[q][0] + a[q][1];
        c2[q] = c1[q] + a[q][2];
        tot[q] = c2[q] + a[q][3];
        val[q] = is_and[q] ? prod : tot[q];
      } else {
        // AND: mantissas multiply (five factors in [0.5, 1): no underflow), exponents add; OR: the children's shares in units
        // of the largest exponent among them (a share 2^-1100 below it counts as nothing, as it would in any arithmetic)
        int pe;
        const double pm = frexp(p[q], &pe);
        int t;
        const double prod = frexp((((pm * a[q][0]) * a[q][1]) * a[q][2]) * a[q][3], &t);
        const int prod_e = pe + ae[q][0] + ae[q][1] + ae[q][2] + ae[q][3] + t;
        const int lo = -(1 << 28);
        const int e0 = a[q][0] != 0.0 ? ae[q][0] : lo, e1 = a[q][1] != 0.0 ? ae[q][1] : lo, e2 = a[q][2] != 0.0 ? ae[q][2] : lo,
                  e3 = a[q][3] != 0.0 ? ae[q][3] : lo;
        const int emax = max(max(e0, e1), max(e2, e3));
        const double s0 = ldexp(a[q][0], max(e0 - emax, -1100)), s1 = ldexp(a[q][1], max(e1 - emax, -1100)),
                     s2 = ldexp(a[q][2], max(e2 - emax, -1100)), s3 = ldexp(a[q][3], max(e3 - emax, -1100));
        c0[q] = s0;
        c1[q] = s0 + s1;
        c2[q] = c1[q] + s2;
        tot[q] = c2[q] + s3;
        int ts;
        const double sm = frexp(tot[q], &ts);
        val[q] = is_and[q] ? prod : sm;
        ve[q] = is_and[q] ? prod_e : (tot[q] != 0.0 ? emax + ts : 0);
      }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const uint32_t at = lane + q * 64 < n ? lane + q * 64 : FX_NODES + 2;
      wr[at] = val[q];
      if (EXT) wre[at] = ve[q];
    }
    FX_LDS_ORDER();
  }
  const double root = vals[((H - 1) & 1u) * FX_VALS + n - 1];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const uint32_t nch = (S.rec[q].x >> 8) & 0xffu;
    const double inf = __builtin_huge_val();
    W.val[q] = tot[q];  // (an OR node's: the sum its shares were summed to; AND nodes make no choice)
    W.t0[q] = nch > 1 ? c0[q] : inf;
    W.t1[q] = nch > 2 ? c1[q] : inf;
    W.t2[q] = nch > 3 ? c2[q] : inf;
  }
  return EXT ? root != 0.0 : root >= 1e-150;
}

// ---- the walk, depth first (forest.hpp:725-758): an AND node is recorded and hands on its children (the first is visited
// next, the others wait), an OR node chooses one child with the uniform of its visit -- the reference's order of draws.
// Everything that steers the walk is wavefront-uniform and lives in scalar registers: the node, the counters, and the pending
// nodes -- a byte each, packed into 128 bits, next to visit lowest (0xff at the bottom ends the walk).  A node's record is
// read from its lane (v_readlane); what ITS node would choose with this visit's uniform is worked out by every lane and the
// visited node's answer read back.  Returns the number of recorded nodes; samp[q]: lane i holds the (i + 64 q)-th.
template <int NS>
__device__ __forceinline__ uint32_t fx_walk(const FxStage& S, const FxWalkTab<NS>& W, const double U, const uint64_t seed,
                                            const uint32_t iter, const uint32_t f, const uint32_t n, uint32_t (&samp)[NS]) {
#pragma unroll
  for (int q = 0; q < NS; ++q) samp[q] = 0;
  unsigned long long pend_lo = 0xffull, pend_hi = 0;
  uint32_t step = 0, node = n - 1, ns = 0;
#define FX_SLOT(arr) (NS == 1 ? arr[0] : (q ? arr[NS - 1] : arr[0]))
  for (;;) {
    const uint32_t q = NS == 1 ? 0u : node >> 6;
    const uint32_t w0 = fx_rl(NS == 1 ? S.rec[0].x : (q ? S.rec[NS - 1].x : S.rec[0].x), node);
    const uint32_t nch = (w0 >> 8) & 0xffu;
    const bool pop = nch == 0;
    if (w0 & 0x80000000u) {
      if (NS == 1 || ns < 64)
        samp[0] = fx_wl(node, ns, samp[0]);
      else
        samp[NS - 1] = fx_wl(node, ns - 64, samp[NS - 1]);
      ++ns;
      if (nch > 1) {  // the other children wait: their bytes go under everything pending
        const uint32_t w1 = fx_rl(NS == 1 ? S.rec[0].y : (q ? S.rec[NS - 1].y : S.rec[0].y), node);
        const uint32_t sh = 8u * (nch - 1);  // 8, 16 or 24
        pend_hi = (pend_hi << sh) | (pend_lo >> (64u - sh));
        pend_lo = (pend_lo << sh) | (w1 & ((1u << sh) - 1u));
      }
      node = w0 & 0xffu;
    } else if (nch) {
      const double u = step < 64 ? fx_rl(U, step) : gibbs_uniform(seed, iter, f, step);
      ++step;
      // the reference subtracts the children's shares from u * value one by one and stops below zero (random.ipp:111-127):
      // child j is chosen when u * value has passed the sums of the shares before it
      const double uv = u * FX_SLOT(W.val);
      const bool g1 = !(uv < FX_SLOT(W.t0)), g2 = g1 && !(uv < FX_SLOT(W.t1)), g3 = g2 && !(uv < FX_SLOT(W.t2));
      const uint32_t sel = g3 ? 24u : (g2 ? 16u : (g1 ? 8u : 0u));
      node = fx_rl((FX_SLOT(W.kp) >> sel) & 0xffu, node);
    }
    if (pop) {
      node = (uint32_t)pend_lo & 0xffu;
      if (node == 0xffu) break;
      pend_lo = (pend_lo >> 8) | (pend_hi << 56);
      pend_hi >>= 8;
    }
  }
#undef FX_SLOT
  return ns;
}

// The register path for a forest of at most NS * 64 nodes: inside pass (plain doubles; with exponents when those underflow),
// walk, and the sample's entries -- rule, norm group, proposal probability of the recorded nodes, NS per lane.  Returns false
// only for a forest without any derivation (root value 0).
template <int NS>
__device__ __forceinline__ bool fx_register_path(const FxStage& S, const double (&p)[FX_NS], const double U, const uint64_t seed,
                                                 const uint32_t iter, const uint32_t n, const uint32_t H, const uint32_t f,
                                                 const uint32_t lane, double* vals, int* vals_e, const uint32_t* lr,
                                                 const uint32_t* ln, const double* lp, uint32_t& ns, uint32_t (&e_r)[FX_NS],
                                                 uint32_t (&e_n)[FX_NS], double (&e_p)[FX_NS], const bool stamp, unsigned long long& t_inside) {
  FxWalkTab<NS> W;
  bool ok = fx_inside<NS, false>(S, p, n, H, lane, vals, vals_e, W);
  if (!ok) ok = fx_inside<NS, true>(S, p, n, H, lane, vals, vals_e, W);
  if (stamp) t_inside = __builtin_readcyclecounter();
  uint32_t samp[NS];
  ns = ok ? fx_walk<NS>(S, W, U, seed, iter, f, n, samp) : 0u;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const bool have = lane + q * 64 < ns;
    const uint32_t nd = have ? samp[q] : 0u;
    e_r[q] = lr[nd];
    e_n[q] = have ? ln[nd] : FX_NONORM;
    e_p[q] = have ? lp[nd] : 1.0;
  }
#pragma unroll
  for (int q = NS; q < FX_NS; ++q) {
    e_r[q] = 0;
    e_n[q] = FX_NONORM;
    e_p[q] = 1.0;
  }
  return ok;
}

__global__ __launch_bounds__(64) void forest_exact_kernel(FExactArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fx_lds[];
  if (A.n_chains > 1) {  // chain c of --crp-restarts: its own state and its own uniforms (FExactArgs::n_chains)
    const uint64_t c = blockIdx.x;
    A.sample_len += c * A.ch_forests;
    A.sample_rules += c * A.ch_sample;
    A.sample_nn += c * A.ch_sample;
    A.p_x += c * A.ch_rules;
    A.normsum += c * A.ch_norms;
    A.ccount += c * A.ch_rules;
    A.csum += c * A.ch_norms;
    A.iter_out += c * 2;
    A.iter += (uint32_t)c * A.iter_stride;
    A.phase_clk = nullptr;
  }
  __shared__ double vals[2 * FX_VALS];  // register path: node values, two buffers; each ends in 1.0, 0.0 and a slot nobody reads
  __shared__ int vals_e[2 * FX_VALS];   // ... their exponents, when plain doubles underflow (the neutral slots: 2^1 x 0.5, 0)
  __shared__ double lp[FX_NODES];     // ... proposal probability of a node's rule
  __shared__ uint32_t lr[FX_NODES], ln[FX_NODES];  // ... its rule and norm group
  const uint32_t lane = threadIdx.x;
  // the LDS path's carve
  const uint32_t n4 = (A.max_n + 3) / 4 * 4, s4 = (A.max_sample + 3) / 4 * 4;
  double* vm = (double*)fx_lds;          // node values: mantissa (an AND node's rule probability until the inside pass)
  double* pp = vm + n4;                  // proposal probability of an AND node's rule
  double* ep = pp + n4;                  // sample entries: probability,
  int* ve = (int*)(ep + s4);             // node values: exponent
  uint32_t* hr = (uint32_t*)(ve + n4);   // rule | bit 31 = AND
  uint32_t* hn = hr + n4;                // norm group
  uint32_t* er = hn + n4;                // sample entries: rule,
  uint32_t* en = er + s4;                //   norm group
  unsigned short* tb = (unsigned short*)(en + s4);
  unsigned short* stk_l = tb + (A.max_tab + 7) / 8 * 8;
  unsigned short* snode = stk_l + A.max_stack;
  if (lane < 2) {
    vals[lane * FX_VALS + FX_NODES] = 1.0;
    vals[lane * FX_VALS + FX_NODES + 1] = 0.0;
    vals_e[lane * FX_VALS + FX_NODES] = 0;  // (1.0 = 1.0 x 2^0: the mantissa slot is shared by both arithmetics)
    vals_e[lane * FX_VALS + FX_NODES + 1] = 0;
  }
  const uint32_t nf = A.n_forests;
  FxProd cheap{1.0, 0}, cnum{1.0, 0}, cden{1.0, 0};  // proposal probability; cache-model probability = cnum / cden
  unsigned long long clk[6] = {0, 0, 0, 0, 0, 0};
  // ---- the pipeline's preamble: descriptors of forests 0 and 1, records of forest 0, whose previous sample leaves the counts ----
  uint4 d = A.xdesc[0], dn = A.xdesc[min(1u, nf - 1)];
  FxStage S;
  fx_request(A.xrec, A.sample_rules, A.sample_nn, d, lane, S);
  S.plen = A.sample_len[0];
  fx_take_out(A, d, lane, S, S.plen);
  double U = gibbs_uniform(A.seed, A.iter, 0u, lane);  // the walk's first 64 uniforms, one per lane (forest 0's)
#pragma unroll
  for (int q = 0; q < FX_NS; ++q)
    if (lane + q * 64 < (d.y & 0xffffu)) {
      lr[lane + q * 64] = S.rec[q].z;
      ln[lane + q * 64] = S.rec[q].w;
    }
  // what the previous forest's cache-model atomics returned (consumed a forest later); ret_on: the entry is a counted rule's
  // (else its factor is ret_p: a fixed rule's probability, or 1 where the lane has no entry)
  double ret_c[FX_NS], ret_s[FX_NS], ret_p[FX_NS];
  bool ret_on[FX_NS];
#pragma unroll
  for (int q = 0; q < FX_NS; ++q) {
    ret_c[q] = ret_s[q] = ret_p[q] = 1.0;
    ret_on[q] = false;
  }
  for (uint32_t f = 0; f < nf; ++f) {
    unsigned long long t0 = A.phase_clk ? __builtin_readcyclecounter() : 0;
    fx_order();  // the counts are as the chain has them: the previous forest's sample in, this forest's previous sample out
#pragma unroll
    for (int q = 0; q < FX_NS; ++q) {
      cnum.mul(ret_on[q] ? ret_c[q] : ret_p[q]);
      cden.mul(ret_on[q] ? ret_s[q] : 1.0);
    }
    const uint32_t n = d.y & 0xffffu, H = d.y >> 16;
    const uint64_t so = ((uint64_t)(d.w & 0xffffu) << 32) | d.z;
    bool slow = ((d.w >> 16) & 1u) != 0;
    uint32_t ns = 0;
    uint32_t e_r[FX_NS], e_n[FX_NS];
    double e_p[FX_NS];
    unsigned long long t1 = 0, t2 = 0, t3 = 0;
    // ---- proposal probability of every AND node's rule (gibbs.hpp:153-157); register path: the lane's own nodes ----
    double p[FX_NS];
#pragma unroll
    for (int q = 0; q < FX_NS; ++q) {
      p[q] = 0.0;
      if (!slow && (S.rec[q].x & 0x80000000u)) {
        const uint32_t r = S.rec[q].z, nn = S.rec[q].w;
        p[q] = nn == FX_NONORM ? A.p_prior[r] : fx_ld(A.p_x + r) / fx_ld(A.normsum + nn);
      }
    }
    // the next forest's records and previous sample set out now: they have the phases below to arrive
    FxStage T;
    fx_request(A.xrec, A.sample_rules, A.sample_nn, dn, lane, T);
    T.plen = A.sample_len[min(f + 1, nf - 1)];
    const uint4 dnn = A.xdesc[min(f + 2, nf - 1)];
    if (!slow) {
#pragma unroll
      for (int q = 0; q < FX_NS; ++q)
        if (lane + q * 64 < n) lp[lane + q * 64] = p[q];
      if (A.phase_clk) t1 = __builtin_readcyclecounter();
      FX_WAVE_SYNC();
      const bool stamp = A.phase_clk != nullptr;
      if (n <= 64 && !((d.w >> 17) & 1u))
        fx_register_path<1>(S, p, U, A.seed, A.iter, n, H, f, lane, vals, vals_e, lr, ln, lp, ns, e_r, e_n, e_p, stamp, t2);
      else
        fx_register_path<FX_NS>(S, p, U, A.seed, A.iter, n, H, f, lane, vals, vals_e, lr, ln, lp, ns, e_r, e_n, e_p, stamp, t2);
      if (A.phase_clk) t3 = __builtin_readcyclecounter();
    }
    if (!slow) {
      // ---- the new sample goes into the counts with the next forest's previous sample coming out; the cache-model counts
      // answer with what they held (gibbs.hpp:866-871, 712-742, 769-792) ----
      // (the next forest's records and previous sample, requested before the inside pass, have long arrived; saying so keeps
      // the compiler from waiting for them below, behind the adds.)  The adds whose results the cache model wants are issued by
      // EVERY lane -- one without a counted entry adds zero to a slot of A.idle -- so that nothing here waits for them.
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
#pragma unroll
      for (int q = 0; q < FX_NS; ++q) {
        const bool on = lane + q * 64 < ns, cnt = on && e_n[q] != FX_NONORM;
        if (on) {
          A.sample_rules[so + lane + q * 64] = e_r[q];
          A.sample_nn[so + lane + q * 64] = e_n[q];
        }
        if (cnt) {
          fx_add(A.p_x + e_r[q], 1.0);
          fx_add(A.normsum + e_n[q], 1.0);
        }
        ret_c[q] = fx_add(cnt ? A.ccount + e_r[q] : A.idle + q * 128 + lane, cnt ? 1.0 : 0.0);
        ret_s[q] = fx_add(cnt ? A.csum + e_n[q] : A.idle + q * 128 + 64 + lane, cnt ? 1.0 : 0.0);
        ret_p[q] = e_p[q];  // (1 where the lane has no entry)
        ret_on[q] = cnt;
      }
      if (f + 1 < nf) fx_take_out(A, dn, lane, T, T.plen);
#pragma unroll
      for (int q = 0; q < FX_NS; ++q) cheap.mul(e_p[q]);
    } else {
      // ================= the LDS path =================
      const uint32_t slot = A.lane_of_forest[f];
      const uint4 s0 = A.slots[2 * (size_t)slot], s1 = A.slots[2 * (size_t)slot + 1];
      const uint32_t words = s1.w >> 15, w4 = (words + 7) / 8;
      const uint4* __restrict__ t4s = (const uint4*)(A.tab + (((uint64_t)s0.y << 32) | s0.x));
      const uint4* __restrict__ hs = (const uint4*)(A.hdr + (((uint64_t)s0.w << 32) | s0.z));
      uint4* t4 = (uint4*)tb;
      for (uint32_t k = lane; k < w4; k += 64) t4[k] = t4s[k];
      for (uint32_t k = lane; k < n; k += 64) {
        const uint4 h = hs[k];
        hr[k] = (h.y & 0x7fffffffu) | (h.x & 0x80000000u);
        hn[k] = h.w;
        double pr = 0.0;
        if (h.x & 0x80000000u) pr = h.w == FX_NONORM ? A.p_prior[h.y] : fx_ld(A.p_x + h.y) / fx_ld(A.normsum + h.w);
        pp[k] = pr;
        vm[k] = pr;
      }
      __syncthreads();
      const uint32_t HH = tb[1];
      const unsigned short* lvl = tb + 4;
      const unsigned short* koff = lvl + HH + 1;
      const unsigned short* kids = koff + n + 1;
      for (uint32_t h = 0; h < HH; ++h) {
        for (uint32_t node = lvl[h] + lane; node < lvl[h + 1]; node += 64) {
          const uint32_t k0 = koff[node], k1 = koff[node + 1];
          double m;
          int e;
          if (hr[node] & 0x80000000u) {  // AND: its rule's probability times its children
            m = frexp(vm[node], &e);
            for (uint32_t k = k0; k < k1; ++k) {
              const uint32_t c = kids[k] & 0x7fffu;
              int t;
              m = frexp(m * vm[c], &t);
              e += ve[c] + t;
            }
          } else {  // OR: the sum of its children, aligned to the larger exponent
            m = 0.0;
            e = 0;
            for (uint32_t k = k0; k < k1; ++k) {
              const uint32_t c = kids[k] & 0x7fffu;
              const double cm = vm[c];
              const int ce = ve[c];
              if (cm == 0.0) continue;
              if (m == 0.0) {
                m = cm;
                e = ce;
              } else {
                const int dd = ce - e;
                int t;
                if (dd <= 0)
                  m = frexp(m + ldexp(cm, dd), &t);
                else {
                  m = frexp(ldexp(m, -dd) + cm, &t);
                  e = ce;
                }
                e += t;
              }
            }
          }
          vm[node] = m;
          ve[node] = e;
        }
        __syncthreads();
      }
      if (lane == 0) {  // one lane walks; tables, values and the stack in LDS
        uint32_t sp = 0, cnt = 0, step = 0;
        stk_l[sp++] = (unsigned short)(n - 1);
        while (sp) {
          const uint32_t node = stk_l[--sp] & 0x7fffu;
          const uint32_t k0 = koff[node], k1 = koff[node + 1];
          if (hr[node] & 0x80000000u) {
            if (cnt < A.max_sample) snode[cnt] = (unsigned short)node;
            ++cnt;
            for (uint32_t k = k1; k-- > k0;)
              if (sp < A.max_stack) stk_l[sp++] = kids[k];
          } else if (k1 > k0) {
            const int ne = ve[node];
            double choice = gibbs_uniform(A.seed, A.iter, f, step++) * vm[node];
            uint32_t pick = k0;
            for (uint32_t k = k0; k < k1; ++k) {
              const uint32_t c = kids[k] & 0x7fffu;
              pick = k;
              choice -= ldexp(vm[c], ve[c] - ne);
              if (choice < 0) break;
            }
            if (sp < A.max_stack) stk_l[sp++] = kids[pick];
          }
        }
        snode[s4] = (unsigned short)(cnt < A.max_sample ? cnt : A.max_sample);
      }
      __syncthreads();
      ns = snode[s4];
      // into the counts: the first FX_NS * 64 entries report their cache-model counts through ret_c / ret_s as the register
      // path's do, the others are waited for at once
#pragma unroll
      for (int q = 0; q < FX_NS; ++q) {
        ret_p[q] = 1.0;
        ret_on[q] = false;
      }
      for (uint32_t i = lane; i < ns; i += 64) {
        const uint32_t node = snode[i], r = hr[node] & 0x7fffffffu, nn = hn[node];
        const double pr = pp[node];
        A.sample_rules[so + i] = r;
        A.sample_nn[so + i] = nn;
        cheap.mul(pr);
        double rc = pr, rs = 1.0;
        if (nn != FX_NONORM) {
          fx_add(A.p_x + r, 1.0);
          fx_add(A.normsum + nn, 1.0);
          rc = fx_add(A.ccount + r, 1.0);
          rs = fx_add(A.csum + nn, 1.0);
        }
        if (i < FX_NS * 64) {
#pragma unroll
          for (int q = 0; q < FX_NS; ++q)
            if ((i >> 6) == (uint32_t)q) {
              ret_c[q] = rc;
              ret_s[q] = rs;
              ret_on[q] = true;
            }
        } else {
          cnum.mul(rc);
          cden.mul(rs);
        }
      }
      if (f + 1 < nf) fx_take_out(A, dn, lane, T, T.plen);
      __syncthreads();
    }
    if (lane == 0) A.sample_len[f] = ns;
    S = T;
    d = dn;
    dn = dnn;
    // what the next forest needs that does not depend on the counts, while the atomics above are on their way
    U = gibbs_uniform(A.seed, A.iter, f + 1, lane);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FX_NS; ++q)
      if (lane + q * 64 < (d.y & 0xffffu)) {
        lr[lane + q * 64] = S.rec[q].z;
        ln[lane + q * 64] = S.rec[q].w;
      }
    if (A.phase_clk) {
      const unsigned long long t4 = __builtin_readcyclecounter();
      if (!slow) {
        clk[0] += t1 - t0;
        clk[1] += t2 - t1;
        clk[2] += t3 - t2;
        clk[3] += t4 - t3;
        clk[4] += 1;
      } else
        clk[5] += 1;
    }
  }
  fx_order();
#pragma unroll
  for (int q = 0; q < FX_NS; ++q) {
    cnum.mul(ret_on[q] ? ret_c[q] : ret_p[q]);
    cden.mul(ret_on[q] ? ret_s[q] : 1.0);
  }
  double cheap_ln = cheap.ln(), cache_ln = cnum.ln() - cden.ln();
  for (int o = 32; o > 0; o >>= 1) {
    cheap_ln += __shfl_down(cheap_ln, o, 64);
    cache_ln += __shfl_down(cache_ln, o, 64);
  }
  if (lane == 0) {
    A.iter_out[0] = cache_ln;
    A.iter_out[1] = cheap_ln;
    if (A.phase_clk)
      for (int k = 0; k < 6; ++k) A.phase_clk[k] += clk[k];
  }
}

__global__ void forest_fold_kernel(double* p_s, double* p_tmax, const double* p_x, double time, uint64_t n) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double tm = p_tmax[r];
  if (time > tm) {
    p_s[r] += (time - tm) * p_x[r];
    p_tmax[r] = time;
  }
}

hipError_t launch_forest_exact(const FExactArgs& A, hipStream_t s) {
  const size_t lds = forest_exact_lds_bytes(A.max_n, A.max_tab, A.max_stack, A.max_sample);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)forest_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(forest_exact_kernel, dim3(A.n_chains > 1 ? A.n_chains : 1u), dim3(64), lds, s, A);
  return hipGetLastError();
}
hipError_t launch_forest_fold(double* p_s, double* p_tmax, const double* p_x, double time, uint64_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(forest_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p_s, p_tmax, p_x, time, n);
  return hipGetLastError();
}

}  // namespace carmel_hip
