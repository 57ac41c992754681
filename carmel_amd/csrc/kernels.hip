// kernels.hip — CDNA4 (gfx950) kernels for carmel's EM hot path.  64-wide wavefronts throughout.
//
// One E-step (forward_backward::estimate, /root/reference/carmel/src/train.cc:763-773) is five stream kernels:
//   trans_w_bucket -> trans_w_tile   the arc weights, permuted from WFST-arc order into lattice (slot) order through
//                                    the blocked transposition tables of lattice.hpp (TransBucket): two coalesced
//                                    passes through LDS instead of one random gather per lattice arc;
//   sweep_lane                       one derivation lattice per LANE (64 lattices per wavefront, record streams
//                                    interleaved row by row so every wave-wide load is one coalesced row): forward
//                                    over the in-arc stream with alpha in the lane's own LDS column, backward in
//                                    place over the out-arc stream, posterior exp(alpha + w + beta - ln p) written
//                                    to the arc's slot (derivations.h:400-449, graph.h:391-402).  Windowed groups
//                                    keep a ring of W rows instead of the whole column (lattices up to 1023 states);
//   sweep_wave                       one lattice per WAVEFRONT (lattices too large for a lane): the 64 lanes take the
//                                    arcs of a level, a state's log-sum-exp through LDS atomics (max, then scaled sum);
//   sweep_bundle / sweep_serial      the level-synchronous workgroup-per-bundle gather sweep (kept for lattices beyond
//                                    the wave kernel's LDS) and the reference-order sweep of cyclic lattices;
//   trans_c_tile -> trans_c_bucket   the posteriors back into arc order, each arc's uses summed in a fixed order.
// or, for a corpus made of small plain lane lattices (LatticeSet::tile_sweep, lattice.hpp), three:
//   trans_w_bucket -> tile_sweep -> trans_c_bucket
//                                    tile_sweep_kernel (tile_sweep.hip) does trans_w_tile's, sweep_lane's and
//                                    trans_c_tile's work on a tile of 8192 lattice positions out of one workgroup's
//                                    LDS: no wcache, no post.
// No kernel on this path issues an atomic: counts are bit-reproducible run to run (the one exception: the partial
// sums of a hub arc split over several buckets meet in one atomic add per piece).
//
// Arithmetic is the log-semiring of graehl/shared/weight.h:737-801 in f64.  The reference adds terms one at a time
// with log1p(exp(-|d|)) and drops addends more than 36 nats smaller; here each state's sum is one streaming logsumexp
// (running max + scaled sum), which differs from that only in the last bits (e^-36 ~ 2e-16).  Counts are LINEAR f64.
//
// The M-step (forward_backward::maximize, train.cc:893-923 -> WFST::normalize, fst.cc:86-244) is mstep_window_kernel
// (one pass, norm groups within a window of consecutive parameters) or the general mstep_group_sum / _big_group /
// _normalize / _tie_* kernels; chain_update / chain_scatter are cascade_parameters::update / distribute_counts
// (cascade.h:286-325, 466-479).
#include "kernels.hpp"
#include "sweep_math.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace carmel_hip {


// the reference's own pairwise add (weight.h:765-801) — used by the serial (cyclic-lattice) sweep so that the
// order-dependent result there is the reference's
__device__ __forceinline__ double lw_add(double a, double b) {
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  double d = a - b;
  if (d > 36.0) return a;
  if (d < -36.0) return b;
  if (d < 0) return b + log1p(exp(d));
  return a + log1p(exp(-d));
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  // hardware global_atomic_add_f64 (no CAS loop)
  unsafeAtomicAdd(p, v);
}


// ---------------- lane sweep: one small lattice per lane, 64 per wavefront ----------------
// Streams are interleaved (record k of lane l at base + k*64 + l): every wave-wide load is one coalesced row.
// The lane's forward values live in its own LDS column col[s*64] (conflict-free for any per-lane s); the backward
// pass overwrites alpha[s] with beta[s] in place — when state s is processed in reverse topological order its
// alpha is read once, and every destination it needs already holds beta.  No barriers, no offsets, no levels.
//
// The wave is latency-bound, not bandwidth-bound, unless its loads run far ahead of their first use, so both passes
// are software pipelines over chunks of U records with all loads value-independent (only the LDS column carries the
// recurrence):
//   forward : records are loaded R chunks ahead; the weight gather logw[arc] for a chunk is issued W chunks ahead
//             of its use, from records that landed R-W iterations earlier.  Each weight is also stored at the arc's
//             position in the BACKWARD stream (wcache), so
//   backward: records (4-byte: destination + flags) and weights are two plain sequential streams, both R ahead.
// Posteriors go to post[] at the backward record's own position (coalesced rows).
// PRE: the weights were already laid out in lattice order (wcache, by the blocked transposition): the forward pass
// reads them there (rows, near-coalesced) instead of gathering logw[arc] and stores nothing.
#define LANE_W(r) (PRE ? wcache[(size_t)(((r).x >> LANE_POS_SHIFT) & LANE_POS_MAX) * 64] : logw[(r).y])
// WIN: windowed groups (LaneGroup::window): state s lives at LDS row s mod window; the forward values are also parked in
// the group's global column (spill) and the backward pass gathers alpha[source] per record from there, two pipeline stages
// like the forward pass's weight gather (the backward record carries its arc's source state), instead of reading it from LDS.
// XC (fused-lane layout, LatticeSet::lane_fused: every group starts on a tile of LANE_FUSED_TILE positions = LANE_FUSED_ROWS rows):
// the backward pass leaves the log posterior of a row in a 16 KB LDS stage instead of writing exp(..) to post[], and after
// every LANE_FUSED_ROWS rows the wavefront sends the tile's items to XC in tile-major item order -- xc[i0 + i] =
// exp(stage[t_pos[i0 + i]]) -- which is what trans_c_tile_kernel would read post[] for: `post` is never written, the tile pass
// of the counts direction never runs, and the count pass finds the same bits in XC.
template <int R, int W, bool PRE, typename LSE = Lse, bool WIN = false, bool XC = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void sweep_lane_kernel(LaneArgs A) {
  constexpr int U = (int)LANE_CHUNK;  // a group's row count is a multiple of U (host padding): chunks are never partial
  static_assert(W >= 1 && W < R, "gather lead must be shorter than the record lead");
  static_assert(!XC || ((int)LANE_FUSED_ROWS % (R * U) == 0), "a tile of the fused layout is whole rounds of the register ring");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const LaneGroup g = A.groups[A.first_group + blockIdx.x];
  const int lane = threadIdx.x;
  const bool active = (uint32_t)lane < g.n_lanes;
  const uint32_t S = active ? A.lane_nstates[g.pair_base + lane] : 0u;
  const double* __restrict__ logw = A.logw;
  double* col = lds + lane;
  const uint32_t wm = WIN ? g.window - 1u : 0xffffffffu;
  double* spill = WIN ? A.spill + (size_t)g.spill_row * 64 + lane : nullptr;
  const uint32_t maxlen = g.maxlen;
  const uint32_t lastk = maxlen - 1;  // prefetches past the end re-read the last row: every load is unconditional
  double* wcache = A.wcache + g.stream_base + lane;
  unsigned long long t_start = 0, t_mid = 0;
  if (A.trace) t_start = __builtin_readcyclecounter();
  // Register rings indexed only by compile-time constants (the chunk loop is unrolled R times): a slot is refilled
  // right after it is consumed and never moved, so no load has to land before its first real use.  All global
  // loads and stores in the loops are unconditional (padding records are harmless: arc 0, backward row 0 of a lane
  // that has padding is itself padding), which keeps the s_waitcnt bookkeeping exact.
  // ---------- forward ----------
  {
    const uint2* __restrict__ f = A.fwd + g.stream_base + lane;            // gather path: {flags, arc id}
    const uint32_t* __restrict__ fx = A.fwdx + g.stream_base + lane;       // PRE: the flags word alone
#define LANE_FREC(k) (PRE ? make_uint2(fx[(size_t)(k) * 64], 0u) : f[(size_t)(k) * 64])
    if (active) col[0] = 0.0;
    if (WIN && active) spill[0] = 0.0;
    uint2 rq[R][U];   // slot j: records of chunk c with c % R == j
    double wq[R][U];  // slot j: their weights
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t k = (uint32_t)(j * U + u);
        rq[j][u] = LANE_FREC(k < maxlen ? k : lastk);
      }
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) wq[j][u] = LANE_W(rq[j][u]);
    LSE acc;
    acc.init();
    uint32_t d = 1;
    double prev = 0.0;  // alpha[d-1] kept in a register: a chain step never waits for the LDS round trip
    // one step = consume chunk kb/U from ring slot j (compile-time), issue the gather W chunks ahead, refill the slot
#define LANE_FWD_STEP(j, kb)                                                                          \
  {                                                                                                   \
    _Pragma("unroll") for (int u = 0; u < U; ++u) wq[((j) + W) % R][u] = LANE_W(rq[((j) + W) % R][u]); \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                   \
      const uint32_t x = rq[j][u].x;                                                                  \
      const uint32_t src = x & LANE_STATE_MASK;                                                       \
      const double w = wq[j][u];                                                                      \
      if (!PRE) wcache[(size_t)((x >> LANE_POS_SHIFT) & LANE_POS_MAX) * 64] = w;                      \
      const double a_src = (src + 1 == d) ? prev : col[(src & wm) * 64];                              \
      acc.add((x & LANE_VALID) ? a_src + w : NEG_INF);                                                \
      if (x & LANE_LAST) {                                                                            \
        prev = acc.value();                                                                           \
        col[(d & wm) * 64] = prev;                                                                    \
        if (WIN) spill[(size_t)d * 64] = prev;                                                        \
        ++d;                                                                                          \
        acc.init();                                                                                   \
      }                                                                                               \
    }                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                   \
      const uint32_t k = (kb) + (uint32_t)(R * U + u);                                                \
      rq[j][u] = LANE_FREC(k < maxlen ? k : lastk);                                                   \
    }                                                                                                 \
  }
    // steady state: whole rounds of R chunks, no control flow between the steps (exact s_waitcnt counts) ...
    uint32_t k0 = 0;
    for (; k0 + R * U <= maxlen; k0 += R * U) {
#pragma unroll
      for (int j = 0; j < R; ++j) LANE_FWD_STEP(j, k0 + (uint32_t)(j * U))
    }
    // ... then the last partial round (ring slots continue from 0)
#pragma unroll
    for (int j = 0; j < R - 1; ++j)
      if (k0 + (uint32_t)(j * U) < maxlen) LANE_FWD_STEP(j, k0 + (uint32_t)(j * U))
#undef LANE_FWD_STEP
#undef LANE_FREC
  }
  if (A.trace) t_mid = __builtin_readcyclecounter();
  // ---------- ln p(pair), corpus scalars, beta at the goal ----------
  double next = NEG_INF;  // beta[s+1] in a register
  if (active) {
    const double lp = col[((S - 1) & wm) * 64];
    const double lwt = A.lane_logw[g.pair_base + lane];
    A.pair_logprob[A.lane_pair[g.pair_base + lane]] = lp;  // the corpus scalars are reduced from these afterwards
    next = (lp == NEG_INF) ? NEG_INF : lwt - lp;  // folds "* weight / prob" (derivations.h:445)
    col[((S - 1) & wm) * 64] = next;
  }
  // ---------- backward + posteriors ----------
  {
    if (!PRE || WIN) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's wcache / spill stores before its re-reads
    const uint32_t* __restrict__ b = A.bwd + g.stream_base + lane;
    double* __restrict__ post = XC ? nullptr : A.post + g.stream_base + lane;
    double* const stage_tile = lds + (size_t)A.lds_rows * 64;  // XC: LANE_FUSED_TILE doubles behind the value rows
    double* const stage = stage_tile + lane;
    uint32_t xq[R][U];
    double wq[R][U];
    double aq[WIN ? R : 1][U];  // WIN: alpha[source] of the records in xq
#define LANE_ASRC(x) spill[(size_t)(((x) >> LANE_POS_SHIFT) & LANE_STATE_MASK) * 64]
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t k = (uint32_t)(j * U + u);
        const size_t kk = (size_t)(k < maxlen ? k : lastk) * 64;
        xq[j][u] = b[kk];
        wq[j][u] = wcache[kk];
      }
    if (WIN) {
#pragma unroll
      for (int j = 0; j < W; ++j)
#pragma unroll
        for (int u = 0; u < U; ++u) aq[WIN ? j : 0][u] = LANE_ASRC(xq[j][u]);
    }
    LSE acc;
    acc.init();
    uint32_t s = S >= 2 ? S - 2 : 0u;
    double al = (!WIN && S >= 2) ? col[s * 64] : NEG_INF;
    // phase 1 of a step is the recurrence (cheap, serial): beta chain + the exponent of every posterior; phase 2 is
    // branch-free: the U exponentials are independent and overlap in the pipeline.  Padding rows get exp(-inf) = 0
    // (never read: count_reduce only visits valid slots).
#define LANE_BWD_STEP(j, kb)                                                                          \
  {                                                                                                   \
    double arg[U];                                                                                    \
    if (WIN) {                                                                                        \
      _Pragma("unroll") for (int u = 0; u < U; ++u)                                                   \
        aq[WIN ? ((j) + W) % R : 0][u] = LANE_ASRC(xq[((j) + W) % R][u]);                             \
    }                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                   \
      const uint32_t x = xq[j][u];                                                                    \
      const uint32_t dst = x & LANE_STATE_MASK;                                                       \
      const double b_dst = (dst == s + 1) ? next : col[(dst & wm) * 64];                              \
      const double t = (x & LANE_VALID) ? wq[j][u] + b_dst : NEG_INF;                                 \
      acc.add(t);                                                                                     \
      arg[u] = (WIN ? aq[WIN ? (j) : 0][u] : al) + t;                                                 \
      if (x & LANE_LAST) {                                                                            \
        next = acc.value();                                                                           \
        col[(s & wm) * 64] = next; /* beta[s] replaces alpha[s] */                                    \
        acc.init();                                                                                   \
        if (s > 0) {                                                                                  \
          --s;                                                                                        \
          if (!WIN) al = col[s * 64];                                                                 \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
    if (XC) { /* (the exponentials stay here: they fill the gaps of the recurrence) */                \
      _Pragma("unroll") for (int u = 0; u < U; ++u) stage[((((kb) + u)) & (LANE_FUSED_ROWS - 1u)) * 64] = K_EXP(arg[u]); \
    } else {                                                                                          \
      _Pragma("unroll") for (int u = 0; u < U; ++u) post[(size_t)((kb) + u) * 64] = K_EXP(arg[u]);    \
    }                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                   \
      const uint32_t k = (kb) + (uint32_t)(R * U + u);                                                \
      const size_t kk = (size_t)(k < maxlen ? k : lastk) * 64;                                        \
      xq[j][u] = b[kk];                                                                               \
      wq[j][u] = wcache[kk];                                                                          \
    }                                                                                                 \
  }
    if (XC) {
      // tile by tile: the tile's position table is requested before its rows are swept (older than every load the steps
      // wait for), its items leave after them
      const uint32_t tile0 = (uint32_t)(g.stream_base / LANE_FUSED_TILE);
      for (uint32_t kt = 0; kt < maxlen; kt += LANE_FUSED_ROWS) {
        const uint32_t tile = tile0 + kt / LANE_FUSED_ROWS;
        const uint64_t i0 = A.xc_tile_base[tile];
        const uint32_t ni = (uint32_t)(A.xc_tile_base[tile + 1] - i0);
        const uint16_t* __restrict__ tp = A.xc_t_pos + i0 + lane;  // (unclamped: DEVBUF_SLACK lies behind the table)
        uint32_t pos2[LANE_FUSED_ROWS / 2];  // two positions a register
#pragma unroll
        for (int u = 0; u < (int)LANE_FUSED_ROWS; u += 2) pos2[u / 2] = (uint32_t)tp[u * 64] | ((uint32_t)tp[(u + 1) * 64] << 16);
        {
          const uint32_t lim = min(kt + LANE_FUSED_ROWS, maxlen);
          uint32_t k0 = kt;
#pragma unroll 1
          for (; k0 + R * U <= lim; k0 += R * U) {  // (a round of the ring at a time, as without XC)
#pragma unroll
            for (int j = 0; j < R; ++j) LANE_BWD_STEP(j, k0 + (uint32_t)(j * U))
          }
#pragma unroll
          for (int j = 0; j < R - 1; ++j)
            if (k0 + (uint32_t)(j * U) < lim) LANE_BWD_STEP(j, k0 + (uint32_t)(j * U))
        }
        double* __restrict__ out = A.xc + i0 + lane;
#pragma unroll
        for (int u0 = 0; u0 < (int)LANE_FUSED_ROWS; u0 += 4) {  // (four at a time: registers)
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = stage_tile[((u0 + u) & 1) ? pos2[(u0 + u) / 2] >> 16 : pos2[(u0 + u) / 2] & 0xffffu];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if ((uint32_t)((u0 + u) * 64 + lane) < ni) out[(u0 + u) * 64] = v[u];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    uint32_t k0 = 0;
    for (; k0 + R * U <= maxlen; k0 += R * U) {
#pragma unroll
      for (int j = 0; j < R; ++j) LANE_BWD_STEP(j, k0 + (uint32_t)(j * U))
    }
#pragma unroll
    for (int j = 0; j < R - 1; ++j)
      if (k0 + (uint32_t)(j * U) < maxlen) LANE_BWD_STEP(j, k0 + (uint32_t)(j * U))
    }
#undef LANE_BWD_STEP
#undef LANE_ASRC
#undef LANE_W
  }
  if (lane == 0) {
    if (A.trace) {
      unsigned long long* o = A.trace + (size_t)(A.first_group + blockIdx.x) * 16;
      o[0] = t_start;
      o[1] = t_mid;
      o[2] = __builtin_readcyclecounter();
      o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) | maxlen;  // HW_ID
    }
  }
}

// ---------------- wave sweep: one lattice per wavefront, the lanes over the arcs of a level ----------------
// (WaveDesc, lattice.hpp.)  All forward / backward values of the lattice live in LDS.  A level's log-sums are formed by
// the 64 lanes together: every arc's term t = value[other end] + weight goes into its state's running maximum with an LDS
// atomic max, then exp(t - max) into its state's sum with an LDS atomic add, and one lane per state finishes
// value = max + log(sum) -- the streaming logsumexp of the lane kernel, term order aside.  The backward pass overwrites
// alpha[s] with beta[s] in place once level(s) is done (every destination lies in a later level and already holds beta)
// and writes each arc's posterior exp(alpha[src] + w + beta'[dst]) at the arc's own row position: coalesced, like its
// weight reads.  The forward pass reads its weights at the arcs' backward positions: a gather, but inside the lattice's own
// stretch of wcache and -- a level's in-arcs being the previous levels' out-arcs -- mostly inside a few rows of it.
// The records and weights of the next level's first row are requested before the current level is finished (they depend
// on nothing the sweep computes).
__device__ __forceinline__ void lds_max_f64(double* p, double v) {
  __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add_f64(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// RING (WaveDesc::ring): the values live in a ring of `ring` LDS slots (state s at s mod ring; no arc spans more states than
// that), the forward values are parked in the lattice's stretch of A.spill and come back one level at a time (al[]) for the
// posteriors.  Records and weights are requested TWO levels ahead of their use.
// GW (WaveArgs::bwd_arc): the weights come straight from the WFST's table -- a forward record's y is the arc id, the backward
// records' arc ids lie beside them (bwd_arc, requested one level before the weight they lead to) -- and no pass lays them out in
// lattice order first: for a table the chip's caches hold, the gathers cost less than writing and re-reading 8 B per lattice arc.
// XD (WaveArgs::xc_idx): an arc's posterior goes straight to its item's place in XC (xc_idx: the tile-major item index of every
// backward position, requested with the record) -- what trans_c_tile would read `post` for; a level's arcs being neighbours in
// the WFST, a row's items mostly fill whole lines of XC.
// (five wavefronts a SIMD: 96 registers -- `long`'s 5000 lattices are resident all at once at five, not at four)
template <bool RING, bool GW, bool XD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) void sweep_wave_kernel(WaveArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const WaveDesc d = A.descs[A.first + blockIdx.x];
  const uint32_t lane = threadIdx.x;
  double* val = lds;
  double* mx = lds + A.max_states;  // (RING: max_states = the ring)
  double* sm = mx + A.max_width;
  double* al = sm + A.max_width;    // RING only: alpha of the states of the level the backward pass is at
  const uint32_t rm = RING ? d.ring - 1u : 0xffffffffu;
  double* __restrict__ spill = RING ? A.spill + d.spill_base : nullptr;
  const uint32_t* __restrict__ lvl = A.level_off + d.level_base;
  const uint32_t* __restrict__ frow = A.frow + d.level_base;
  const uint32_t* __restrict__ brow = A.brow + d.level_base;
  const uint2* __restrict__ f = A.fwd + d.fwd_base + lane;
  const uint32_t* __restrict__ b = A.bwd + d.bwd_base + lane;
  const double* __restrict__ wc = GW ? A.logw : A.wcache + d.bwd_base;
  const uint32_t* __restrict__ ba = GW ? A.bwd_arc + d.bwd_base + lane : nullptr;
  const uint32_t amax = A.n_arcs - 1u;  // (a padding record's arc id is 0xffffffff)
  double* __restrict__ post = XD ? A.xc : A.post + d.bwd_base + lane;
  const uint32_t* __restrict__ xi = XD ? A.xc_idx + d.bwd_base + lane : nullptr;
  const uint32_t S = d.n_states, NL = d.n_levels;
  if (RING) {
    for (uint32_t s = lane; s <= rm; s += 64) val[s] = NEG_INF;
  } else {
    for (uint32_t s = lane; s < S; s += 64) val[s] = NEG_INF;
  }
  for (uint32_t i = lane; i < d.max_width; i += 64) {
    mx[i] = NEG_INF;
    sm[i] = 0.0;
  }
  __syncthreads();
  if (lane == 0) {
    val[0] = 0.0;
    if (RING) spill[0] = 0.0;
  }
  __syncthreads();
  // ---------- forward ----------
  {
    const uint32_t rend = frow[NL];
    // the first rows of this level (r0) and of the next three (r1 .. r3): records three levels ahead, their weights two --
    // a record's weight is a gather through the record (wc[rec.y]), and issued in the same step as the record's own load it
    // made every step wait out a full memory round trip; now the gather goes through the record requested a step earlier
#define WAVE_FROW(k) ((k) <= NL ? frow[(k)] : rend)
    // The records and weights in flight live in rings indexed by compile-time constants -- the level loop is unrolled four times
    // and slot (l - 1) % 4 belongs to level l -- and are never moved: rotated through named registers (rec0 = rec1, w0 = w1 ...)
    // every step had to wait for ALL its outstanding loads, the ones just issued for three levels ahead included, before it could
    // move their destinations (s_waitcnt vmcnt(0) a level: a memory round trip per level step).  The row starts are scalars and rotate.
    uint32_t r0 = frow[1], r1 = WAVE_FROW(2), r2 = WAVE_FROW(3), r3 = WAVE_FROW(4);
    // records three levels ahead, weights two: a weight's gather goes through the record requested a step earlier and has two
    // steps to arrive (measured the other way round -- the record two steps, the gather one: the sweep 1.33 -> 1.53 ms on `long`;
    // and five ahead / three with the loops unrolled six times: 1.59, 9700 instructions against the instruction cache)
    uint2 rc[4];
    double wv[4];
    rc[0] = f[(size_t)r0 * 64];
    rc[1] = f[(size_t)(r1 < rend ? r1 : r0) * 64];
    rc[2] = f[(size_t)(r2 < rend ? r2 : r0) * 64];
    wv[0] = wc[rc[0].y];
    wv[1] = wc[rc[1].y];
    uint32_t l = 1;
#define WAVE_FWD_STEP(SL)                                                                                     \
  {                                                                                                           \
    const uint32_t s0 = lvl[l], ns = lvl[l + 1] - s0;                                                         \
    const uint32_t r4 = WAVE_FROW(l + 4);                                                                     \
    rc[((SL) + 3) % 4] = f[(size_t)(r3 < rend ? r3 : r0) * 64];                                               \
    wv[((SL) + 2) % 4] = wc[rc[((SL) + 2) % 4].y];                                                            \
    const uint32_t x0 = rc[(SL)].x;                                                                           \
    const bool v0 = (x0 & WAVE_VALID) != 0;                                                                   \
    const uint32_t dr0 = (x0 >> 16) & 0x3fffu;                                                                \
    const double t0 = v0 ? val[(x0 & 0xffffu) & rm] + wv[(SL)] : NEG_INF;                                     \
    if (t0 > NEG_INF) lds_max_f64(&mx[dr0], t0);                                                              \
    for (uint32_t r = r0 + 1; r < r1; ++r) {                                                                  \
      const uint2 q = f[(size_t)r * 64];                                                                      \
      const double t = (q.x & WAVE_VALID) ? val[(q.x & 0xffffu) & rm] + wc[q.y] : NEG_INF;                    \
      if (t > NEG_INF) lds_max_f64(&mx[(q.x >> 16) & 0x3fffu], t);                                            \
    }                                                                                                         \
    __syncthreads();                                                                                          \
    if (t0 > NEG_INF) lds_add_f64(&sm[dr0], K_EXP(t0 - mx[dr0]));                                             \
    for (uint32_t r = r0 + 1; r < r1; ++r) {                                                                  \
      const uint2 q = f[(size_t)r * 64];                                                                      \
      const double t = (q.x & WAVE_VALID) ? val[(q.x & 0xffffu) & rm] + wc[q.y] : NEG_INF;                    \
      const uint32_t dr = (q.x >> 16) & 0x3fffu;                                                              \
      if (t > NEG_INF) lds_add_f64(&sm[dr], K_EXP(t - mx[dr]));                                               \
    }                                                                                                         \
    __syncthreads();                                                                                          \
    /* (RING: a level is at most a wavefront wide -- one pass, and written as one: in front of a LOOP without loads the   \
       compiler waits for every load in flight, the records and weights requested for the levels ahead included) */      \
    if (RING) {                                                                                               \
      if (lane < ns) {                                                                                        \
        const double m = mx[lane], a = sm[lane];                                                              \
        const double v = (m == NEG_INF) ? NEG_INF : (a == 1.0 ? m : m + K_LOG(a));                            \
        val[(s0 + lane) & rm] = v;                                                                            \
        spill[s0 + lane] = v;                                                                                 \
        mx[lane] = NEG_INF;                                                                                   \
        sm[lane] = 0.0;                                                                                       \
      }                                                                                                       \
    } else                                                                                                    \
    for (uint32_t i = lane; i < ns; i += 64) {                                                                \
      const double m = mx[i], a = sm[i];                                                                      \
      const double v = (m == NEG_INF) ? NEG_INF : (a == 1.0 ? m : m + K_LOG(a));                              \
      val[(s0 + i) & rm] = v;                                                                                 \
      mx[i] = NEG_INF;                                                                                        \
      sm[i] = 0.0;                                                                                            \
    }                                                                                                         \
    __syncthreads();                                                                                          \
    r0 = r1;                                                                                                  \
    r1 = r2;                                                                                                  \
    r2 = r3;                                                                                                  \
    r3 = r4;                                                                                                  \
    ++l;                                                                                                      \
  }
    while (l + 3 < NL) {
      WAVE_FWD_STEP(0)
      WAVE_FWD_STEP(1)
      WAVE_FWD_STEP(2)
      WAVE_FWD_STEP(3)
    }
    if (l < NL) WAVE_FWD_STEP(0)
    if (l < NL) WAVE_FWD_STEP(1)
    if (l < NL) WAVE_FWD_STEP(2)
#undef WAVE_FWD_STEP
#undef WAVE_FROW
  }
  // ---------- ln p(pair); beta'[goal] = ln(weight) - ln p folds "* weight / prob" (derivations.h:445) ----------
  if (lane == 0) {
    const double lp = val[(S - 1) & rm];
    A.pair_logprob[d.pair] = lp;
    val[(S - 1) & rm] = (lp == NEG_INF) ? NEG_INF : d.logw - lp;
  }
  if (RING) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's spill stores before its re-reads
  __syncthreads();
  // ---------- backward + posteriors ----------
  {
    const uint32_t rend = brow[NL];
    // (as in the forward pass: the step loop unrolled three times, slot (k - 1) % 3 belongs to step k; the arc ids three steps ahead,
    // the weights two)
#define WAVE_BROW(j) ((j) <= NL ? brow[(j)] : rend)
    uint32_t r0 = brow[1], r1 = brow[2], r2 = WAVE_BROW(3);
    const size_t p1 = (size_t)(r1 < rend ? r1 : r0) * 64;
    uint32_t xs[3], qs[3], gs[3];
    double ws[3], avs[3];
    xs[0] = b[(size_t)r0 * 64];
    xs[1] = b[p1];
    qs[0] = XD ? xi[(size_t)r0 * 64] : 0u;
    qs[1] = XD ? xi[p1] : 0u;
    qs[2] = 0u;
    xs[2] = 0u;
#define WAVE_POST(r, q, v)                 \
  {                                        \
    if (!XD)                               \
      post[(size_t)(r) * 64] = (v);        \
    else if ((q) != 0xffffffffu)           \
      post[(q)] = (v);                     \
  }
#define WAVE_BW(p, g) (GW ? wc[min((g), amax)] : wc[(p) + lane])
    gs[0] = gs[1] = 0u;
    gs[2] = GW ? ba[(size_t)(r2 < rend ? r2 : r0) * 64] : 0u;  // (the arcs of step 3: its weight is requested in step 1)
    ws[0] = WAVE_BW((size_t)r0 * 64, ba[(size_t)r0 * 64]);
    ws[1] = WAVE_BW(p1, ba[p1]);
    ws[2] = 0.0;
    // RING: alpha of the level of this step and of the next two, one state per lane (levels are at most a wavefront wide)
    avs[0] = avs[1] = avs[2] = NEG_INF;
    if (RING) {
      const uint32_t la = NL - 2, sa = lvl[la], na = lvl[la + 1] - sa;
      avs[0] = lane < na ? spill[sa + lane] : NEG_INF;
      if (NL > 2) {
        const uint32_t sb = lvl[la - 1], nb = lvl[la] - sb;
        avs[1] = lane < nb ? spill[sb + lane] : NEG_INF;
      }
      al[lane] = avs[0];
      __syncthreads();
    }
    uint32_t k = 1;
#define WAVE_BWD_STEP(SL)                                                                                             \
  {                                                                                                                   \
    constexpr int S3 = (SL) % 3, N1 = ((SL) + 1) % 3, N2 = ((SL) + 2) % 3;                                            \
    const uint32_t l = NL - 1 - k;                                                                                    \
    const uint32_t s0 = lvl[l], ns = lvl[l + 1] - s0;                                                                 \
    const uint32_t r3 = WAVE_BROW(k + 3);                                                                             \
    const size_t p2 = (size_t)(r2 < rend ? r2 : r0) * 64;                                                             \
    xs[N2] = b[p2];                                                                                                   \
    if (XD) qs[N2] = xi[p2];                                                                                          \
    if (GW) gs[S3] = ba[(size_t)(r3 < rend ? r3 : r0) * 64];                                                          \
    ws[N2] = WAVE_BW(p2, gs[N2]);                                                                                     \
    avs[N2] = NEG_INF;                                                                                                \
    if (RING && l >= 2) {                                                                                             \
      const uint32_t sc = lvl[l - 2], nc = lvl[l - 1] - sc;                                                           \
      avs[N2] = lane < nc ? spill[sc + lane] : NEG_INF;                                                               \
    }                                                                                                                 \
    const uint32_t x0 = xs[S3];                                                                                       \
    const bool v0 = (x0 & WAVE_VALID) != 0;                                                                           \
    const uint32_t sr0 = (x0 >> 16) & 0x3fffu;                                                                        \
    const double t0 = v0 ? ws[S3] + val[(x0 & 0xffffu) & rm] : NEG_INF;                                               \
    const double al0 = v0 ? (RING ? al[sr0] : val[s0 + sr0]) : NEG_INF;                                               \
    if (t0 > NEG_INF) lds_max_f64(&mx[sr0], t0);                                                                      \
    WAVE_POST(r0, qs[S3], K_EXP(al0 + t0)); /* exp(-inf) = 0 on padding and dead arcs */                              \
    for (uint32_t r = r0 + 1; r < r1; ++r) {                                                                          \
      const uint32_t q = b[(size_t)r * 64];                                                                           \
      const bool v = (q & WAVE_VALID) != 0;                                                                           \
      const uint32_t sr = (q >> 16) & 0x3fffu;                                                                        \
      const double t = v ? WAVE_BW((size_t)r * 64, ba[(size_t)r * 64]) + val[(q & 0xffffu) & rm] : NEG_INF;          \
      const double a = v ? (RING ? al[sr] : val[s0 + sr]) : NEG_INF;                                                  \
      if (t > NEG_INF) lds_max_f64(&mx[sr], t);                                                                       \
      WAVE_POST(r, xi[(size_t)r * 64], K_EXP(a + t));                                                                 \
    }                                                                                                                 \
    __syncthreads();                                                                                                  \
    if (t0 > NEG_INF) lds_add_f64(&sm[sr0], K_EXP(t0 - mx[sr0]));                                                     \
    for (uint32_t r = r0 + 1; r < r1; ++r) {                                                                          \
      const uint32_t q = b[(size_t)r * 64];                                                                           \
      const uint32_t sr = (q >> 16) & 0x3fffu;                                                                        \
      const double t = (q & WAVE_VALID) ? WAVE_BW((size_t)r * 64, ba[(size_t)r * 64]) + val[(q & 0xffffu) & rm] : NEG_INF; \
      if (t > NEG_INF) lds_add_f64(&sm[sr], K_EXP(t - mx[sr]));                                                       \
    }                                                                                                                 \
    __syncthreads();                                                                                                  \
    if (RING) { /* (one pass; see the forward step) */                                                                \
      if (lane < ns) {                                                                                                \
        const double m = mx[lane], a = sm[lane];                                                                      \
        val[(s0 + lane) & rm] = (m == NEG_INF) ? NEG_INF : (a == 1.0 ? m : m + K_LOG(a));                             \
        mx[lane] = NEG_INF;                                                                                           \
        sm[lane] = 0.0;                                                                                               \
      }                                                                                                               \
    } else                                                                                                            \
    for (uint32_t i = lane; i < ns; i += 64) {                                                                        \
      const double m = mx[i], a = sm[i];                                                                              \
      val[(s0 + i) & rm] = (m == NEG_INF) ? NEG_INF : (a == 1.0 ? m : m + K_LOG(a)); /* beta[s] (full form: replaces alpha[s]) */ \
      mx[i] = NEG_INF;                                                                                                \
      sm[i] = 0.0;                                                                                                    \
    }                                                                                                                 \
    if (RING) al[lane] = avs[N1]; /* the next step's level */                                                         \
    __syncthreads();                                                                                                  \
    r0 = r1;                                                                                                          \
    r1 = r2;                                                                                                          \
    r2 = r3;                                                                                                          \
    ++k;                                                                                                              \
  }
    while (k + 2 < NL) {
      WAVE_BWD_STEP(0)
      WAVE_BWD_STEP(1)
      WAVE_BWD_STEP(2)
    }
    if (k < NL) WAVE_BWD_STEP(0)
    if (k < NL) WAVE_BWD_STEP(1)
#undef WAVE_BWD_STEP
#undef WAVE_BROW
#undef WAVE_BW
#undef WAVE_POST
  }
}

// ---------------- bundle sweep: one workgroup per bundle of lattices, level-synchronous ----------------
// forward (alpha in LDS), then backward in place: a state's alpha is read once when its level is processed and
// replaced by its beta; destinations lie in later levels and already hold beta.  Posteriors go to post[] at the
// out-arc's position.  USE_LDS=false: the values live in global scratch (lattices above the LDS cap).
template <int BLOCK, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void sweep_bundle_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const BundleDesc d = A.bundles[A.first_bundle + blockIdx.x];
  const int tid = threadIdx.x;
  const uint32_t ns = d.n_states;
  const uint2* __restrict__ ia = A.in_arcs + d.in_base;
  const uint2* __restrict__ oa = A.out_arcs + d.out_base;
  const uint32_t* __restrict__ ioff = A.in_off + d.off_base;
  const uint32_t* __restrict__ ooff = A.out_off + d.off_base;
  const uint32_t* __restrict__ lvl = A.level_off + d.level_base;
  const double* __restrict__ logw = A.logw;
  double* post = A.post + d.out_base;
  double* val = USE_LDS ? lds : (A.val_g + d.off_base);
  // ---- forward ----
  for (uint32_t s = tid; s < ns; s += BLOCK) val[s] = NEG_INF;
  __syncthreads();
  for (uint32_t p = tid; p < d.n_pairs; p += BLOCK) val[A.pair_start[d.pair_base + p]] = 0.0;
  __syncthreads();
  for (uint32_t l = 1; l < d.n_levels; ++l) {
    const uint32_t s0 = lvl[l], s1 = lvl[l + 1];
    for (uint32_t s = s0 + tid; s < s1; s += BLOCK) {
      const uint32_t a0 = ioff[s], a1 = ioff[s + 1];
      Lse acc;
      acc.init();
      for (uint32_t a = a0; a < a1; ++a) {
        const uint2 r = ia[a];
        acc.add(val[r.x] + logw[r.y]);
      }
      val[s] = acc.value();
    }
    __syncthreads();
  }
  // per pair: ln p(pair) = alpha[goal]; beta[goal] = ln(weight) - ln p(pair) folds "* weight / prob"
  // (derivations.h:445) into the sweep.  Goals have no out-arcs, so nothing else reads their alpha.
  for (uint32_t p = tid; p < d.n_pairs; p += BLOCK) {
    const uint32_t f = A.pair_final[d.pair_base + p];
    const double lp = val[f];
    const double lwt = A.pair_logw[d.pair_base + p];
    A.pair_logprob[A.pair_id[d.pair_base + p]] = lp;  // the corpus scalars are reduced from these afterwards
    val[f] = (lp == NEG_INF) ? NEG_INF : lwt - lp;
  }
  __syncthreads();
  // ---- backward + posteriors ----
  for (uint32_t l = d.n_levels; l-- > 0;) {
    const uint32_t s0 = lvl[l], s1 = lvl[l + 1];
    for (uint32_t s = s0 + tid; s < s1; s += BLOCK) {
      const uint32_t a0 = ooff[s], a1 = ooff[s + 1];
      if (a0 == a1) continue;  // goal states keep beta[goal]
      const double al = val[s];
      Lse acc;
      acc.init();
      for (uint32_t a = a0; a < a1; ++a) {
        const uint2 r = oa[a];
        const double t = logw[r.y] + val[r.x];
        acc.add(t);
        post[a] = exp(al + t);
      }
      val[s] = acc.value();
    }
    __syncthreads();
  }
}

// Cyclic lattices (derivations.h:726-728 "Forward/backward will miss some paths"): one lane per lattice walks
// the states in the reference's own order with the reference's pairwise adds, so the paths it drops and the
// values it ends with are the reference's.  States are stored in forward order; out lists and reversed-graph
// lists are in the reference's list order (lattice.cpp).
__global__ void sweep_serial_kernel(SweepArgs A, uint32_t n_bundles) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_bundles) return;
  const BundleDesc d = A.bundles[A.first_bundle + b];
  const uint2* ia = A.in_arcs + d.in_base;
  const uint2* oa = A.out_arcs + d.out_base;
  const uint32_t* ioff = A.in_off + d.off_base;
  const uint32_t* ooff = A.out_off + d.off_base;
  double* f = A.val_g + d.off_base;
  double* bb = A.val2_g + d.off_base;
  double* post = A.post + d.out_base;
  const uint32_t ns = d.n_states;
  const uint32_t st = A.pair_start[d.pair_base], fin = A.pair_final[d.pair_base];
  for (uint32_t s = 0; s < ns; ++s) {
    f[s] = NEG_INF;
    bb[s] = NEG_INF;
  }
  f[st] = 0.0;
  for (uint32_t s = 0; s < ns; ++s) {  // graph.h:391-402, order = reversed DFS post-order
    for (uint32_t a = ooff[s]; a < ooff[s + 1]; ++a) {
      const uint2 r = oa[a];
      f[r.x] = lw_add(f[r.x], f[s] + A.logw[r.y]);
    }
  }
  const double prob = f[fin];
  A.pair_logprob[A.pair_id[d.pair_base]] = prob;
  const double lwt = A.pair_logw[d.pair_base];
  bb[fin] = 0.0;
  for (uint32_t s = ns; s-- > 0;) {
    for (uint32_t a = ioff[s]; a < ioff[s + 1]; ++a) {
      const uint2 r = ia[a];
      bb[r.x] = lw_add(bb[r.x], bb[s] + A.logw[r.y]);
    }
  }
  for (uint32_t s = 0; s < ns; ++s)
    for (uint32_t a = ooff[s]; a < ooff[s + 1]; ++a) {
      const uint2 r = oa[a];
      post[a] = (prob != NEG_INF) ? exp(A.logw[r.y] + f[s] + bb[r.x] + lwt - prob) : 0.0;
    }
}

// ---------------- corpus scalars (train.cc:326-332): sum ln p, sum weight * ln p, pairs ----------------
// Reduced from pair_logprob[] in a fixed order (bit-reproducible).  Every wave adding into three shared doubles would
// serialise ~3 atomics per wave on one address: at config-4 scale that alone held the sweep kernel at 0.4 ms.
// pair_w[p] = the pair's weight, negative for pairs without a derivation (dropped from the corpus).
#define SCALAR_BLOCKS 256
__global__ __launch_bounds__(256) void scalars_partial_kernel(const double* __restrict__ pair_logprob,
                                                              const double* __restrict__ pair_w, uint64_t n_pairs,
                                                              double* __restrict__ partial) {
  __shared__ double sh[3][4];
  double a = 0.0, b = 0.0, c = 0.0;
  for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n_pairs; p += (uint64_t)SCALAR_BLOCKS * 256) {
    const double w = pair_w[p];
    if (w < 0.0) continue;
    const double lp = pair_logprob[p];
    a += lp;
    b += lp * w;
    c += 1.0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_down(a, o, 64);
    b += __shfl_down(b, o, 64);
    c += __shfl_down(c, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = a;
    sh[1][threadIdx.x >> 6] = b;
    sh[2][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}
// (all four scalar slots are written: nothing has to clear them before an E-step)
__global__ void scalars_final_kernel(const double* __restrict__ partial, double* __restrict__ scalars) {
  if (threadIdx.x < 3) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};  // four independent chains of loads, added in a fixed order
    for (int k = 0; k < SCALAR_BLOCKS; k += 4)
      for (int j = 0; j < 4; ++j) v[j] += partial[(k + j) * 3 + threadIdx.x];
    scalars[threadIdx.x] = (v[0] + v[1]) + (v[2] + v[3]);
  } else if (threadIdx.x == 3)
    scalars[3] = 0.0;
}

// ---------------- expected counts: per-arc sum of posteriors ----------------
// The posterior slots that use WFST arc a are slot_pos[arc_off[a] .. arc_off[a+1]) (sorted on the host once — the
// topology never changes between iterations).  One thread per arc: reads of arc_off / slot_pos and the store of
// counts[a] are coalesced and dense (every arc is written, so no memset), the only random traffic is one 8-byte
// gather per slot.  Arcs with more than COUNT_HOT slots (a hub arc of a small transducer can have millions) are
// left to count_reduce_hot_kernel: one workgroup per arc.
#define COUNT_HOT 64
__global__ __launch_bounds__(256) void count_reduce_kernel(ReduceArgs R) {
  for (uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; a < R.n_arcs;
       a += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t j0 = R.arc_off[a], j1 = R.arc_off[a + 1];
    double c = 0.0;
    if (j1 - j0 <= COUNT_HOT)
      for (uint64_t j = j0; j < j1; ++j) c += R.post[R.slot_pos[j]];
    R.counts[a] = c;  // hot arcs: zero here, count_reduce_hot_kernel adds its chunk sums
  }
}
// hot arcs are cut into chunks of HOT_CHUNK slots (hot_chunks[3c] = arc, [3c+1] = first slot, [3c+2] = end): one
// workgroup per chunk, one atomic per chunk
__global__ __launch_bounds__(256) void count_reduce_hot_kernel(ReduceArgs R) {
  __shared__ double sh[4];
  const uint64_t a = R.hot_chunks[3 * (uint64_t)blockIdx.x];
  const uint64_t j0 = R.hot_chunks[3 * (uint64_t)blockIdx.x + 1], j1 = R.hot_chunks[3 * (uint64_t)blockIdx.x + 2];
  double c = 0.0;
  for (uint64_t j = j0 + threadIdx.x; j < j1; j += 256) c += R.post[R.slot_pos[j]];
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomic_add_f64(R.counts + a, sh[0] + sh[1] + sh[2] + sh[3]);
}

// ---------------- M-step (fst.cc:86-244 for normal + locked arcs; train.cc:134-182) ----------------

// digamma(x), x > 0: the recurrence psi(x) = psi(x + 1) - 1/x up to x >= 10, then the asymptotic series
// ln x - 1/(2x) - sum_k B_2k / (2k x^2k) (Abramowitz & Stegun 6.3.18; the reference calls boost::math::digamma at 8
// digits, digamma.hpp:24-30 -- a third-party dependency absent from the tree, restated from the published series)
__device__ __forceinline__ double digamma_pos(double x) {
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double z = 1.0 / (x * x);
  // B_2k / 2k: 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12
  const double ser = z * (1.0 / 12 - z * (1.0 / 120 - z * (1.0 / 252 - z * (1.0 / 240 - z * (1.0 / 132 - z * (691.0 / 32760 - z / 12))))));
  return r + log(x) - 0.5 / x - ser;
}
// ln of mean_field_scale::operator() (mean_field_scale.hpp:40-52): linear (alpha is NaN) -> ln x; else
// exp(digamma(x + alpha)), continued linearly to 0 below x + alpha = .0002
__device__ __forceinline__ double ln_scale(double x, double alpha) {
  if (alpha != alpha) return x > 0.0 ? log(x) : NEG_INF;
  const double xa = x + alpha, floor_x = .0002;
  if (xa < floor_x) return xa > 0.0 ? digamma_pos(floor_x) + log(xa / floor_x) : NEG_INF;
  return digamma_pos(xa);
}

// unnormalised linear value of parameter k: prep_new_weights (train.cc:134-153) + "w += addc" of normalize pass 1
// (fst.cc:125).  Locked arcs keep their weight (plus addc, as the reference does).
__device__ __forceinline__ double mstep_value(const MstepArgs& M, uint64_t k, int use_counts, uint32_t ng) {
  const bool locked = (M.group[k] == 0u);
  double v;
  if (locked || !use_counts)
    v = exp(M.logw[k]);
  else
    v = M.counts[k] + (M.prior ? M.prior[k] : 0.0);
  return M.add_count ? v + M.add_count[ng] : v;
}

// ln((1 - sum of locked) / sum of normal), kept as a difference of logs so that a lone arc (v == sum) comes out as
// exactly 1 (fst.cc:213-230); -inf when nothing is left to distribute
__device__ __forceinline__ double mstep_scale(double sn, double sl, double alpha = __builtin_nan("")) {
  const double remain = 1.0 - sl;
  if (alpha == alpha) return (remain > 0.0 && sn > 0.0) ? log(remain) - ln_scale(sn, alpha) : NEG_INF;  // fst.cc:217-221
  return (remain > 0.0 && sn > 0.0) ? (sl == 0.0 ? -log(sn) : log(remain) - log(sn)) : NEG_INF;
}
// new weight of member k (locked arcs keep v) and its |change| in the real domain (weight.h:837-856)
__device__ __forceinline__ double mstep_update(const MstepArgs& M, uint64_t k, int use_counts, uint32_t g, double sc,
                                               double& mx) {
  const double old = M.logw[k];
  if (M.save_old == 1) M.old_logw[k] = old;
  const double v = mstep_value(M, k, use_counts, g);
  double nw;
  if (M.group[k] == 0u) {
    nw = v > 0.0 ? log(v) : NEG_INF;
  } else if (M.tie_of && M.tie_of[k] != 0xffffffffu) {
    const double tw = M.tie_tab[3 * M.n_ties + M.tie_of[k]];
    nw = tw > 0.0 ? log(tw) : NEG_INF;
    mx = fmax(mx, fabs(tw - exp(M.save_old ? old : M.old_logw[k])));
  } else {
    if (M.dig_alpha && M.dig_alpha[g] == M.dig_alpha[g])
      nw = sc != NEG_INF ? ln_scale(v, M.dig_alpha[g]) + sc : NEG_INF;
    else
      nw = (sc != NEG_INF && v > 0.0) ? log(v) + sc : NEG_INF;
    mx = fmax(mx, fabs(exp(nw) - exp(M.save_old ? old : M.old_logw[k])));
  }
  return nw;
}
__device__ __forceinline__ void mstep_block_max(double mx, unsigned long long* out) {
  __shared__ double shm[4];
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) *out = (unsigned long long)__double_as_longlong(fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3])));
}
// The M-step of WFST::normalize (fst.cc:86-244) in one pass: one thread per norm group sums its members (listed in
// norm_perm[group_off[g] .. group_off[g+1]), contiguous for the per-state groups of JOINT / CONDITIONAL) and then
// writes their new weights -- no per-group scale array, no second read of the counts, no atomics.  The largest
// |change| leaves as one partial per workgroup (every wave pushing an atomicMax onto one address is the same
// serialisation that the corpus scalars had).
#define MSTEP_GRID 2048
// Tied arcs (!N), fst.cc:107-152.
// pass A, one thread per norm group: the sum over its unlocked members (normal and tied) and over its locked ones
__global__ __launch_bounds__(256) void mstep_tie_sums_kernel(MstepArgs M, int use_counts) {
  for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < M.n_groups; g += (uint64_t)gridDim.x * 256) {
    const uint64_t j0 = M.group_off[g], j1 = M.group_off[g + 1];
    double su = 0.0, sl = 0.0;
    for (uint64_t j = j0; j < j1; ++j) {
      const uint64_t k = M.norm_perm[j];
      const double v = mstep_value(M, k, use_counts, (uint32_t)g);
      if (M.group[k] == 0u)
        sl += v;
      else
        su += v;
    }
    M.gscale[g] = su;  // scratch until mstep_group_sum_kernel writes the real scale
    M.glocked[g] = sl;
  }
}
// pass B, one thread per parameter: a tied one adds its value to its tie's arc total, its group's unlocked sum to the
// tie's state total, and its group's locked sum to the tie's maximum.  Ties are rare: atomics.
__global__ __launch_bounds__(256) void mstep_tie_accum_kernel(MstepArgs M, int use_counts) {
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < M.n; k += (uint64_t)gridDim.x * 256) {
    const uint32_t tie = M.tie_of[k];
    const uint32_t g = M.norm_of[k];
    if (tie == 0xffffffffu || g == 0xffffffffu) continue;
    atomic_add_f64(M.tie_tab + tie, mstep_value(M, k, use_counts, g));
    atomic_add_f64(M.tie_tab + M.n_ties + tie, M.gscale[g]);
    atomicMax((unsigned long long*)(M.tie_tab + 2 * M.n_ties + tie), (unsigned long long)__double_as_longlong(M.glocked[g]));
  }
}
// the tie's weight (fst.cc:169-195): total / (state total / (1 - max locked)), 0 when nothing can be given
__global__ void mstep_tie_weight_kernel(MstepArgs M) {
  const uint64_t tie = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tie >= M.n_ties) return;
  const double total = M.tie_tab[tie], gmax = M.tie_tab[2 * M.n_ties + tie];
  double norm = M.tie_tab[M.n_ties + tie], w = 0.0;
  if (!(gmax > 1.0)) {
    if (gmax != 0.0) norm /= (1.0 - gmax);
    if (total != 0.0) {
      const double a = M.tie_alpha ? M.tie_alpha[tie] : __builtin_nan("");
      w = (a == a) ? exp(ln_scale(total, a) - ln_scale(norm, a)) : total / norm;  // scale(groupTotal) / scale(groupNorm)
    }
  }
  M.tie_tab[3 * M.n_ties + tie] = w;
}
// pass 0, one thread per norm group: gscale[g] from the sums over its normal and locked members (listed in
// norm_perm[group_off[g] .. group_off[g+1]), contiguous for the per-state groups of JOINT / CONDITIONAL); no atomics
__global__ __launch_bounds__(256) void mstep_group_sum_kernel(MstepArgs M, int use_counts) {
  for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < M.n_groups; g += (uint64_t)gridDim.x * 256) {
    const uint64_t j0 = M.group_off[g], j1 = M.group_off[g + 1];
    if (j1 - j0 > MSTEP_BIG_GROUP) continue;  // mstep_big_group_kernel
    double sn = 0.0, sl = 0.0;
    for (uint64_t j = j0; j < j1; ++j) {
      const uint64_t k = M.norm_perm[j];
      const double v = mstep_value(M, k, use_counts, (uint32_t)g);
      if (M.group[k] == 0u)
        sl += v;
      else if (M.tie_of && M.tie_of[k] != 0xffffffffu)
        sl += M.tie_tab[3 * M.n_ties + M.tie_of[k]];  // a tied arc's share is reserved like a locked one (fst.cc:169-195)
      else
        sn += v;
    }
    M.gscale[g] = mstep_scale(sn, sl, M.dig_alpha ? M.dig_alpha[g] : __builtin_nan(""));
  }
}
// pass 1, one thread per parameter (coalesced): the new weight; the largest |change| leaves as one partial per
// workgroup (every wave pushing an atomicMax onto one address is the same serialisation the corpus scalars had)
__global__ __launch_bounds__(256) void mstep_normalize_kernel(MstepArgs M, int use_counts) {
  double mx = 0.0;
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < M.n; k += (uint64_t)MSTEP_GRID * 256) {
    const uint32_t ng = M.norm_of[k];
    if (ng == 0xffffffffu) {  // member normalised by NONE keeps its weights (cascade.h:339-350)
      if (M.save_old == 1) M.old_logw[k] = M.logw[k];
      continue;
    }
    M.logw[k] = mstep_update(M, k, use_counts, ng, M.gscale[ng], mx);
  }
  mstep_block_max(mx, M.max_partial + blockIdx.x);
}
// The whole M-step in one pass when every norm group lives inside a window of `span` consecutive parameters (the
// per-state groups of JOINT / CONDITIONAL normalisation over a state-major arc table: span < out-degree): a workgroup
// stages 256 + 2 * span values and group ids in LDS, each thread adds up the members of its own group in ascending
// parameter order -- the same order, hence the same bits, as the group-major pass above -- and writes its new weight.
// One coalesced read of the counts, one write of the weights; no group table, no scale array.
#define MSTEP_WINDOW_MAX 64
template <bool NEED_LW>
__global__ __launch_bounds__(256) void mstep_window_kernel(MstepArgs M, int use_counts, uint32_t span) {
  __shared__ double v_sh[256 + 2 * MSTEP_WINDOW_MAX];
  __shared__ uint16_t g_sh[256 + 2 * MSTEP_WINDOW_MAX];  // MstepArgs::code16
  uint32_t blk = M.block_first + blockIdx.x;
  if (M.n_ranges) {
    uint32_t r = 0;
    while (r + 1 < M.n_ranges && blockIdx.x >= M.range_cum[r + 1]) ++r;
    blk = M.range_first[r] + (blockIdx.x - M.range_cum[r]);
  }
  const int64_t base = (int64_t)blk * 256 - (int64_t)span;
  // every global load of the workgroup is issued up front (tile element, halo element, own old weight): one round
  // trip per workgroup instead of one per dependent step.  NEED_LW: some member's value is its current weight (no
  // counts in use, or locked members) -- read from the snapshot lw_src, never from the array being rewritten.
  const uint64_t kown = (uint64_t)blk * 256 + threadIdx.x;
  const double old_own = kown < M.n ? M.logw[kown] : 0.0;
  const uint32_t mask_own = (M.mask32 && kown < M.n) ? M.mask32[kown] : 0u;
  const uint32_t lock_own = (NEED_LW && M.mask32 && kown < M.n) ? M.lockmask32[kown] : 0u;
  const unsigned long long mask_own64 = (M.mask64 && kown < M.n) ? M.mask64[kown] : 0ull;
  const unsigned long long lock_own64 = (NEED_LW && M.mask64 && kown < M.n) ? M.lockmask64[kown] : 0ull;
  int64_t kk[2] = {base + threadIdx.x, base + 256 + threadIdx.x};
  bool in[2];
  uint16_t code[2];
  double cv[2], pv[2], lw[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    in[h] = kk[h] >= 0 && (uint64_t)kk[h] < M.n && (h == 0 || threadIdx.x < 2 * span);
    const uint64_t k = in[h] ? (uint64_t)kk[h] : 0;
    code[h] = M.code16[k];
    lw[h] = NEED_LW ? M.lw_src[k] : 0.0;
    cv[h] = use_counts ? M.counts[k] : 0.0;
    pv[h] = (use_counts && M.prior) ? M.prior[k] : 0.0;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1 && threadIdx.x >= 2 * span) break;
    double v = 0.0;
    uint16_t g = 0xffffu;
    if (in[h] && code[h] != 0xffffu) {
      g = code[h];
      const bool locked = (g & 0x4000u) != 0;
      v = (NEED_LW && (locked || !use_counts)) ? exp(lw[h]) : cv[h] + pv[h];  // mstep_value
      if (M.add_count) v += M.add_count[M.norm_of[kk[h]]];
    }
    v_sh[threadIdx.x + h * 256] = v;
    g_sh[threadIdx.x + h * 256] = g;
  }
  __syncthreads();
  double mx = 0.0;
  const uint64_t k = (uint64_t)blk * 256 + threadIdx.x;
  if (k < M.n) {
    const uint32_t me = threadIdx.x + span;
    const uint16_t gid = g_sh[me];
    if (gid == 0xffffu) {  // member normalised by NONE keeps its weights (cascade.h:339-350)
      if (M.save_old == 1) M.old_logw[k] = old_own;
    } else {
      const uint16_t want = gid & 0x3fffu;
      double sn = 0.0, sl = 0.0;
      if (M.mask32) {  // the members are known: add them up in ascending order (the order of the scan below)
        for (uint32_t m = mask_own; m; m &= m - 1) sn += v_sh[me + (uint32_t)__builtin_ctz(m) - 15u];
        if (NEED_LW)
          for (uint32_t m = lock_own; m; m &= m - 1) sl += v_sh[me + (uint32_t)__builtin_ctz(m) - 15u];
      } else if (M.mask64) {
        for (unsigned long long m = mask_own64; m; m &= m - 1) sn += v_sh[me + (uint32_t)__builtin_ctzll(m) - 31u];
        if (NEED_LW)
          for (unsigned long long m = lock_own64; m; m &= m - 1) sl += v_sh[me + (uint32_t)__builtin_ctzll(m) - 31u];
      } else
        for (uint32_t j = me - span; j <= me + span; ++j) {
          const uint16_t gj = g_sh[j];
          if (gj == 0xffffu || (gj & 0x3fffu) != want) continue;
          if (gj & 0x4000u)
            sl += v_sh[j];
          else
            sn += v_sh[j];
        }
      // new weight straight from the sums: one division, one log (and one exp for the old weight) per parameter
      const double old = old_own;
      if (M.save_old == 1) M.old_logw[k] = old;
      const double v = v_sh[me];
      double nw;
      if (gid & 0x4000u) {
        nw = v > 0.0 ? log(v) : NEG_INF;
      } else {
        const double remain = 1.0 - sl;
        const bool ok = remain > 0.0 && sn > 0.0 && v > 0.0;
        const double lin = ok ? (sl == 0.0 ? v / sn : v * remain / sn) : 0.0;  // a lone arc: v / v == 1 exactly
        nw = ok ? log(lin) : NEG_INF;
        mx = fmax(mx, fabs(lin - exp(M.save_old ? old : M.old_logw[k])));
      }
      M.logw[k] = nw;
    }
  }
  // more workgroups than partial slots: fold by atomicMax (2048 addresses, a few adds each: no serialisation)
  __shared__ double shm[4];
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double m = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
    if (m > 0.0) atomicMax(M.max_partial + (blk % MSTEP_GRID), (unsigned long long)__double_as_longlong(m));
  }
}
// pass 0 for big groups (e.g. JOINT normalisation of a state with 10^5 arcs): one workgroup per group
__global__ __launch_bounds__(256) void mstep_big_group_kernel(MstepArgs M, int use_counts) {
  __shared__ double sh[2][4];
  const uint64_t g = M.big_groups[blockIdx.x];
  const uint64_t j0 = M.group_off[g], j1 = M.group_off[g + 1];
  double sn = 0.0, sl = 0.0;
  for (uint64_t j = j0 + threadIdx.x; j < j1; j += 256) {
    const uint64_t k = M.norm_perm[j];
    const double v = mstep_value(M, k, use_counts, (uint32_t)g);
    if (M.group[k] == 0u)
      sl += v;
    else if (M.tie_of && M.tie_of[k] != 0xffffffffu)
      sl += M.tie_tab[3 * M.n_ties + M.tie_of[k]];
    else
      sn += v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sn += __shfl_down(sn, o, 64);
    sl += __shfl_down(sl, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = sn;
    sh[1][threadIdx.x >> 6] = sl;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    M.gscale[g] = mstep_scale(sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3],
                              M.dig_alpha ? M.dig_alpha[g] : __builtin_nan(""));
}
// (the partials are cleared for the next pass here and the result is stored, not folded: no memset per M-step)
__global__ __launch_bounds__(256) void mstep_max_final_kernel(unsigned long long* partial, uint64_t n,
                                                              unsigned long long* bits, unsigned long long* box,
                                                              unsigned long long box_seq) {
  __shared__ unsigned long long shm[4];
  unsigned long long m = 0;  // non-negative doubles order like their bit patterns
  for (uint64_t k = threadIdx.x; k < n; k += 256) {
    m = partial[k] > m ? partial[k] : m;
    partial[k] = 0ull;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_down(m, o, 64);
    m = other > m ? other : m;
  }
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) m = shm[k] > m ? shm[k] : m;
    *bits = m;
    if (box) {  // the host's mailbox: the value, then the sequence number it waits for
      __hip_atomic_store(box, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(box + 1, box_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// overrelax (train.cc:157-171): w = old * (em/old)^rate for unlocked arcs with old > 0
__global__ void overrelax_kernel(double* logw, const double* old_logw, double* em_logw, const uint32_t* group,
                                 double rate, uint64_t n) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
    const double em = logw[k];
    em_logw[k] = em;
    if (group[k] != 0u && old_logw[k] != NEG_INF) logw[k] = old_logw[k] + (em - old_logw[k]) * rate;
  }
}

__global__ void max_change_kernel(const double* logw, const double* old_logw, const uint32_t* group,
                                  unsigned long long* max_change_bits, uint64_t n) {
  double mx = 0.0;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
    if (group[k] != 0u) mx = fmax(mx, fabs(exp(logw[k]) - exp(old_logw[k])));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(max_change_bits, (unsigned long long)__double_as_longlong(mx));
}

// ---------------- cascade gather / scatter (cascade.h:426-433, 286-325) ----------------
// composed arc weight = sum of ln weights of its chain's parameters
__global__ void chain_update_kernel(double* arc_logw, const uint32_t* arc_chain, const uint64_t* chain_off,
                                    const uint64_t* chain_param, const double* param_logw, uint64_t n_arcs) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_arcs; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = arc_chain[k];
    double w = 0.0;
    bool zero = false;
    for (uint64_t j = chain_off[c]; j < chain_off[c + 1]; ++j) {
      const double pw = param_logw[chain_param[j]];
      if (pw == NEG_INF) zero = true;
      w += pw;
    }
    arc_logw[k] = zero ? NEG_INF : w;
  }
}
// parameter counts += (composed count + composed prior) for every unlocked parameter of the arc's chain
__global__ void chain_scatter_kernel(double* param_counts, const double* arc_counts, double arc_prior,
                                     const double* arc_prior_w, const uint32_t* arc_chain, const uint64_t* chain_off,
                                     const uint64_t* chain_param, const uint32_t* param_group, uint64_t n_arcs) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_arcs; k += (uint64_t)gridDim.x * blockDim.x) {
    const double c = arc_counts[k] + (arc_prior_w ? arc_prior + arc_prior_w[k] : arc_prior);
    if (!(c > 0.0)) continue;
    const uint32_t ch = arc_chain[k];
    for (uint64_t j = chain_off[ch]; j < chain_off[ch + 1]; ++j) {
      const uint64_t p = chain_param[j];
      if (param_group[p] != 0u) atomic_add_f64(param_counts + p, c);
    }
  }
}

__global__ void add_f64_kernel(double* dst, const double* src, uint64_t n) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[k] += src[k];
}
__global__ void fill_f64_kernel(double* p, double v, uint64_t n) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) p[k] = v;
}

// Run-length source indices (TransArgs::tr_* / br_*): the items of a tile arrive as one run per bucket (of a bucket: one
// per tile), so instead of 4 bytes per item the kernels read {first item, first source} per RUN and find an item's run
// without searching: the run starts are set as bits of a mask in LDS (one bit per item), a prefix count per mask word
// follows, and item i belongs to run  pref[i / 32] + popc(mask[i / 32] up to bit i % 32) - 1  -- two LDS reads and a
// popcount, the same for every item (a binary search was eleven dependent LDS reads per item and made both kernels slower
// than the per-item indices).  r_src holds source - first item, so the item's source is r_src[run] + i.
#define TRANS_RUN_WORDS ((TRANS_TILE > TRANS_BUCKET ? TRANS_TILE : TRANS_BUCKET) / 32)
struct RunLds {
  uint32_t* r_src;   // TRANS_RUN_CAP
  uint32_t* mask;    // TRANS_TILE / 32 words
  uint32_t* pref;    // ... run starts before word w
};
__device__ __forceinline__ RunLds run_lds(double* after_tile) {
  RunLds R;
  R.r_src = (uint32_t*)after_tile;
  R.mask = R.r_src + TRANS_RUN_CAP;
  R.pref = R.mask + TRANS_RUN_WORDS;
  return R;
}
// (called by all 1024 threads; contains barriers)
__device__ __forceinline__ void run_stage(const RunLds& R, const uint16_t* __restrict__ rel, const uint32_t* __restrict__ src,
                                          uint32_t nr) {
  constexpr uint32_t WORDS = TRANS_RUN_WORDS, PER = WORDS / 64;
  for (uint32_t w = threadIdx.x; w < WORDS; w += 1024) R.mask[w] = 0u;
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < nr; r += 1024) {
    const uint32_t first = rel[r];
    R.r_src[r] = src[r] - first;
    atomicOr(&R.mask[first >> 5], 1u << (first & 31u));
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    uint32_t c[PER], tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      c[k] = tot;
      tot += __popc(R.mask[threadIdx.x * PER + k]);
    }
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(inc, o, 64);
      if ((int)threadIdx.x >= o) inc += v;
    }
    const uint32_t before = inc - tot;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) R.pref[threadIdx.x * PER + k] = before + c[k];
  }
  __syncthreads();
}
__device__ __forceinline__ uint32_t run_source(const RunLds& R, uint32_t i) {
  const uint32_t w = i >> 5;
  const uint32_t run = R.pref[w] + __popc(R.mask[w] & (0xffffffffu >> (31u - (i & 31u)))) - 1u;
  return R.r_src[run] + i;
}
// weights, pass 1: one workgroup per arc bucket.  The bucket's weights go to LDS (coalesced read), its items leave in
// position-sorted order (coalesced write), picking their weight out of LDS.
// SC (TransArgs::scatter bit 0): the items leave for their TILE-major place instead (one contiguous run per tile: a
// scattered write that nobody waits for), and pass 2 reads its tile's items as one sequential stretch -- the dependent
// round trip "index, then gather" moves from the reading pass, which waits for it, to the writing pass, which does not.
template <bool SC, bool RL, int KB = TRANS_KB>  // KB: rounds of 1024 threads a bucket holds at most (LatticeSet::bucket / 1024)
__global__ __launch_bounds__(1024) void trans_w_bucket_kernel(TransArgs T) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const RunLds R = run_lds(lds + KB * 1024);
  uint32_t bloc = blockIdx.x;
  if (SC) {
    bloc = xcd_chunked(blockIdx.x, T.bucket_count);  // the grid is rounded up to a multiple of 8
    if (bloc >= T.bucket_count) return;
  }
  const uint32_t bucket = T.bucket_first + bloc;
  const TransBucket B = T.buckets[bucket];
  // every loop below is a fixed 16 x 1024 sweep with its loads issued as one batch (a bucket / tile holds at most
  // 16384 items): one dependent round trip per phase instead of one per iteration
  double w[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const uint32_t a = threadIdx.x + k * 1024;
    w[k] = a < B.n_arcs ? T.logw[B.arc_lo + a] : 0.0;
  }
  uint16_t ia[KB];
  uint32_t dst[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const uint32_t j = threadIdx.x + k * 1024;
    ia[k] = j < B.n_items ? T.b_arc[B.item_base + j] : (uint16_t)0;
    if (SC && !RL) dst[k] = j < B.n_items ? T.b_src[B.item_base + j] : 0u;
  }
#pragma unroll
  for (int k = 0; k < KB; ++k) lds[threadIdx.x + k * 1024] = w[k];
  if (SC && RL) {
    const uint32_t r0 = T.br_off[bucket];
    run_stage(R, T.br_rel + r0, T.br_src + r0, T.br_off[bucket + 1] - r0);
  } else
    __syncthreads();
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const uint32_t j = threadIdx.x + k * 1024;
    if (j < B.n_items) {
      if (SC)
        T.x[RL ? run_source(R, j) : dst[k]] = lds[ia[k]];
      else
        T.x[B.item_base + j] = lds[ia[k]];
    }
  }
}
// weights, pass 2: one workgroup per tile of positions.  The tile's items arrive as runs (one per bucket), are placed
// in LDS at their position and the tile is written to wcache in one coalesced sweep.
#ifndef TRANS_TILE_WAVES
#define TRANS_TILE_WAVES 4  // waves per SIMD the tile kernels are compiled for (4: one workgroup per CU)
#endif
#ifndef TRANS_WB_WAVES
#define TRANS_WB_WAVES 4
#endif
// SQ: pass 1 scattered (see trans_w_bucket_kernel<true, ..>): the tile's items are x[tile_base .. ) in item order.
template <bool RL, bool SQ>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(TRANS_TILE_WAVES, TRANS_TILE_WAVES))) void trans_w_tile_kernel(TransArgs T) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const RunLds R = run_lds(lds + TRANS_TILE);
  const uint32_t tloc = xcd_chunked(blockIdx.x, T.tile_count);  // the grid is rounded up to a multiple of 8
  if (tloc >= T.tile_count) return;
  const uint32_t tile = T.tile_first + tloc;
  const uint64_t p0 = (uint64_t)tile * T.tile;
  if (p0 >= T.n_wcache) return;  // tiles of bundle positions: the bundle sweep gathers its weights itself
  const uint32_t np = (uint32_t)min((uint64_t)T.tile, T.n_wcache - p0);
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k) lds[threadIdx.x + k * 1024] = 0.0;
  if (RL && !SQ) {
    const uint32_t r0 = T.tr_off[tile];
    run_stage(R, T.tr_rel + r0, T.tr_src + r0, T.tr_off[tile + 1] - r0);
  } else
    __syncthreads();
  const uint64_t i0 = T.tile_base[tile];
  const uint32_t ni = (uint32_t)(T.tile_base[tile + 1] - i0);
  uint32_t src[TRANS_KT];
  uint16_t pos[TRANS_KT];
  double v[TRANS_KT];
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k) {
    const uint32_t i = threadIdx.x + k * 1024;
    if (SQ)
      src[k] = 0u;
    else if (RL)
      src[k] = i < ni ? run_source(R, i) : 0u;
    else
      src[k] = i < ni ? T.t_src[i0 + i] : 0u;
    pos[k] = i < ni ? T.t_pos[i0 + i] : (uint16_t)0;
  }
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k) {
    if (SQ) {
      const uint32_t i = threadIdx.x + k * 1024;
      v[k] = i < ni ? T.x[i0 + i] : 0.0;
    } else
      v[k] = T.x[src[k]];
  }
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k)
    if (threadIdx.x + k * 1024 < ni) lds[pos[k]] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k) {
    const uint32_t q = threadIdx.x + k * 1024;
    if (q < np) T.wcache[p0 + q] = lds[q];
  }
}
// counts, pass 1: one workgroup per tile: posteriors to LDS (coalesced), items out in item order (tile-major).
// SC (TransArgs::scatter bit 1): items out to their BUCKET-major place (one contiguous run per bucket), pass 2 reads a
// bucket's items as one sequential stretch.
template <bool SC, bool RL>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(TRANS_TILE_WAVES, TRANS_TILE_WAVES))) void trans_c_tile_kernel(TransArgs T) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const RunLds R = run_lds(lds + TRANS_TILE);
  uint32_t tloc = blockIdx.x;
  if (SC) {
    tloc = xcd_chunked(blockIdx.x, T.tile_count);  // the grid is rounded up to a multiple of 8
    if (tloc >= T.tile_count) return;
  }
  const uint32_t tile = T.tile_first + tloc;
  const uint64_t p0 = (uint64_t)tile * T.tile;
  const uint32_t np = (uint32_t)min((uint64_t)T.tile, T.n_post - p0);
  const uint64_t i0 = T.tile_base[tile];
  const uint32_t ni = (uint32_t)(T.tile_base[tile + 1] - i0);
  double v[TRANS_KT];
  uint16_t pos[TRANS_KT];
  uint32_t dst[TRANS_KT];
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k) {
    const uint32_t q = threadIdx.x + k * 1024;
    v[k] = q < np ? T.post[p0 + q] : 0.0;
    pos[k] = q < ni ? T.t_pos[i0 + q] : (uint16_t)0;
    if (SC && !RL) dst[k] = q < ni ? T.t_src[i0 + q] : 0u;
  }
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k) lds[threadIdx.x + k * 1024] = v[k];
  if (SC && RL) {
    const uint32_t r0 = T.tr_off[tile];
    run_stage(R, T.tr_rel + r0, T.tr_src + r0, T.tr_off[tile + 1] - r0);
  } else
    __syncthreads();
#pragma unroll
  for (int k = 0; k < TRANS_KT; ++k) {
    const uint32_t i = threadIdx.x + k * 1024;
    if (i < ni) {
      if (SC)
        T.xc[RL ? run_source(R, i) : dst[k]] = lds[pos[k]];
      else
        T.xc[i0 + i] = lds[pos[k]];
    }
  }
}
// The tile passes for the fused-lane layout (LatticeSet::lane_fused: tiles of LANE_FUSED_TILE = 2048 positions, a lane group
// starting on each): the same two kernels at an eighth of the size -- 256 threads, 8 items each, 16 KB of LDS, eight workgroups
// to a CU, so one workgroup's load phase runs under another's store phase.  Per-item indices (a tile has ~230 runs of 9).
#define TRANS_SMALL_THREADS 256
#define TRANS_SMALL_K ((int)(LANE_FUSED_TILE / TRANS_SMALL_THREADS))
template <bool SQ>
__global__ __launch_bounds__(TRANS_SMALL_THREADS) void trans_w_tile_small_kernel(TransArgs T) {
  __shared__ double lds[LANE_FUSED_TILE];
  constexpr int NT = TRANS_SMALL_THREADS, K = TRANS_SMALL_K;
  const uint32_t tloc = xcd_chunked(blockIdx.x, T.tile_count);  // the grid is rounded up to a multiple of 8
  if (tloc >= T.tile_count) return;
  const uint32_t tile = T.tile_first + tloc;
  const uint64_t p0 = (uint64_t)tile * LANE_FUSED_TILE;
  if (p0 >= T.n_wcache) return;
  const uint32_t np = (uint32_t)min((uint64_t)LANE_FUSED_TILE, T.n_wcache - p0);
  const uint64_t i0 = T.tile_base[tile];
  const uint32_t ni = (uint32_t)(T.tile_base[tile + 1] - i0);
  uint32_t src[K];
  uint16_t pos[K];
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t i = threadIdx.x + k * NT;
    src[k] = (!SQ && i < ni) ? T.t_src[i0 + i] : 0u;
    pos[k] = i < ni ? T.t_pos[i0 + i] : (uint16_t)0;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) lds[threadIdx.x + k * NT] = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t i = threadIdx.x + k * NT;
    if (SQ)
      v[k] = i < ni ? T.x[i0 + i] : 0.0;
    else
      v[k] = T.x[src[k]];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (threadIdx.x + k * NT < ni) lds[pos[k]] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t q = threadIdx.x + k * NT;
    if (q < np) T.wcache[p0 + q] = lds[q];
  }
}
// (gather form only: the fused-lane layout keeps the counts direction's random access in the bucket pass)
__global__ __launch_bounds__(TRANS_SMALL_THREADS) void trans_c_tile_small_kernel(TransArgs T) {
  __shared__ double lds[LANE_FUSED_TILE];
  constexpr int NT = TRANS_SMALL_THREADS, K = TRANS_SMALL_K;
  const uint32_t tile = T.tile_first + blockIdx.x;
  const uint64_t p0 = (uint64_t)tile * LANE_FUSED_TILE;
  const uint32_t np = p0 < T.n_post ? (uint32_t)min((uint64_t)LANE_FUSED_TILE, T.n_post - p0) : 0u;
  const uint64_t i0 = T.tile_base[tile];
  const uint32_t ni = (uint32_t)(T.tile_base[tile + 1] - i0);
  double v[K];
  uint16_t pos[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t q = threadIdx.x + k * NT;
    v[k] = q < np ? T.post[p0 + q] : 0.0;
    pos[k] = q < ni ? T.t_pos[i0 + q] : (uint16_t)0;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) lds[threadIdx.x + k * NT] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t i = threadIdx.x + k * NT;
    if (i < ni) T.xc[i0 + i] = lds[pos[k]];
  }
}
// counts, pass 2: one workgroup per arc bucket: its items (runs, one per tile) are placed in LDS in arc-sorted order,
// then one thread per arc adds up its contiguous range in a fixed order -- no atomics, bit-reproducible.  A bucket
// that is a piece of a split arc reduces the piece and adds it atomically.
// SQ: pass 1 scattered (trans_c_tile_kernel<true, ..>): the bucket's items are xc[item_base .. ) in item order.
template <bool RL, bool SQ, int KB = TRANS_KB>
__global__ __launch_bounds__(1024) void trans_c_bucket_kernel(TransArgs T) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const RunLds R = run_lds(lds + KB * 1024);
  __shared__ double part[16];
  __shared__ uint32_t big[512];
  __shared__ uint32_t n_big;
  const uint32_t bloc = xcd_chunked(blockIdx.x, T.bucket_count);  // the grid is rounded up to a multiple of 8
  if (bloc >= T.bucket_count) return;
  const uint32_t bucket = T.bucket_first + bloc;
  const TransBucket B = T.buckets[bucket];
  // every global load of the workgroup is issued before the first barrier: item indices, the items, and the item
  // ranges of this thread's arcs (the per-arc loop below then runs out of registers and LDS alone)
  uint16_t r0[KB], r1[KB];
  const bool single = (B.flags & TRANS_SINGLE) != 0;
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const uint32_t a = threadIdx.x + k * 1024;
    const bool ok = !single && a < B.n_arcs;
    r0[k] = ok ? T.a_off[B.arc_lo + a] : (uint16_t)0;
    r1[k] = (ok && a + 1 < B.n_arcs) ? T.a_off[B.arc_lo + a + 1] : (uint16_t)0;
  }
  {
    uint32_t src[KB];
    uint16_t rk[KB];
    double v[KB];
    if (RL && !SQ) {
      const uint32_t r0 = T.br_off[bucket];
      run_stage(R, T.br_rel + r0, T.br_src + r0, T.br_off[bucket + 1] - r0);
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const uint32_t j = threadIdx.x + k * 1024;
      if (SQ)
        src[k] = 0u;
      else if (RL)
        src[k] = j < B.n_items ? run_source(R, j) : 0u;
      else
        src[k] = j < B.n_items ? T.b_src[B.item_base + j] : 0u;
      rk[k] = j < B.n_items ? T.b_rank[B.item_base + j] : (uint16_t)0;
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (SQ) {
        const uint32_t j = threadIdx.x + k * 1024;
        v[k] = j < B.n_items ? T.xc[B.item_base + j] : 0.0;
      } else
        v[k] = T.xc[src[k]];
    }
#pragma unroll
    for (int k = 0; k < KB; ++k)
      if (threadIdx.x + k * 1024 < B.n_items) lds[rk[k]] = v[k];
  }
  __syncthreads();
  if (single) {
    double v = 0.0;
    for (uint32_t j = threadIdx.x; j < B.n_items; j += 1024) v += lds[j];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int k = 0; k < 16; ++k) tot += part[k];
      if (B.flags & TRANS_SPLIT)
        atomic_add_f64(T.counts + B.arc_lo, tot);
      else
        T.counts[B.arc_lo] = tot;
    }
    return;
  }
  // one thread per arc; arcs with more than 32 items are left to whole waves afterwards (fixed summation order
  // either way).  (A thread's 16 arcs advancing together, one item of each per step -- 16 independent LDS reads per
  // step instead of a chain per arc -- was measured 50 % slower on config 4, 86 -> 132 us: the loop is not what the
  // kernel waits for, and the 48 extra registers cost more than the chains.)
  if (threadIdx.x == 0) n_big = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const uint32_t a = threadIdx.x + k * 1024;
    if (a >= B.n_arcs) break;
    const uint32_t q0 = r0[k], q1 = a + 1 < B.n_arcs ? (uint32_t)r1[k] : B.n_items;
    if (q1 - q0 > 32) {
      const uint32_t q = atomicAdd(&n_big, 1u);
      if (q < 512) {
        big[q] = a;
        continue;
      }
    }
    double v = 0.0;
    for (uint32_t r = q0; r < q1; ++r) v += lds[r];
    T.counts[B.arc_lo + a] = v;
  }
  __syncthreads();
  const uint32_t nb = n_big < 512 ? n_big : 512;
  for (uint32_t q = threadIdx.x >> 6; q < nb; q += 16) {
    const uint32_t a = big[q];
    const uint32_t q0 = T.a_off[B.arc_lo + a], q1 = a + 1 < B.n_arcs ? (uint32_t)T.a_off[B.arc_lo + a + 1] : B.n_items;
    double v = 0.0;
    for (uint32_t r = q0 + (threadIdx.x & 63); r < q1; r += 64) v += lds[r];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) T.counts[B.arc_lo + a] = v;
  }
}
__global__ void zero_list_kernel(double* p, const uint32_t* idx, uint32_t n) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) p[idx[k]] = 0.0;
}

// ---------------- launch helpers (called from engine.cpp, compiled in this TU) ----------------
static inline int grid_for(uint64_t n, int block) {
  uint64_t g = (n + block - 1) / block;
  if (g > 256ull * 16) g = 256ull * 16;
  if (g < 1) g = 1;
  return (int)g;
}

template <int R, int W, bool PRE, bool WIN = false, bool XC = false>
static hipError_t launch_lane_variant(const LaneArgs& A, unsigned grid, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)sweep_lane_kernel<R, W, PRE, Lse, WIN, XC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((sweep_lane_kernel<R, W, PRE, Lse, WIN, XC>), dim3(grid), dim3(64), lds, stream, A);
  return hipGetLastError();
}

hipError_t launch_lane_sweep(const LaneArgs& A0, const LatticeSet::LaneClass& lc, hipStream_t stream, bool fused) {
  LaneArgs A = A0;
  A.first_group = lc.first;
  A.lds_rows = lc.max_states;
  size_t lds = (size_t)lc.max_states * 64 * sizeof(double);
  if (fused) {
    // the posteriors leave through a stage of one tile behind the value rows (sweep_lane_kernel<.., XC>)
    if (!A.pre_weights || !A.xc || !A.xc_tile_base || !A.xc_t_pos) return hipErrorInvalidValue;
    lds += (size_t)LANE_FUSED_TILE * sizeof(double);
    if (lc.windowed) {
      if (!A.spill) return hipErrorInvalidValue;
      return launch_lane_variant<4, 2, true, true, true>(A, lc.count, lds, stream);
    }
    return launch_lane_variant<4, 2, true, false, true>(A, lc.count, lds, stream);
  }
  // ring depths per form, measured (tagging cascade x400, windowed: <4,2> 495 us, <4,1> 511, <3,1> 534, <3,2> 643; streaming
  // with weights in lattice order: two chunks ahead is enough, and the smaller ring leaves more registers / less code)
  if (lc.windowed) {
    if (!A.spill) return hipErrorInvalidValue;
    if (A.pre_weights) return launch_lane_variant<4, 2, true, true>(A, lc.count, lds, stream);
    return launch_lane_variant<4, 2, false, true>(A, lc.count, lds, stream);
  }
  if (A.pre_weights) return launch_lane_variant<3, 1, true>(A, lc.count, lds, stream);
  return launch_lane_variant<4, 2, false>(A, lc.count, lds, stream);
}

template <bool RING>
static void launch_wave_variant(const WaveArgs& A, uint32_t count, size_t lds, bool big_lds, hipStream_t stream) {
  const bool gw = A.bwd_arc != nullptr, xd = A.xc_idx != nullptr;
  // (the attribute is per kernel and per device: set where a launch needs more than the kernel has been given there)
  static size_t lds_set[64][2][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const bool set_lds = big_lds && lds > lds_set[dev][gw][xd];
  if (set_lds) lds_set[dev][gw][xd] = lds;
#define WAVE_LAUNCH(GW, XD)                                                                                                       \
  {                                                                                                                               \
    if (set_lds) (void)hipFuncSetAttribute((const void*)sweep_wave_kernel<RING, GW, XD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((sweep_wave_kernel<RING, GW, XD>), dim3(count), dim3(64), lds, stream, A);                                 \
  }
  if (gw && xd) WAVE_LAUNCH(true, true)
  else if (gw) WAVE_LAUNCH(true, false)
  else if (xd) WAVE_LAUNCH(false, true)
  else WAVE_LAUNCH(false, false)
#undef WAVE_LAUNCH
}
hipError_t launch_wave_sweep(const WaveArgs& A0, const LatticeSet::WaveClass& wc, hipStream_t stream) {
  WaveArgs A = A0;
  A.first = wc.first;
  A.max_width = wc.max_width;
  if (A.bwd_arc && (!A.logw || !A.n_arcs)) return hipErrorInvalidValue;
  if (A.xc_idx && !A.xc) return hipErrorInvalidValue;
  if (wc.ring) {
    if (!A.spill) return hipErrorInvalidValue;
    A.max_states = wc.ring;
    const size_t lds = ((size_t)wc.ring + 3 * (size_t)wc.max_width) * sizeof(double);
    launch_wave_variant<true>(A, wc.count, lds, false, stream);
    return hipGetLastError();
  }
  A.max_states = wc.max_states;
  const size_t lds = ((size_t)wc.max_states + 2 * (size_t)wc.max_width) * sizeof(double);
  launch_wave_variant<false>(A, wc.count, lds, lds > 64 * 1024, stream);
  return hipGetLastError();
}

hipError_t launch_sweep(const SweepArgs& A0, const LatticeSet::LaunchClass& lc, hipStream_t stream) {
  SweepArgs A = A0;
  A.first_bundle = lc.first;
  if (lc.serial) {
    int block = 64;
    int grid = (int)((lc.count + block - 1) / block);
    hipLaunchKernelGGL(sweep_serial_kernel, dim3(grid), dim3(block), 0, stream, A, lc.count);
    return hipGetLastError();
  }
  size_t lds = (size_t)lc.max_states * sizeof(double);
  if (lc.max_states == 0) {
    hipLaunchKernelGGL((sweep_bundle_kernel<1024, false>), dim3(lc.count), dim3(1024), 0, stream, A);
  } else if (lc.block == 64) {
    hipLaunchKernelGGL((sweep_bundle_kernel<64, true>), dim3(lc.count), dim3(64), lds, stream, A);
  } else if (lc.block == 256) {
    hipLaunchKernelGGL((sweep_bundle_kernel<256, true>), dim3(lc.count), dim3(256), lds, stream, A);
  } else {
    (void)hipFuncSetAttribute((const void*)sweep_bundle_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((sweep_bundle_kernel<1024, true>), dim3(lc.count), dim3(1024), lds, stream, A);
  }
  return hipGetLastError();
}

#define TRANS_SET_LDS(K, BYTES) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES)
static void trans_lds_attr() {
  // per device (hipFuncSetAttribute is): a process that drives a second GPU sets the kernels up there too
  static unsigned char done_dev[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
  unsigned char& done = done_dev[dev];
  if (__atomic_load_n(&done, __ATOMIC_ACQUIRE)) return;
  const int lds = (int)(TRANS_TILE > TRANS_BUCKET ? TRANS_TILE : TRANS_BUCKET) * 8;
  const int lds_rl = lds + TRANS_RUN_LDS;
  TRANS_SET_LDS((trans_w_bucket_kernel<false, false>), lds);
  TRANS_SET_LDS((trans_w_bucket_kernel<true, false>), lds);
  TRANS_SET_LDS((trans_w_bucket_kernel<true, true>), lds_rl);
  TRANS_SET_LDS((trans_w_tile_kernel<false, false>), lds);
  TRANS_SET_LDS((trans_w_tile_kernel<true, false>), lds_rl);
  TRANS_SET_LDS((trans_w_tile_kernel<false, true>), lds);
  TRANS_SET_LDS((trans_c_tile_kernel<false, false>), lds);
  TRANS_SET_LDS((trans_c_tile_kernel<true, false>), lds);
  TRANS_SET_LDS((trans_c_tile_kernel<true, true>), lds_rl);
  TRANS_SET_LDS((trans_c_bucket_kernel<false, false>), lds);
  TRANS_SET_LDS((trans_c_bucket_kernel<true, false>), lds_rl);
  TRANS_SET_LDS((trans_c_bucket_kernel<false, true>), lds);
  {  // the half-size bucket kernels (their static LDS takes them past the default limit as well)
    constexpr int KH = TRANS_KB / 2;
    TRANS_SET_LDS((trans_w_bucket_kernel<false, false, KH>), lds);
    TRANS_SET_LDS((trans_w_bucket_kernel<true, false, KH>), lds);
    TRANS_SET_LDS((trans_w_bucket_kernel<true, true, KH>), lds_rl);
    TRANS_SET_LDS((trans_c_bucket_kernel<false, false, KH>), lds);
    TRANS_SET_LDS((trans_c_bucket_kernel<true, false, KH>), lds_rl);
    TRANS_SET_LDS((trans_c_bucket_kernel<false, true, KH>), lds);
  }
  __atomic_store_n(&done, (unsigned char)1, __ATOMIC_RELEASE);
}
hipError_t launch_trans_w_bucket_range(const TransArgs& T0, uint32_t first, uint32_t count, hipStream_t stream) {
  trans_lds_attr();
  if (!T0.n_buckets || !count) return hipSuccess;
  TransArgs T = T0;
  T.bucket_first = first;
  T.bucket_count = count;
  const dim3 g8((count + 7) / 8 * 8);
  if (T.bucket == TRANS_BUCKET / 2) {  // half-size buckets (the fused-lane layout): 64 KB of LDS, two workgroups to a CU
    constexpr int KH = TRANS_KB / 2;
    if (!(T.scatter & 1u))
      hipLaunchKernelGGL((trans_w_bucket_kernel<false, false, KH>), dim3(count), dim3(1024), KH * 1024 * 8, stream, T);
    else if (T.use_runs)
      hipLaunchKernelGGL((trans_w_bucket_kernel<true, true, KH>), g8, dim3(1024), KH * 1024 * 8 + TRANS_RUN_LDS, stream, T);
    else
      hipLaunchKernelGGL((trans_w_bucket_kernel<true, false, KH>), g8, dim3(1024), KH * 1024 * 8, stream, T);
    return hipGetLastError();
  }
  if (T.bucket != TRANS_BUCKET) return hipErrorInvalidValue;
  if (!(T.scatter & 1u))
    hipLaunchKernelGGL((trans_w_bucket_kernel<false, false>), dim3(count), dim3(1024), TRANS_BUCKET * 8, stream, T);
  else if (T.use_runs)
    hipLaunchKernelGGL((trans_w_bucket_kernel<true, true>), g8, dim3(1024), TRANS_BUCKET * 8 + TRANS_RUN_LDS, stream, T);
  else
    hipLaunchKernelGGL((trans_w_bucket_kernel<true, false>), g8, dim3(1024), TRANS_BUCKET * 8, stream, T);
  return hipGetLastError();
}
hipError_t launch_trans_w_bucket(const TransArgs& T, hipStream_t stream) { return launch_trans_w_bucket_range(T, 0, T.n_buckets, stream); }
hipError_t launch_trans_w_tiles(const TransArgs& T0, uint32_t tile_first, uint32_t tile_count, hipStream_t stream) {
  trans_lds_attr();
  if (!T0.n_buckets || !tile_count) return hipSuccess;
  TransArgs T = T0;
  T.tile_first = tile_first;
  T.tile_count = tile_count;
  const dim3 g8((tile_count + 7) / 8 * 8);
  if (T.tile == LANE_FUSED_TILE) {  // the fused-lane layout's small tiles
    if (T.scatter & 1u)
      hipLaunchKernelGGL((trans_w_tile_small_kernel<true>), g8, dim3(TRANS_SMALL_THREADS), 0, stream, T);
    else
      hipLaunchKernelGGL((trans_w_tile_small_kernel<false>), g8, dim3(TRANS_SMALL_THREADS), 0, stream, T);
    return hipGetLastError();
  }
  if (T.tile != TRANS_TILE && T.tile != TILE_SWEEP_TILE) return hipErrorInvalidValue;
  if (T.scatter & 1u)
    hipLaunchKernelGGL((trans_w_tile_kernel<false, true>), g8, dim3(1024), TRANS_TILE * 8, stream, T);
  else if (T.use_runs)
    hipLaunchKernelGGL((trans_w_tile_kernel<true, false>), g8, dim3(1024), TRANS_TILE * 8 + TRANS_RUN_LDS, stream, T);
  else
    hipLaunchKernelGGL((trans_w_tile_kernel<false, false>), g8, dim3(1024), TRANS_TILE * 8, stream, T);
  return hipGetLastError();
}
hipError_t launch_trans_c_tiles(const TransArgs& T0, uint32_t tile_first, uint32_t tile_count, hipStream_t stream) {
  trans_lds_attr();
  if (!T0.n_buckets || !tile_count) return hipSuccess;
  TransArgs T = T0;
  T.tile_first = tile_first;
  T.tile_count = tile_count;
  const dim3 g8((tile_count + 7) / 8 * 8);
  if (T.tile == LANE_FUSED_TILE) {
    if (T.scatter & 2u) return hipErrorInvalidValue;  // (trans_args keeps the bit off under this layout)
    hipLaunchKernelGGL(trans_c_tile_small_kernel, dim3(tile_count), dim3(TRANS_SMALL_THREADS), 0, stream, T);
    return hipGetLastError();
  }
  if (T.tile != TRANS_TILE && T.tile != TILE_SWEEP_TILE) return hipErrorInvalidValue;
  if (!(T.scatter & 2u))
    hipLaunchKernelGGL((trans_c_tile_kernel<false, false>), dim3(tile_count), dim3(1024), TRANS_TILE * 8, stream, T);
  else if (T.use_runs)
    hipLaunchKernelGGL((trans_c_tile_kernel<true, true>), g8, dim3(1024), TRANS_TILE * 8 + TRANS_RUN_LDS, stream, T);
  else
    hipLaunchKernelGGL((trans_c_tile_kernel<true, false>), g8, dim3(1024), TRANS_TILE * 8, stream, T);
  return hipGetLastError();
}
hipError_t launch_zero_list(double* p, const uint32_t* idx, uint32_t n, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(zero_list_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, idx, n);
  return hipGetLastError();
}
hipError_t launch_trans_c_bucket_range(const TransArgs& T0, uint32_t first, uint32_t count, hipStream_t stream) {
  trans_lds_attr();
  if (!T0.n_buckets || !count) return hipSuccess;
  TransArgs T = T0;
  T.bucket_first = first;
  T.bucket_count = count;
  const dim3 g8((count + 7) / 8 * 8);
  if (T.bucket == TRANS_BUCKET / 2) {
    constexpr int KH = TRANS_KB / 2;
    if (T.scatter & 2u)
      hipLaunchKernelGGL((trans_c_bucket_kernel<false, true, KH>), g8, dim3(1024), KH * 1024 * 8, stream, T);
    else if (T.use_runs)
      hipLaunchKernelGGL((trans_c_bucket_kernel<true, false, KH>), g8, dim3(1024), KH * 1024 * 8 + TRANS_RUN_LDS, stream, T);
    else
      hipLaunchKernelGGL((trans_c_bucket_kernel<false, false, KH>), g8, dim3(1024), KH * 1024 * 8, stream, T);
    return hipGetLastError();
  }
  if (T.bucket != TRANS_BUCKET) return hipErrorInvalidValue;
  if (T.scatter & 2u)
    hipLaunchKernelGGL((trans_c_bucket_kernel<false, true>), g8, dim3(1024), TRANS_BUCKET * 8, stream, T);
  else if (T.use_runs)
    hipLaunchKernelGGL((trans_c_bucket_kernel<true, false>), g8, dim3(1024), TRANS_BUCKET * 8 + TRANS_RUN_LDS, stream, T);
  else
    hipLaunchKernelGGL((trans_c_bucket_kernel<false, false>), g8, dim3(1024), TRANS_BUCKET * 8, stream, T);
  return hipGetLastError();
}
hipError_t launch_trans_c_bucket(const TransArgs& T, const uint32_t* split_arcs, uint32_t n_split, hipStream_t stream) {
  if (!T.n_buckets) return hipSuccess;
  hipError_t e = launch_zero_list(T.counts, split_arcs, n_split, stream);
  if (e != hipSuccess) return e;
  return launch_trans_c_bucket_range(T, 0, T.n_buckets, stream);
}
__global__ void gather_idx_kernel(double* small, const double* src, const uint32_t* idx, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) small[k] = src[idx[k]];
}
__global__ void scatter_idx_kernel(double* dst, const double* small, const uint32_t* idx, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) dst[idx[k]] = small[k];
}
hipError_t launch_gather_idx(double* small, const double* src, const uint32_t* idx, uint32_t n, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(gather_idx_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, small, src, idx, n);
  return hipGetLastError();
}
hipError_t launch_scatter_idx(double* dst, const double* small, const uint32_t* idx, uint32_t n, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(scatter_idx_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, small, idx, n);
  return hipGetLastError();
}

hipError_t launch_count_reduce(const ReduceArgs& R, hipStream_t stream) {
  if (!R.n_arcs) return hipSuccess;
  hipLaunchKernelGGL(count_reduce_kernel, dim3(grid_for(R.n_arcs, 256)), dim3(256), 0, stream, R);
  if (R.n_hot_chunks)
    hipLaunchKernelGGL(count_reduce_hot_kernel, dim3((unsigned)R.n_hot_chunks), dim3(256), 0, stream, R);
  return hipGetLastError();
}

hipError_t launch_scalars(const double* pair_logprob, const double* pair_w, uint64_t n_pairs, double* partial,
                          double* scalars, hipStream_t s) {
  hipLaunchKernelGGL(scalars_partial_kernel, dim3(SCALAR_BLOCKS), dim3(256), 0, s, pair_logprob, pair_w, n_pairs, partial);
  hipLaunchKernelGGL(scalars_final_kernel, dim3(1), dim3(64), 0, s, partial, scalars);
  return hipGetLastError();
}
hipError_t launch_add(double* dst, const double* src, uint64_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(add_f64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, dst, src, n);
  return hipGetLastError();
}
hipError_t launch_fill(double* p, double v, uint64_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(fill_f64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, v, n);
  return hipGetLastError();
}
hipError_t launch_mstep(const MstepArgs& M, int use_counts, hipStream_t s) {
  if (!M.n) {  // a model without parameters: nothing moves, but the caller waits for THIS M-step's sequence number in the mailbox
    hipLaunchKernelGGL(mstep_max_final_kernel, dim3(1), dim3(256), 0, s, M.max_partial, (uint64_t)MSTEP_GRID, M.max_change_bits, M.box, M.box_seq);
    return hipGetLastError();
  }
  if (M.window_span && !(M.n_ties && M.tie_of)) {
    if (M.lw_src)
      hipLaunchKernelGGL(mstep_window_kernel<true>, dim3((unsigned)((M.n + 255) / 256)), dim3(256), 0, s, M, use_counts, M.window_span);
    else
      hipLaunchKernelGGL(mstep_window_kernel<false>, dim3((unsigned)((M.n + 255) / 256)), dim3(256), 0, s, M, use_counts, M.window_span);
  } else {
    if (M.n_ties && M.tie_of) {
      hipError_t e = hipMemsetAsync(M.tie_tab, 0, 4 * M.n_ties * sizeof(double), s);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(mstep_tie_sums_kernel, dim3(grid_for(M.n_groups, 256)), dim3(256), 0, s, M, use_counts);
      hipLaunchKernelGGL(mstep_tie_accum_kernel, dim3(grid_for(M.n, 256)), dim3(256), 0, s, M, use_counts);
      hipLaunchKernelGGL(mstep_tie_weight_kernel, dim3((unsigned)((M.n_ties + 255) / 256)), dim3(256), 0, s, M);
    }
    if (M.n_groups) hipLaunchKernelGGL(mstep_group_sum_kernel, dim3(grid_for(M.n_groups, 256)), dim3(256), 0, s, M, use_counts);
    if (M.n_big) hipLaunchKernelGGL(mstep_big_group_kernel, dim3((unsigned)M.n_big), dim3(256), 0, s, M, use_counts);
    hipLaunchKernelGGL(mstep_normalize_kernel, dim3(MSTEP_GRID), dim3(256), 0, s, M, use_counts);
  }
  hipLaunchKernelGGL(mstep_max_final_kernel, dim3(1), dim3(256), 0, s, M.max_partial, (uint64_t)MSTEP_GRID,
                     M.max_change_bits, M.box, M.box_seq);
  return hipGetLastError();
}
hipError_t launch_mstep_window_range(const MstepArgs& M0, int use_counts, uint32_t block_first, uint32_t n_blocks, hipStream_t s) {
  if (!n_blocks) return hipSuccess;
  MstepArgs M = M0;
  M.block_first = block_first;
  if (M.lw_src)
    hipLaunchKernelGGL(mstep_window_kernel<true>, dim3(n_blocks), dim3(256), 0, s, M, use_counts, M.window_span);
  else
    hipLaunchKernelGGL(mstep_window_kernel<false>, dim3(n_blocks), dim3(256), 0, s, M, use_counts, M.window_span);
  return hipGetLastError();
}
hipError_t launch_mstep_max_final(const MstepArgs& M, hipStream_t s) {
  hipLaunchKernelGGL(mstep_max_final_kernel, dim3(1), dim3(256), 0, s, M.max_partial, (uint64_t)MSTEP_GRID, M.max_change_bits, M.box,
                     M.box_seq);
  return hipGetLastError();
}
hipError_t launch_overrelax(double* logw, const double* old_logw, double* em_logw, const uint32_t* group, double rate,
                            uint64_t n, hipStream_t s) {
  hipLaunchKernelGGL(overrelax_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, logw, old_logw, em_logw, group, rate, n);
  return hipGetLastError();
}
hipError_t launch_max_change(const double* logw, const double* old_logw, const uint32_t* group,
                             unsigned long long* bits, uint64_t n, hipStream_t s) {
  hipLaunchKernelGGL(max_change_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, logw, old_logw, group, bits, n);
  return hipGetLastError();
}
hipError_t launch_chain_update(double* arc_logw, const uint32_t* arc_chain, const uint64_t* chain_off,
                               const uint64_t* chain_param, const double* param_logw, uint64_t n_arcs, hipStream_t s) {
  hipLaunchKernelGGL(chain_update_kernel, dim3(grid_for(n_arcs, 256)), dim3(256), 0, s, arc_logw, arc_chain, chain_off,
                     chain_param, param_logw, n_arcs);
  return hipGetLastError();
}
hipError_t launch_chain_scatter(double* param_counts, const double* arc_counts, double arc_prior, const double* arc_prior_w,
                                const uint32_t* arc_chain, const uint64_t* chain_off, const uint64_t* chain_param,
                                const uint32_t* param_group, uint64_t n_arcs, hipStream_t s) {
  hipLaunchKernelGGL(chain_scatter_kernel, dim3(grid_for(n_arcs, 256)), dim3(256), 0, s, param_counts, arc_counts,
                     arc_prior, arc_prior_w, arc_chain, chain_off, chain_param, param_group, n_arcs);
  return hipGetLastError();
}
__global__ void counts_to_logw_kernel(double* logw, const double* counts, const double* prior, const uint32_t* group,
                                      uint64_t n) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
    if (group[k] != 0u) {
      const double v = counts[k] + (prior ? prior[k] : 0.0);
      logw[k] = v > 0.0 ? log(v) : NEG_INF;
    }
}
hipError_t launch_counts_to_logw(double* logw, const double* counts, const double* prior, const uint32_t* group, uint64_t n,
                                 hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(counts_to_logw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, logw, counts, prior, group, n);
  return hipGetLastError();
}

}  // namespace carmel_hip
