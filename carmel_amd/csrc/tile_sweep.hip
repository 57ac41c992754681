// tile_sweep.hip — the E-step's middle for corpora of small plain lane lattices (LatticeSet::tile_sweep, lattice.hpp): the
// weights' way into lattice order, the lane sweeps (derivations.h:400-449, graph.h:391-402) and the posteriors' way out as
// ONE persistent kernel per tile of 8192 lattice positions, out of a workgroup's LDS.  kernels.hip's trans_w_tile /
// sweep_lane / trans_c_tile do the same work through wcache / post in HBM (and still do, for every other corpus, and
// behind CARMEL_HIP_TILE_SWEEP_KERNEL=0): same results.  DESIGN.md 4a "The tile sweep" has the measurements.
#include "kernels.hpp"
#include "sweep_math.hpp"
#include <algorithm>

namespace carmel_hip {

// ---------------- tile sweep: weights in, lane sweeps, posteriors out -- one kernel per tile of small lattices ----------------
// (LatticeSet::tile_sweep.)  Where every lattice of the corpus is a plain lane lattice of at most TILE_SWEEP_ROWS arcs, the lane
// groups are laid out so that none straddles a tile of TILE_SWEEP_TILE positions, and the three middle kernels of the E-step
// become one: the workgroup of a tile places the tile's weights (its stretch of X) in LDS at their lane positions -- what
// trans_w_tile writes to wcache --, its wavefronts sweep the tile's groups out of LDS (the log posterior of an arc replaces
// its weight; the whole workgroup exponentiates the tile in place afterwards), and the tile's items leave for XC where
// trans_c_tile sends them.  Per lattice arc the E-step no longer writes and re-reads wcache (16 B + the sweep's two reads of
// it) nor post (16 B): what is left between the two bucket passes is X in, the position table, the destinations, XC out --
// and one packed record per position for the tiles that hold a lattice that is not a single path.  LDS of a tile: its
// positions' weights / posteriors (8 B), their records (4 B), and the forward / backward values of its groups
// (LaneGroup::spill_row = a group's first row of 64).
// A wavefront that sweeps out of LDS has nobody to hide behind (a tile has two to sixteen groups, a CU one tile): what it
// costs is what it issues.  So the topology is in LDS with the weights (no memory round trip inside a sweep), what is static
// about a row is decided when the records are packed (pack_tile_records_kernel): whether the arc's other end is the state
// just finished (its value is in a register: no column read) and whether the arc is its state's only one (the state's
// value is one addition: no log-sum-exp); and a group of single paths is swept without records at all (tile_chain_sweep).
#define TS_SRC(x) ((x) & 0xffu)
#define TS_POS(x) (((x) >> 8) & 63u)
#define TS_FV 0x4000u
#define TS_FL 0x8000u
#define TS_DST(x) (((x) >> 16) & 0xffu)
#define TS_BV 0x1000000u
#define TS_BL 0x2000000u
#define TS_FCHAIN 0x4000000u   // forward: the source is the state finished last (or the row is padding)
#define TS_FEASY 0x8000000u    // forward: the only in-arc of its state (or padding)
#define TS_BCHAIN 0x10000000u  // backward: the destination is the state finished last (or padding)
#define TS_BEASY 0x20000000u   // backward: the only out-arc of its state (or padding)
// one thread per lane of a group walks the lane's rows: the two records of a row in one word + the static properties above;
// chain[gi] bit 0 = every lattice of the group is a single path (states 0 .. len in a row: forward row k is the arc k -> k + 1, whose
// backward row is maxlen - 1 - k): tile_chain_sweep needs no records at all
__global__ void pack_tile_records_kernel(const LaneGroup* __restrict__ groups, uint32_t n_groups, const uint32_t* __restrict__ lane_nstates,
                                         const uint32_t* __restrict__ fwdx, const uint32_t* __restrict__ bwd, uint32_t* __restrict__ out,
                                         uint32_t* __restrict__ chain) {
  const uint32_t gi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (gi >= n_groups) return;
  const LaneGroup g = groups[gi];
  const uint32_t S = lane < g.n_lanes ? lane_nstates[g.pair_base + lane] : 0u;
  uint32_t d = 1, s = S >= 2 ? S - 2 : 0u;
  bool fresh_f = true, fresh_b = true;
  const uint32_t len = S ? S - 1 : 0u;
  bool path = true;
  for (uint32_t k = 0; k < g.maxlen; ++k) {
    const size_t p = g.stream_base + (size_t)k * 64 + lane;
    const uint32_t f = fwdx[p], b = bwd[p];
    // (the backward stream of a lane ends with the group's last row: its padding comes first)
    if (k < len)
      path = path && (f & LANE_VALID) && (f & LANE_LAST) && (f & LANE_STATE_MASK) == k && ((f >> LANE_POS_SHIFT) & LANE_POS_MAX) == g.maxlen - 1 - k;
    else
      path = path && !(f & (LANE_VALID | LANE_LAST));
    if (k >= g.maxlen - len)
      path = path && (b & LANE_VALID) && (b & LANE_LAST) && (b & LANE_STATE_MASK) == g.maxlen - k;
    else
      path = path && !(b & (LANE_VALID | LANE_LAST));
    uint32_t x = (f & 0xffu) | (((f >> LANE_POS_SHIFT) & 63u) << 8) | ((f & LANE_VALID) ? TS_FV : 0u) | ((f & LANE_LAST) ? TS_FL : 0u) |
                 ((b & 0xffu) << 16) | ((b & LANE_VALID) ? TS_BV : 0u) | ((b & LANE_LAST) ? TS_BL : 0u);
    const bool f_pad = !(f & (LANE_VALID | LANE_LAST)), b_pad = !(b & (LANE_VALID | LANE_LAST));
    if (f_pad || (f & LANE_STATE_MASK) + 1 == d) x |= TS_FCHAIN;
    if (f_pad || ((f & LANE_VALID) && (f & LANE_LAST) && fresh_f)) x |= TS_FEASY;
    if (b_pad || (b & LANE_STATE_MASK) == s + 1) x |= TS_BCHAIN;
    if (b_pad || ((b & LANE_VALID) && (b & LANE_LAST) && fresh_b)) x |= TS_BEASY;
    if (!f_pad) fresh_f = false;
    if (f & LANE_LAST) {
      ++d;
      fresh_f = true;
    }
    if (!b_pad) fresh_b = false;
    if (b & LANE_LAST) {
      if (s > 0) --s;
      fresh_b = true;
    }
    out[p] = x;
  }
  // ... and how many leading backward rows are padding in some lane of the group (its lattices are nearly of one length)
  uint32_t pad = lane < g.n_lanes ? g.maxlen - len : 0u;
  for (int o = 32; o > 0; o >>= 1) pad = max(pad, (uint32_t)__shfl_xor((int)pad, o, 64));
  const bool all_paths = __all(path);
  if (lane == 0) chain[gi] = (all_paths ? 1u : 0u) | (pad << 8);
}
// tile_chain[t] = every group of tile t is a group of single paths: the tile's records are never looked at (nor fetched)
__global__ void tile_chain_kernel(const uint32_t* __restrict__ tile_group, const uint32_t* __restrict__ chain, uint32_t n_tiles,
                                  uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  uint32_t all = 1u;
  for (uint32_t g = tile_group[t]; g < tile_group[t + 1]; ++g) all &= chain[g];
  out[t] = all & 1u;
}
struct TileLane {  // a lane's lattice: states, ln(pair weight), corpus pair (requested a group ahead of its sweep)
  uint32_t S, pair;
  double lwt;
};
// (group gi's lanes are entries gi * 64 .. of the per-lane arrays -- LaneGroup::pair_base, both builders: the request does
// not have to wait for the group's descriptor; lanes past n_lanes hold S = 0)
__device__ __forceinline__ TileLane tile_lane(const LaneArgs& A, uint32_t gi, const int lane) {
  TileLane L;
  const size_t k = (size_t)gi * 64 + lane;
  L.S = A.lane_nstates[k];
  L.lwt = A.lane_logw[k];
  L.pair = A.lane_pair[k];
  return L;
}
// one group: records, weights / posteriors in the lane's columns of LDS rows (recl, rows), values in its column `col`: the
// arithmetic of sweep_lane_kernel, operation for operation (an "easy" row is Lse's own result for a single term).
// Returns ln p(pair): the caller stores it -- a store inside the sweep would sit in the wavefront's in-order queue of vector
// memory operations in front of whatever the wavefront waits for next.
__device__ __forceinline__ double tile_group_sweep(const LaneGroup& g, const TileLane& L, const int lane, double* col, double* rows,
                                                   const uint32_t* recl) {
  constexpr int U = (int)LANE_CHUNK;
  const bool active = (uint32_t)lane < g.n_lanes;
  const uint32_t S = active ? L.S : 0u;
  const double lwt = L.lwt;
  const uint32_t maxlen = g.maxlen;
  // ---------- forward ----------
  if (active) col[0] = 0.0;
  {
    Lse acc;
    acc.init();
    uint32_t d = 1;
    double prev = 0.0;
    uint32_t x1[U], x2[U];  // records of the next chunk and of the one after
    double w1[U];           // weights of the next chunk
#pragma unroll
    for (int u = 0; u < U; ++u) x1[u] = recl[u * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) x2[u] = recl[((U < maxlen ? U : 0) + u) * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) w1[u] = rows[TS_POS(x1[u]) * 64];
    for (uint32_t kb = 0; kb < maxlen; kb += U) {
      uint32_t x[U];
      double w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x[u] = x1[u];
        w[u] = w1[u];
        x1[u] = x2[u];
      }
      const uint32_t k2 = kb + 2 * U < maxlen ? kb + 2 * U : kb;  // (past the end: any row of the group)
#pragma unroll
      for (int u = 0; u < U; ++u) x2[u] = recl[(k2 + u) * 64];
#pragma unroll
      for (int u = 0; u < U; ++u) w1[u] = rows[TS_POS(x1[u]) * 64];
      if (__all(((x[0] & x[1] & x[2] & x[3]) & (TS_FCHAIN | TS_FEASY)) == (TS_FCHAIN | TS_FEASY))) {
        // every row of the chunk, in every lane: the only in-arc of its state, out of the state before it (or padding)
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (x[u] & TS_FV) {
            prev += w[u];
            col[d * 64] = prev;
            ++d;
          }
        continue;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t xx = x[u];
        double a_src = prev;
        if (!__all(xx & TS_FCHAIN)) {
          const double a = col[TS_SRC(xx) * 64];
          a_src = (xx & TS_FCHAIN) ? prev : a;
        }
        if (__all(xx & TS_FEASY)) {
          if (xx & TS_FV) {
            prev = a_src + w[u];
            col[d * 64] = prev;
            ++d;
          }
        } else {
          acc.add((xx & TS_FV) ? a_src + w[u] : NEG_INF);
          if (xx & TS_FL) {
            prev = acc.value();
            col[d * 64] = prev;
            ++d;
            acc.init();
          }
        }
      }
    }
  }
  // ---------- ln p(pair), beta at the goal ----------
  double next = NEG_INF, lp_out = NEG_INF;
  if (active) {
    const double lp = col[(S - 1) * 64];
    lp_out = lp;
    next = (lp == NEG_INF) ? NEG_INF : lwt - lp;
    col[(S - 1) * 64] = next;
  }
  // ---------- backward + posteriors ----------
  {
    Lse acc;
    acc.init();
    uint32_t s = S >= 2 ? S - 2 : 0u;
    uint32_t x1[U];
    double w1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x1[u] = recl[u * 64];
      w1[u] = rows[u * 64];
    }
    for (uint32_t kb = 0; kb < maxlen; kb += U) {
      double t[U], al[U];
      uint32_t x[U];
      double w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x[u] = x1[u];
        w[u] = w1[u];
      }
      const uint32_t kn = kb + U < maxlen ? kb + U : kb;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x1[u] = recl[(kn + u) * 64];
        w1[u] = rows[(kn + u) * 64];
      }
      if (__all(((x[0] & x[1] & x[2] & x[3]) & (TS_BCHAIN | TS_BEASY)) == (TS_BCHAIN | TS_BEASY))) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          al[u] = col[s * 64];
          t[u] = NEG_INF;
          if (x[u] & TS_BV) {
            next += w[u];
            t[u] = next;
            col[s * 64] = next;
            if (s > 0) --s;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t xx = x[u];
          al[u] = col[s * 64];  // alpha[s]: requested before this row can replace it with beta[s]; not waited for until the chunk's end
          double b_dst = next;
          if (!__all(xx & TS_BCHAIN)) {
            const double b = col[TS_DST(xx) * 64];
            b_dst = (xx & TS_BCHAIN) ? next : b;
          }
          t[u] = (xx & TS_BV) ? w[u] + b_dst : NEG_INF;
          if (__all(xx & TS_BEASY)) {
            if (xx & TS_BV) {
              next = t[u];
              col[s * 64] = next;
              if (s > 0) --s;
            }
          } else {
            acc.add(t[u]);
            if (xx & TS_BL) {
              next = acc.value();
              col[s * 64] = next;
              acc.init();
              if (s > 0) --s;
            }
          }
        }
      }
      // the rows' log posteriors; the exponentials are left to the whole workgroup (the tile's way out)
#pragma unroll
      for (int u = 0; u < U; ++u) rows[(kb + u) * 64] = (S >= 2 ? al[u] : NEG_INF) + t[u];
    }
  }
  return lp_out;
}
// a group of single paths (pack_tile_records_kernel's chain flag): the same additions in the same order, and nothing else --
// no records, no branches, no validity tests.  The group's lattices are nearly of one length (lanes are sorted by length);
// the few padding rows a lane has are given the weight 0, so that every lane walks all maxlen rows: past its lattice's end
// the forward chain keeps adding 0 (and writes the unchanged value to column rows nobody reads: the layout gives a group
// maxlen + 1 of them), before its first backward row the backward chain does.  (What the zeros can change is the sign of a
// zero: -0.0 + 0.0 is +0.0.)
__device__ __forceinline__ double tile_chain_sweep(const LaneGroup& g, const TileLane& L, uint32_t max_pad, const int lane, double* col,
                                                   double* rows) {
  constexpr int U = (int)LANE_CHUNK;
  const bool active = (uint32_t)lane < g.n_lanes;
  const uint32_t S = active ? L.S : 0u;
  const uint32_t len = S ? S - 1 : 0u;
  const uint32_t maxlen = g.maxlen;
  const uint32_t off = maxlen - len;  // backward rows before this one are padding
  for (uint32_t k = 0; k < max_pad; ++k)
    if (k < off) rows[k * 64] = 0.0;
  col[0] = 0.0;
  double prev = 0.0;
  // Two chunks per round, each chunk's LDS reads requested while the other is worked on, into registers of its own: every
  // wait inside the round is for a counted number of operations.
  {
    const double* wr = rows + (size_t)(maxlen - 1) * 64;  // the arc k -> k + 1 lies at backward row maxlen - 1 - k
    double* cw = col + 64;
    double wa[U], wb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wa[u] = wr[-u * 64];
    for (uint32_t kb = 0; kb < maxlen; kb += 2 * U) {
      const bool two = kb + U < maxlen;  // (maxlen is a multiple of U, not of 2 U)
      if (two) {
#pragma unroll
        for (int u = 0; u < U; ++u) wb[u] = wr[-(U + u) * 64];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        prev += wa[u];
        cw[u * 64] = prev;
      }
      if (kb + 2 * U < maxlen) {
#pragma unroll
        for (int u = 0; u < U; ++u) wa[u] = wr[-(2 * U + u) * 64];
      }
      if (two) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          prev += wb[u];
          cw[(U + u) * 64] = prev;
        }
      }
      wr -= 2 * U * 64;
      cw += 2 * U * 64;
    }
  }
  const double lp = prev;  // = col[(S - 1) * 64] (and every later row of the column)
  double next = (!active || lp == NEG_INF) ? NEG_INF : L.lwt - lp;
  {
    // backward row k is the arc out of state maxlen - 1 - k; the rows before a lane's first are its padding
    double* wr = rows;
    double* ca = col + (size_t)(maxlen - 1) * 64;
    double wa[U], aa[U], wb[U], ab[U];  // the weights, and the forward values of the states the chunk will overwrite
#pragma unroll
    for (int u = 0; u < U; ++u) {
      wa[u] = wr[u * 64];
      aa[u] = ca[-u * 64];
    }
    for (uint32_t kb = 0; kb < maxlen; kb += 2 * U) {
      const bool two = kb + U < maxlen;
      if (two) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          wb[u] = wr[(U + u) * 64];
          ab[u] = ca[-(U + u) * 64];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        next = wa[u] + next;
        ca[-u * 64] = next;
        wr[u * 64] = aa[u] + next;
      }
      if (kb + 2 * U < maxlen) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          wa[u] = wr[(2 * U + u) * 64];
          aa[u] = ca[-(2 * U + u) * 64];
        }
      }
      if (two) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          next = wb[u] + next;
          ca[-(U + u) * 64] = next;
          wr[(U + u) * 64] = ab[u] + next;
        }
      }
      wr += 2 * U * 64;
      ca -= 2 * U * 64;
    }
  }
  return lp;
}
// The kernel is persistent and its wavefronts are specialised.  A workgroup per CU walks its share of the tiles; half of its
// wavefronts SWEEP (a group each at a time; they also fetch the next tile's records, when it needs any, once their sweeps
// are done), the other half MOVE: while tile t is swept they request tile t + 1 (its stretch of X, its position table)
// into their registers, sweep a group of single paths themselves, request tile t's destinations; when the sweeps are done
// and the tile is exponentiated they send its posteriors out and place tile t + 1 in LDS.  The counter of a wavefront's
// outstanding loads is in order, so a sweeping wavefront that also held the next tile's requests would wait for all of
// them at its first own load; a moving wavefront has nothing else to wait for.
#define TILE_SWEEP_THREADS 512
#define TILE_SWEEP_MOVERS 256
typedef uint32_t ts_u32x4 __attribute__((ext_vector_type(4)));
struct TileIn {  // a tile on its way in: what every moving thread holds of it (loaded values as they arrive: nothing computes
                 // on them before the tile is placed, so nothing waits for them)
  double v[TILE_SWEEP_TILE / TILE_SWEEP_MOVERS];
  uint16_t pos[TILE_SWEEP_TILE / TILE_SWEEP_MOVERS];
};
template <bool SCAT>
__device__ __forceinline__ void tile_request(const TransArgs& T, const LaneArgs& A, uint32_t tile, uint64_t i0, uint32_t ni, uint32_t m,
                                             TileIn& in) {
  constexpr int NM = TILE_SWEEP_MOVERS, KT = (int)(TILE_SWEEP_TILE / NM);
  // every load is unconditional and unclamped: a mover issues ~100 of them while its tile is swept, and at a lone wavefront's
  // issue rate the address arithmetic of a clamp per load was what the sweepers ended up waiting for.  Items past the tile's
  // are the next tile's (or the slack behind the arrays, DevBuf::alloc): read and ignored.
  const uint16_t* __restrict__ tp = T.t_pos + i0 + m;
#pragma unroll
  for (int k = 0; k < KT; ++k) in.pos[k] = tp[k * NM];
  if (SCAT) {
    const double* __restrict__ xp = T.x + i0 + m;
#pragma unroll
    for (int k = 0; k < KT; ++k) in.v[k] = xp[k * NM];
  } else {
    const uint32_t last = ni ? ni - 1 : 0u;  // (a gather's index has to be a real one)
    const uint32_t* __restrict__ sp = T.t_src + (ni ? i0 : 0);
    uint32_t src[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) src[k] = sp[min(m + k * NM, last)];
#pragma unroll
    for (int k = 0; k < KT; ++k) in.v[k] = T.x[src[k]];
  }
}
// the sweeps leave an arc's LOG posterior at its position; the exponentials are the whole workgroup's (every position of the
// tile, padding included: sixteen independent ones a thread)
__device__ __forceinline__ void tile_exp_in_place(double* lds) {
  constexpr int K = (int)(TILE_SWEEP_TILE / TILE_SWEEP_THREADS);
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += 4) {  // four side by side (a mover holds the next tile in its registers meanwhile)
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = lds[threadIdx.x + (k0 + k) * TILE_SWEEP_THREADS];
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[threadIdx.x + (k0 + k) * TILE_SWEEP_THREADS] = K_EXP(v[k]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// the walk of a workgroup over its tiles: workgroup b runs on XCD b % 8 and takes the tiles of that XCD's contiguous eighth
// (xcd_chunked), gridDim.x / 8 apart; the scalars of a tile (its items, its groups) are requested two tiles ahead.
// Wave-uniform values that are REQUESTED AHEAD (a tile's scalars, a group's descriptor) are loaded through an address the
// compiler cannot prove uniform (ts_zero: a zero it cannot see through): a uniform vector load is moved to scalar registers
// the moment it is issued, i.e. waited for on the spot; this way it stays a pending vector register until its use, where
// ts_sc / ts_uniform make it scalar.
__device__ __forceinline__ uint32_t ts_zero() {
  uint32_t z = 0;
  asm volatile("" : "+v"(z));
  return z;
}
__device__ __forceinline__ uint32_t ts_sc(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t ts_sc(uint64_t v) { return ((uint64_t)ts_sc((uint32_t)(v >> 32)) << 32) | ts_sc((uint32_t)v); }
__device__ __forceinline__ LaneGroup ts_uniform(const LaneGroup& v) {
  LaneGroup g;
  g.stream_base = ts_sc(v.stream_base);
  g.maxlen = ts_sc(v.maxlen);
  g.n_lanes = ts_sc(v.n_lanes);
  g.pair_base = ts_sc(v.pair_base);
  g.max_states = ts_sc(v.max_states);
  g.window = 0;
  g.spill_row = ts_sc(v.spill_row);
  return g;
}
struct TileWalk {
  uint32_t tile, ni, g0, g1, chain;  // (g0, g1, chain, i0, i1: as loaded until ts_uniform)
  uint64_t i0, i1;
  bool ok;
};
__device__ __forceinline__ TileWalk tile_walk_at(const TransArgs& T, const LaneArgs& A, const uint32_t* __restrict__ tile_group, uint32_t vidx,
                                                 bool ok, uint32_t z) {
  TileWalk w;
  const uint32_t tloc = xcd_chunked(vidx, T.tile_count);
  w.ok = ok && tloc < T.tile_count && vidx / 8 < (T.tile_count + 7) / 8;
  w.tile = T.tile_first + (w.ok ? tloc : 0u);
  w.i0 = T.tile_base[w.tile + z];
  w.i1 = T.tile_base[w.tile + 1 + z];
  w.g0 = tile_group[w.tile + z];
  w.g1 = tile_group[w.tile + 1 + z];
  w.chain = A.tile_chain[w.tile + z];
  w.ni = 0;
  return w;
}
__device__ __forceinline__ TileWalk ts_uniform(const TileWalk& v) {
  TileWalk w;
  w.tile = v.tile;
  w.ok = v.ok;
  w.i0 = ts_sc(v.i0);
  w.i1 = ts_sc(v.i1);
  w.ni = (uint32_t)(w.i1 - w.i0);
  w.g0 = ts_sc(v.g0);
  w.g1 = ts_sc(v.g1);
  w.chain = ts_sc(v.chain);
  return w;
}
template <bool SCAT>
__global__ __launch_bounds__(TILE_SWEEP_THREADS) void tile_sweep_kernel(TransArgs T, LaneArgs A, const uint32_t* __restrict__ tile_group) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NM = TILE_SWEEP_MOVERS, KT = (int)(TILE_SWEEP_TILE / NM), NW = (TILE_SWEEP_THREADS - NM) / 64;
  uint32_t* const recs = (uint32_t*)(lds + TILE_SWEEP_TILE);                // one packed record per position
  double* const alpha = lds + TILE_SWEEP_TILE + TILE_SWEEP_TILE / 2;        // TILE_SWEEP_ALPHA_ROWS rows of values
  const int lane = threadIdx.x & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t vidx = blockIdx.x;
  const uint32_t z = ts_zero();
  // (on the way in: the count pass's atomics start from zero -- TransArgs::zero_list)
  for (uint32_t i = blockIdx.x * TILE_SWEEP_THREADS + threadIdx.x; i < T.n_zero; i += gridDim.x * TILE_SWEEP_THREADS) T.counts[T.zero_list[i]] = 0.0;
  TileWalk cur = tile_walk_at(T, A, tile_group, vidx, true, z);
  if (!cur.ok) return;
  cur = ts_uniform(cur);
  TileWalk nxt = ts_uniform(tile_walk_at(T, A, tile_group, vidx + gridDim.x, true, z));
  TileWalk nn_raw = tile_walk_at(T, A, tile_group, vidx + 2 * gridDim.x, nxt.ok, z);  // (as loaded)
  // Both kinds of wavefront meet at the same four barriers per tile: (a) the tile is placed, (b) it is swept, (b') its posteriors
  // are exponentiated, (c) it is read out.
  if (wv >= (uint32_t)NW) {
    // ================= movers =================
    const uint32_t m = threadIdx.x - (uint32_t)(TILE_SWEEP_THREADS - NM);
    TileIn in;
    tile_request<SCAT>(T, A, cur.tile, cur.i0, cur.ni, m, in);
    // A mover sweeps as well while it has nothing to do but wait for its requests: group NW + (its number) of the tile, when
    // that is a group of single paths (tile_chain_sweep asks the memory for nothing; a sweeper takes it otherwise).  Its
    // descriptor and lanes are requested a tile ahead, behind the tile's own requests, and are there when the tile is placed.
    const uint32_t mw = wv - (uint32_t)NW;
    LaneGroup gm_raw;
    TileLane lm;
    uint32_t cm_raw;
    {
      const uint32_t gk = cur.g0 + NW + mw < cur.g1 ? cur.g0 + NW + mw : cur.g0;
      gm_raw = A.groups[gk + z];
      lm = tile_lane(A, gk, lane);
      cm_raw = A.chain[gk + z];
    }
    for (;;) {
      // the tile's weights to their lane positions
      uint32_t pos2[KT / 2];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k & 1)
          pos2[k / 2] |= (uint32_t)in.pos[k] << 16;
        else
          pos2[k / 2] = in.pos[k];
        if (m + k * NM < cur.ni) lds[in.pos[k]] = in.v[k];
      }
      __syncthreads();  // (a)
      // requests while the tile is swept: the next tile (and, once its own sweep is done, this tile's destinations)
      if (nxt.ok) tile_request<SCAT>(T, A, nxt.tile, nxt.i0, nxt.ni, m, in);
      {
        const LaneGroup gm = ts_uniform(gm_raw);  // (this tile's: there since the tile was placed)
        const TileLane lm_cur = lm;
        const uint32_t cm = ts_sc(cm_raw);
        const bool mine = cur.g0 + NW + mw < cur.g1 && (cm & 1u);
        const uint32_t gk = nxt.g0 + NW + mw < nxt.g1 ? nxt.g0 + NW + mw : nxt.g0;
        gm_raw = A.groups[gk + z];
        lm = tile_lane(A, gk, lane);
        cm_raw = A.chain[gk + z];
        if (mine) {
          const uint32_t off = (uint32_t)(gm.stream_base - (uint64_t)cur.tile * TILE_SWEEP_TILE) + lane;
          const double lp = tile_chain_sweep(gm, lm_cur, cm >> 8, lane, alpha + (size_t)gm.spill_row * 64 + lane, lds + off);
          if ((uint32_t)lane < gm.n_lanes) A.pair_logprob[lm_cur.pair] = lp;
        }
      }
      uint32_t dst[KT];
      __builtin_amdgcn_sched_barrier(0);  // (not before the sweep: its registers)
      if (SCAT) {
        const uint32_t* __restrict__ sp = T.t_src + cur.i0 + m;  // (unclamped, as the tile's requests)
#pragma unroll
        for (int k = 0; k < KT; ++k) dst[k] = sp[k * NM];
      }
      __syncthreads();  // (b)
      tile_exp_in_place(lds);
      __syncthreads();  // (b')
      // the tile's posteriors out
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const uint32_t i = m + k * NM;
        const uint32_t q = (k & 1) ? pos2[k / 2] >> 16 : pos2[k / 2] & 0xffffu;
        if (i < cur.ni) T.xc[SCAT ? (uint64_t)dst[k] : cur.i0 + i] = lds[q];
      }
      if (!nxt.ok) break;
      __syncthreads();  // (c)
      vidx += gridDim.x;
      cur = nxt;
      nxt = ts_uniform(nn_raw);
      nn_raw = tile_walk_at(T, A, tile_group, vidx + 2 * gridDim.x, nxt.ok, z);
    }
  } else {
    // ================= sweepers =================
    constexpr int KR = (int)(TILE_SWEEP_TILE / 4 / (TILE_SWEEP_THREADS - NM));  // 16-byte pieces of a tile's records per sweeping thread
    LaneGroup g;
    TileLane L;
    uint32_t chain;
    {
      const uint32_t gk = cur.g0 + wv < cur.g1 ? cur.g0 + wv : cur.g0;
      g = ts_uniform(A.groups[gk + z]);
      L = tile_lane(A, gk, lane);
      chain = ts_sc(A.chain[gk + z]);
      if (!cur.chain) {  // the first tile's records as they are (the stream covers whole tiles); a tile of single paths needs none
        ts_u32x4 rr[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k)
          rr[k] = *(const ts_u32x4*)(A.rec2 + (uint64_t)cur.tile * TILE_SWEEP_TILE + (threadIdx.x + k * (TILE_SWEEP_THREADS - NM)) * 4);
#pragma unroll
        for (int k = 0; k < KR; ++k) *(ts_u32x4*)(recs + (threadIdx.x + k * (TILE_SWEEP_THREADS - NM)) * 4) = rr[k];
      }
    }
    for (;;) {
      unsigned long long t0 = 0, t1 = 0, t2 = 0;
      if (A.trace) t0 = __builtin_readcyclecounter();
      const uint64_t p0 = (uint64_t)cur.tile * TILE_SWEEP_TILE;
      // the first group of the next tile: its descriptor and lanes arrive while this tile is swept
      const uint32_t gk_n = nxt.g0 + wv < nxt.g1 ? nxt.g0 + wv : nxt.g0;
      const LaneGroup g_n = A.groups[gk_n + z];  // (as loaded: made uniform at its use)
      const TileLane L_n = tile_lane(A, gk_n, lane);
      const uint32_t chain_n = A.chain[gk_n + z];
      __syncthreads();  // (a)
      if (A.trace) t1 = __builtin_readcyclecounter();
      for (uint32_t gi = cur.g0 + wv; gi < cur.g1; gi += NW) {
        const uint32_t gk = gi + NW < cur.g1 ? gi + NW : gi;  // the wavefront's next group of this tile, requested before this one's sweep
        const LaneGroup g_next = A.groups[gk + z];
        const TileLane L_next = tile_lane(A, gk, lane);
        const uint32_t chain_next = A.chain[gk + z];
        const uint32_t off = (uint32_t)(g.stream_base - p0) + lane;
        const bool movers = gi == cur.g0 + NW + wv && (chain & 1u);  // a group of single paths among the tile's second four: a mover's
        double lp = 0.0;
        if (!movers) {
          if (chain & 1u)
            lp = tile_chain_sweep(g, L, chain >> 8, lane, alpha + (size_t)g.spill_row * 64 + lane, lds + off);
          else
            lp = tile_group_sweep(g, L, lane, alpha + (size_t)g.spill_row * 64 + lane, lds + off, recs + off);
        }
        const bool active = !movers && (uint32_t)lane < g.n_lanes;
        const uint32_t pair = L.pair;
        g = ts_uniform(g_next);
        chain = ts_sc(chain_next);
        L = L_next;
        if (active) A.pair_logprob[pair] = lp;  // (after the waits for the next group's requests: see tile_group_sweep)
      }
      if (A.trace && lane == 0) A.trace[(size_t)cur.tile * 16 + 8 + wv] = __builtin_readcyclecounter() - t1;  // this wavefront's sweeps
      // the next tile's records (unless it is all single paths): on their way while this tile's posteriors leave
      const bool recs_n = nxt.ok && !nxt.chain;
      ts_u32x4 rr[KR];
      if (recs_n) {
#pragma unroll
        for (int k = 0; k < KR; ++k)
          rr[k] = *(const ts_u32x4*)(A.rec2 + (uint64_t)nxt.tile * TILE_SWEEP_TILE + (threadIdx.x + k * (TILE_SWEEP_THREADS - NM)) * 4);
      }
      __syncthreads();  // (b)
      if (A.trace) t2 = __builtin_readcyclecounter();
      tile_exp_in_place(lds);
      __syncthreads();  // (b')
      // (the movers read the posteriors out; nobody needs this tile's records any more)
      if (recs_n) {
#pragma unroll
        for (int k = 0; k < KR; ++k) *(ts_u32x4*)(recs + (threadIdx.x + k * (TILE_SWEEP_THREADS - NM)) * 4) = rr[k];
      }
      if (A.trace && threadIdx.x == 0) {  // experiment (CARMEL_HIP_LANE_TRACE): cycles per phase, as the first sweeping wavefront sees them
        unsigned long long* o = A.trace + (size_t)cur.tile * 16;
        o[0] = t0;
        o[1] = t1 - t0;
        o[2] = t2 - t1;
        o[4] = ((unsigned long long)(cur.g1 - cur.g0) << 32) | cur.ni;
        o[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) | g.maxlen;
      }
      if (!nxt.ok) break;
      __syncthreads();  // (c)
      if (A.trace && threadIdx.x == 0) A.trace[(size_t)cur.tile * 16 + 3] = __builtin_readcyclecounter() - t2;
      vidx += gridDim.x;
      cur = nxt;
      nxt = ts_uniform(nn_raw);
      nn_raw = tile_walk_at(T, A, tile_group, vidx + 2 * gridDim.x, nxt.ok, z);
      g = ts_uniform(g_n);
      L = L_n;
      chain = ts_sc(chain_n);
    }
  }
}
#define TRANS_SET_LDS(K, BYTES) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES)
hipError_t launch_tile_sweep(const TransArgs& T0, const LaneArgs& A, const uint32_t* tile_group, uint32_t tile_first, uint32_t tile_count,
                             hipStream_t stream) {
  if (!T0.n_buckets || !tile_count) return hipSuccess;
  if (T0.tile != TILE_SWEEP_TILE || !A.pre_weights || !A.rec2 || !A.chain || !A.tile_chain) return hipErrorInvalidValue;
  // the movers' unclamped requests read up to a tile's worth of items past the last tile's: 8 B (x), 2 B (t_pos), 4 B (t_src) each
  if (T0.slack_bytes < TILE_SWEEP_TILE * sizeof(double)) return hipErrorInvalidValue;
  // per device: the dynamic-LDS attribute of a kernel and the CU count are the device's, not the process's
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  static int dev_cus[64] = {0};  // 0: this device has not been set up (written once per device; racing writers store the same value)
  const int lds = (int)TILE_SWEEP_LDS;
  if (!dev_cus[dev]) {
    if (hipFuncSetAttribute((const void*)tile_sweep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)tile_sweep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return hipErrorInvalidValue;  // (the caller falls back to the three kernels on the same layout)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
    __atomic_store_n(&dev_cus[dev], prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8, __ATOMIC_RELEASE);
  }
  const int n_cu = __atomic_load_n(&dev_cus[dev], __ATOMIC_ACQUIRE);
  TransArgs T = T0;
  T.tile_first = tile_first;
  T.tile_count = tile_count;
  // persistent: a workgroup per CU (its LDS is a CU's), fewer when there are fewer tiles; a multiple of 8 (XCDs)
  const dim3 g8(std::min<uint32_t>((uint32_t)n_cu, (tile_count + 7) / 8 * 8));
  // (the scattering forms of both directions go together: TransArgs::scatter is 3 where the corpus has run-length indices)
  if ((T.scatter & 3u) == 3u && T.use_runs)
    hipLaunchKernelGGL((tile_sweep_kernel<true>), g8, dim3(TILE_SWEEP_THREADS), lds, stream, T, A, tile_group);
  else if (!(T.scatter & 3u))
    hipLaunchKernelGGL((tile_sweep_kernel<false>), g8, dim3(TILE_SWEEP_THREADS), lds, stream, T, A, tile_group);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_pack_tile_records(const LaneGroup* groups, uint32_t n_groups, const uint32_t* lane_nstates, const uint32_t* fwdx,
                                    const uint32_t* bwd, uint32_t* out, uint32_t* chain, const uint32_t* tile_group, uint32_t n_tiles,
                                    uint32_t* tile_chain, hipStream_t stream) {
  if (n_groups)
    hipLaunchKernelGGL(pack_tile_records_kernel, dim3((n_groups + 3) / 4), dim3(256), 0, stream, groups, n_groups, lane_nstates, fwdx, bwd, out,
                       chain);
  if (n_tiles) hipLaunchKernelGGL(tile_chain_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, tile_group, chain, n_tiles, tile_chain);
  return hipGetLastError();
}
}  // namespace carmel_hip
