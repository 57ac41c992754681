// lattice_gpu.hip — derivation-lattice construction and layout ON THE GPU (7 "hard part a").
//
// What it computes is what lattice.cpp computes on the host -- the reference's derivations::compute + prune
// (/root/reference/carmel/src/derivations.h:479-513, 572-629, 640-704) for every training pair, then the lane-group
// record streams, the posterior slots sorted by WFST arc and the blocked transposition tables -- and it produces the
// SAME device image bit for bit (tests/test_lattice_gpu.py compares checksums of every array), because every
// order-defining rule of the host builder is kept:
//   * a pair's lattice is explored depth first from (0, start, 0); from a node the four label classes (e,e),
//     (e,out[o]), (in[i],e), (in[i],out[o]) in that order, within a class the WFST arcs in arc-id order; state ids are
//     DFS pre-order; an arc is kept iff its destination is not known to be dead (derivations.h:656-701);
//   * states that cannot reach the goal are dropped, ids stably compacted (derivations.h:572-629);
//   * lane numbering is (longest-path level, id); records as documented in lattice.hpp.
// How: one THREAD per pair runs that depth-first search with an explicit stack in a private slice of scratch memory
// (10^6 pairs: 131 072 threads in flight, each a short walk over the (state, in, out)-sorted arc index); pairs are
// sorted into lane groups with one radix sort; one wave per group interleaves its 64 record streams; the (arc, slot)
// items are radix-sorted by arc for the per-arc slot lists, once more by (bucket, position) and once by tile for the
// transposition tables.  Only two things touch the host: the launch classes (a few thousand group descriptors) and
// the trainer's bookkeeping.
//
// Scope: corpora all of whose lattices are acyclic, have at most `lane_states` states (the one-lattice-per-lane layout)
// and fit the per-thread caps below.  Any pair outside that -- a cycle, a big lattice -- makes the whole build fall
// back to the host builder (carmel_hip_build_lattices then behaves exactly as before).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <functional>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#include "engine.hpp"
#include "options.hpp"

namespace {

// per-pair capacities of the exploration (GArgs::cx ...): explored states, edges kept during exploration, arcs of the pruned
// lattice (stride of the per-pair record arrays), hash slots (2 * cx).  Two sizes: the small one covers corpora of plain
// one-per-lane lattices (config 4: 17 KB of scratch per thread), the large one corpora with windowed groups -- lattices of
// up to 1 023 states (the tagging cascade, config c4a).
struct GCaps {
  uint32_t cx, ce, ck, ch, threads;
};
const GCaps G_SMALL = {256, 448, 256, 512, 131072};
const GCaps G_LARGE = {1024, 2560, 1536, 2048, 65536};

enum { PF_HAS = 1, PF_FALLBACK = 2 };

struct GArgs {
  // WFST: (state, in, out)-sorted arc index (HostWfst::build_index)
  const uint64_t* idx_off;
  const uint64_t* idx_key;
  const uint32_t* idx_arc;
  const uint32_t* arc_dst;
  uint32_t final_state;
  // corpus
  const uint64_t* in_off;
  const uint64_t* out_off;
  const uint32_t* in_sym;
  const uint32_t* out_sym;
  uint64_t n_pairs;
  uint32_t lane_states;
  uint32_t cx, ce, ck, ch;   // capacities (GCaps)
  uint32_t span_min;         // lattices above this many states get their span (BuildOptions::lane_window_min)
  uint16_t* pp_span;         // max over the arcs of (dst - src) in the (level, id) numbering; 0 = not computed
  uint16_t* pp_L;            // levels
  // per-thread scratch
  uint8_t* scratch;
  uint32_t scratch_stride;
  // per-pair results
  uint16_t* pp_E;
  uint16_t* pp_S;
  uint8_t* pp_flags;
  uint32_t* pp_xs;   // explored states
  uint32_t* pp_xa;   // explored arcs
  uint32_t* rec_fwd; // [pair * GK + k]: forward record word, the backward row still relative to the lattice's own
  uint32_t* rec_bwd;
  uint32_t* rec_arc; // WFST arc of the backward record k
};

__device__ __forceinline__ uint32_t st_hash(uint32_t i, uint32_t s, uint32_t o) {
  uint64_t h = ((uint64_t)i << 40) ^ ((uint64_t)o << 20) ^ s;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return (uint32_t)h;
}

// [lo, hi) of the arcs of state s labelled key = in << 32 | out in the sorted index
__device__ __forceinline__ void key_range(const GArgs& G, uint32_t s, uint64_t key, uint32_t& lo, uint32_t& hi) {
  uint64_t a = G.idx_off[s], b = G.idx_off[s + 1];
  while (a < b) {
    const uint64_t m = (a + b) >> 1;
    if (G.idx_key[m] < key) a = m + 1; else b = m;
  }
  uint64_t e = a;
  const uint64_t end = G.idx_off[s + 1];
  while (e < end && G.idx_key[e] == key) ++e;
  lo = (uint32_t)a;
  hi = (uint32_t)e;
}

__global__ __launch_bounds__(256) void explore_kernel(GArgs G) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint8_t* base = G.scratch + (size_t)tid * G.scratch_stride;
  const uint32_t GX = G.cx, GE = G.ce, GK = G.ck, GH = G.ch;
  // carve the private slice
  uint32_t* st_i = (uint32_t*)base;             // [GX]
  uint32_t* st_s = st_i + GX;
  uint32_t* st_o = st_s + GX;
  uint32_t* f_pos = st_o + GX;                  // frames [GX]
  uint32_t* f_end = f_pos + GX;
  uint32_t* f_parc = f_end + GX;
  uint32_t* e_arc = f_parc + GX;                // edges [GE]
  uint16_t* hash = (uint16_t*)(e_arc + GE);     // [GH]
  uint16_t* f_sid = hash + GH;                  // [GX]
  uint16_t* f_pdst = f_sid + GX;
  uint16_t* e_src = f_pdst + GX;                // [GE]
  uint16_t* e_dst = e_src + GE;
  uint16_t* wa = e_dst + GE;                    // work arrays [GE + 2] each
  uint16_t* wb = wa + GE + 2;
  uint16_t* wc = wb + GE + 2;
  uint16_t* wd = wc + GE + 2;
  uint16_t* we = wd + GE + 2;
  uint16_t* wf = we + GE + 2;
  uint16_t* level = wf + GE + 2;                // [GX]
  uint8_t* removed = (uint8_t*)(level + GX);    // [GX]
  uint8_t* f_phase = removed + GX;              // [GX]
  uint8_t* f_flags = f_phase + GX;              // bit0 dead, bit1 pending
  for (uint64_t p = tid; p < G.n_pairs; p += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t* in = G.in_sym + G.in_off[p];
    const uint32_t* out = G.out_sym + G.out_off[p];
    const uint32_t n_in = (uint32_t)(G.in_off[p + 1] - G.in_off[p]), n_out = (uint32_t)(G.out_off[p + 1] - G.out_off[p]);
    const uint32_t gi = n_in, gs = G.final_state, go = n_out;
    for (uint32_t k = 0; k < GH; ++k) hash[k] = 0xffffu;
    uint32_t n_states = 0, n_edges = 0, depth = 0, explored_arcs = 0;
    bool overflow = false;
    // ---- explore (lattice.cpp explore(): derivations.h:640-704) ----
    auto new_state = [&](uint32_t i, uint32_t s, uint32_t o) -> uint32_t {
      uint32_t h = st_hash(i, s, o) & (GH - 1);
      for (;;) {
        const uint32_t id = hash[h];
        if (id == 0xffffu) {
          if (n_states == GX || depth == GX) {
            overflow = true;
            return 0;
          }
          const uint32_t nid = n_states++;
          st_i[nid] = i;
          st_s[nid] = s;
          st_o[nid] = o;
          hash[h] = (uint16_t)nid;
          removed[nid] = 0;
          f_sid[depth] = (uint16_t)nid;
          f_phase[depth] = 0xff;  // not started
          f_pos[depth] = f_end[depth] = 0;
          f_flags[depth] = (i == gi && s == gs && o == go) ? 0 : 1;  // dead unless it is the goal
          ++depth;
          return nid | 0x80000000u;
        }
        if (st_i[id] == i && st_s[id] == s && st_o[id] == o) return id;
        h = (h + 1) & (GH - 1);
      }
    };
    new_state(0, 0, 0);
    while (depth && !overflow) {
      const uint32_t fi = depth - 1;
      if (f_flags[fi] & 2) {  // the child pushed for f_parc has finished
        if (!removed[f_pdst[fi]]) {
          if (n_edges == GE) {
            overflow = true;
            break;
          }
          e_src[n_edges] = f_sid[fi];
          e_dst[n_edges] = f_pdst[fi];
          e_arc[n_edges] = f_parc[fi];
          ++n_edges;
          f_flags[fi] &= ~1;
        }
        f_flags[fi] &= ~2;
        ++f_pos[fi];
      }
      for (;;) {
        const uint32_t sid = f_sid[fi];
        const int ph = (int)(int8_t)f_phase[fi];  // -1 not started
        if (f_pos[fi] < f_end[fi]) {
          const uint32_t arc = G.idx_arc[f_pos[fi]];
          ++explored_arcs;
          const uint32_t ni = st_i[sid] + (ph >= 2 ? 1u : 0u), no = st_o[sid] + ((ph == 1 || ph == 3) ? 1u : 0u);
          const uint32_t r = new_state(ni, G.arc_dst[arc], no);
          if (overflow) break;
          if (r & 0x80000000u) {
            f_flags[fi] |= 2;
            f_pdst[fi] = (uint16_t)(r & 0x7fffffffu);
            f_parc[fi] = arc;
            break;
          }
          if (!removed[r]) {
            if (n_edges == GE) {
              overflow = true;
              break;
            }
            e_src[n_edges] = (uint16_t)sid;
            e_dst[n_edges] = (uint16_t)r;
            e_arc[n_edges] = arc;
            ++n_edges;
            f_flags[fi] &= ~1;
          }
          ++f_pos[fi];
          continue;
        }
        // next label class
        const int nph = ph + 1;
        f_phase[fi] = (uint8_t)nph;
        if (nph > 3) {
          removed[sid] = f_flags[fi] & 1;
          --depth;
          break;
        }
        const uint32_t ci = st_i[sid], co = st_o[sid];
        const bool useO = co < n_out, useI = ci < n_in;
        uint32_t si = 0, so = 0;
        bool ok = true;
        switch (nph) {
          case 0: break;
          case 1:
            ok = useO;
            if (ok) so = out[co];
            break;
          case 2:
            ok = useI;
            if (ok) si = in[ci];
            break;
          case 3:
            ok = useI && useO;
            if (ok) {
              si = in[ci];
              so = out[co];
            }
            break;
        }
        if (ok) {
          uint32_t lo, hi;
          key_range(G, st_s[sid], ((uint64_t)si << 32) | so, lo, hi);
          f_pos[fi] = lo;
          f_end[fi] = hi;
        } else
          f_pos[fi] = f_end[fi] = 0;
      }
    }
    G.pp_xs[p] = n_states;
    G.pp_xa[p] = explored_arcs;
    G.pp_E[p] = 0;
    G.pp_S[p] = 0;
    G.pp_span[p] = 0;
    G.pp_L[p] = 0;
    if (overflow) {
      G.pp_flags[p] = PF_FALLBACK;
      continue;
    }
    // goal?
    uint32_t goal = 0xffffffffu;
    {
      uint32_t h = st_hash(gi, gs, go) & (GH - 1);
      for (;;) {
        const uint32_t id = hash[h];
        if (id == 0xffffu) break;
        if (st_i[id] == gi && st_s[id] == gs && st_o[id] == go) {
          goal = id;
          break;
        }
        h = (h + 1) & (GH - 1);
      }
    }
    if (goal == 0xffffffffu) {
      G.pp_flags[p] = 0;
      continue;
    }
    // ---- co-reachability over the kept edges (build_pair_lattice_impl) ----
    const uint32_t nst = n_states;
    uint16_t* indeg_off = wa;  // [nst + 1]
    uint16_t* rsrc = wb;       // [n_edges]
    uint16_t* cur = wc;
    uint16_t* keep = wd;       // old -> new or 0xffff
    for (uint32_t s = 0; s <= nst; ++s) indeg_off[s] = 0;
    for (uint32_t e = 0; e < n_edges; ++e) indeg_off[e_dst[e] + 1]++;
    for (uint32_t s = 0; s < nst; ++s) indeg_off[s + 1] += indeg_off[s];
    for (uint32_t s = 0; s < nst; ++s) cur[s] = indeg_off[s];
    for (uint32_t e = 0; e < n_edges; ++e) rsrc[cur[e_dst[e]]++] = e_src[e];
    for (uint32_t s = 0; s < nst; ++s) keep[s] = 0xffffu;
    {
      uint16_t* stk = wc;
      uint32_t sp = 0;
      stk[sp++] = (uint16_t)goal;
      keep[goal] = 0;
      while (sp) {
        const uint32_t v = stk[--sp];
        for (uint32_t k = indeg_off[v]; k < indeg_off[v + 1]; ++k) {
          const uint32_t u = rsrc[k];
          if (keep[u] == 0xffffu) {
            keep[u] = 0;
            stk[sp++] = (uint16_t)u;
          }
        }
      }
    }
    uint32_t kept = 0;
    for (uint32_t s = 0; s < nst; ++s)
      if (keep[s] != 0xffffu && !removed[s]) keep[s] = (uint16_t)kept++;
      else keep[s] = 0xffffu;
    // kept edges, compacted in place (order preserved)
    uint32_t E = 0;
    for (uint32_t e = 0; e < n_edges; ++e)
      if (keep[e_src[e]] != 0xffffu && keep[e_dst[e]] != 0xffffu) {
        e_src[E] = keep[e_src[e]];
        e_dst[E] = keep[e_dst[e]];
        e_arc[E] = e_arc[e];
        ++E;
      }
    const uint32_t S = kept;
    if (S > 1023u || E > GK || E > LANE_POS_MAX || keep[0] == 0xffffu) {  // not a one-lattice-per-lane case, plain or windowed
      G.pp_flags[p] = PF_HAS | PF_FALLBACK;
      continue;
    }
    // ---- levels: Kahn, level = longest path from the start ----
    uint16_t* indeg = wa;  // [S]
    uint16_t* ooff = wb;   // [S + 1]
    uint16_t* odst = wc;   // [E]
    uint16_t* q = wd;      // [S]
    uint16_t* cur2 = we;
    for (uint32_t s = 0; s < S; ++s) indeg[s] = 0;
    for (uint32_t s = 0; s <= S; ++s) ooff[s] = 0;
    for (uint32_t e = 0; e < E; ++e) {
      indeg[e_dst[e]]++;
      ooff[e_src[e] + 1]++;
    }
    for (uint32_t s = 0; s < S; ++s) ooff[s + 1] += ooff[s];
    for (uint32_t s = 0; s < S; ++s) cur2[s] = ooff[s];
    for (uint32_t e = 0; e < E; ++e) odst[cur2[e_src[e]]++] = e_dst[e];
    for (uint32_t s = 0; s < S; ++s) level[s] = 0;
    uint32_t qn = 0, done = 0;
    for (uint32_t s = 0; s < S; ++s)
      if (indeg[s] == 0) q[qn++] = (uint16_t)s;
    while (done < qn) {
      const uint32_t u = q[done++];
      const uint32_t lu = level[u];
      for (uint32_t k = ooff[u]; k < ooff[u + 1]; ++k) {
        const uint32_t v = odst[k];
        if (level[v] < lu + 1) level[v] = (uint16_t)(lu + 1);
        if (--indeg[v] == 0) q[qn++] = (uint16_t)v;
      }
    }
    if (done != S) {  // a cycle: the reference's order-dependent sweep is the host builder's business
      G.pp_flags[p] = PF_HAS | PF_FALLBACK;
      continue;
    }
    // ---- lane records (lattice.cpp lwork): topological numbering by (level, id) ----
    uint16_t* newid = wa;  // [S]
    uint16_t* ioff = wb;   // [S + 1]
    uint16_t* ooff2 = wc;  // [S + 1]
    uint16_t* ie = wd;     // [E]
    uint16_t* oe = we;     // [E]
    uint16_t* fpos = wf;   // [E]
    {
      // counting sort of the states by level, stable in id
      uint16_t* cnt = wd;  // [S + 1] (levels < S)
      for (uint32_t l = 0; l <= S; ++l) cnt[l] = 0;
      for (uint32_t s = 0; s < S; ++s) cnt[level[s] + 1]++;
      for (uint32_t l = 0; l < S; ++l) cnt[l + 1] += cnt[l];
      for (uint32_t s = 0; s < S; ++s) newid[s] = cnt[level[s]]++;
    }
    for (uint32_t s = 0; s <= S; ++s) ioff[s] = ooff2[s] = 0;
    for (uint32_t e = 0; e < E; ++e) {
      ioff[newid[e_dst[e]] + 1]++;
      ooff2[newid[e_src[e]] + 1]++;
    }
    for (uint32_t s = 0; s < S; ++s) {
      ioff[s + 1] += ioff[s];
      ooff2[s + 1] += ooff2[s];
    }
    {
      uint16_t* c = (uint16_t*)f_pos;  // the frames are free now: cursors [S]
      for (uint32_t s = 0; s < S; ++s) c[s] = ioff[s];
      for (uint32_t e = 0; e < E; ++e) ie[c[newid[e_dst[e]]]++] = (uint16_t)e;
      for (uint32_t s = 0; s < S; ++s) c[s] = ooff2[s];
      for (uint32_t e = 0; e < E; ++e) oe[c[newid[e_src[e]]]++] = (uint16_t)e;
    }
    if (S > G.span_min) {  // how far apart are an arc's ends in this numbering?  (windowed groups, lattice.cpp)
      uint32_t sp = 1;
      for (uint32_t e = 0; e < E; ++e) sp = max(sp, (uint32_t)newid[e_dst[e]] - (uint32_t)newid[e_src[e]]);
      G.pp_span[p] = (uint16_t)sp;
    }
    {
      uint32_t nl = 0;
      for (uint32_t s2 = 0; s2 < S; ++s2) nl = max(nl, (uint32_t)level[s2] + 1u);
      G.pp_L[p] = (uint16_t)nl;
    }
    uint32_t* rf = G.rec_fwd + p * GK;
    uint32_t* rb = G.rec_bwd + p * GK;
    uint32_t* ra = G.rec_arc + p * GK;
    uint32_t pos = 0;
    for (uint32_t sidx = S; sidx-- > 0;) {
      if (sidx == S - 1) continue;  // the goal has no out-arcs
      for (uint32_t k = ooff2[sidx]; k < ooff2[sidx + 1]; ++k) {
        const uint32_t e = oe[k];
        fpos[e] = (uint16_t)pos;
        rb[pos] = (uint32_t)newid[e_dst[e]] | (sidx << LANE_POS_SHIFT) | LANE_VALID | (k + 1 == ooff2[sidx + 1] ? LANE_LAST : 0u);
        ra[pos] = e_arc[e];
        ++pos;
      }
    }
    pos = 0;
    for (uint32_t d = 1; d < S; ++d)
      for (uint32_t k = ioff[d]; k < ioff[d + 1]; ++k) {
        const uint32_t e = ie[k];
        rf[pos++] = (uint32_t)newid[e_src[e]] | ((uint32_t)fpos[e] << LANE_POS_SHIFT) | LANE_VALID |
                    (k + 1 == ioff[d + 1] ? LANE_LAST : 0u);
      }
    G.pp_E[p] = (uint16_t)E;
    G.pp_S[p] = (uint16_t)S;
    G.pp_flags[p] = PF_HAS;
  }
}

// the ring a windowed lattice needs (lattice.cpp window_of): the power of two above its span, 0 = not windowed
__device__ __forceinline__ uint32_t window_of(uint32_t span, uint32_t S, uint32_t use_window, uint32_t lane_window) {
  if (!use_window || !span) return 0u;
  uint32_t w = 8;
  while (w < span + 1) w <<= 1;
  return (w <= lane_window && w < S) ? w : 0u;
}
struct ClassArgs {
  const uint16_t* E;
  const uint16_t* S;
  const uint16_t* span;
  const uint16_t* L;
  const uint8_t* flags;
  uint64_t n;
  uint32_t use_window, lane_window, lane_states;
  double wave_min_width;  // (BuildOptions::wave_lane_min_width)
  uint32_t wave_lane_arcs;  // (BuildOptions::wave_lane_arcs)
};
// what a pair is: 0 = plain lane, w = windowed lane with a ring of w rows, 0xff = neither (the host builder's business);
// counts {plain, windowed, other, windowed AND above lane_states AND wide enough for a wavefront of its own}
__global__ void classify_kernel(ClassArgs C, uint8_t* win, unsigned long long* out) {
  unsigned long long v[4] = {0, 0, 0, 0};
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < C.n; p += (uint64_t)gridDim.x * blockDim.x) {
    uint8_t k = 0;
    if (C.flags[p] & PF_HAS) {
      const uint32_t S = C.S[p], E = C.E[p];
      const uint32_t w = window_of(C.span[p], S, C.use_window, C.lane_window);
      if (w) {
        k = (uint8_t)w;
        ++v[1];
        if (S > C.lane_states && C.L[p] > 1 && (double)E >= C.wave_min_width * (double)(C.L[p] - 1)) {
          ++v[3];
          if (E > C.wave_lane_arcs) ++v[2];  // (wide and long: a wavefront of its own whatever the corpus' size -- the host builder's)
        }
      } else if (S <= C.lane_states)
        ++v[0];
      else {
        k = 0xff;
        ++v[2];
      }
    }
    win[p] = k;
  }
  for (int q = 0; q < 4; ++q) {
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_down(v[q], o, 64);
    if ((threadIdx.x & 63) == 0 && v[q]) atomicAdd(out + q, v[q]);
  }
}

// sort key of a pair: plain lattices first -- more arcs first, then more states, then corpus order (the host's stable_sort)
// --, then the windowed ones -- wider ring first, then the same; pairs without a derivation last
__global__ void pair_key_kernel(const uint16_t* E, const uint16_t* S, const uint8_t* flags, const uint8_t* win, uint64_t n,
                                unsigned long long* key) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const unsigned long long w = win[p];
  key[p] = (flags[p] & PF_HAS) ? ((w ? (1ull << 62) | ((0x7full - w) << 55) : 0ull) | ((unsigned long long)(0x7ffu - E[p]) << 44) |
                                  ((unsigned long long)(0x3ffu - S[p]) << 34) | p)
                               : ~0ull;
}

// statistics: {pairs kept, fallback pairs, explored states, explored arcs, kept states, kept arcs, kept arcs of the lattices
// above win_min states}
__global__ void pair_stats_kernel(const uint16_t* E, const uint16_t* S, const uint8_t* flags, const uint32_t* xs, const uint32_t* xa,
                                  uint64_t n, uint32_t win_min, unsigned long long* out) {
  unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
    v[0] += (flags[p] & PF_HAS) ? 1 : 0;
    v[1] += (flags[p] & PF_FALLBACK) ? 1 : 0;
    v[2] += xs[p];
    v[3] += xa[p];
    v[4] += S[p];
    v[5] += E[p];
    v[6] += S[p] > win_min ? E[p] : 0;
  }
  for (int k = 0; k < 7; ++k) {
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    if ((threadIdx.x & 63) == 0 && v[k]) atomicAdd(out + k, v[k]);
  }
}

// lane slot -> index in the sorted pair list: the plain lattices fill groups [0, ng_plain), the windowed ones start a group
// of their own (lattice.cpp: a group never mixes the two); 2^63 = an empty slot
struct SlotMap {
  uint64_t n_plain, n_win;
  uint32_t ng_plain;
  __device__ __forceinline__ unsigned long long at(uint32_t group, uint32_t lane) const {
    if (group < ng_plain) {
      const uint64_t j = (uint64_t)group * 64 + lane;
      return j < n_plain ? j : ~0ull;
    }
    const uint64_t r = (uint64_t)(group - ng_plain) * 64 + lane;
    return r < n_win ? n_plain + r : ~0ull;
  }
};

// one wave per lane group: its descriptor (rows, widest lattice, widest ring)
__global__ __launch_bounds__(64) void group_dims_kernel(const unsigned long long* sorted_key, const uint16_t* E, const uint16_t* S,
                                                        const uint8_t* win, SlotMap M, uint32_t* g_maxlen, uint32_t* g_maxstates,
                                                        uint32_t* g_items, uint32_t* g_window) {
  const unsigned long long j = M.at(blockIdx.x, threadIdx.x);
  uint32_t e = 0, s = 0, w = 0;
  if (j != ~0ull) {
    const uint32_t p = (uint32_t)(sorted_key[j] & 0xffffffffull);
    e = E[p];
    s = S[p];
    w = win[p];
  }
  uint32_t me = e, ms = s, te = e, mw = w;
  for (int o = 32; o > 0; o >>= 1) {
    me = max(me, (uint32_t)__shfl_down(me, o, 64));
    ms = max(ms, (uint32_t)__shfl_down(ms, o, 64));
    mw = max(mw, (uint32_t)__shfl_down(mw, o, 64));
    te += (uint32_t)__shfl_down(te, o, 64);
  }
  if (threadIdx.x == 0) {
    g_maxlen[blockIdx.x] = (max(me, 1u) + LANE_CHUNK - 1) / LANE_CHUNK * LANE_CHUNK;
    g_maxstates[blockIdx.x] = ms;
    g_items[blockIdx.x] = te;
    g_window[blockIdx.x] = mw;
  }
}

// one wave per lane group: interleave the 64 record streams, emit the (arc, slot) items
__global__ __launch_bounds__(64) void interleave_kernel(const LaneGroup* groups, const unsigned long long* sorted_key, const uint16_t* E,
                                                        const uint16_t* S, const uint32_t* rec_fwd, const uint32_t* rec_bwd,
                                                        const uint32_t* rec_arc, const double* pair_weight, SlotMap M, uint32_t GK,
                                                        const unsigned long long* g_item_off, uint32_t* lane_fwdx, uint32_t* lane_bwd,
                                                        uint32_t* lane_pair, uint32_t* lane_nstates, double* lane_logw,
                                                        unsigned long long* items) {
  const LaneGroup g = groups[blockIdx.x];
  const uint32_t lane = threadIdx.x;
  const uint64_t j = (uint64_t)g.pair_base + lane;  // lane slot
  const unsigned long long sj = M.at(blockIdx.x, lane);
  uint32_t e = 0, p = 0;
  const bool active = lane < g.n_lanes && sj != ~0ull;
  if (active) {
    p = (uint32_t)(sorted_key[sj] & 0xffffffffull);
    e = E[p];
    lane_pair[j] = p;
    lane_nstates[j] = S[p];
    lane_logw[j] = pair_weight ? pair_weight[p] : 0.0;  // (ln of the pair's weight, taken on the host: the host builder's bits)
  } else if (lane < 64) {
    lane_pair[j] = 0xffffffffu;
    lane_nstates[j] = 0;
    lane_logw[j] = 0.0;
  }
  // item offset of this lane = the group's base + the arcs of the lanes before it
  uint32_t pre = e;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(pre, o, 64);
    if ((int)lane >= o) pre += v;
  }
  unsigned long long it = g_item_off[blockIdx.x] + (pre - e);
  if (!active) return;
  const uint32_t shift = g.maxlen - e;  // the backward stream is right-aligned in the group's rows
  uint32_t* f = lane_fwdx + g.stream_base + lane;
  uint32_t* b = lane_bwd + g.stream_base + lane;
  const uint32_t* rf = rec_fwd + (size_t)p * GK;
  const uint32_t* rb = rec_bwd + (size_t)p * GK;
  const uint32_t* ra = rec_arc + (size_t)p * GK;
  for (uint32_t k = 0; k < e; ++k) {
    const uint32_t x = rf[k];
    const uint32_t bp = ((x >> LANE_POS_SHIFT) & LANE_POS_MAX) + shift;
    f[(size_t)k * 64] = (x & ~(LANE_POS_MAX << LANE_POS_SHIFT)) | (bp << LANE_POS_SHIFT);
    b[(size_t)(shift + k) * 64] = rb[k];
    items[it + k] = ((unsigned long long)ra[k] << 32) | (unsigned long long)(g.stream_base + (uint64_t)(shift + k) * 64 + lane);
  }
}

// items sorted by (arc, slot): slot list, first item of every arc
__global__ void slots_kernel(const unsigned long long* items, uint64_t n, uint64_t n_arcs, uint64_t* slot_pos, uint64_t* arc_off) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = (uint32_t)(items[i] >> 32);
  slot_pos[i] = items[i] & 0xffffffffull;
  const uint32_t prev = i ? (uint32_t)(items[i - 1] >> 32) : 0u;
  for (uint64_t x = (i ? (uint64_t)prev + 1 : 0); x <= a; ++x) arc_off[x] = i;  // arcs without items in between start here too
  if (i + 1 == n)
    for (uint64_t x = (uint64_t)a + 1; x <= n_arcs; ++x) arc_off[x] = n;
}

// end (exclusive) of the transposition bucket that starts at arc a, for every arc that is not heavy (build_transpose's
// greedy rule: at most TRANS_BUCKET items and arcs, stop before a heavy arc)
__global__ void bucket_end_kernel(const uint64_t* arc_off, uint64_t n_arcs, uint32_t* b_end, uint32_t BK /* LatticeSet::bucket */) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_arcs) return;
  const uint64_t limit = arc_off[a] + BK;
  uint64_t lo = a, hi = min(n_arcs, a + (uint64_t)BK);  // largest e in [a, hi] with arc_off[e] <= limit
  while (lo < hi) {
    const uint64_t m = (lo + hi + 1) >> 1;
    if (arc_off[m] <= limit) lo = m; else hi = m - 1;
  }
  b_end[a] = (uint32_t)lo;
}
// the arcs with more than TRANS_HEAVY items (unordered; sorted afterwards)
__global__ void heavy_list_kernel(const uint64_t* arc_off, uint64_t n_arcs, uint32_t* list, uint32_t cap, uint32_t* n) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_arcs || arc_off[a + 1] - arc_off[a] <= TRANS_HEAVY) return;
  const uint32_t k = atomicAdd(n, 1u);
  if (k < cap) list[k] = (uint32_t)a;
}
// the greedy chain itself: sequential by definition, a few thousand steps
__global__ void bucket_chain_kernel(const uint64_t* arc_off, const uint32_t* b_end, uint64_t n_arcs, const uint32_t* heavy,
                                    uint32_t n_heavy, TransBucket* buckets, uint32_t* split_arcs, uint32_t cap,
                                    uint32_t* n_out /* buckets, split arcs, overflow */, uint32_t BK /* LatticeSet::bucket */) {
  if (blockIdx.x || threadIdx.x) return;
  uint64_t a = 0;
  uint32_t nb = 0, ns = 0, hp = 0;
  bool over = false;
  while (a < n_arcs) {
    const uint64_t c = arc_off[a + 1] - arc_off[a];
    if (c > TRANS_HEAVY) {
      const bool split = c > BK;
      if (split) {
        if (ns < cap) split_arcs[ns] = (uint32_t)a;
        ++ns;
      }
      for (uint64_t g = arc_off[a]; g < arc_off[a + 1]; g += BK) {
        if (nb < cap)
          buckets[nb] = TransBucket{g, (uint32_t)min((uint64_t)BK, arc_off[a + 1] - g), (uint32_t)a, 1u,
                                    TRANS_SINGLE | (split ? TRANS_SPLIT : 0u)};
        else
          over = true;
        ++nb;
      }
      ++a;
      continue;
    }
    // the greedy loop stops before a heavy arc: the first one at or after a, from the sorted list
    uint64_t e = b_end[a];
    while (hp < n_heavy && heavy[hp] < a) ++hp;
    if (hp < n_heavy && heavy[hp] < e) e = heavy[hp];
    if (e == a) e = a + 1;  // (cannot happen: a itself is not heavy and fits)
    if (nb < cap)
      buckets[nb] = TransBucket{arc_off[a], (uint32_t)(arc_off[e] - arc_off[a]), (uint32_t)a, (uint32_t)(e - a), 0u};
    else
      over = true;
    ++nb;
    a = e;
  }
  n_out[0] = nb;
  n_out[1] = ns;
  n_out[2] = over || ns > cap;
}

__global__ void a_off_kernel(const TransBucket* buckets, uint32_t n_buckets, const uint64_t* arc_off, uint16_t* a_off) {
  const TransBucket B = buckets[blockIdx.x];
  if (B.flags & TRANS_SINGLE) return;
  for (uint32_t a = threadIdx.x; a < B.n_arcs; a += blockDim.x) a_off[B.arc_lo + a] = (uint16_t)(arc_off[B.arc_lo + a] - B.item_base);
}

// per item (arc-sorted index i): the key (bucket, slot) under which the bucket-major order sorts it
__global__ void bucket_key_kernel(const TransBucket* buckets, uint32_t n_buckets, const uint64_t* slot_pos, uint64_t n,
                                  unsigned long long* key, uint32_t* val) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = n_buckets - 1;  // last bucket with item_base <= i
  while (lo < hi) {
    const uint32_t m = (lo + hi + 1) >> 1;
    if (buckets[m].item_base <= i) lo = m; else hi = m - 1;
  }
  key[i] = ((unsigned long long)lo << 32) | (unsigned long long)slot_pos[i];
  val[i] = (uint32_t)i;
}
// J = bucket-major index after that sort
__global__ void bucket_major_kernel(const TransBucket* buckets, const unsigned long long* key_sorted, const uint32_t* val_sorted,
                                    const unsigned long long* items_sorted, uint64_t n, uint32_t tile_sz, uint16_t* b_arc, uint16_t* b_rank,
                                    uint32_t* tile_key, uint32_t* J_val) {
  const uint64_t J = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (J >= n) return;
  const uint32_t b = (uint32_t)(key_sorted[J] >> 32);
  const uint32_t i = val_sorted[J];
  const TransBucket B = buckets[b];
  b_rank[J] = (uint16_t)(i - B.item_base);
  b_arc[J] = (B.flags & TRANS_SINGLE) ? (uint16_t)0 : (uint16_t)((uint32_t)(items_sorted[i] >> 32) - B.arc_lo);
  tile_key[J] = (uint32_t)((key_sorted[J] & 0xffffffffull) / tile_sz);
  J_val[J] = (uint32_t)J;
}
// I = tile-major index: stable sort of the bucket-major sequence by tile
__global__ void tile_major_kernel(const uint32_t* tile_sorted, const uint32_t* J_sorted, const unsigned long long* key_sorted, uint64_t n,
                                  uint64_t n_tiles, uint32_t tile_sz, uint32_t* t_src, uint16_t* t_pos, uint32_t* b_src, uint64_t* tile_base) {
  const uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= n) return;
  const uint32_t J = J_sorted[I], t = tile_sorted[I];
  t_src[I] = J;
  t_pos[I] = (uint16_t)((key_sorted[J] & 0xffffffffull) - (uint64_t)t * tile_sz);
  b_src[J] = (uint32_t)I;
  const uint32_t prev = I ? tile_sorted[I - 1] : 0u;
  for (uint64_t x = (I ? (uint64_t)prev + 1 : 0); x <= t; ++x) tile_base[x] = I;
  if (I + 1 == n)
    for (uint64_t x = (uint64_t)t + 1; x <= n_tiles; ++x) tile_base[x] = n;
}

// order-independent checksum of an array of 32-bit words: sum of mix(index, word)
__global__ void fingerprint_kernel(const uint32_t* w, uint64_t n, unsigned long long* out) {
  unsigned long long v = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long z = (k << 32) ^ w[k];
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    v += z ^ (z >> 31);
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

template <class K>
hipError_t sort_keys(DevBuf<char>& tmp, const K* in, K* out, uint64_t n, int end_bit, hipStream_t s) {
  size_t bytes = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, in, out, (int)n, 0, end_bit, s);
  if (e != hipSuccess) return e;
  if (tmp.n < bytes && (e = tmp.alloc(bytes)) != hipSuccess) return e;
  return hipcub::DeviceRadixSort::SortKeys(tmp.p, bytes, in, out, (int)n, 0, end_bit, s);
}
template <class K, class V>
hipError_t sort_pairs(DevBuf<char>& tmp, const K* kin, K* kout, const V* vin, V* vout, uint64_t n, int end_bit, hipStream_t s) {
  size_t bytes = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, (int)n, 0, end_bit, s);
  if (e != hipSuccess) return e;
  if (tmp.n < bytes && (e = tmp.alloc(bytes)) != hipSuccess) return e;
  return hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, kin, kout, vin, vout, (int)n, 0, end_bit, s);
}
static int bits_for(uint64_t v) {
  int b = 1;
  while (b < 64 && (v >> b)) ++b;
  return b;
}

}  // namespace

// ---- run-length form of the transposition's source indices (TransArgs::tr_* / br_*) ----
namespace {
__global__ void run_flags_kernel(const uint32_t* src, uint64_t n, uint32_t* flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || src[i] != src[i - 1] + 1u) ? 1u : 0u;
}
__global__ void run_starts_kernel(const uint64_t* starts, uint64_t n_seg, uint64_t n, uint32_t* flags) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_seg && starts[s] < n) flags[starts[s]] = 1u;
}
__global__ void run_emit_kernel(const uint32_t* src, const uint32_t* flags, const uint32_t* rid, const uint64_t* starts, uint64_t n_seg,
                                uint64_t n, uint16_t* run_rel, uint32_t* run_src) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flags[i]) return;
  uint64_t lo = 0, hi = n_seg - 1;  // last segment with starts[seg] <= i (empty segments share a start: the last one wins)
  while (lo < hi) {
    const uint64_t m = (lo + hi + 1) >> 1;
    if (starts[m] <= i) lo = m; else hi = m - 1;
  }
  run_rel[rid[i]] = (uint16_t)(i - starts[lo]);
  run_src[rid[i]] = src[i];
}
__global__ void run_off_kernel(const uint32_t* rid, const uint64_t* starts, uint64_t n_seg, uint64_t n, uint32_t total, uint32_t* seg_off,
                               uint32_t* max_runs) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n_seg) return;
  const uint32_t mine = (s == n_seg || starts[s] >= n) ? total : rid[starts[s]];
  seg_off[s] = mine;
  if (s < n_seg) {
    const uint32_t next = (s + 1 == n_seg || starts[s + 1] >= n) ? total : rid[starts[s + 1]];
    atomicMax(max_runs, next - mine);
  }
}
__global__ void bucket_starts_kernel(const TransBucket* b, uint32_t nb, uint64_t n, uint64_t* starts) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nb) starts[k] = b[k].item_base;
  if (k == nb) starts[k] = n;
}
int run_table(hipStream_t s, DevBuf<char>& tmp, const uint32_t* src, uint64_t n, const uint64_t* starts, uint64_t n_seg, DevBuf<uint32_t>& off,
              DevBuf<uint16_t>& rel, DevBuf<uint32_t>& rsrc, uint32_t& max_runs) {
  DevBuf<uint32_t> flags, rid, d_max;
  HIPCHK(flags.alloc(n + 1));
  HIPCHK(rid.alloc(n + 1));
  HIPCHK(d_max.alloc(1));
  HIPCHK(hipMemsetAsync(d_max.p, 0, 4, s));
  const unsigned g = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(run_flags_kernel, dim3(g), dim3(256), 0, s, src, n, flags.p);
  hipLaunchKernelGGL(run_starts_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, s, starts, n_seg, n, flags.p);
  size_t bytes = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flags.p, rid.p, (int)n, s));
  if (tmp.n < bytes) HIPCHK(tmp.alloc(bytes));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, flags.p, rid.p, (int)n, s));
  uint32_t last_rid = 0, last_flag = 0;
  HIPCHK(hipMemcpyAsync(&last_rid, rid.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(&last_flag, flags.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const uint32_t total = last_rid + last_flag;
  HIPCHK(off.alloc(n_seg + 1));
  HIPCHK(rel.alloc(total));
  HIPCHK(rsrc.alloc(total));
  hipLaunchKernelGGL(run_emit_kernel, dim3(g), dim3(256), 0, s, src, flags.p, rid.p, starts, n_seg, n, rel.p, rsrc.p);
  hipLaunchKernelGGL(run_off_kernel, dim3((unsigned)((n_seg + 256) / 256)), dim3(256), 0, s, rid.p, starts, n_seg, n, total, off.p, d_max.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(&max_runs, d_max.p, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}
}  // namespace

// the WFST arc of every item in tile-major order: item J of bucket B (bucket-major: where X holds its weight) belongs to arc
// B.arc_lo + b_arc[J] and lies at tile-major index b_src[J]
__global__ __launch_bounds__(1024) void tile_arc_kernel(const TransBucket* buckets, const uint16_t* b_arc, const uint32_t* b_src, uint32_t* t_arc) {
  const TransBucket B = buckets[blockIdx.x];
  for (uint32_t j = threadIdx.x; j < B.n_items; j += 1024) t_arc[b_src[B.item_base + j]] = B.arc_lo + b_arc[B.item_base + j];
}
// the item (tile-major index: its place in XC) of every wave position; idx[] arrives filled with 0xffffffff
__global__ __launch_bounds__(1024) void wave_item_kernel(const uint64_t* tile_base, const uint16_t* t_pos, uint32_t tile_first, uint32_t tile_sz,
                                                         uint64_t n_wave, uint32_t* idx) {
  const uint32_t tile = tile_first + blockIdx.x;
  const uint64_t i0 = tile_base[tile], i1 = tile_base[tile + 1];
  for (uint64_t i = i0 + threadIdx.x; i < i1; i += 1024) {
    const uint64_t p = (uint64_t)blockIdx.x * tile_sz + t_pos[i];
    if (p < n_wave) idx[p] = (uint32_t)i;
  }
}
// rows (64 positions) with an item, and those of them whose items lie within 128 places of XC of one another (a wavefront
// walks its share of the rows and adds its two totals once: same-address atomics serialise)
__global__ __launch_bounds__(256) void wave_row_span_kernel(const uint32_t* idx, uint64_t n_rows, unsigned long long* out) {
  const uint64_t w0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
  unsigned long long rows = 0, compact = 0;
  for (uint64_t row = w0; row < n_rows; row += nw) {
    const uint32_t q = idx[row * 64 + (threadIdx.x & 63)];
    uint32_t lo = q, hi = q == 0xffffffffu ? 0u : q;
    for (int o = 32; o; o >>= 1) {
      lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
      hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    if (lo != 0xffffffffu) {
      ++rows;
      if (hi - lo < 128u) ++compact;
    }
  }
  if ((threadIdx.x & 63) == 0 && rows) {
    atomicAdd(out, rows);
    atomicAdd(out + 1, compact);
  }
}
static int build_run_tables_impl(carmel_hip_trainer* t);
static int build_wave_items(carmel_hip_trainer* t);
// after either builder: the run tables, or the tiles' arc ids; the wave positions' items
int build_run_tables(carmel_hip_trainer* t) {
  t->t_t_arc.release();
  t->wave_xc_idx.release();
  int rc = build_run_tables_impl(t);
  if (rc) return rc;
  rc = build_wave_items(t);
  if (rc) return rc;
  struct Account {  // what this function leaves allocated counts as the lattices' device memory
    carmel_hip_trainer* t;
    ~Account() { t->device_bytes += t->t_t_arc.bytes() + t->wave_xc_idx.bytes() + t->tr_off.bytes() + t->tr_rel.bytes() + t->tr_src.bytes() +
                                    t->br_off.bytes() + t->br_rel.bytes() + t->br_src.bytes(); }
  } account{t};
  // A WFST whose weights the last-level cache holds (128 MB of them in its 256 MB), whose arcs lie in many lattices each (four
  // items an arc and more) and a transposition with per-item indices: the tile pass (or the tile sweep) can fetch a tile's
  // weights from the table itself -- through t_t_arc, the arc of every tile-major item -- instead of from X, and the bucket pass
  // that writes X (8 B per item written, 8 B read back from HBM where the table's lines come from the cache) has nothing left to
  // do.  Measured (tools/r5_tile_gather.sh): c4a (16 items an arc) 4.27 -> 3.96 ms an iteration, amb 1.21 -> 1.06; config 2
  // (0.56 items an arc: a line fetched per item for 8 bytes of it) 0.083 -> 0.087, hence the rule.
  // CARMEL_HIP_TILE_GATHER=0 / 1: never / whatever the sizes (A/B: the same values at the same places).
  if (!t->use_transpose || t->use_runs || !t->t_buckets.n || !t->t_t_src.n || !t->t_b_src.n || !t->t_b_arc.n) return CARMEL_HIP_OK;
  if (!t->wcache.n) return CARMEL_HIP_OK;  // (no lane record: no tile pass in the weights' direction)
  const char* env = lib_opt("tile_gather");
  if (env ? atoi(env) == 0 : (t->w.n_arcs * sizeof(double) > (128ull << 20) || t->t_t_src.n < 4 * t->w.n_arcs)) return CARMEL_HIP_OK;
  HIPCHK(t->t_t_arc.alloc(t->t_t_src.n));
  hipLaunchKernelGGL(tile_arc_kernel, dim3((unsigned)t->t_buckets.n), dim3(1024), 0, t->stream, t->t_buckets.p, t->t_b_arc.p, t->t_b_src.p,
                     t->t_t_arc.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(t->stream));
  if (lib_opt("timing"))
    fprintf(stderr, "timing: tile weights from the table: %zu items over %llu arcs\n", t->t_t_arc.n, (unsigned long long)t->w.n_arcs);
  return CARMEL_HIP_OK;
}
// One-per-wavefront lattices and nothing behind them (no bundle positions in their tiles), per-item indices: the sweep can send
// every posterior to its item's place in XC itself (sweep_wave_kernel<.., XD>) -- `post` is neither written nor read back and
// trans_c_tile has no wave tile to do -- where that scattered write fills lines: where most rows' items are neighbours in XC (a
// level's arcs are neighbours in the WFST, so they share a bucket and sit side by side in a tile's run: `long`, every row).
// CARMEL_HIP_WAVE_XC=0 / 1: never / whatever the rows look like (A/B: the same values at the same places of XC).
static int build_wave_items(carmel_hip_trainer* t) {
  if (!t->use_transpose || t->use_runs || !t->wave_records || t->out_arcs.n || !t->t_tile_base.n || !t->t_t_pos.n) return CARMEL_HIP_OK;
  const char* env = lib_opt("wave_xc");
  if (env && atoi(env) == 0) return CARMEL_HIP_OK;
  const uint32_t tile = t->lat.tile;
  if (!tile || t->wave_slot_base % tile) return CARMEL_HIP_OK;
  const uint32_t first = (uint32_t)(t->wave_slot_base / tile), n_tiles = (uint32_t)(t->t_tile_base.n - 1);
  if (first >= n_tiles) return CARMEL_HIP_OK;
  hipStream_t s = t->stream;
  HIPCHK(t->wave_xc_idx.alloc(t->wave_records));
  HIPCHK(hipMemsetAsync(t->wave_xc_idx.p, 0xff, t->wave_xc_idx.bytes(), s));
  hipLaunchKernelGGL(wave_item_kernel, dim3(n_tiles - first), dim3(1024), 0, s, t->t_tile_base.p, t->t_t_pos.p, first, tile, (uint64_t)t->wave_records,
                     t->wave_xc_idx.p);
  DevBuf<unsigned long long> d;
  HIPCHK(d.alloc(2));
  HIPCHK(hipMemsetAsync(d.p, 0, 16, s));
  const uint64_t n_rows = t->wave_records / 64;
  hipLaunchKernelGGL(wave_row_span_kernel, dim3((unsigned)std::min<uint64_t>((n_rows + 3) / 4, 2048)), dim3(256), 0, s, t->wave_xc_idx.p, n_rows, d.p);
  unsigned long long h[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const bool on = env ? true : (h[0] && h[1] * 4 >= h[0] * 3);
  if (lib_opt("timing"))
    fprintf(stderr, "timing: wave posteriors straight to XC: %llu of %llu rows compact -> %s\n", h[1], h[0], on ? "on" : "off");
  if (!on) t->wave_xc_idx.release();
  return CARMEL_HIP_OK;
}
// derive the run tables from t_t_src / t_b_src (which stay, for the A/B switch and the checksums)
static int build_run_tables_impl(carmel_hip_trainer* t) {
  t->use_runs = false;
  if (!t->use_transpose || !t->t_buckets.n || !t->t_t_src.n) return CARMEL_HIP_OK;
  // On where the runs are long enough to pay and the corpus is large enough for its traffic to matter (config 4: twelve
  // items per run; 0.16 GB of 2.41 less per E-step, trans_w_tile 108 -> 99 us, trans_c_bucket unchanged); corpora whose items
  // spread over more (tile, bucket) cells than a quarter of their items keep the per-item indices, and so do small ones
  // (config 2: the three extra barriers cost more than the bytes).  CARMEL_HIP_TRANS_RUNS=0 / 1 forces it off / on.
  const char* env = lib_opt("trans_runs");
  if (env && atoi(env) == 0) return CARMEL_HIP_OK;
  {
    const double tiles = (double)(t->t_tile_base.n - 1), cells = tiles * (double)t->t_buckets.n;
    if (!(env && atoi(env) == 1) && ((double)t->t_t_src.n < 4.0 * cells || tiles < 512.0)) return CARMEL_HIP_OK;
    // a corpus whose items lie mostly in one-per-wavefront lattices over a table the cache holds: the wave sweeps' gathered weights
    // and direct posteriors, and the tile passes' weights from the table (below), want per-item indices and are worth more than the
    // runs -- `mix` (84 % of its arcs in long lattices beside lane groups): E-step 1.99 ms with runs, 1.72 without
    if (!env && t->wave_bwd_arc.n && (double)t->wave_records * 2.0 > (double)t->t_t_src.n) return CARMEL_HIP_OK;
  }
  hipStream_t s = t->stream;
  const uint64_t n = t->t_t_src.n;
  const uint64_t n_tiles = t->t_tile_base.n - 1;
  const uint32_t nb = (uint32_t)t->t_buckets.n;
  DevBuf<char> tmp;
  DevBuf<uint64_t> bstarts;
  HIPCHK(bstarts.alloc((size_t)nb + 1));
  hipLaunchKernelGGL(bucket_starts_kernel, dim3((nb + 256) / 256), dim3(256), 0, s, t->t_buckets.p, nb, n, bstarts.p);
  uint32_t max_t = 0, max_b = 0;
  int rc = run_table(s, tmp, t->t_t_src.p, n, t->t_tile_base.p, n_tiles, t->tr_off, t->tr_rel, t->tr_src, max_t);
  if (rc) return rc;
  rc = run_table(s, tmp, t->t_b_src.p, n, bstarts.p, nb, t->br_off, t->br_rel, t->br_src, max_b);
  if (rc) return rc;
  t->use_runs = max_t <= TRANS_RUN_CAP && max_b <= TRANS_RUN_CAP && max_t > 0 && max_b > 0;
  if (!t->use_runs) {
    t->tr_off.release();
    t->tr_rel.release();
    t->tr_src.release();
    t->br_off.release();
    t->br_rel.release();
    t->br_src.release();
  }
  if (lib_opt("timing"))
    fprintf(stderr, "timing: transposition runs: %zu per %llu items, at most %u per tile / %u per bucket -> %s\n", t->tr_src.n,
            (unsigned long long)n, max_t, max_b, t->use_runs ? "run-length indices" : "per-item indices");
  return CARMEL_HIP_OK;
}

// checksums of the lattice image in device memory: lets a test assert that the GPU builder and the host builder leave
// the very same bytes behind.  out[16].
int carmel_hip_debug_lattice_fingerprint_impl(carmel_hip_trainer* t, uint64_t* out) {
  hipStream_t s = t->stream;
  DevBuf<unsigned long long> acc;
  HIPCHK(acc.alloc(16));
  HIPCHK(hipMemsetAsync(acc.p, 0, 16 * 8, s));
  auto fp = [&](int k, const void* p, size_t bytes) {
    if (p && bytes >= 4) hipLaunchKernelGGL(fingerprint_kernel, dim3(1024), dim3(256), 0, s, (const uint32_t*)p, (uint64_t)(bytes / 4), acc.p + k);
  };
  fp(0, t->lane_groups.p, t->lane_groups.bytes());
  fp(1, t->lane_fwdx.p, t->lane_fwdx.bytes());
  fp(2, t->lane_bwd.p, t->lane_bwd.bytes());
  fp(3, t->lane_pair.p, t->lane_pair.bytes());
  fp(4, t->lane_nstates.p, t->lane_nstates.bytes());
  fp(5, t->lane_logw.p, t->lane_logw.bytes());
  fp(6, t->t_buckets.p, t->t_buckets.bytes());
  fp(7, t->t_tile_base.p, t->t_tile_base.bytes());
  fp(8, t->t_b_arc.p, t->t_b_arc.bytes() & ~(size_t)3);
  fp(9, t->t_b_rank.p, t->t_b_rank.bytes() & ~(size_t)3);
  fp(10, t->t_b_src.p, t->t_b_src.bytes());
  fp(11, t->t_t_pos.p, t->t_t_pos.bytes() & ~(size_t)3);
  fp(12, t->t_t_src.p, t->t_t_src.bytes());
  fp(13, t->t_a_off.p, t->t_a_off.bytes() & ~(size_t)3);
  fp(14, t->t_split_arcs.p, t->t_split_arcs.bytes());
  fp(15, t->pair_w.p, t->pair_w.bytes());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, acc.p, 16 * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}

// Returns CARMEL_HIP_OK with done = true when the lattices were built on the GPU; done = false: not a case for this
// builder (the caller runs the host builder).
// The tables of the blocked transposition from the (arc << 32 | slot) items of a lattice set, on the device: slots by arc
// (arc_off, slot_pos), the buckets, the bucket-major and tile-major orders.  Shared by the device builder below and by the
// host builder's layouts (one-per-wavefront lattices, bundles: gpu_tables_for_host_layout) -- the host's own counting sorts of
// the same (lattice.cpp build_transpose) give the same bytes.  n_sort >= n_items keys are sorted; the ones past n_items carry
// the arc id n_arcs (slots that hold no arc) and are ignored afterwards.
static int gpu_tables_from_items(carmel_hip_trainer* t, DevBuf<char>& tmp, DevBuf<unsigned long long>& items, DevBuf<unsigned long long>& items_sorted,
                                 uint64_t n_items, uint64_t n_sort, uint64_t n_post, const std::function<void(const char*)>& lap) {
  const HostWfst& w = t->w;
  const LatticeSet& L = t->lat;
  hipStream_t s = t->stream;
  // ---- slots by arc ----
  HIPCHK(sort_keys(tmp, items.p, items_sorted.p, n_sort, 32 + bits_for(w.n_arcs + 1), s));
  HIPCHK(t->arc_off.alloc(w.n_arcs + 1));
  HIPCHK(t->slot_pos.alloc(n_items));
  hipLaunchKernelGGL(slots_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, items_sorted.p, n_items, w.n_arcs,
                     t->slot_pos.p, t->arc_off.p);
  t->hot_chunks.release();  // (the gather formulation is not offered on top of this builder)
  lap("slots by arc");
  // ---- transposition tables ----
  DevBuf<uint32_t> b_end, d_nout;
  HIPCHK(b_end.alloc(w.n_arcs));
  HIPCHK(d_nout.alloc(4));
  hipLaunchKernelGGL(bucket_end_kernel, dim3((unsigned)((w.n_arcs + 255) / 256)), dim3(256), 0, s, t->arc_off.p, w.n_arcs, b_end.p, L.bucket);
  const uint32_t cap = (uint32_t)(2 * (n_items / L.bucket) + w.n_arcs / L.bucket + 2 * (n_items / TRANS_HEAVY) + 16);
  HIPCHK(t->t_buckets.alloc(cap));
  HIPCHK(t->t_split_arcs.alloc(cap));
  DevBuf<uint32_t> heavy, heavy_sorted;
  HIPCHK(heavy.alloc(cap));
  HIPCHK(heavy_sorted.alloc(cap));
  HIPCHK(hipMemsetAsync(d_nout.p, 0, 16, s));
  hipLaunchKernelGGL(heavy_list_kernel, dim3((unsigned)((w.n_arcs + 255) / 256)), dim3(256), 0, s, t->arc_off.p, w.n_arcs, heavy.p, cap,
                     d_nout.p + 3);
  uint32_t n_heavy = 0;
  HIPCHK(hipMemcpyAsync(&n_heavy, d_nout.p + 3, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (n_heavy > cap) return fail(CARMEL_HIP_ERR_HIP, "gpu lattice build: heavy-arc list overflow");
  if (n_heavy) HIPCHK(sort_keys(tmp, heavy.p, heavy_sorted.p, n_heavy, 32, s));
  hipLaunchKernelGGL(bucket_chain_kernel, dim3(1), dim3(1), 0, s, t->arc_off.p, b_end.p, w.n_arcs, heavy_sorted.p, n_heavy,
                     t->t_buckets.p, t->t_split_arcs.p, cap, d_nout.p, L.bucket);
  uint32_t h_nout[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(h_nout, d_nout.p, 12, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (h_nout[2]) return fail(CARMEL_HIP_ERR_HIP, "gpu lattice build: bucket table overflow");
  const uint32_t n_buckets = h_nout[0], n_split = h_nout[1];
  t->t_buckets.n = n_buckets;       // (capacity stays; .n is what the engine reads)
  t->t_split_arcs.n = n_split;
  lap("buckets");
  const uint64_t n_tiles = (n_post + L.tile - 1) / L.tile;
  HIPCHK(t->t_a_off.alloc(w.n_arcs));
  HIPCHK(hipMemsetAsync(t->t_a_off.p, 0, w.n_arcs * 2, s));
  hipLaunchKernelGGL(a_off_kernel, dim3(n_buckets), dim3(256), 0, s, t->t_buckets.p, n_buckets, t->arc_off.p, t->t_a_off.p);
  DevBuf<unsigned long long> bkey, bkey_sorted;
  DevBuf<uint32_t> bval, bval_sorted, tile_key, tile_sorted, Jv, J_sorted;
  HIPCHK(bkey.alloc(n_items));
  HIPCHK(bkey_sorted.alloc(n_items));
  HIPCHK(bval.alloc(n_items));
  HIPCHK(bval_sorted.alloc(n_items));
  hipLaunchKernelGGL(bucket_key_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, t->t_buckets.p, n_buckets,
                     t->slot_pos.p, n_items, bkey.p, bval.p);
  HIPCHK(sort_pairs(tmp, bkey.p, bkey_sorted.p, bval.p, bval_sorted.p, n_items, 32 + bits_for(n_buckets), s));
  HIPCHK(t->t_b_arc.alloc(n_items));
  HIPCHK(t->t_b_rank.alloc(n_items));
  HIPCHK(t->t_b_src.alloc(n_items));
  HIPCHK(t->t_t_pos.alloc(n_items));
  HIPCHK(t->t_t_src.alloc(n_items));
  HIPCHK(t->t_tile_base.alloc(n_tiles + 1));
  HIPCHK(tile_key.alloc(n_items));
  HIPCHK(tile_sorted.alloc(n_items));
  HIPCHK(Jv.alloc(n_items));
  HIPCHK(J_sorted.alloc(n_items));
  hipLaunchKernelGGL(bucket_major_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, t->t_buckets.p, bkey_sorted.p,
                     bval_sorted.p, items_sorted.p, n_items, L.tile, t->t_b_arc.p, t->t_b_rank.p, tile_key.p, Jv.p);
  HIPCHK(sort_pairs(tmp, tile_key.p, tile_sorted.p, Jv.p, J_sorted.p, n_items, bits_for(n_tiles), s));
  hipLaunchKernelGGL(tile_major_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, tile_sorted.p, J_sorted.p,
                     bkey_sorted.p, n_items, n_tiles, L.tile, t->t_t_src.p, t->t_t_pos.p, t->t_b_src.p, t->t_tile_base.p);
  HIPCHK(hipGetLastError());
  HIPCHK(t->t_x.alloc(n_items));
  HIPCHK(t->t_xc.alloc(n_items));
  lap("transposition tables");
  return CARMEL_HIP_OK;
}

// ... for a layout of the HOST builder (lattice.cpp with BuildOptions::device_tables: one-per-wavefront lattices, bundles,
// mixtures): the arc of every posterior slot -- [lane records | padding to a tile | wave records | bundle out-arcs] -- goes to
// the device, the items are formed and sorted here.  lane_arc / wave_arc: 0xffffffff where a record holds no arc.
__global__ void host_layout_items_kernel(const uint32_t* lane_arc, uint64_t n_lane, const uint32_t* wave_arc, uint64_t wave_base, uint64_t n_wave,
                                         const uint2* out_arcs, uint64_t n_out, uint32_t n_arcs, unsigned long long* items,
                                         unsigned long long* n_valid) {
  const uint64_t n_post = wave_base + n_wave + n_out;
  unsigned long long mine = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_post; k += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t a = 0xffffffffu;
    if (k < n_lane)
      a = lane_arc[k];
    else if (k >= wave_base && k < wave_base + n_wave)
      a = wave_arc[k - wave_base];
    else if (k >= wave_base + n_wave)
      a = out_arcs[k - wave_base - n_wave].y;
    const bool ok = a != 0xffffffffu;
    items[k] = ((unsigned long long)(ok ? a : n_arcs) << 32) | (unsigned long long)k;
    mine += ok;
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_valid, mine);
}
int gpu_tables_for_host_layout(carmel_hip_trainer* t, const std::vector<uint32_t>& lane_arc, const std::vector<uint32_t>& wave_arc) {
  const LatticeSet& L = t->lat;
  hipStream_t s = t->stream;
  const bool timing = lib_opt("timing") != nullptr;
  auto last = std::chrono::steady_clock::now();
  std::function<void(const char*)> lap = [&](const char* what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "timing: gpu lattice build: %-34s %.4f s\n", what, std::chrono::duration<double>(now - last).count());
    last = now;
  };
  const uint64_t n_lane = lane_arc.size(), n_wave = wave_arc.size(), n_out = L.out_arcs.size(), n_post = L.n_post;
  if (n_post != L.wave_slot_base + n_wave + n_out || n_post >= (1ull << 32) || !n_post) return fail(CARMEL_HIP_ERR_ARG, "gpu_tables_for_host_layout: inconsistent layout");
  DevBuf<uint32_t> d_lane, d_wave;
  DevBuf<unsigned long long> items, items_sorted, d_n;
  DevBuf<char> tmp;
  HIPCHK(d_lane.upload(lane_arc, s));
  HIPCHK(d_wave.upload(wave_arc, s));
  HIPCHK(items.alloc(n_post));
  HIPCHK(items_sorted.alloc(n_post));
  HIPCHK(d_n.alloc(1));
  HIPCHK(hipMemsetAsync(d_n.p, 0, 8, s));
  hipLaunchKernelGGL(host_layout_items_kernel, dim3((unsigned)std::min<uint64_t>((n_post + 255) / 256, 65536)), dim3(256), 0, s, d_lane.p, n_lane,
                     d_wave.p, L.wave_slot_base, n_wave, (const uint2*)t->out_arcs.p, n_out, (uint32_t)t->w.n_arcs, items.p, d_n.p);
  unsigned long long n_items = 0;
  HIPCHK(hipMemcpyAsync(&n_items, d_n.p, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  lap("items of the host layout");
  if (!n_items) return fail(CARMEL_HIP_ERR_ARG, "gpu_tables_for_host_layout: no arc in any lattice");
  return gpu_tables_from_items(t, tmp, items, items_sorted, n_items, n_post, n_post, lap);
}

int gpu_build_lattices(carmel_hip_trainer* t, const BuildOptions& opt, uint8_t* has_derivation, carmel_hip_lattice_stats* stats,
                       bool& done) {
  done = false;
  const HostWfst& w = t->w;
  const HostCorpus& c = t->corpus;
  const uint64_t np = c.n_pairs;
  const GCaps caps = opt.gpu_large_caps ? G_LARGE : G_SMALL;
  const uint32_t GX = caps.cx, GE = caps.ce, GK = caps.ck, GH = caps.ch;
  if (!np || !opt.lane_states || opt.lane_states > GX || !opt.prune) return CARMEL_HIP_OK;
  if (np >= 0xfffffff0ull || w.n_arcs >= 0xfffffff0ull) return CARMEL_HIP_OK;
  const auto t0 = std::chrono::steady_clock::now();
  const bool timing = lib_opt("timing") != nullptr;
  auto lap = [&, last = t0](const char* what) mutable {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "timing: gpu lattice build: %-34s %.4f s\n", what, std::chrono::duration<double>(now - last).count());
    last = now;
  };
  hipStream_t s = t->stream;
  // ---- inputs to the device ----
  DevBuf<uint64_t> d_idx_off, d_idx_key, d_in_off, d_out_off;
  DevBuf<uint32_t> d_idx_arc, d_dst, d_in, d_out;
  DevBuf<double> d_pw;
  HIPCHK(d_idx_off.upload(w.idx_off, s));
  HIPCHK(d_idx_key.upload(w.idx_key, s));
  HIPCHK(d_idx_arc.upload(w.idx_arc, s));
  HIPCHK(d_dst.upload(w.dst, s));
  HIPCHK(d_in_off.upload(c.in_off, s));
  HIPCHK(d_out_off.upload(c.out_off, s));
  {
    std::vector<uint32_t> a = c.in_sym, b = c.out_sym;
    if (a.empty()) a.push_back(0);
    if (b.empty()) b.push_back(0);
    HIPCHK(d_in.upload(a, s));
    HIPCHK(d_out.upload(b, s));
  }
  {
    std::vector<double> lw(c.weight.size());
    for (size_t k = 0; k < lw.size(); ++k) lw[k] = c.weight[k] > 0 ? std::log(c.weight[k]) : -std::numeric_limits<double>::infinity();
    HIPCHK(d_pw.upload(lw, s));
  }
  // ---- one thread per pair: explore, prune, levels, lane records ----
  const uint32_t stride = (uint32_t)(((4 * (6 * GX + GE) + 2 * (GH + 3 * GX + 2 * GE + 6 * (GE + 2)) + 3 * GX) + 63) / 64 * 64);
  const uint32_t n_threads = (uint32_t)std::min<uint64_t>(caps.threads, (np + 255) / 256 * 256);
  DevBuf<uint8_t> scratch, pp_flags, pp_win;
  DevBuf<uint16_t> pp_E, pp_S, pp_span, pp_L;
  HIPCHK(pp_span.alloc(np));
  HIPCHK(pp_L.alloc(np));
  HIPCHK(pp_win.alloc(np));
  DevBuf<uint32_t> pp_xs, pp_xa, rec_fwd, rec_bwd, rec_arc;
  HIPCHK(scratch.alloc((size_t)n_threads * stride));
  HIPCHK(pp_flags.alloc(np));
  HIPCHK(pp_E.alloc(np));
  HIPCHK(pp_S.alloc(np));
  HIPCHK(pp_xs.alloc(np));
  HIPCHK(pp_xa.alloc(np));
  HIPCHK(rec_fwd.alloc(np * GK));
  HIPCHK(rec_bwd.alloc(np * GK));
  HIPCHK(rec_arc.alloc(np * GK));
  GArgs G;
  G.idx_off = d_idx_off.p;
  G.idx_key = d_idx_key.p;
  G.idx_arc = d_idx_arc.p;
  G.arc_dst = d_dst.p;
  G.final_state = w.final_state;
  G.in_off = d_in_off.p;
  G.out_off = d_out_off.p;
  G.in_sym = d_in.p;
  G.out_sym = d_out.p;
  G.n_pairs = np;
  G.lane_states = opt.lane_states;
  G.cx = GX;
  G.ce = GE;
  G.ck = GK;
  G.ch = GH;
  G.span_min = opt.lane_window ? opt.lane_window_min : 0xffffffffu;
  G.pp_span = pp_span.p;
  G.pp_L = pp_L.p;
  G.scratch = scratch.p;
  G.scratch_stride = stride;
  G.pp_E = pp_E.p;
  G.pp_S = pp_S.p;
  G.pp_flags = pp_flags.p;
  G.pp_xs = pp_xs.p;
  G.pp_xa = pp_xa.p;
  G.rec_fwd = rec_fwd.p;
  G.rec_bwd = rec_bwd.p;
  G.rec_arc = rec_arc.p;
  lap("uploads");
  hipLaunchKernelGGL(explore_kernel, dim3(n_threads / 256), dim3(256), 0, s, G);
  HIPCHK(hipGetLastError());
  DevBuf<unsigned long long> d_stats;
  HIPCHK(d_stats.alloc(8));
  HIPCHK(hipMemsetAsync(d_stats.p, 0, 64, s));
  hipLaunchKernelGGL(pair_stats_kernel, dim3(1024), dim3(256), 0, s, pp_E.p, pp_S.p, pp_flags.p, pp_xs.p, pp_xa.p, np, opt.lane_window_min, d_stats.p);
  unsigned long long hs[8];
  HIPCHK(hipMemcpyAsync(hs, d_stats.p, 64, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  lap("explore + prune + records");
  if (hs[1]) return CARMEL_HIP_OK;  // some pair is not a case for this builder: the host builder does the whole corpus
  const uint64_t n_kept = hs[0], n_items = hs[5];
  if (!n_kept || n_items >= (1ull << 32) || np >= (1ull << 32)) return CARMEL_HIP_OK;
  // a corpus with a tenth of its arcs in lattices above lane_window_min states gets windowed lane groups (lattice.cpp, the
  // same rule); what every pair is -- plain lane, windowed lane with a ring of w rows, neither -- follows from its span
  const uint32_t use_window = (opt.lane_window && hs[6] * 10 >= hs[5]) ? 1u : 0u;
  unsigned long long hc[4] = {0, 0, 0, 0};
  {
    ClassArgs CA;
    CA.E = pp_E.p;
    CA.S = pp_S.p;
    CA.span = pp_span.p;
    CA.L = pp_L.p;
    CA.flags = pp_flags.p;
    CA.n = np;
    CA.use_window = use_window;
    CA.lane_window = opt.lane_window;
    CA.lane_states = opt.lane_states;
    CA.wave_min_width = opt.wave_lane_min_width;
    CA.wave_lane_arcs = opt.wave ? opt.wave_lane_arcs : 0xffffffffu;
    HIPCHK(hipMemsetAsync(d_stats.p, 0, 64, s));
    hipLaunchKernelGGL(classify_kernel, dim3(1024), dim3(256), 0, s, CA, pp_win.p, d_stats.p);
    HIPCHK(hipMemcpyAsync(hc, d_stats.p, 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  const uint64_t n_plain = hc[0], n_win = hc[1];
  if (hc[2]) return CARMEL_HIP_OK;  // lattices no lane takes: bundles / one-per-wavefront lattices are laid out on the host
  // (the host builder gives wide windowed lattices a wavefront of their own when the corpus is too small to fill the chip
  // one per lane: lattice.cpp, BuildOptions::wave_lane_threshold)
  if (opt.wave && hc[3] && n_plain + n_win < opt.wave_lane_threshold) return CARMEL_HIP_OK;
  // ---- lane groups ----
  DevBuf<unsigned long long> key, key_sorted;
  DevBuf<char> tmp;
  HIPCHK(key.alloc(np));
  HIPCHK(key_sorted.alloc(np));
  hipLaunchKernelGGL(pair_key_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, pp_E.p, pp_S.p, pp_flags.p, pp_win.p, np, key.p);
  HIPCHK(sort_keys(tmp, key.p, key_sorted.p, np, 64, s));
  const size_t ng_plain = (size_t)((n_plain + 63) / 64), ng = ng_plain + (size_t)((n_win + 63) / 64);
  SlotMap SM;
  SM.n_plain = n_plain;
  SM.n_win = n_win;
  SM.ng_plain = (uint32_t)ng_plain;
  DevBuf<uint32_t> g_maxlen, g_maxstates, g_items, g_window;
  HIPCHK(g_maxlen.alloc(ng));
  HIPCHK(g_maxstates.alloc(ng));
  HIPCHK(g_items.alloc(ng));
  HIPCHK(g_window.alloc(ng));
  hipLaunchKernelGGL(group_dims_kernel, dim3((unsigned)ng), dim3(64), 0, s, key_sorted.p, pp_E.p, pp_S.p, pp_win.p, SM, g_maxlen.p,
                     g_maxstates.p, g_items.p, g_window.p);
  std::vector<uint32_t> h_maxlen(ng), h_maxstates(ng), h_items(ng), h_window(ng);
  HIPCHK(hipMemcpyAsync(h_maxlen.data(), g_maxlen.p, ng * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h_maxstates.data(), g_maxstates.p, ng * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h_items.data(), g_items.p, ng * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h_window.data(), g_window.p, ng * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  LatticeSet& L = t->lat;
  L = LatticeSet();
  L.lane_groups.resize(ng);
  std::vector<unsigned long long> h_item_off(ng);
  uint64_t spill_rows = 0;
  {
    unsigned long long acc = 0;
    for (size_t g = 0; g < ng; ++g) {
      LaneGroup& Gd = L.lane_groups[g];
      std::memset(&Gd, 0, sizeof Gd);
      Gd.n_lanes = (uint32_t)(g < ng_plain ? std::min<uint64_t>(64, n_plain - g * 64) : std::min<uint64_t>(64, n_win - (g - ng_plain) * 64));
      Gd.pair_base = (uint32_t)(g * 64);
      Gd.maxlen = h_maxlen[g];
      const uint32_t win = g < ng_plain ? 0u : h_window[g];
      Gd.max_states = win ? win : h_maxstates[g];
      Gd.window = win;
      if (win) {
        if (spill_rows + h_maxstates[g] > 0xffffffffull) return CARMEL_HIP_OK;
        Gd.spill_row = (uint32_t)spill_rows;
        spill_rows += h_maxstates[g];
      }
      h_item_off[g] = acc;
      acc += h_items[g];
    }
  }
  L.lane_spill_rows = spill_rows;
  const uint64_t n_rec = assign_lane_classes(L, opt, true);  // launch classes, pieces, stream bases: the host builder's own rule
  if (n_rec >= (1ull << 32)) return CARMEL_HIP_OK;
  lap("pair sort + lane groups");
  // ---- device image: lane streams ----
  HIPCHK(t->lane_groups.upload(L.lane_groups, s));
  if (L.tile_sweep)
    HIPCHK(t->tile_group.upload(L.tile_group, s));
  else
    t->tile_group.release();
  DevBuf<unsigned long long> d_item_off, items, items_sorted;
  HIPCHK(d_item_off.upload(h_item_off, s));
  HIPCHK(t->lane_fwdx.alloc(n_rec));
  t->lane_fwd.release();
  HIPCHK(t->lane_bwd.alloc(n_rec));
  HIPCHK(hipMemsetAsync(t->lane_fwdx.p, 0, n_rec * 4, s));
  HIPCHK(hipMemsetAsync(t->lane_bwd.p, 0, n_rec * 4, s));
  HIPCHK(t->lane_pair.alloc(ng * 64));
  HIPCHK(t->lane_nstates.alloc(ng * 64));
  HIPCHK(t->lane_logw.alloc(ng * 64));
  HIPCHK(items.alloc(n_items));
  HIPCHK(items_sorted.alloc(n_items));
  hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)ng), dim3(64), 0, s, t->lane_groups.p, key_sorted.p, pp_E.p, pp_S.p, rec_fwd.p,
                     rec_bwd.p, rec_arc.p, d_pw.p, SM, GK, d_item_off.p, t->lane_fwdx.p, t->lane_bwd.p, t->lane_pair.p,
                     t->lane_nstates.p, t->lane_logw.p, items.p);
  if (spill_rows)
    HIPCHK(t->lane_spill.alloc(spill_rows * 64));
  else
    t->lane_spill.release();
  HIPCHK(hipGetLastError());
  if (L.tile_sweep) {
    HIPCHK(t->lane_rec2.alloc(n_rec));
    HIPCHK(t->lane_chain.alloc(ng));
    HIPCHK(t->tile_chain.alloc(t->tile_group.n - 1));
    HIPCHK(hipMemsetAsync(t->lane_rec2.p, 0, n_rec * 4, s));
    HIPCHK(launch_pack_tile_records(t->lane_groups.p, (uint32_t)t->lane_groups.n, t->lane_nstates.p, t->lane_fwdx.p, t->lane_bwd.p, t->lane_rec2.p, t->lane_chain.p,
                                    t->tile_group.p, (uint32_t)(t->tile_group.n - 1), t->tile_chain.p, s));
    if (lib_opt("timing")) {
      std::vector<uint32_t> ch(t->lane_chain.n);
      HIPCHK(hipMemcpyAsync(ch.data(), t->lane_chain.p, ch.size() * 4, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      size_t n1 = 0;
      for (uint32_t c : ch) n1 += c & 1u;  // (bit 0; the padding field sits above it)
      fprintf(stderr, "timing: tile sweep: %zu tiles, %zu groups, %zu of them single paths\n", L.tile_group.size() - 1, ch.size(), n1);
    }
  } else {
    t->lane_rec2.release();
    t->lane_chain.release();
    t->tile_chain.release();
  }
  lap("record streams");
  {
    int rc = gpu_tables_from_items(t, tmp, items, items_sorted, n_items, n_items, /*n_post: no bundle arcs on this path*/ n_rec, lap);
    if (rc) return rc;
  }
  const uint64_t n_post = n_rec;
  // ---- the rest of the trainer's image ----
  t->use_transpose = true;
  t->lane_records = n_rec;
  t->wave_descs.release();  // (a one-per-lane corpus has no one-per-wavefront lattices)
  t->wave_fwd.release();
  t->wave_bwd.release();
  t->wave_bwd_arc.release();
  t->wave_level_off.release();
  t->wave_frow.release();
  t->wave_brow.release();
  t->wave_spill.release();
  t->wave_slot_base = n_rec;
  t->wave_records = 0;
  if (L.lane_fused)  // (allocated on first use by the A/B switch's three kernels: engine.cpp)
    t->post.release();
  else
    HIPCHK(t->post.alloc(n_post));
  HIPCHK(t->wcache.alloc(n_rec));
  t->bundles.release();
  t->in_arcs.release();
  t->out_arcs.release();
  t->in_off.release();
  t->out_off.release();
  t->level_off.release();
  t->pair_start.release();
  t->pair_final.release();
  t->pair_id.release();
  t->pair_logw.release();
  t->alpha_g.release();
  t->beta_g.release();
  std::vector<uint8_t> flags(np);
  HIPCHK(hipMemcpyAsync(flags.data(), pp_flags.p, np, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  L.has_deriv.resize(np);
  std::vector<double> pw(np);
  for (uint64_t p = 0; p < np; ++p) {
    L.has_deriv[p] = (flags[p] & PF_HAS) ? 1 : 0;
    pw[p] = L.has_deriv[p] ? (c.weight.empty() ? 1.0 : c.weight[p]) : -1.0;
  }
  if (has_derivation) std::memcpy(has_derivation, L.has_deriv.data(), np);
  HIPCHK(t->pair_w.upload(pw, s));
  HIPCHK(t->scalar_partial.alloc(3 * 256));
  HIPCHK(t->pair_logprob.alloc(np));
  HIPCHK(launch_fill(t->pair_logprob.p, -std::numeric_limits<double>::infinity(), np, s));
  // last pair's statistics (derivations::statistics, see carmel_hip_lattice_stats)
  uint32_t last_xs = 0;
  HIPCHK(hipMemcpyAsync(&last_xs, pp_xs.p + (np - 1), 4, hipMemcpyDeviceToHost, s));
  uint64_t last_kept = np;
  while (last_kept > 0 && !L.has_deriv[last_kept - 1]) --last_kept;
  uint16_t lastS = 0, lastE = 0;
  if (last_kept) {
    HIPCHK(hipMemcpyAsync(&lastS, pp_S.p + (last_kept - 1), 2, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&lastE, pp_E.p + (last_kept - 1), 2, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  L.n_kept = n_kept;
  L.explored_states = hs[2];
  L.explored_arcs = hs[3];
  L.total_states = hs[4];
  L.total_arcs = hs[5];
  L.lane_states = hs[4];
  L.lane_arcs = hs[5];
  L.n_post = n_post;
  L.last_pre_states = last_xs;
  L.last_post_states = lastS;
  L.last_post_arcs = lastE;
  {
    uint64_t ml = 0;
    for (auto& g : L.lane_groups) ml = std::max<uint64_t>(ml, g.max_states);
    L.max_levels = ml;  // (an upper bound: levels are not kept per pair on this path)
  }
  t->device_bytes = t->lane_groups.bytes() + t->lane_fwdx.bytes() + t->lane_bwd.bytes() + t->lane_pair.bytes() + t->lane_nstates.bytes() +
                    t->lane_logw.bytes() + t->post.bytes() + t->wcache.bytes() + t->arc_off.bytes() + t->slot_pos.bytes() +
                    t->t_b_arc.bytes() + t->t_b_rank.bytes() + t->t_t_pos.bytes() + t->t_b_src.bytes() + t->t_t_src.bytes() +
                    t->t_x.bytes() + t->t_xc.bytes() + t->pair_logprob.bytes() + t->lane_spill.bytes();
  t->have_lattices = true;
  ++t->lattice_epoch;
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->n_pairs = np;
    stats->n_pairs_kept = n_kept;
    stats->explored_states = L.explored_states;
    stats->explored_arcs = L.explored_arcs;
    stats->kept_states = L.total_states;
    stats->kept_arcs = L.total_arcs;
    stats->n_bundles = L.lane_groups.size();
    stats->max_levels = L.max_levels;
    stats->device_bytes = t->device_bytes;
    stats->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->last_pair_explored_states = L.last_pre_states;
    stats->last_pair_kept_states = L.last_post_states;
    stats->last_pair_kept_arcs = L.last_post_arcs;
    stats->n_windowed_pairs = n_win;
    stats->n_windowed_pairs = 0;
    for (auto& g : L.lane_groups)
      if (g.window) stats->n_windowed_pairs += g.n_lanes;
  }
  lap("finish");
  done = true;
  return CARMEL_HIP_OK;
}
