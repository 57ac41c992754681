// lattice.cpp — see lattice.hpp.  Host C++ only (no HIP): builds lattices and the batched-CSR image.
#include "lattice.hpp"
#include "options.hpp"
#include <algorithm>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <atomic>

namespace carmel_hip {
void build_transpose(LatticeSet& out, uint64_t n_arcs, int nt);


void HostWfst::build_index() {
  idx_off.assign((size_t)n_states + 1, 0);
  for (uint64_t k = 0; k < n_arcs; ++k) idx_off[src[k] + 1]++;
  for (uint32_t s = 0; s < n_states; ++s) idx_off[s + 1] += idx_off[s];
  idx_key.resize(n_arcs);
  idx_arc.resize(n_arcs);
  std::vector<uint64_t> cur(idx_off.begin(), idx_off.end() - 1);
  for (uint64_t k = 0; k < n_arcs; ++k) {
    uint64_t p = cur[src[k]]++;
    idx_key[p] = ((uint64_t)in[k] << 32) | out[k];
    idx_arc[p] = (uint32_t)k;
  }
  // sort each state's slice by key, ties by arc id (stable) == wfst_io_index's per-(in,out) push_back order
  // (derivations.h:150-154)
  std::vector<uint32_t> perm;
  std::vector<uint64_t> tk;
  std::vector<uint32_t> ta;
  for (uint32_t s = 0; s < n_states; ++s) {
    uint64_t a = idx_off[s], b = idx_off[s + 1];
    size_t n = (size_t)(b - a);
    if (n < 2) continue;
    bool sorted = true;
    for (size_t i = 1; i < n; ++i)
      if (idx_key[a + i] < idx_key[a + i - 1]) {
        sorted = false;
        break;
      }
    if (sorted) continue;
    perm.resize(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return idx_key[a + x] < idx_key[a + y]; });
    tk.resize(n);
    ta.resize(n);
    for (size_t i = 0; i < n; ++i) {
      tk[i] = idx_key[a + perm[i]];
      ta[i] = idx_arc[a + perm[i]];
    }
    std::copy(tk.begin(), tk.end(), idx_key.begin() + a);
    std::copy(ta.begin(), ta.end(), idx_arc.begin() + a);
  }
}

// Given out.lane_groups with maxlen / max_states / n_lanes / pair_base filled in: the launch classes (by LDS need), their
// pieces, and every group's stream_base.  Returns the number of records of the lane streams (padding included).
uint64_t assign_lane_classes(LatticeSet& out, const BuildOptions& opt, bool only_lanes) {
  const size_t ng = out.lane_groups.size();
  out.lane_classes.clear();
  out.tile = TRANS_TILE;
  out.bucket = TRANS_BUCKET;
  out.tile_sweep = false;
  out.lane_fused = false;
  out.tile_group.clear();
  // tile sweep: plain groups only, each within a tile of positions and of value rows
  bool sweepable = opt.tile_sweep && only_lanes && ng > 0 && opt.lane_chunks <= 1;
  for (size_t g = 0; sweepable && g < ng; ++g) {
    const LaneGroup& G = out.lane_groups[g];
    if (G.window || G.maxlen > TILE_SWEEP_ROWS || std::max(G.max_states, G.maxlen + 1) > TILE_SWEEP_ALPHA_ROWS) sweepable = false;
  }
  if (sweepable) {
    out.tile = TILE_SWEEP_TILE;
    out.tile_sweep = true;
    out.lane_tiles_aligned = true;
    uint64_t base = 0;
    uint32_t rows = 0, vrows = 0, mx = 0;
    for (size_t g = 0; g < ng; ++g) {
      LaneGroup& G = out.lane_groups[g];
      const uint32_t need = std::max(G.max_states, G.maxlen + 1);  // (tile_chain_sweep writes a column row per stream row)
      if (g == 0 || (uint64_t)(rows + G.maxlen) * 64 > TILE_SWEEP_TILE || vrows + need > TILE_SWEEP_ALPHA_ROWS ||
          g - out.tile_group.back() >= TILE_SWEEP_GROUPS) {
        base = (base + TILE_SWEEP_TILE - 1) / TILE_SWEEP_TILE * TILE_SWEEP_TILE;
        out.tile_group.push_back((uint32_t)g);
        rows = vrows = 0;
      }
      G.stream_base = base;
      G.spill_row = vrows;
      base += (uint64_t)G.maxlen * 64;
      rows += G.maxlen;
      vrows += need;
      mx = std::max(mx, G.max_states);
    }
    out.tile_group.push_back((uint32_t)ng);
    base = (base + TILE_SWEEP_TILE - 1) / TILE_SWEEP_TILE * TILE_SWEEP_TILE;
    // a small corpus (config 2: 287 buckets of 8192 items for 256 CUs that take two each) fills the chip with half-size buckets
    if (base / TRANS_BUCKET < 768) out.bucket = TRANS_BUCKET / 2;
    LatticeSet::LaneClass lc;  // one class: the tile kernel needs none, the lane kernel (A/B) takes the largest column
    lc.first = 0;
    lc.count = (uint32_t)ng;
    lc.max_states = mx;
    lc.tile_first = 0;
    lc.tile_count = (uint32_t)(base / TILE_SWEEP_TILE);
    out.lane_classes.push_back(lc);
    return base;
  }
  // fused lanes: small tiles, every group on a tile boundary of its own (LANE_FUSED_TILE, lattice.hpp)
  const bool fused = opt.lane_fused && only_lanes && ng > 0;
  const uint64_t TS = fused ? LANE_FUSED_TILE : TRANS_TILE;
  out.bucket = TRANS_BUCKET;
  if (fused) {
    out.tile = LANE_FUSED_TILE;
    out.lane_fused = true;
    out.bucket = TRANS_BUCKET / 2;
  }
  // classes: contiguous runs of groups sharing one LDS size (512 B per state per wave)
  std::vector<LatticeSet::LaneClass> classes;
  {
    size_t i = 0;
    while (i < ng) {
      uint32_t mx = out.lane_groups[i].max_states;
      size_t j = i + 1;
      while (j < ng) {
        uint32_t m = out.lane_groups[j].max_states;
        if ((out.lane_groups[j].window != 0) != (out.lane_groups[i].window != 0)) break;  // two kernels
        if (m > mx) mx = m;
        // a class costs a launch (ramp-up + tail); it only pays when the LDS saved buys occupancy that matters:
        // below ~20 KB per wave (8 waves per CU) the sweep is already bound by the random-gather rate
        const uint32_t min_split = 48u;
        if (j - i >= 256 && mx > min_split && (uint64_t)m * 3 <= (uint64_t)mx * 2) break;
        ++j;
      }
      LatticeSet::LaneClass lc;
      lc.first = (uint32_t)i;
      lc.count = (uint32_t)(j - i);
      lc.max_states = mx;
      lc.windowed = out.lane_groups[i].window != 0;
      classes.push_back(lc);
      i = j;
    }
  }
  // pieces: every class is cut into lane_chunks pieces of about equal record count.  When there is more than one
  // piece (several classes, or chunks), each piece's stream starts on a tile boundary of the blocked transposition, so
  // that the tile passes can be launched per piece and the pieces can run side by side
  uint64_t base = 0;
  const uint32_t want_chunks = std::max<uint32_t>(1, opt.lane_chunks);
  std::vector<uint32_t> nchs;
  size_t n_pieces = 0;
  for (const auto& lc : classes) {
    uint64_t rows = 0;
    for (uint32_t g = lc.first; g < lc.first + lc.count; ++g) rows += out.lane_groups[g].maxlen;
    // no chunk smaller than 16 tiles' worth of records: a launch has to fill the chip
    nchs.push_back((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want_chunks, rows * 64 / (16ull * TRANS_TILE))));
    n_pieces += nchs.back();
  }
  const bool align = n_pieces > 1 || fused;
  out.lane_tiles_aligned = align;
  for (size_t ci = 0; ci < classes.size(); ++ci) {
    const auto& lc = classes[ci];
    uint64_t rows = 0;
    for (uint32_t g = lc.first; g < lc.first + lc.count; ++g) rows += out.lane_groups[g].maxlen;
    const uint32_t nch = nchs[ci];
    uint32_t g = lc.first;
    uint64_t done = 0;
    for (uint32_t k = 0; k < nch; ++k) {
      const uint64_t goal = rows * (k + 1) / nch;
      LatticeSet::LaneClass piece = lc;
      piece.first = g;
      if (align) base = (base + TS - 1) / TS * TS;
      piece.tile_first = (uint32_t)(base / TS);
      while (g < lc.first + lc.count && (done < goal || k + 1 == nch)) {
        LaneGroup& G = out.lane_groups[g];
        if (fused) base = (base + TS - 1) / TS * TS;
        G.stream_base = base;
        base += (uint64_t)G.maxlen * 64;
        done += G.maxlen;
        ++g;
      }
      piece.count = g - piece.first;
      piece.max_states = lc.max_states;  // one LDS size per class keeps the occupancy of its pieces equal
      piece.tile_count = (uint32_t)((base + TS - 1) / TS) - piece.tile_first;
      if (piece.count) out.lane_classes.push_back(piece);
    }
  }
  if (align) base = (base + TS - 1) / TS * TS;  // bundle positions start on a tile too
  return base;
}

namespace {

// open-addressing map (i, s, o) -> lattice state id, reused across pairs by one thread
struct StateMap {
  struct Slot {
    uint32_t i, s, o, id;
  };
  std::vector<Slot> slots;
  std::vector<uint32_t> stamp;
  uint32_t gen = 0;
  size_t mask = 0, used = 0;
  void reset(size_t want) {
    size_t cap = 64;
    while (cap < want * 2) cap <<= 1;
    if (cap > slots.size()) {
      slots.assign(cap, Slot());
      stamp.assign(cap, 0);
      gen = 0;
    }
    mask = slots.size() - 1;
    used = 0;
    if (++gen == 0) {
      std::fill(stamp.begin(), stamp.end(), 0);
      gen = 1;
    }
  }
  static inline uint64_t hash(uint32_t i, uint32_t s, uint32_t o) {
    uint64_t h = ((uint64_t)i << 40) ^ ((uint64_t)o << 20) ^ s;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
  }
  void grow() {
    std::vector<Slot> old;
    std::vector<uint32_t> ost;
    old.swap(slots);
    ost.swap(stamp);
    uint32_t og = gen;
    slots.assign(old.size() * 2, Slot());
    stamp.assign(old.size() * 2, 0);
    mask = slots.size() - 1;
    gen = 1;
    used = 0;
    for (size_t k = 0; k < old.size(); ++k)
      if (ost[k] == og) {
        bool ins;
        find_or_insert(old[k].i, old[k].s, old[k].o, old[k].id, ins);
      }
  }
  // returns id; inserted=true when (i,s,o) was new and got new_id
  uint32_t find_or_insert(uint32_t i, uint32_t s, uint32_t o, uint32_t new_id, bool& inserted) {
    if ((used + 1) * 2 > slots.size()) grow();
    size_t p = (size_t)hash(i, s, o) & mask;
    for (;;) {
      if (stamp[p] != gen) {
        stamp[p] = gen;
        slots[p] = Slot{i, s, o, new_id};
        ++used;
        inserted = true;
        return new_id;
      }
      const Slot& sl = slots[p];
      if (sl.i == i && sl.s == s && sl.o == o) {
        inserted = false;
        return sl.id;
      }
      p = (p + 1) & mask;
    }
  }
  bool find(uint32_t i, uint32_t s, uint32_t o, uint32_t& id) const {
    size_t p = (size_t)hash(i, s, o) & mask;
    for (;;) {
      if (stamp[p] != gen) return false;
      const Slot& sl = slots[p];
      if (sl.i == i && sl.s == s && sl.o == o) {
        id = sl.id;
        return true;
      }
      p = (p + 1) & mask;
    }
  }
};

struct Frame {
  uint32_t sid, i, s, o;
  int phase;          // -1 not started, 0..3 label classes, 4 done
  uint64_t pos, end;  // current slice of the arc index
  uint32_t ni, no;    // child positions for the current phase
  bool dead, pending;
  uint32_t pend_dst, pend_arc;
};

struct Scratch {
  StateMap map;
  std::vector<Frame> stack;
  std::vector<uint8_t> removed;
  std::vector<PairLattice::E> edges;
  std::vector<uint32_t> a, b, c, d;
  std::vector<uint32_t> span_cnt, span_id;
};

inline void key_range(const HostWfst& w, uint32_t s, uint64_t key, uint64_t& lo, uint64_t& hi) {
  uint64_t a = w.idx_off[s], b = w.idx_off[s + 1];
  const uint64_t* k = w.idx_key.data();
  if (b - a <= 8) {
    while (a < b && k[a] < key) ++a;
    uint64_t e = a;
    while (e < b && k[e] == key) ++e;
    lo = a;
    hi = e;
    return;
  }
  const uint64_t* p = std::lower_bound(k + a, k + b, key);
  lo = (uint64_t)(p - k);
  uint64_t e = lo;
  while (e < b && k[e] == key) ++e;
  hi = e;
}

}  // namespace

// Exploration in the reference's order (derivations.h:640-704): depth first; from a node the four label classes
// (e,e), (e,out[o]), (in[i],e), (in[i],out[o]) in that order; within a class the WFST arcs in arc-id order; an arc
// is kept iff its destination is not (yet) known to be dead.  State ids come out in DFS pre-order like the
// reference's, which is what lets a cyclic lattice be swept in the reference's order later.
static void explore(const HostWfst& w, const uint32_t* in, uint32_t n_in, const uint32_t* out, uint32_t n_out,
                    Scratch& sc, uint32_t& n_states, bool& found_goal, uint32_t& goal_id, uint64_t& explored_arcs) {
  const uint32_t EPS = 0;
  sc.map.reset(256);
  sc.stack.clear();
  sc.removed.clear();
  sc.edges.clear();
  explored_arcs = 0;
  n_states = 0;
  const uint32_t gi = n_in, gs = w.final_state, go = n_out;
  auto new_state = [&](uint32_t i, uint32_t s, uint32_t o) -> uint32_t {
    bool ins;
    uint32_t id = sc.map.find_or_insert(i, s, o, n_states, ins);
    if (ins) {
      ++n_states;
      sc.removed.push_back(0);
      Frame f;
      f.sid = id;
      f.i = i;
      f.s = s;
      f.o = o;
      f.phase = -1;
      f.pos = f.end = 0;
      f.ni = f.no = 0;
      f.dead = !(i == gi && s == gs && o == go);
      f.pending = false;
      f.pend_dst = f.pend_arc = 0;
      sc.stack.push_back(f);
      return id | 0x80000000u;  // high bit: newly created (a frame was pushed)
    }
    return id;
  };
  new_state(0, 0, 0);
  while (!sc.stack.empty()) {
    size_t fi = sc.stack.size() - 1;
    {
      Frame& F = sc.stack[fi];
      if (F.pending) {  // the child pushed for F.pend_arc has finished
        if (!sc.removed[F.pend_dst]) {
          sc.edges.push_back(PairLattice::E{F.sid, F.pend_dst, F.pend_arc});
          F.dead = false;
        }
        F.pending = false;
        ++F.pos;
      }
    }
    bool pushed = false;
    for (;;) {
      Frame& F = sc.stack[fi];
      if (F.pos < F.end) {
        uint32_t arc = w.idx_arc[F.pos];
        ++explored_arcs;
        uint32_t r = new_state(F.ni, w.dst[arc], F.no);  // may reallocate sc.stack
        Frame& G = sc.stack[fi];
        if (r & 0x80000000u) {
          G.pending = true;
          G.pend_dst = r & 0x7fffffffu;
          G.pend_arc = arc;
          pushed = true;
          break;
        }
        if (!sc.removed[r]) {
          sc.edges.push_back(PairLattice::E{G.sid, r, arc});
          G.dead = false;
        }
        ++G.pos;
        continue;
      }
      // next label class
      ++F.phase;
      if (F.phase > 3) {
        sc.removed[F.sid] = F.dead;
        sc.stack.pop_back();
        break;
      }
      bool useO = F.o < n_out, useI = F.i < n_in;
      uint32_t si = EPS, so = EPS;
      bool ok = true;
      F.ni = F.i;
      F.no = F.o;
      switch (F.phase) {
        case 0:
          break;
        case 1:
          ok = useO;
          if (ok) {
            so = out[F.o];
            F.no = F.o + 1;
          }
          break;
        case 2:
          ok = useI;
          if (ok) {
            si = in[F.i];
            F.ni = F.i + 1;
          }
          break;
        case 3:
          ok = useI && useO;
          if (ok) {
            si = in[F.i];
            so = out[F.o];
            F.ni = F.i + 1;
            F.no = F.o + 1;
          }
          break;
      }
      if (ok)
        key_range(w, F.s, ((uint64_t)si << 32) | so, F.pos, F.end);
      else
        F.pos = F.end = 0;
    }
    (void)pushed;
  }
  found_goal = sc.map.find(gi, gs, go, goal_id);
}

void build_pair_lattice_impl(const HostWfst& w, const uint32_t* in, uint32_t n_in, const uint32_t* out,
                             uint32_t n_out, bool /*prune*/, PairLattice& lat, bool& has_deriv, Scratch& sc) {
  uint32_t nst = 0, goal = 0;
  bool found = false;
  uint64_t explored_arcs = 0;
  explore(w, in, n_in, out, n_out, sc, nst, found, goal, explored_arcs);
  lat = PairLattice();
  lat.explored_states = nst;
  lat.explored_arcs = explored_arcs;
  has_deriv = found;
  if (!found) return;
  // co-reachability over kept edges (drops the junk cycles the reference's remove[] marking lets through; they
  // carry zero backward mass there, so nothing observable changes)
  std::vector<uint32_t>& indeg_off = sc.a;  // CSR of reversed edges
  indeg_off.assign((size_t)nst + 1, 0);
  for (auto& e : sc.edges) indeg_off[e.dst + 1]++;
  for (uint32_t s = 0; s < nst; ++s) indeg_off[s + 1] += indeg_off[s];
  std::vector<uint32_t>& rsrc = sc.b;
  rsrc.resize(sc.edges.size());
  {
    std::vector<uint32_t>& cur = sc.c;
    cur.assign(indeg_off.begin(), indeg_off.end() - 1);
    for (auto& e : sc.edges) rsrc[cur[e.dst]++] = e.src;
  }
  std::vector<uint32_t>& keep_id = sc.d;  // old -> new or ~0
  keep_id.assign(nst, 0xffffffffu);
  {
    std::vector<uint32_t>& st = sc.c;
    st.clear();
    st.push_back(goal);
    keep_id[goal] = 0;
    while (!st.empty()) {
      uint32_t v = st.back();
      st.pop_back();
      for (uint32_t k = indeg_off[v]; k < indeg_off[v + 1]; ++k) {
        uint32_t u = rsrc[k];
        if (keep_id[u] == 0xffffffffu) {
          keep_id[u] = 0;
          st.push_back(u);
        }
      }
    }
  }
  uint32_t kept = 0;
  for (uint32_t s = 0; s < nst; ++s)
    if (keep_id[s] != 0xffffffffu && !sc.removed[s]) keep_id[s] = kept++;
    else keep_id[s] = 0xffffffffu;
  lat.n_states = kept;
  lat.start = keep_id[0];
  lat.fin = keep_id[goal];
  lat.edges.reserve(sc.edges.size());
  for (auto& e : sc.edges)
    if (keep_id[e.src] != 0xffffffffu && keep_id[e.dst] != 0xffffffffu)
      lat.edges.push_back(PairLattice::E{keep_id[e.src], keep_id[e.dst], e.arc});
  // levels: Kahn over the kept graph, level = longest path from the start
  std::vector<uint32_t>& indeg = sc.a;
  indeg.assign(kept, 0);
  for (auto& e : lat.edges) indeg[e.dst]++;
  std::vector<uint32_t>& ooff = sc.b;
  ooff.assign((size_t)kept + 1, 0);
  for (auto& e : lat.edges) ooff[e.src + 1]++;
  for (uint32_t s = 0; s < kept; ++s) ooff[s + 1] += ooff[s];
  std::vector<uint32_t> odst(lat.edges.size());
  {
    std::vector<uint32_t>& cur = sc.c;
    cur.assign(ooff.begin(), ooff.end() - 1);
    for (auto& e : lat.edges) odst[cur[e.src]++] = e.dst;  // insertion order per state preserved
  }
  lat.level.assign(kept, 0);
  std::vector<uint32_t>& q = sc.c;
  q.clear();
  for (uint32_t s = 0; s < kept; ++s)
    if (indeg[s] == 0) q.push_back(s);
  size_t done = 0;
  uint32_t maxlev = 0;
  while (done < q.size()) {
    uint32_t u = q[done++];
    uint32_t lu = lat.level[u];
    for (uint32_t k = ooff[u]; k < ooff[u + 1]; ++k) {
      uint32_t v = odst[k];
      if (lat.level[v] < lu + 1) lat.level[v] = lu + 1;
      if (--indeg[v] == 0) q.push_back(v);
    }
    if (lu > maxlev) maxlev = lu;
  }
  if (done == kept) {
    lat.n_levels = maxlev + 1;
    return;
  }
  // cyclic: the reference sweeps in reversed DFS post-order from the start, walking each state's out-arc list
  // newest-first (graph.h:91-94 push_front, :267-283 order_from) and skipping back edges.  level[] := position
  // in that forward order, so the serial kernel can reproduce the reference's values exactly.
  lat.cyclic = true;
  std::vector<uint8_t> begun(kept, 0), fin(kept, 0);
  std::vector<uint32_t> post;
  post.reserve(kept);
  std::vector<std::pair<uint32_t, uint32_t> > st;  // (state, next arc index counting from the end)
  st.push_back({lat.start, 0});
  begun[lat.start] = 1;
  while (!st.empty()) {
    uint32_t u = st.back().first;
    uint32_t k = st.back().second;
    uint32_t deg = ooff[u + 1] - ooff[u];
    if (k < deg) {
      st.back().second++;
      uint32_t v = odst[ooff[u + 1] - 1 - k];  // newest first
      if (!fin[v] && !begun[v]) {
        begun[v] = 1;
        st.push_back({v, 0});
      }
    } else {
      fin[u] = 1;
      post.push_back(u);
      st.pop_back();
    }
  }
  // every kept state is reachable from the start, so post covers all of them
  for (uint32_t p = 0; p < post.size(); ++p) lat.level[post[p]] = (uint32_t)(post.size() - 1 - p);
  lat.n_levels = kept;
}

void build_pair_lattice(const HostWfst& w, const uint32_t* in, uint32_t n_in, const uint32_t* out, uint32_t n_out,
                        bool prune, PairLattice& lat, bool& has_deriv) {
  Scratch sc;
  build_pair_lattice_impl(w, in, n_in, out, n_out, prune, lat, has_deriv, sc);
}

namespace {

// Lay one bundle out: states level-major (within a level: pair order, then the pair's own state order), in-arcs
// grouped by destination, out-arcs grouped by source.
struct BundlePlan {
  std::vector<uint32_t> pairs;  // indices into the kept-pair list
  uint64_t n_states = 0, n_arcs = 0;
  uint32_t n_levels = 0;
  bool cyclic = false;
};

}  // namespace

bool build_lattices(const HostWfst& w, const HostCorpus& c, const BuildOptions& opt, LatticeSet& out,
                    std::string& err) {
  if (w.idx_off.size() != (size_t)w.n_states + 1) {
    err = "WFST index not built";
    return false;
  }
  const uint64_t np = c.n_pairs;
  const bool timing = lib_opt("timing") != nullptr;  // phase times on stderr
  auto tick = std::chrono::steady_clock::now();
  auto phase = [&](const char* name) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "timing: lattice build: %-28s %8.3f s\n", name, std::chrono::duration<double>(now - tick).count());
    tick = now;
  };
  std::vector<PairLattice> lats(np);
  out = LatticeSet();
  out.has_deriv.assign(np, 0);
  // default: at most 32 threads -- measured on a 256-core host (config 4, 10^6 pairs): 8 threads 1.44 s, 32 threads
  // 0.71 s, 64 threads 0.99 s, 256 threads 1.88 s for the per-pair phase (allocator and memory-system contention)
  int nt = opt.threads > 0 ? opt.threads : std::min(32, (int)std::thread::hardware_concurrency());
  if (nt < 1) nt = 1;
  if ((uint64_t)nt > np) nt = (int)std::max<uint64_t>(1, np);
  {
    std::atomic<uint64_t> next(0);
    // pairs are handed out in batches of 256 on large corpora; a corpus of a few thousand LONG pairs (`long`: 5 000 lattices of up
    // to 4 800 states) would be twenty batches for thirty-two threads, so the batch shrinks until every thread gets about eight
    const uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(256, np / ((uint64_t)nt * 8)));
    auto work = [&]() {
      Scratch sc;
      for (;;) {
        uint64_t p0 = next.fetch_add(batch);
        if (p0 >= np) break;
        uint64_t p1 = std::min(np, p0 + batch);
        for (uint64_t p = p0; p < p1; ++p) {
          bool hd = false;
          build_pair_lattice_impl(w, c.in_sym.data() + c.in_off[p], (uint32_t)(c.in_off[p + 1] - c.in_off[p]),
                                  c.out_sym.data() + c.out_off[p], (uint32_t)(c.out_off[p + 1] - c.out_off[p]),
                                  opt.prune, lats[p], hd, sc);
          out.has_deriv[p] = hd;
          PairLattice& L = lats[p];
          if (hd && !L.cyclic && opt.lane_window && L.n_states > opt.lane_window_min && L.n_states <= LANE_STATE_MASK) {
            // the lane layout numbers the states by (level, id): how far apart are the ends of an arc there?
            std::vector<uint32_t>& cnt = sc.span_cnt;
            std::vector<uint32_t>& nid = sc.span_id;
            cnt.assign((size_t)L.n_levels + 2, 0);
            for (uint32_t st = 0; st < L.n_states; ++st) cnt[L.level[st] + 1]++;
            for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
            nid.resize(L.n_states);
            for (uint32_t st = 0; st < L.n_states; ++st) nid[st] = cnt[L.level[st]]++;
            uint32_t sp = 1;
            for (auto& e : L.edges) sp = std::max(sp, nid[e.dst] - nid[e.src]);
            L.span = sp;
          }
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  std::vector<uint32_t> kept;
  for (uint64_t p = 0; p < np; ++p) {
    out.explored_states += lats[p].explored_states;
    out.explored_arcs += lats[p].explored_arcs;
    out.last_pre_states = lats[p].explored_states;
    if (out.has_deriv[p]) {
      kept.push_back((uint32_t)p);
      out.last_post_states = lats[p].n_states;
      out.last_post_arcs = lats[p].edges.size();
    }
  }
  out.n_kept = kept.size();
  phase("derivations (per pair)");
  // ---- pack into bundles ----
  // small lattices: sort by (levels, states) so that a bundle's members have similar depth (level-synchronous
  // sweeps idle the lanes of members that ran out of levels)
  std::vector<uint32_t> small, big, cyc, lane, lane_win;
  // ring of LDS rows a windowed lattice needs: the power of two above its span
  // Windows cost a launch class, a second kernel and 16 B of traffic per state: they are used when lattices above
  // lane_window_min states carry a tenth of the corpus' arcs (the tagging cascade: nearly all; config 4: none -- and the
  // GPU builder, which lays out plain groups only, applies the same rule to decide whether the corpus is its case)
  uint64_t arcs_all = 0, arcs_big = 0;
  for (uint32_t p : kept) {
    arcs_all += lats[p].edges.size();
    if (lats[p].n_states > opt.lane_window_min) arcs_big += lats[p].edges.size();
  }
  const bool use_window = opt.lane_window && arcs_big * 10 >= arcs_all;
  auto window_of = [&](const PairLattice& L) -> uint32_t {
    if (!use_window || !L.span) return 0u;
    uint32_t w = 8;
    while (w < L.span + 1) w <<= 1;
    return (w <= opt.lane_window && w < L.n_states) ? w : 0u;
  };
  // one lattice per wavefront (WaveDesc): what no lane takes, when it is wide enough to feed 64 lanes; and what a windowed
  // lane would take, when the corpus is too small to fill the chip one lattice per lane
  std::vector<uint32_t> wave;
  auto wave_fits = [&](const PairLattice& L, double min_width) {
    if (!opt.wave || L.cyclic || L.n_states < 2 || L.n_states > WAVE_MAX_STATES || L.n_levels < 2) return false;
    if ((double)L.edges.size() < min_width * (double)(L.n_levels - 1)) return false;
    std::vector<uint32_t> cnt((size_t)L.n_levels, 0);
    for (uint32_t st = 0; st < L.n_states; ++st)
      if (++cnt[L.level[st]] > WAVE_MAX_WIDTH) return false;
    return true;
  };
  uint64_t n_windowed = 0;
  for (uint32_t p : kept) {
    const PairLattice& L = lats[p];
    if (!L.cyclic && opt.lane_states && L.edges.size() <= LANE_POS_MAX && L.n_states <= LANE_STATE_MASK &&
        (window_of(L) || L.n_states <= opt.lane_states))
      ++n_windowed;  // every lattice a lane would take, plain or windowed: is that enough one-per-lane work to fill the chip?
  }
  for (uint32_t p : kept) {
    const PairLattice& L = lats[p];
    const bool plain_lane = opt.lane_states && L.n_states <= opt.lane_states && L.n_states <= LANE_STATE_MASK && L.edges.size() <= LANE_POS_MAX;
    if (L.cyclic)
      cyc.push_back(p);
    else if (opt.lane_states && window_of(L) && L.edges.size() <= LANE_POS_MAX) {
      if (!plain_lane && (n_windowed < opt.wave_lane_threshold || L.edges.size() > opt.wave_lane_arcs) && wave_fits(L, opt.wave_lane_min_width))
        wave.push_back(p);
      else
        lane_win.push_back(p);
    } else if (plain_lane)
      lane.push_back(p);
    else if (wave_fits(L, opt.wave_min_width))
      wave.push_back(p);
    else if (L.n_states * 4 <= opt.small_states)
      small.push_back(p);
    else
      big.push_back(p);
  }
  std::stable_sort(small.begin(), small.end(), [&](uint32_t a, uint32_t b) {
    if (lats[a].n_levels != lats[b].n_levels) return lats[a].n_levels > lats[b].n_levels;
    return lats[a].n_states > lats[b].n_states;
  });
  std::stable_sort(big.begin(), big.end(), [&](uint32_t a, uint32_t b) { return lats[a].n_states < lats[b].n_states; });
  // ---- lane groups: one lattice per lane, 64 per wavefront ----
  if (!lane.empty() || !lane_win.empty()) {
    std::stable_sort(lane.begin(), lane.end(), [&](uint32_t a, uint32_t b) {
      if (lats[a].edges.size() != lats[b].edges.size()) return lats[a].edges.size() > lats[b].edges.size();
      return lats[a].n_states > lats[b].n_states;
    });
    // windowed lattices: by window first (a group's ring is its widest member's), then by length
    std::stable_sort(lane_win.begin(), lane_win.end(), [&](uint32_t a, uint32_t b) {
      const uint32_t wa = window_of(lats[a]), wb = window_of(lats[b]);
      if (wa != wb) return wa > wb;
      if (lats[a].edges.size() != lats[b].edges.size()) return lats[a].edges.size() > lats[b].edges.size();
      return lats[a].n_states > lats[b].n_states;
    });
    // plain groups, then windowed groups (a group never mixes the two): `lane` becomes one slot per lane, 64 per group
    const size_t ng_plain = (lane.size() + 63) / 64, ng = ng_plain + (lane_win.size() + 63) / 64;
    {
      std::vector<uint32_t> slots(ng * 64, 0xffffffffu);
      std::copy(lane.begin(), lane.end(), slots.begin());
      std::copy(lane_win.begin(), lane_win.end(), slots.begin() + ng_plain * 64);
      lane.swap(slots);
    }
    out.lane_groups.resize(ng);
    out.lane_pair.assign(ng * 64, 0xffffffffu);
    out.lane_nstates.assign(ng * 64, 0);
    out.lane_logw.assign(ng * 64, 0.0);
    uint64_t spill_rows = 0;
    for (size_t g = 0; g < ng; ++g) {
      LaneGroup& G = out.lane_groups[g];
      std::memset(&G, 0, sizeof G);
      size_t l0 = g * 64, l1 = l0;
      while (l1 < l0 + 64 && lane[l1] != 0xffffffffu) ++l1;
      G.n_lanes = (uint32_t)(l1 - l0);
      G.pair_base = (uint32_t)l0;
      uint32_t ml = 0, ms = 0, win = 0;
      for (size_t l = l0; l < l1; ++l) {
        ml = std::max<uint32_t>(ml, (uint32_t)lats[lane[l]].edges.size());
        ms = std::max(ms, lats[lane[l]].n_states);
        if (g >= ng_plain) win = std::max(win, window_of(lats[lane[l]]));
      }
      G.maxlen = (std::max<uint32_t>(ml, 1) + LANE_CHUNK - 1) / LANE_CHUNK * LANE_CHUNK;
      G.max_states = win ? win : ms;
      G.window = win;
      if (win) {
        if (spill_rows + ms > 0xffffffffull) {
          err = "too many windowed lattice states for one trainer";
          return false;
        }
        G.spill_row = (uint32_t)spill_rows;
        spill_rows += ms;
      }
    }
    out.lane_spill_rows = spill_rows;
    const uint64_t base = assign_lane_classes(out, opt, wave.empty() && small.empty() && big.empty() && cyc.empty());
    out.lane_fwd.assign(base, uint2_t{0, 0});
    out.lane_bwd.assign(base, uint2_t{0, 0});
    std::atomic<size_t> nextg(0);
    auto lwork = [&]() {
      std::vector<uint32_t> order, newid, ioff, ooff, cur;
      std::vector<uint32_t> ie, oe, fpos;
      for (;;) {
        size_t g = nextg.fetch_add(4);
        if (g >= ng) break;
        size_t ge = std::min(ng, g + 4);
        for (; g < ge; ++g) {
          const LaneGroup& G = out.lane_groups[g];
          for (uint32_t l = 0; l < G.n_lanes; ++l) {
            uint32_t p = lane[G.pair_base + l];
            const PairLattice& L = lats[p];
            const uint32_t S = L.n_states;
            out.lane_pair[G.pair_base + l] = p;
            out.lane_nstates[G.pair_base + l] = S;
            double wt = c.weight.empty() ? 1.0 : c.weight[p];
            out.lane_logw[G.pair_base + l] = wt > 0 ? std::log(wt) : -std::numeric_limits<double>::infinity();
            // topological numbering: by (level, id); the start is alone on level 0, the goal alone on the last
            order.resize(S);
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return L.level[a] < L.level[b]; });
            newid.resize(S);
            for (uint32_t k = 0; k < S; ++k) newid[order[k]] = k;
            const size_t E = L.edges.size();
            ioff.assign((size_t)S + 1, 0);
            ooff.assign((size_t)S + 1, 0);
            for (auto& e : L.edges) {
              ioff[newid[e.dst] + 1]++;
              ooff[newid[e.src] + 1]++;
            }
            for (uint32_t k = 0; k < S; ++k) {
              ioff[k + 1] += ioff[k];
              ooff[k + 1] += ooff[k];
            }
            ie.resize(E);
            oe.resize(E);
            cur.assign(ioff.begin(), ioff.end() - 1);
            for (uint32_t k = 0; k < E; ++k) ie[cur[newid[L.edges[k].dst]]++] = k;
            cur.assign(ooff.begin(), ooff.end() - 1);
            for (uint32_t k = 0; k < E; ++k) oe[cur[newid[L.edges[k].src]]++] = k;
            uint2_t* f = out.lane_fwd.data() + G.stream_base + l;
            uint2_t* b = out.lane_bwd.data() + G.stream_base + l;
            // backward stream first: out-arcs by source in reverse topological order.  The position of an edge there
            // is also its posterior slot and the place where the forward pass leaves its weight (wcache), so the
            // forward record carries it.
            // The backward stream is RIGHT-aligned in the group's rows (it starts at row maxlen - E): for a chain the
            // arc at forward row k then sits at backward row maxlen-1-k in every lane, so the forward pass's weight
            // stores (wcache) are coalesced rows too; for general lattices they are near-coalesced.
            fpos.resize(E);  // edge -> row in the backward stream
            size_t pos = (size_t)G.maxlen - E;
            for (uint32_t sidx = S; sidx-- > 0;) {
              if (sidx == S - 1) continue;  // the goal has no out-arcs
              for (uint32_t k = ooff[sidx]; k < ooff[sidx + 1]; ++k) {
                const auto& e = L.edges[oe[k]];
                uint32_t x = newid[e.dst] | (sidx << LANE_POS_SHIFT) | LANE_VALID | (k + 1 == ooff[sidx + 1] ? LANE_LAST : 0u);
                fpos[oe[k]] = (uint32_t)pos;
                b[(pos++) * 64] = uint2_t{x, e.arc};
              }
            }
            pos = 0;
            for (uint32_t d = 1; d < S; ++d)
              for (uint32_t k = ioff[d]; k < ioff[d + 1]; ++k) {
                const auto& e = L.edges[ie[k]];
                uint32_t x = newid[e.src] | (fpos[ie[k]] << LANE_POS_SHIFT) | LANE_VALID |
                             (k + 1 == ioff[d + 1] ? LANE_LAST : 0u);
                f[(pos++) * 64] = uint2_t{x, e.arc};
              }
          }
        }
      }
    };
    std::vector<std::thread> th;
    int nt3 = (int)std::min<size_t>((size_t)nt, std::max<size_t>(1, ng / 4));
    for (int t = 1; t < nt3; ++t) th.emplace_back(lwork);
    lwork();
    for (auto& t : th) t.join();
    for (uint32_t p : lane) {
      if (p == 0xffffffffu) continue;
      out.lane_states += lats[p].n_states;
      out.lane_arcs += lats[p].edges.size();
      out.max_levels = std::max<uint64_t>(out.max_levels, lats[p].n_levels);
    }
  }
  // ---- wave lattices: one per wavefront (WaveDesc) ----
  if (!wave.empty()) {
    // launch order: largest first (the long ones start early, the short ones fill the tail); classes by LDS need
    std::stable_sort(wave.begin(), wave.end(), [&](uint32_t a, uint32_t b) {
      if (lats[a].n_states != lats[b].n_states) return lats[a].n_states > lats[b].n_states;
      return lats[a].edges.size() > lats[b].edges.size();
    });
    const size_t nw = wave.size();
    out.waves.resize(nw);
    std::vector<uint64_t> frows(nw + 1, 0), brows(nw + 1, 0), lbase(nw + 1, 0);
    {
      std::atomic<size_t> next(0);
      auto count = [&]() {
        std::vector<uint32_t> fi, bo, width, lo_, nid_;
        for (;;) {
          const size_t k = next.fetch_add(16);
          if (k >= nw) break;
          for (size_t q = k; q < std::min(nw, k + 16); ++q) {
            const PairLattice& L = lats[wave[q]];
            fi.assign(L.n_levels, 0);
            bo.assign(L.n_levels, 0);
            width.assign(L.n_levels, 0);
            for (auto& e : L.edges) {
              fi[L.level[e.dst]]++;
              bo[L.level[e.src]]++;
            }
            for (uint32_t st = 0; st < L.n_states; ++st) width[L.level[st]]++;
            uint64_t rf = 0, rb = 0;
            uint32_t mw = 0;
            for (uint32_t l = 0; l < L.n_levels; ++l) {
              rf += (fi[l] + 63) / 64;
              rb += (bo[l] + 63) / 64;
              mw = std::max(mw, width[l]);
            }
            frows[q + 1] = rf;
            brows[q + 1] = rb;
            lbase[q + 1] = (uint64_t)L.n_levels + 1;
            out.waves[q].max_width = mw;
            // how far apart are an arc's ends in the level-major numbering?  (ring form)
            uint32_t ring = 0;
            if (mw <= WAVE_RING_WIDTH && opt.wave_ring) {
              lo_.assign((size_t)L.n_levels + 1, 0);
              for (uint32_t l = 0; l < L.n_levels; ++l) lo_[l + 1] = lo_[l] + width[l];
              nid_.resize(L.n_states);
              for (uint32_t st = 0; st < L.n_states; ++st) nid_[st] = lo_[L.level[st]]++;
              uint32_t sp = 1;
              for (auto& e : L.edges) sp = std::max(sp, nid_[e.dst] - nid_[e.src]);
              uint32_t r = 8;
              while (r < sp) r <<= 1;
              if (r <= WAVE_RING_MAX && r < L.n_states) ring = r;
            }
            out.waves[q].ring = ring;
          }
        }
      };
      std::vector<std::thread> th;
      const int ntw = (int)std::min<size_t>((size_t)nt, std::max<size_t>(1, nw / 16));
      for (int t = 1; t < ntw; ++t) th.emplace_back(count);
      count();
      for (auto& t : th) t.join();
    }
    {
      // launch order: ring lattices first (widest ring first), then the full-LDS ones; inside, the largest first
      std::vector<uint32_t> perm(nw);
      std::iota(perm.begin(), perm.end(), 0u);
      std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
        const uint32_t ra = out.waves[a].ring, rb = out.waves[b].ring;
        if ((ra != 0) != (rb != 0)) return ra != 0;
        return ra > rb;
      });
      std::vector<uint32_t> wave2(nw);
      std::vector<WaveDesc> d2(nw);
      std::vector<uint64_t> f2(nw + 1, 0), b2(nw + 1, 0), l2(nw + 1, 0);
      for (size_t q = 0; q < nw; ++q) {
        wave2[q] = wave[perm[q]];
        d2[q] = out.waves[perm[q]];
        f2[q + 1] = frows[perm[q] + 1];
        b2[q + 1] = brows[perm[q] + 1];
        l2[q + 1] = lbase[perm[q] + 1];
      }
      wave.swap(wave2);
      out.waves.swap(d2);
      frows.swap(f2);
      brows.swap(b2);
      lbase.swap(l2);
    }
    uint64_t spill = 0;
    std::vector<uint64_t> spill_base(nw, 0);
    for (size_t q = 0; q < nw; ++q) {
      frows[q + 1] += frows[q];
      brows[q + 1] += brows[q];
      lbase[q + 1] += lbase[q];
      if (out.waves[q].ring) {
        spill_base[q] = spill;
        spill += lats[wave[q]].n_states;
      }
    }
    out.wave_spill_states = spill;
    if (lbase[nw] > 0xffffffffull || brows[nw] * 64 > (1ull << 40)) {
      err = "wave lattice set too large";
      return false;
    }
    out.wave_gather = opt.wave_gather;
    out.wave_fwd.assign(frows[nw] * 64, uint2_t{0, 0});
    out.wave_bwd.assign(brows[nw] * 64, 0u);
    out.wave_bwd_arc.assign(brows[nw] * 64, 0xffffffffu);
    out.wave_level_off.resize(lbase[nw]);
    out.wave_frow.resize(lbase[nw]);
    out.wave_brow.resize(lbase[nw]);
    {
      std::atomic<size_t> next(0);
      auto fill = [&]() {
        std::vector<uint32_t> order, newid, ioff, ooff, cur, ie, oe, bpos;
        for (;;) {
          const size_t k = next.fetch_add(8);
          if (k >= nw) break;
          for (size_t q = k; q < std::min(nw, k + 8); ++q) {
            const uint32_t p = wave[q];
            const PairLattice& L = lats[p];
            WaveDesc& D = out.waves[q];
            const uint32_t mw = D.max_width, ring = D.ring;
            std::memset(&D, 0, sizeof D);
            D.max_width = mw;
            D.ring = ring;
            D.spill_base = spill_base[q];
            D.fwd_base = frows[q] * 64;
            D.bwd_base = brows[q] * 64;
            D.n_states = L.n_states;
            D.n_levels = L.n_levels;
            D.level_base = (uint32_t)lbase[q];
            D.pair = p;
            const double wt = c.weight.empty() ? 1.0 : c.weight[p];
            D.logw = wt > 0 ? std::log(wt) : -std::numeric_limits<double>::infinity();
            D.n_arcs = L.edges.size();
            const uint32_t S = L.n_states, NL = L.n_levels;
            const size_t E = L.edges.size();
            order.resize(S);
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return L.level[a] < L.level[b]; });
            newid.resize(S);
            for (uint32_t i = 0; i < S; ++i) newid[order[i]] = i;
            uint32_t* lo = out.wave_level_off.data() + D.level_base;
            uint32_t* fr = out.wave_frow.data() + D.level_base;
            uint32_t* br = out.wave_brow.data() + D.level_base;
            std::fill(lo, lo + NL + 1, 0u);
            for (uint32_t st = 0; st < S; ++st) lo[L.level[st] + 1]++;
            for (uint32_t l = 0; l < NL; ++l) lo[l + 1] += lo[l];
            ioff.assign((size_t)S + 1, 0);
            ooff.assign((size_t)S + 1, 0);
            for (auto& e : L.edges) {
              ioff[newid[e.dst] + 1]++;
              ooff[newid[e.src] + 1]++;
            }
            for (uint32_t i = 0; i < S; ++i) {
              ioff[i + 1] += ioff[i];
              ooff[i + 1] += ooff[i];
            }
            ie.resize(E);
            oe.resize(E);
            cur.assign(ioff.begin(), ioff.end() - 1);
            for (uint32_t i = 0; i < E; ++i) ie[cur[newid[L.edges[i].dst]]++] = i;
            cur.assign(ooff.begin(), ooff.end() - 1);
            for (uint32_t i = 0; i < E; ++i) oe[cur[newid[L.edges[i].src]]++] = i;
            // backward stream: step k = source level NL - 1 - k (the goal's level has no out-arcs: step 0 is empty)
            bpos.resize(E);
            uint32_t* B = out.wave_bwd.data() + D.bwd_base;
            uint32_t* BA = out.wave_bwd_arc.data() + D.bwd_base;
            uint32_t row = 0;
            for (uint32_t kk = 0; kk < NL; ++kk) {
              br[kk] = row;
              const uint32_t l = NL - 1 - kk;
              const uint32_t a0 = ooff[lo[l]], a1 = ooff[lo[l + 1]];
              for (uint32_t a = a0; a < a1; ++a) {
                const auto& e = L.edges[oe[a]];
                const uint32_t pos = row * 64 + (a - a0);
                bpos[oe[a]] = pos;
                B[pos] = newid[e.dst] | ((newid[e.src] - lo[l]) << 16) | WAVE_VALID;
                BA[pos] = e.arc;
              }
              row += (a1 - a0 + 63) / 64;
            }
            br[NL] = row;
            uint2_t* F = out.wave_fwd.data() + D.fwd_base;
            row = 0;
            for (uint32_t l = 0; l < NL; ++l) {
              fr[l] = row;
              const uint32_t a0 = ioff[lo[l]], a1 = ioff[lo[l + 1]];
              for (uint32_t a = a0; a < a1; ++a) {
                const auto& e = L.edges[ie[a]];
                F[(size_t)row * 64 + (a - a0)] = uint2_t{newid[e.src] | ((newid[e.dst] - lo[l]) << 16) | WAVE_VALID, opt.wave_gather ? e.arc : bpos[ie[a]]};
              }
              row += (a1 - a0 + 63) / 64;
            }
            fr[NL] = row;
          }
        }
      };
      std::vector<std::thread> th;
      const int ntw = (int)std::min<size_t>((size_t)nt, std::max<size_t>(1, nw / 8));
      for (int t = 1; t < ntw; ++t) th.emplace_back(fill);
      fill();
      for (auto& t : th) t.join();
    }
    // classes: one per ring size, then the full-LDS lattices by LDS need (8 B per state + 16 B per state of the widest
    // level), largest first
    size_t i = 0;
    while (i < nw && out.waves[i].ring) {
      size_t j = i;
      uint32_t ms = 0;
      while (j < nw && out.waves[j].ring == out.waves[i].ring) ms = std::max(ms, out.waves[j++].n_states);
      LatticeSet::WaveClass wc{(uint32_t)i, (uint32_t)(j - i), ms, WAVE_RING_WIDTH};
      wc.ring = out.waves[i].ring;
      // rings up to 128 values share a launch (a lattice keeps its own ring inside the launch's LDS, WaveDesc::ring: 1 KB of 2.5 at
      // most): a launch per ring size left `mix`'s one lattice of ring 32 alone in front of its 1985 lattices of ring 16 -- on
      // streams that shared a hardware queue, 0.3 ms of an empty chip
      if (!out.wave_classes.empty() && out.wave_classes.back().ring && std::max(out.wave_classes.back().ring, wc.ring) <= 128) {
        LatticeSet::WaveClass& b = out.wave_classes.back();
        b.count += wc.count;
        b.max_states = std::max(b.max_states, wc.max_states);
        b.ring = std::max(b.ring, wc.ring);
      } else
        out.wave_classes.push_back(wc);
      i = j;
    }
    const uint32_t caps[] = {WAVE_MAX_STATES, 8192, 4096, 2048, 1024, 512, 0};
    for (int kc = 0; caps[kc] && i < nw; ++kc) {
      size_t j = i;
      uint32_t ms = 0, mw = 0;
      while (j < nw && out.waves[j].n_states > caps[kc + 1]) {
        ms = std::max(ms, out.waves[j].n_states);
        mw = std::max(mw, out.waves[j].max_width);
        ++j;
      }
      if (j > i) out.wave_classes.push_back(LatticeSet::WaveClass{(uint32_t)i, (uint32_t)(j - i), ms, mw});
      i = j;
    }
    for (uint32_t p : wave) {
      out.wave_states += lats[p].n_states;
      out.wave_arcs += lats[p].edges.size();
      out.max_levels = std::max<uint64_t>(out.max_levels, lats[p].n_levels);
    }
  }
  std::vector<BundlePlan> plans;
  {
    BundlePlan cur;
    for (uint32_t p : small) {
      const PairLattice& L = lats[p];
      if (!cur.pairs.empty() && (cur.pairs.size() >= opt.small_pairs || cur.n_states + L.n_states > opt.small_states)) {
        plans.push_back(cur);
        cur = BundlePlan();
      }
      cur.pairs.push_back(p);
      cur.n_states += L.n_states;
      cur.n_arcs += L.edges.size();
      cur.n_levels = std::max(cur.n_levels, L.n_levels);
    }
    if (!cur.pairs.empty()) plans.push_back(cur);
  }
  size_t n_small = plans.size();
  for (uint32_t p : big) {
    BundlePlan b;
    b.pairs.push_back(p);
    b.n_states = lats[p].n_states;
    b.n_arcs = lats[p].edges.size();
    b.n_levels = lats[p].n_levels;
    plans.push_back(b);
  }
  size_t n_acyclic = plans.size();
  for (uint32_t p : cyc) {
    BundlePlan b;
    b.pairs.push_back(p);
    b.n_states = lats[p].n_states;
    b.n_arcs = lats[p].edges.size();
    b.n_levels = lats[p].n_levels;
    b.cyclic = true;
    plans.push_back(b);
  }
  out.n_cyclic = cyc.size();
  // launch classes
  auto add_class = [&](size_t first, size_t count, uint32_t block, uint32_t max_states, bool serial) {
    if (!count) return;
    LatticeSet::LaunchClass lc;
    lc.first = (uint32_t)first;
    lc.count = (uint32_t)count;
    lc.block = block;
    lc.max_states = max_states;
    lc.serial = serial;
    out.classes.push_back(lc);
  };
  if (n_small) {
    uint64_t mx = 0;
    for (size_t b = 0; b < n_small; ++b) mx = std::max(mx, plans[b].n_states);
    add_class(0, n_small, 64, (uint32_t)mx, false);
  }
  {
    // big lattices ascending by states: classes at 8K-state (64 KiB) and lds_states_max boundaries, then global
    size_t i = n_small;
    const uint32_t caps[2] = {8192, opt.lds_states_max};
    for (int k = 0; k < 2 && i < n_acyclic; ++k) {
      size_t j = i;
      uint64_t mx = 0;
      while (j < n_acyclic && plans[j].n_states <= caps[k]) {
        mx = std::max(mx, plans[j].n_states);
        ++j;
      }
      add_class(i, j - i, k == 0 ? 256 : 1024, (uint32_t)mx, false);
      i = j;
    }
    if (i < n_acyclic) add_class(i, n_acyclic - i, 1024, 0, false);
  }
  add_class(n_acyclic, plans.size() - n_acyclic, 64, 0, true);
  // offsets
  size_t nb = plans.size();
  out.bundles.resize(nb);
  uint64_t arc_base = 0, off_base = 0, lev_base = 0, pair_base = 0;
  for (size_t b = 0; b < nb; ++b) {
    BundleDesc& d = out.bundles[b];
    std::memset(&d, 0, sizeof d);
    d.in_base = d.out_base = arc_base;
    d.off_base = off_base;
    d.n_states = (uint32_t)plans[b].n_states;
    d.n_levels = plans[b].n_levels;
    d.level_base = (uint32_t)lev_base;
    d.pair_base = (uint32_t)pair_base;
    d.n_pairs = (uint32_t)plans[b].pairs.size();
    d.flags = plans[b].cyclic ? 1u : 0u;
    d.n_arcs = plans[b].n_arcs;
    arc_base += plans[b].n_arcs;
    off_base += plans[b].n_states + 1;
    lev_base += (uint64_t)plans[b].n_levels + 1;
    pair_base += plans[b].pairs.size();
    out.max_levels = std::max<uint64_t>(out.max_levels, plans[b].n_levels);
    if (lev_base > 0xffffffffull || pair_base > 0xffffffffull) {
      err = "lattice set too large for 32-bit level/pair tables";
      return false;
    }
  }
  out.total_arcs = arc_base + out.lane_arcs + out.wave_arcs;
  out.total_states = off_base - nb + out.lane_states + out.wave_states;
  out.in_arcs.resize(arc_base);
  out.out_arcs.resize(arc_base);
  out.in_off.resize(off_base);
  out.out_off.resize(off_base);
  out.level_off.resize(lev_base);
  out.pair_start.resize(pair_base);
  out.pair_final.resize(pair_base);
  if (opt.keep_state_ids) out.state_orig.assign(out.out_off.size(), 0u);
  out.pair_id.resize(pair_base);
  out.pair_logw.resize(pair_base);
  {
    std::atomic<size_t> next(0);
    auto work = [&]() {
      std::vector<uint32_t> lvl_cnt, newid, cur;
      std::vector<std::vector<uint32_t> > maps;
      for (;;) {
        size_t b = next.fetch_add(8);
        if (b >= nb) break;
        size_t be = std::min(nb, b + 8);
        for (; b < be; ++b) {
          const BundlePlan& P = plans[b];
          const BundleDesc& d = out.bundles[b];
          uint32_t nl = d.n_levels;
          // count states per level across members
          lvl_cnt.assign((size_t)nl + 1, 0);
          for (uint32_t p : P.pairs) {
            const PairLattice& L = lats[p];
            for (uint32_t s = 0; s < L.n_states; ++s) lvl_cnt[L.level[s] + 1]++;
          }
          for (uint32_t l = 0; l < nl; ++l) lvl_cnt[l + 1] += lvl_cnt[l];
          uint32_t* lo = out.level_off.data() + d.level_base;
          for (uint32_t l = 0; l <= nl; ++l) lo[l] = lvl_cnt[l];
          cur.assign(lvl_cnt.begin(), lvl_cnt.end() - 1);
          maps.resize(P.pairs.size());
          for (size_t m = 0; m < P.pairs.size(); ++m) {
            const PairLattice& L = lats[P.pairs[m]];
            maps[m].resize(L.n_states);
            for (uint32_t s = 0; s < L.n_states; ++s) maps[m][s] = cur[L.level[s]]++;
            if (opt.keep_state_ids)
              for (uint32_t s = 0; s < L.n_states; ++s) out.state_orig[d.off_base + maps[m][s]] = s;
            out.pair_start[d.pair_base + m] = maps[m][L.start];
            out.pair_final[d.pair_base + m] = maps[m][L.fin];
            out.pair_id[d.pair_base + m] = P.pairs[m];
            double wt = c.weight.empty() ? 1.0 : c.weight[P.pairs[m]];
            out.pair_logw[d.pair_base + m] = wt > 0 ? std::log(wt) : -std::numeric_limits<double>::infinity();
          }
          uint32_t* ioff = out.in_off.data() + d.off_base;
          uint32_t* ooff = out.out_off.data() + d.off_base;
          std::fill(ioff, ioff + d.n_states + 1, 0u);
          std::fill(ooff, ooff + d.n_states + 1, 0u);
          for (size_t m = 0; m < P.pairs.size(); ++m) {
            const PairLattice& L = lats[P.pairs[m]];
            for (auto& e : L.edges) {
              ioff[maps[m][e.dst] + 1]++;
              ooff[maps[m][e.src] + 1]++;
            }
          }
          for (uint32_t s = 0; s < d.n_states; ++s) {
            ioff[s + 1] += ioff[s];
            ooff[s + 1] += ooff[s];
          }
          uint2_t* ia = out.in_arcs.data() + d.in_base;
          uint2_t* oa = out.out_arcs.data() + d.out_base;
          std::vector<uint32_t>& ci = newid;
          ci.assign(ioff, ioff + d.n_states);
          std::vector<uint32_t> co(ooff, ooff + d.n_states);
          if (!P.cyclic) {
            for (size_t m = 0; m < P.pairs.size(); ++m) {
              const PairLattice& L = lats[P.pairs[m]];
              for (auto& e : L.edges) {
                uint32_t s = maps[m][e.src], t = maps[m][e.dst];
                ia[ci[t]++] = uint2_t{s, e.arc};
                oa[co[s]++] = uint2_t{t, e.arc};
              }
            }
          } else {
            // reference list orders (one member): out-list of a state = newest insertion first; the reversed
            // graph's list of v = reverse of (old state id ascending, list order) — derivations.h:546-550,
            // graph.cc:41-57.  Old state ids are the PairLattice ids (DFS pre-order, stably compacted).
            const PairLattice& L = lats[P.pairs[0]];
            const auto& mp = maps[0];
            std::vector<uint32_t> eo((size_t)L.n_states + 1, 0);
            for (auto& e : L.edges) eo[e.src + 1]++;
            for (uint32_t s = 0; s < L.n_states; ++s) eo[s + 1] += eo[s];
            std::vector<uint32_t> byl(L.edges.size());
            {
              std::vector<uint32_t> cc(eo.begin(), eo.end() - 1);
              for (uint32_t k = 0; k < L.edges.size(); ++k) byl[cc[L.edges[k].src]++] = k;
            }
            // list order per old state: reverse insertion
            std::vector<uint32_t> listorder;
            listorder.reserve(L.edges.size());
            for (uint32_t s = 0; s < L.n_states; ++s)
              for (uint32_t k = eo[s + 1]; k-- > eo[s];) listorder.push_back(byl[k]);
            for (uint32_t k : listorder) {
              const auto& e = L.edges[k];
              oa[co[mp[e.src]]++] = uint2_t{mp[e.dst], e.arc};
            }
            // reversed lists: walking listorder pushes to the FRONT of r[dst]; final list = reverse of that walk
            for (size_t q = listorder.size(); q-- > 0;) {
              const auto& e = L.edges[listorder[q]];
              ia[ci[mp[e.dst]]++] = uint2_t{mp[e.src], e.arc};
            }
          }
        }
      }
    };
    std::vector<std::thread> th;
    int nt2 = (int)std::min<size_t>((size_t)nt, std::max<size_t>(1, nb / 8));
    for (int t = 1; t < nt2; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  phase("layout (lanes, bundles)");
  // ---- posterior slots sorted by WFST arc id (counting sort) ----
  {
    // positions: [lane records | (up to a tile boundary) wave records | bundle out-arcs]
    const uint64_t nlane_rec = out.lane_bwd.size();
    const uint64_t nwave = out.wave_bwd.size();
    out.wave_slot_base = nwave ? (nlane_rec + out.tile - 1) / out.tile * out.tile : nlane_rec;
    const uint64_t nlane = out.wave_slot_base + nwave;  // first bundle slot
    out.n_post = nlane + out.out_arcs.size();
    if (opt.device_tables && out.n_post && out.n_post < (1ull << 32)) {  // (the engine sorts the slots and builds the tables on the device)
      out.tables_deferred = true;
      phase("slots by arc (left to the device)");
      return true;
    }
    // counting sort by arc id, stable in slot order.  Threads own contiguous arc ranges: each scans every record (a
    // sequential read) and handles the records of its own arcs, so counters and output stay private and local.
    std::vector<uint64_t> cnt(w.n_arcs + 1, 0);
    const int ns = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)nt, (nlane_rec + nwave + out.out_arcs.size()) / (1u << 16)));
    auto arc_lo = [&](int t) { return (uint64_t)((__uint128_t)w.n_arcs * (uint64_t)t / (uint64_t)ns); };
    auto in_threads = [&](const std::function<void(int)>& f) {
      std::vector<std::thread> th;
      for (int t = 1; t < ns; ++t) th.emplace_back(f, t);
      f(0);
      for (auto& x : th) x.join();
    };
    in_threads([&](int t) {
      const uint64_t lo = arc_lo(t), hi = arc_lo(t + 1);
      for (uint64_t k = 0; k < nlane_rec; ++k) {
        const uint2_t r = out.lane_bwd[k];
        if ((r.x & LANE_VALID) && r.y >= lo && r.y < hi) cnt[r.y + 1]++;
      }
      for (const uint32_t a : out.wave_bwd_arc)
        if (a != 0xffffffffu && a >= lo && a < hi) cnt[(uint64_t)a + 1]++;
      for (const auto& r : out.out_arcs)
        if (r.y >= lo && r.y < hi) cnt[r.y + 1]++;
    });
    for (uint64_t a = 0; a < w.n_arcs; ++a) cnt[a + 1] += cnt[a];
    const uint64_t total = cnt[w.n_arcs];
    out.arc_off = cnt;  // offsets before the fill pass advances the cursors
    out.slot_pos.resize(total);
    in_threads([&](int t) {
      const uint64_t lo = arc_lo(t), hi = arc_lo(t + 1);
      for (uint64_t k = 0; k < nlane_rec; ++k) {
        const uint2_t r = out.lane_bwd[k];
        if ((r.x & LANE_VALID) && r.y >= lo && r.y < hi) out.slot_pos[cnt[r.y]++] = k;
      }
      for (uint64_t k = 0; k < nwave; ++k) {
        const uint32_t a = out.wave_bwd_arc[k];
        if (a != 0xffffffffu && a >= lo && a < hi) out.slot_pos[cnt[a]++] = out.wave_slot_base + k;
      }
      for (uint64_t k = 0; k < out.out_arcs.size(); ++k) {
        const uint32_t a = out.out_arcs[k].y;
        if (a >= lo && a < hi) out.slot_pos[cnt[a]++] = nlane + k;
      }
    });
    for (uint64_t a = 0; a < w.n_arcs; ++a)
      if (out.arc_off[a + 1] - out.arc_off[a] > 64)
        for (uint64_t j = out.arc_off[a]; j < out.arc_off[a + 1]; j += 4096) {
          out.hot_chunks.push_back(a);
          out.hot_chunks.push_back(j);
          out.hot_chunks.push_back(std::min(out.arc_off[a + 1], j + 4096));
        }
  }
  phase("slots by arc (counting sort)");
  build_transpose(out, w.n_arcs, nt);
  phase("transposition tables");
  return true;
}

// see TransBucket (lattice.hpp)
void build_transpose(LatticeSet& out, uint64_t n_arcs, int nt) {
  const uint64_t N = out.slot_pos.size();
  out.t_buckets.clear();
  out.t_split_arcs.clear();
  if (N >= (1ull << 32) || N == 0) return;  // 32-bit item indices; the engine then keeps the gather path
  const uint64_t BK = out.bucket;
  {
    uint64_t a = 0;
    while (a < n_arcs) {
      uint64_t c = out.arc_off[a + 1] - out.arc_off[a];
      if (c > TRANS_HEAVY) {
        // a hub arc gets buckets of its own (one if it fits, else pieces whose sums are added atomically): its sum
        // is a workgroup-wide reduction instead of one thread's serial loop
        const bool split = c > BK;
        if (split) out.t_split_arcs.push_back((uint32_t)a);
        for (uint64_t g = out.arc_off[a]; g < out.arc_off[a + 1]; g += BK)
          out.t_buckets.push_back(TransBucket{g, (uint32_t)std::min<uint64_t>(BK, out.arc_off[a + 1] - g),
                                              (uint32_t)a, 1u, TRANS_SINGLE | (split ? TRANS_SPLIT : 0u)});
        ++a;
        continue;
      }
      uint64_t e = a, items = 0;
      while (e < n_arcs && e - a < BK) {
        uint64_t ce = out.arc_off[e + 1] - out.arc_off[e];
        if (ce > TRANS_HEAVY || items + ce > BK) break;
        items += ce;
        ++e;
      }
      out.t_buckets.push_back(TransBucket{out.arc_off[a], (uint32_t)items, (uint32_t)a, (uint32_t)(e - a), 0u});
      a = e;
    }
  }
  out.t_a_off.assign(n_arcs, 0);
  for (const TransBucket& B : out.t_buckets)
    if (!(B.flags & TRANS_SINGLE))
      for (uint32_t a = 0; a < B.n_arcs; ++a) out.t_a_off[B.arc_lo + a] = (uint16_t)(out.arc_off[B.arc_lo + a] - B.item_base);
  out.t_b_arc.assign(N, 0);
  out.t_b_rank.assign(N, 0);
  out.t_b_src.assign(N, 0);
  out.t_t_pos.assign(N, 0);
  out.t_t_src.assign(N, 0);
  const uint64_t tile_sz = out.tile;
  const uint64_t n_tiles = (out.n_post + tile_sz - 1) / tile_sz;
  out.t_tile_base.assign(n_tiles + 1, 0);
  std::vector<uint64_t> pos_of(N);  // position of the item at bucket-major index J
  {
    std::atomic<size_t> next(0);
    auto work = [&]() {
      std::vector<std::pair<uint64_t, uint32_t>> items;  // (position, rank in arc-sorted order)
      for (;;) {
        size_t b = next.fetch_add(16);
        if (b >= out.t_buckets.size()) break;
        size_t be = std::min(out.t_buckets.size(), b + 16);
        for (; b < be; ++b) {
          const TransBucket& B = out.t_buckets[b];
          items.resize(B.n_items);
          for (uint32_t r = 0; r < B.n_items; ++r) items[r] = {out.slot_pos[B.item_base + r], r};
          std::sort(items.begin(), items.end());
          uint64_t a = B.arc_lo;
          // arc of rank r: walk arc_off (ranks are arc-sorted); build rank -> local arc first
          std::vector<uint16_t> arc_of_rank(B.n_items);
          if (B.flags & TRANS_SINGLE) {
            std::fill(arc_of_rank.begin(), arc_of_rank.end(), (uint16_t)0);
          } else {
            for (uint32_t r = 0; r < B.n_items; ++r) {
              while (out.arc_off[a + 1] <= B.item_base + r) ++a;
              arc_of_rank[r] = (uint16_t)(a - B.arc_lo);
            }
          }
          for (uint32_t j = 0; j < B.n_items; ++j) {
            const uint64_t J = B.item_base + j;
            out.t_b_arc[J] = arc_of_rank[items[j].second];
            out.t_b_rank[J] = (uint16_t)items[j].second;
            pos_of[J] = items[j].first;
          }
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  // tile-major order: stable counting sort of the bucket-major sequence by tile.  Threads own contiguous ranges of J: a
  // histogram per thread, cursors from the prefix over (tile, thread) -- thread 0's items of a tile first --, then every
  // thread scatters its own range: the same order as one thread walking J = 0 .. N - 1
  {
    const int ns = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(nt, 1), N / (1u << 18)));
    auto j_lo = [&](int t) { return (uint64_t)((__uint128_t)N * (uint64_t)t / (uint64_t)ns); };
    std::vector<std::vector<uint64_t>> hist((size_t)ns, std::vector<uint64_t>(n_tiles, 0));
    auto in_threads = [&](const std::function<void(int)>& f) {
      std::vector<std::thread> th;
      for (int t = 1; t < ns; ++t) th.emplace_back(f, t);
      f(0);
      for (auto& x : th) x.join();
    };
    in_threads([&](int t) {
      std::vector<uint64_t>& h = hist[(size_t)t];
      for (uint64_t J = j_lo(t), e = j_lo(t + 1); J < e; ++J) h[pos_of[J] / tile_sz]++;
    });
    uint64_t acc = 0;
    for (uint64_t tile = 0; tile < n_tiles; ++tile) {
      out.t_tile_base[tile] = acc;
      for (int t = 0; t < ns; ++t) {
        const uint64_t c = hist[(size_t)t][tile];
        hist[(size_t)t][tile] = acc;  // becomes thread t's cursor for this tile
        acc += c;
      }
    }
    out.t_tile_base[n_tiles] = acc;
    in_threads([&](int t) {
      std::vector<uint64_t>& cur = hist[(size_t)t];
      for (uint64_t J = j_lo(t), e = j_lo(t + 1); J < e; ++J) {
        const uint64_t tile = pos_of[J] / tile_sz, I = cur[tile]++;
        out.t_t_src[I] = (uint32_t)J;
        out.t_t_pos[I] = (uint16_t)(pos_of[J] - tile * tile_sz);
        out.t_b_src[J] = (uint32_t)I;
      }
    });
  }
}

}  // namespace carmel_hip
