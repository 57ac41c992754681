// unrolled.cpp: eligibility test and tables of the unrolled (position x state) sweep -- see unrolled.hpp.
#include "unrolled.hpp"
#include "options.hpp"

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <unordered_map>

namespace carmel_hip {


bool build_unrolled(const HostWfst& w, const HostCorpus& c, int threads, UnrolledModel& M) {
  M = UnrolledModel();
  auto no = [&](const char* why) {
    M.why = why;
    return false;
  };
  const uint32_t S = w.n_states;
  if (S == 0 || S > UNROLLED_WIDE_MAX_STATES) return no("more than 1024 states");
  // which tape carries the symbols?
  bool in_used = false, out_used = false;
  for (uint64_t a = 0; a < w.n_arcs; ++a) {
    if (w.in[a]) in_used = true;
    if (w.out[a]) out_used = true;
  }
  if (in_used && out_used) return no("arcs read and write");
  M.tape = in_used ? 0 : 1;
  const std::vector<uint32_t>& lab = in_used ? w.in : w.out;
  M.S = S;
  M.start = 0;
  M.fin = w.final_state;
  // dense symbol ids
  std::unordered_map<uint32_t, uint32_t> dense;
  for (uint64_t a = 0; a < w.n_arcs; ++a)
    if (lab[a] && !dense.count(lab[a])) {
      uint32_t id = (uint32_t)dense.size();
      dense.emplace(lab[a], id);
    }
  const uint32_t V = (uint32_t)dense.size();
  if (V == 0) return no("no symbol arcs");
  if (V > 65535) return no("more than 65535 symbols");
  M.V = V;
  // *e*:*e* arcs: must be acyclic; order them by the longest *e*-path into their source
  std::vector<uint32_t> eps;
  for (uint64_t a = 0; a < w.n_arcs; ++a)
    if (!lab[a]) eps.push_back((uint32_t)a);
  std::vector<uint32_t> depth(S, 0);
  {
    std::vector<uint32_t> indeg(S, 0);
    for (uint32_t a : eps) indeg[w.dst[a]]++;
    std::vector<uint32_t> q;
    for (uint32_t s = 0; s < S; ++s)
      if (!indeg[s]) q.push_back(s);
    size_t done = 0;
    std::vector<std::vector<uint32_t>> outs(S);
    for (uint32_t a : eps) outs[w.src[a]].push_back(a);
    while (done < q.size()) {
      uint32_t s = q[done++];
      for (uint32_t a : outs[s]) {
        uint32_t d = w.dst[a];
        depth[d] = std::max(depth[d], depth[s] + 1);
        if (--indeg[d] == 0) q.push_back(d);
      }
    }
    if (q.size() != S) return no("*e*:*e* cycle");
  }
  std::stable_sort(eps.begin(), eps.end(), [&](uint32_t a, uint32_t b) { return depth[w.src[a]] < depth[w.src[b]]; });
  for (uint32_t a : eps) {
    M.e_arc.push_back(a);
    M.e_src.push_back((uint16_t)w.src[a]);
    M.e_dst.push_back((uint16_t)w.dst[a]);
  }
  // ELL slabs
  std::vector<std::vector<uint32_t>> by_sym(V);
  for (uint64_t a = 0; a < w.n_arcs; ++a)
    if (lab[a]) by_sym[dense[lab[a]]].push_back((uint32_t)a);
  M.f_off.assign(V + 1, 0);
  M.b_off.assign(V + 1, 0);
  std::vector<uint32_t> cnt(S), fdeg(V, 0), bdeg(V, 0);
  uint64_t fsum = 0, bsum = 0;
  uint32_t fmax = 0, bmax = 0;
  for (uint32_t x = 0; x < V; ++x) {
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (uint32_t a : by_sym[x]) fdeg[x] = std::max(fdeg[x], ++cnt[w.dst[a]]);
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (uint32_t a : by_sym[x]) bdeg[x] = std::max(bdeg[x], ++cnt[w.src[a]]);
    fsum += fdeg[x];
    bsum += bdeg[x];
    fmax = std::max(fmax, fdeg[x]);
    bmax = std::max(bmax, bdeg[x]);
  }
  // slabs of one size when that costs at most 30 % more rows: the sweep's loop bounds and addresses are then the same
  // for every symbol (scalar arithmetic; no per-lane clamping or masking of the rows)
  if ((uint64_t)fmax * V * 10 <= fsum * 13 && (uint64_t)bmax * V * 10 <= bsum * 13 && !lib_opt("unrolled_ragged")) {
    M.f_deg_u = fmax;
    M.b_deg_u = bmax;
    std::fill(fdeg.begin(), fdeg.end(), fmax);
    std::fill(bdeg.begin(), bdeg.end(), bmax);
  }
  for (uint32_t x = 0; x < V; ++x) {
    M.f_off[x + 1] = M.f_off[x] + fdeg[x] * S;
    M.b_off[x + 1] = M.b_off[x] + bdeg[x] * S;
  }
  M.f_arc.assign(M.f_off[V], 0xffffffffu);
  M.f_src.assign(M.f_off[V], 0);
  M.b_arc.assign(M.b_off[V], 0xffffffffu);
  M.b_dst.assign(M.b_off[V], 0);
  for (uint32_t x = 0; x < V; ++x) {
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (uint32_t a : by_sym[x]) {
      const uint32_t k = M.f_off[x] + (cnt[w.dst[a]]++) * S + w.dst[a];
      M.f_arc[k] = a;
      M.f_src[k] = (uint16_t)w.src[a];
    }
    std::fill(cnt.begin(), cnt.end(), 0u);
    const uint32_t bd = (M.b_off[x + 1] - M.b_off[x]) / S;
    for (uint32_t a : by_sym[x]) {
      // row of the arc in its source's column: rotated by the source, so that the lanes of one row tend to point at
      // different destinations / parameters (their posterior adds then do not collide in LDS)
      const uint32_t row = (cnt[w.src[a]]++ + w.src[a]) % bd;
      const uint32_t k = M.b_off[x] + row * S + w.src[a];
      M.b_arc[k] = a;
      M.b_dst[k] = (uint16_t)w.dst[a];
    }
  }
  // reachability tables as state sets (W 64-bit words per set): T[x][src] = destinations, TR[x][dst] = sources;
  // multiplicities for the stats
  const uint32_t W = (S + 63) / 64;
  if ((uint64_t)V * S * W * 16 > (1ull << 30)) return no("reachability tables too large");
  auto setbit = [&](uint64_t* m, uint32_t s) { m[s >> 6] |= 1ull << (s & 63); };
  auto getbit = [&](const uint64_t* m, uint32_t s) { return (m[s >> 6] >> (s & 63)) & 1ull; };
  std::vector<uint64_t> T((size_t)V * S * W, 0), TR((size_t)V * S * W, 0), E((size_t)S * W, 0), ER((size_t)S * W, 0);
  bool multi = false;
  for (uint32_t x = 0; x < V; ++x)
    for (uint32_t a : by_sym[x]) {
      uint64_t* row = &T[((size_t)x * S + w.src[a]) * W];
      if (getbit(row, w.dst[a])) multi = true;
      setbit(row, w.dst[a]);
      setbit(&TR[((size_t)x * S + w.dst[a]) * W], w.src[a]);
    }
  std::vector<uint16_t> mult;  // [x][src][dst] when some (x, src, dst) has several arcs
  if (multi) {
    if ((uint64_t)V * S * S * 2 > (1ull << 30)) return no("parallel arcs on too large a model");
    mult.assign((size_t)V * S * S, 0);
    for (uint32_t x = 0; x < V; ++x)
      for (uint32_t a : by_sym[x]) mult[((size_t)x * S + w.src[a]) * S + w.dst[a]]++;
  }
  for (uint32_t a : eps) {
    setbit(&E[(size_t)w.src[a] * W], w.dst[a]);
    setbit(&ER[(size_t)w.dst[a] * W], w.src[a]);
  }
  // m |= everything reachable from m along adj (rows of W words per state)
  auto closure = [&](uint64_t* m, const std::vector<uint64_t>& adj, std::vector<uint64_t>& tmp) {
    if (eps.empty()) return;
    for (;;) {
      tmp.assign(W, 0);
      for (uint32_t wd = 0; wd < W; ++wd)
        for (uint64_t r = m[wd]; r;) {
          const uint32_t s = wd * 64 + (uint32_t)__builtin_ctzll(r);
          r &= r - 1;
          for (uint32_t k = 0; k < W; ++k) tmp[k] |= adj[(size_t)s * W + k];
        }
      bool grew = false;
      for (uint32_t k = 0; k < W; ++k) {
        if (tmp[k] & ~m[k]) grew = true;
        m[k] |= tmp[k];
      }
      if (!grew) return;
    }
  };
  // ---- corpus ----
  const uint64_t n = c.n_pairs;
  M.has_deriv.assign(n, 0);
  const std::vector<uint64_t>& t_off = M.tape ? c.out_off : c.in_off;
  const std::vector<uint32_t>& t_sym = M.tape ? c.out_sym : c.in_sym;
  const std::vector<uint64_t>& o_off = M.tape ? c.in_off : c.out_off;
  std::vector<uint64_t> st_states(n, 0), st_arcs(n, 0), st_expl(n, 0);
  int nt = threads > 0 ? threads : std::min(32, (int)std::thread::hardware_concurrency());  // see lattice.cpp
  if (nt < 1) nt = 1;
  std::atomic<uint64_t> next(0);
  std::atomic<uint32_t> maxlen(0);
  auto work = [&]() {
    std::vector<uint64_t> F, B, tmp, live(W), dm(W);
    std::vector<uint16_t> xs;
    for (;;) {
      uint64_t p0 = next.fetch_add(256);
      if (p0 >= n) break;
      for (uint64_t p = p0; p < std::min(n, p0 + 256); ++p) {
        if (o_off[p + 1] != o_off[p]) continue;  // the other string must be empty
        const uint64_t L = t_off[p + 1] - t_off[p];
        xs.resize(L);
        bool known = true;
        for (uint64_t o = 0; o < L; ++o) {
          auto it = dense.find(t_sym[t_off[p] + o]);
          if (it == dense.end()) {
            known = false;
            break;
          }
          xs[o] = (uint16_t)it->second;
        }
        if (!known) continue;
        F.assign((L + 1) * W, 0);
        B.assign((L + 1) * W, 0);
        setbit(&F[0], M.start);
        closure(&F[0], E, tmp);
        uint64_t expl = 0;
        for (uint64_t o = 0; o < L; ++o) {
          uint64_t* nx = &F[(o + 1) * W];
          for (uint32_t wd = 0; wd < W; ++wd)
            for (uint64_t r = F[o * W + wd]; r;) {
              const uint32_t s = wd * 64 + (uint32_t)__builtin_ctzll(r);
              r &= r - 1;
              const uint64_t* row = &T[((size_t)xs[o] * S + s) * W];
              for (uint32_t k = 0; k < W; ++k) {
                nx[k] |= row[k];
                expl += (uint64_t)__builtin_popcountll(row[k]);
              }
            }
          closure(nx, E, tmp);
        }
        st_expl[p] = expl;
        if (!getbit(&F[L * W], M.fin)) continue;
        M.has_deriv[p] = 1;
        setbit(&B[L * W], M.fin);
        closure(&B[L * W], ER, tmp);
        for (uint64_t o = L; o-- > 0;) {
          uint64_t* pv = &B[o * W];
          for (uint32_t wd = 0; wd < W; ++wd)
            for (uint64_t r = B[(o + 1) * W + wd]; r;) {
              const uint32_t d = wd * 64 + (uint32_t)__builtin_ctzll(r);
              r &= r - 1;
              const uint64_t* row = &TR[((size_t)xs[o] * S + d) * W];
              for (uint32_t k = 0; k < W; ++k) pv[k] |= row[k];
            }
          closure(pv, ER, tmp);
        }
        uint64_t ns = 0, na = 0;
        for (uint64_t o = 0; o <= L; ++o) {
          for (uint32_t k = 0; k < W; ++k) {
            live[k] = F[o * W + k] & B[o * W + k];
            ns += (uint64_t)__builtin_popcountll(live[k]);
          }
          for (uint32_t wd = 0; wd < W; ++wd)
            for (uint64_t r = live[wd]; r;) {
              const uint32_t s = wd * 64 + (uint32_t)__builtin_ctzll(r);
              r &= r - 1;
              for (uint32_t k = 0; k < W; ++k)  // *e*:*e* arcs inside the position (single arcs assumed)
                na += (uint64_t)__builtin_popcountll(E[(size_t)s * W + k] & live[k]);
              if (o < L) {
                const uint64_t* row = &T[((size_t)xs[o] * S + s) * W];
                for (uint32_t k = 0; k < W; ++k) dm[k] = row[k] & F[(o + 1) * W + k] & B[(o + 1) * W + k];
                if (!multi)
                  for (uint32_t k = 0; k < W; ++k) na += (uint64_t)__builtin_popcountll(dm[k]);
                else
                  for (uint32_t k = 0; k < W; ++k)
                    for (uint64_t q = dm[k]; q;) {
                      const uint32_t d = k * 64 + (uint32_t)__builtin_ctzll(q);
                      q &= q - 1;
                      na += mult[((size_t)xs[o] * S + s) * S + d];
                    }
              }
            }
        }
        st_states[p] = ns;
        st_arcs[p] = na;
        uint32_t cur = maxlen.load();
        while ((uint32_t)L > cur && !maxlen.compare_exchange_weak(cur, (uint32_t)L)) {
        }
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  M.max_len = maxlen.load();
  M.seq_off.assign(1, 0);
  // pairs in order of decreasing length: the pairs that share a wavefront are neighbours in this list
  std::vector<uint32_t> by_len;
  for (uint64_t p = 0; p < n; ++p) {
    M.explored_arcs += st_expl[p];
    if (M.has_deriv[p]) by_len.push_back((uint32_t)p);
  }
  std::stable_sort(by_len.begin(), by_len.end(),
                   [&](uint32_t x, uint32_t y) { return t_off[x + 1] - t_off[x] > t_off[y + 1] - t_off[y]; });
  for (uint32_t p : by_len) {
    M.lattice_states += st_states[p];
    M.lattice_arcs += st_arcs[p];
    M.pair_id.push_back((uint32_t)p);
    M.pair_weight.push_back(c.weight.empty() ? 1.0 : c.weight[p]);
    for (uint64_t o = t_off[p]; o < t_off[p + 1]; ++o) M.seq_sym.push_back((uint16_t)dense[t_sym[o]]);
    M.seq_off.push_back(M.seq_sym.size());
  }
  M.ok = true;
  return true;
}

}  // namespace carmel_hip
