// engine_unrolled.cpp: the trainer's side of the unrolled sweep (unrolled.hpp): eligibility, upload, E-step.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "dense.hpp"
#include "options.hpp"
#include "engine.hpp"
#include "unrolled_args.hpp"
#include <numeric>

// The rank-1 dense form (dense.hpp).  Eligible: a cascade, at most DENSE_MAX_STATES states, no *e*:*e* arcs, and for every
// arc (s -> s', symbol c) the parameters of its chain that collect NO counts are the same for all c (they make A[s][s']),
// the ones that do are the same for all s (at most one: B[c][s'] and its accumulator), and an arc exists exactly when both
// its A and its B entry do.  Returns false (with t->dense clear) when the model is not of that shape.
static int dense_try_build(carmel_hip_trainer* t, const UnrolledModel& M, const std::vector<uint32_t>& slot_of, uint32_t n_slots,
                           bool& built) {
  built = false;
  t->dense = false;
  if (const char* e = lib_opt("dense"))
    if (atoi(e) == 0) return CARMEL_HIP_OK;
  const uint32_t S = M.S, V = M.V, SP = dense_padded_states(S);
  if (!t->cascade || !SP || S > DENSE_MAX_STATES || M.e_arc.size() > 16 || V == 0 || V > 4096) return CARMEL_HIP_OK;
  for (uint32_t a : M.e_arc) {  // *e*:*e* arcs are swept but collect nothing here: all their parameters must be count-free
    const uint32_t ch = t->w.group[a];
    for (uint64_t j = t->h_chain_off[ch]; j < t->h_chain_off[ch + 1]; ++j)
      if (slot_of[t->h_chain_param[j]] != 0xffffffffu) return CARMEL_HIP_OK;
  }
  if ((size_t)V * SP * 10 + (size_t)n_slots * 8 > 60 * 1024) return CARMEL_HIP_OK;  // B, its slots and the accumulators live in LDS
  std::vector<std::vector<uint32_t> > la((size_t)SP * SP), ub((size_t)V * SP);
  std::vector<uint8_t> a_has((size_t)SP * SP, 0), b_has((size_t)V * SP, 0);
  std::vector<uint8_t> seen((size_t)S * S * V, 0);
  std::vector<uint32_t> L, U;
  uint64_t n_seen = 0;
  for (uint32_t x = 0; x < V; ++x)
    for (uint32_t e = M.f_off[x]; e < M.f_off[x + 1]; ++e) {
      const uint32_t a = M.f_arc[e];
      if (a == 0xffffffffu) continue;
      const uint32_t dst = (e - M.f_off[x]) % S, src = M.f_src[e];
      uint8_t& sn = seen[((size_t)src * S + dst) * V + x];
      if (sn) return CARMEL_HIP_OK;  // two arcs between the same states with the same symbol
      sn = 1;
      ++n_seen;
      L.clear();
      U.clear();
      const uint32_t ch = t->w.group[a];
      for (uint64_t j = t->h_chain_off[ch]; j < t->h_chain_off[ch + 1]; ++j) {
        const uint32_t p = (uint32_t)t->h_chain_param[j];
        (slot_of[p] == 0xffffffffu ? L : U).push_back(p);
      }
      std::sort(L.begin(), L.end());
      std::sort(U.begin(), U.end());
      if (U.size() > 1) return CARMEL_HIP_OK;
      const size_t ka = (size_t)src * SP + dst, kb = (size_t)x * SP + dst;
      if (!a_has[ka]) {
        a_has[ka] = 1;
        la[ka] = L;
      } else if (la[ka] != L)
        return CARMEL_HIP_OK;
      if (!b_has[kb]) {
        b_has[kb] = 1;
        ub[kb] = U;
      } else if (ub[kb] != U)
        return CARMEL_HIP_OK;
    }
  {  // rank-1 support: every (A entry, B entry) pair that meets in a destination state is an arc
    uint64_t want = 0;
    for (uint32_t d = 0; d < S; ++d) {
      uint64_t na = 0, nb = 0;
      for (uint32_t s0 = 0; s0 < S; ++s0) na += a_has[(size_t)s0 * SP + d];
      for (uint32_t x = 0; x < V; ++x) nb += b_has[(size_t)x * SP + d];
      want += na * nb;
    }
    if (want != n_seen || !n_seen) return CARMEL_HIP_OK;
  }
  // strings: sorted by length, 64 per wavefront
  const size_t np = M.pair_id.size();
  std::vector<uint32_t> order(np);
  std::iota(order.begin(), order.end(), 0u);
  auto len_of = [&](uint32_t k) { return (uint32_t)(M.seq_off[k + 1] - M.seq_off[k]); };
  for (uint32_t k = 0; k < np; ++k)
    if (len_of(k) == 0) return CARMEL_HIP_OK;  // (an empty string has no position to sweep)
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return len_of(a) > len_of(b); });
  const uint32_t ng = (uint32_t)((np + 63) / 64);
  std::vector<uint64_t> sym_off(ng + 1, 0), vbuf_off(ng + 1, 0);
  for (uint32_t g = 0; g < ng; ++g) {
    const uint64_t tmax = len_of(order[(size_t)g * 64]);
    sym_off[g + 1] = sym_off[g] + tmax * 64;
    vbuf_off[g + 1] = vbuf_off[g] + tmax * SP * 64;
  }
  if (vbuf_off[ng] * 8 > (16ull << 30)) return CARMEL_HIP_OK;
  std::vector<uint16_t> sym(sym_off[ng], 0);
  std::vector<uint32_t> len((size_t)ng * 64, 0), pair((size_t)ng * 64, 0);
  std::vector<double> weight((size_t)ng * 64, 0.0);
  for (size_t k = 0; k < np; ++k) {
    const uint32_t q = order[k], g = (uint32_t)(k / 64), l = (uint32_t)(k % 64), T = len_of(q);
    len[k] = T;
    pair[k] = M.pair_id[q];
    weight[k] = M.pair_weight[q];
    for (uint32_t tt = 0; tt < T; ++tt) sym[sym_off[g] + (size_t)tt * 64 + l] = M.seq_sym[M.seq_off[q] + tt];
  }
  std::vector<uint32_t> a_off(la.size() + 1, 0), a_par, b_off(ub.size() + 1, 0), b_par;
  for (size_t k = 0; k < la.size(); ++k) {
    a_par.insert(a_par.end(), la[k].begin(), la[k].end());
    a_off[k + 1] = (uint32_t)a_par.size();
  }
  std::vector<uint16_t> bslot(ub.size(), (uint16_t)0xffffu);
  for (size_t k = 0; k < ub.size(); ++k) {
    b_par.insert(b_par.end(), ub[k].begin(), ub[k].end());
    b_off[k + 1] = (uint32_t)b_par.size();
    if (ub[k].size() == 1) bslot[k] = (uint16_t)slot_of[ub[k][0]];
  }
  if (a_par.empty()) a_par.push_back(0);
  if (b_par.empty()) b_par.push_back(0);
  hipStream_t s = t->stream;
  HIPCHK(t->d_a_off.upload(a_off, s));
  HIPCHK(t->d_a_par.upload(a_par, s));
  HIPCHK(t->d_b_off.upload(b_off, s));
  HIPCHK(t->d_b_par.upload(b_par, s));
  HIPCHK(t->d_a_has.upload(a_has, s));
  HIPCHK(t->d_b_has.upload(b_has, s));
  HIPCHK(t->d_Bslot.upload(bslot, s));
  HIPCHK(t->d_sym.upload(sym, s));
  HIPCHK(t->d_sym_off.upload(sym_off, s));
  HIPCHK(t->d_vbuf_off.upload(vbuf_off, s));
  HIPCHK(t->d_len.upload(len, s));
  HIPCHK(t->d_pair.upload(pair, s));
  HIPCHK(t->d_weight.upload(weight, s));
  HIPCHK(t->d_A.alloc((size_t)SP * SP));
  HIPCHK(t->d_AT.alloc((size_t)SP * SP));
  HIPCHK(t->d_B.alloc((size_t)V * SP));
  HIPCHK(t->d_vbuf.alloc(vbuf_off[ng]));
  HIPCHK(t->d_zbuf.alloc(sym_off[ng]));
  HIPCHK(t->d_afbuf.alloc((size_t)ng * 64));
  HIPCHK(t->d_partial.alloc((size_t)ng * n_slots));
  HIPCHK(hipStreamSynchronize(s));
  t->d_SP = SP;
  t->d_groups = ng;
  t->dense = true;
  if (lib_opt("timing"))
    fprintf(stderr, "timing: dense sweep S=%u padded=%u symbols=%u eps=%zu slots=%u strings=%zu positions=%llu groups=%u parked_bytes=%llu\n", S, SP,
            V, M.e_arc.size(), n_slots, np, (unsigned long long)M.seq_sym.size(), ng, (unsigned long long)(vbuf_off[ng] * 8));
  built = true;
  return CARMEL_HIP_OK;
}

// Tries to set the trainer up for the unrolled sweep.  Returns CARMEL_HIP_OK with t->unrolled set when the model and
// the corpus are eligible, CARMEL_HIP_OK with t->unrolled clear when they are not (the caller then builds explicit
// lattices), an error code on a HIP failure.
int unrolled_try_build(carmel_hip_trainer* t, int host_threads, uint8_t* has_derivation, carmel_hip_lattice_stats* stats) {
  t->unrolled = false;
  if (!t->allow_unrolled) return CARMEL_HIP_OK;  // carmel_hip_set_layout_policy: explicit lattices on every rank
  bool forced = false;  // CARMEL_HIP_UNROLLED: 0 = never, 1 = whenever eligible, unset = when it pays (density test below)
  if (const char* e = lib_opt("unrolled")) {
    if (atoi(e) == 0) return CARMEL_HIP_OK;
    forced = true;
  }
  auto t0 = std::chrono::steady_clock::now();
  UnrolledModel& M = t->um;
  if (!build_unrolled(t->w, t->corpus, host_threads, M)) return CARMEL_HIP_OK;
  // accumulator slots: the arc itself, or the unlocked parameters of its chain
  const uint64_t n_arcs = t->w.n_arcs;
  std::vector<uint16_t> arc_slot((size_t)n_arcs * UNROLLED_MAX_CHAIN, (uint16_t)UNROLLED_NO_SLOT);
  uint32_t n_slots = 0;
  std::vector<double> uses, wprior;
  std::vector<uint32_t> slot_of;  // cascade: parameter -> accumulator slot (0xffffffff: locked)
  if (!t->cascade) {
    if (n_arcs > UNROLLED_MAX_SLOTS) return CARMEL_HIP_OK;
    n_slots = (uint32_t)n_arcs;
    for (uint64_t a = 0; a < n_arcs; ++a) arc_slot[a * UNROLLED_MAX_CHAIN] = (uint16_t)a;
  } else {
    // one accumulator per UNLOCKED parameter (a locked language model of 20 000 arcs under a 729-parameter channel
    // needs 729 slots)
    slot_of.assign(t->n_params, 0xffffffffu);
    for (uint64_t p = 0; p < t->n_params; ++p)
      if (t->h_param_group[p] != CARMEL_HIP_LOCKED_GROUP) slot_of[p] = n_slots++;
    if (n_slots > UNROLLED_MAX_SLOTS || n_slots > n_arcs) return CARMEL_HIP_OK;
    uses.assign(t->n_params, 0.0);
    if (!t->h_arc_prior_w.empty()) wprior.assign(t->n_params, 0.0);  // carmel -U
    for (uint64_t a = 0; a < n_arcs; ++a) {
      const uint32_t ch = t->w.group[a];
      uint32_t k = 0;
      for (uint64_t j = t->h_chain_off[ch]; j < t->h_chain_off[ch + 1]; ++j) {
        const uint64_t p = t->h_chain_param[j];
        if (slot_of[p] == 0xffffffffu) continue;
        if (k == UNROLLED_MAX_CHAIN) return CARMEL_HIP_OK;  // longer chains: explicit lattices
        arc_slot[a * UNROLLED_MAX_CHAIN + k++] = (uint16_t)slot_of[p];
        uses[p] += 1.0;
        if (!wprior.empty()) wprior[p] += t->h_arc_prior_w[a];
      }
    }
  }
  // LDS: accumulators + per wave (max_len + 1) rows of S values, the scales, one row of beta
  const bool wide = M.S > UNROLLED_MAX_STATES;  // a workgroup per pair, a thread per state (unrolled_wide_kernel)
  uint32_t n_waves = wide ? 1u : unrolled_waves(n_slots, M.max_len, M.S);
  if (wide && ((size_t)n_slots + 2 * (size_t)M.S + 16 + M.max_len + 2) * sizeof(double) > 150 * 1024) n_waves = 0;
  if (!n_waves) return CARMEL_HIP_OK;  // a pair too long for LDS: explicit lattices
  if (M.pair_id.empty()) return fail(CARMEL_HIP_ERR_NO_DERIV, "No training example had a derivation");
  if (!forced) {
    // The sweep visits every table entry of every position, reachable or not.  That pays when most (position, state)
    // nodes are live (decipherment: every state at every position); on sparse lattices explicit storage does orders
    // of magnitude less work -- kept unless the explicit lattices would not fit comfortably.
    double dense = 0.0;  // table entries the sweep would visit
    for (size_t k = 0; k < M.seq_sym.size(); ++k) dense += (double)(M.f_off[M.seq_sym[k] + 1] - M.f_off[M.seq_sym[k]]);
    const double density = dense > 0 ? (double)M.lattice_arcs / dense : 1.0;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const bool explicit_fits = (double)M.lattice_arcs * 64.0 < 0.25 * (double)free_b && M.lattice_arcs < (1ull << 31);
    if (density < 0.05 && explicit_fits) return CARMEL_HIP_OK;
  }
  hipStream_t s = t->stream;
  std::vector<uint16_t> e_slot(M.e_arc.size() * UNROLLED_MAX_CHAIN, (uint16_t)UNROLLED_NO_SLOT);
  for (size_t k = 0; k < M.e_arc.size(); ++k)
    for (uint32_t j = 0; j < UNROLLED_MAX_CHAIN; ++j) e_slot[k * UNROLLED_MAX_CHAIN + j] = arc_slot[(size_t)M.e_arc[k] * UNROLLED_MAX_CHAIN + j];
  {  // table offsets in rows of S entries (the kernel's per-lane arithmetic stays free of divisions)
    std::vector<uint32_t> fr(M.f_off.size()), br(M.b_off.size());
    for (size_t k = 0; k < fr.size(); ++k) fr[k] = M.f_off[k] / M.S;
    for (size_t k = 0; k < br.size(); ++k) br[k] = M.b_off[k] / M.S;
    HIPCHK(t->u_f_off.upload(fr, s));
    HIPCHK(t->u_b_off.upload(br, s));
  }
  HIPCHK(t->u_f_arc.upload(M.f_arc, s));
  HIPCHK(t->u_b_arc.upload(M.b_arc, s));
  HIPCHK(t->u_e_arc.upload(M.e_arc, s));
  {
    // packed tables: weight (refreshed every E-step), other end, slots -- one 16-byte load per entry
    std::vector<URec> fr(M.f_arc.size()), br(M.b_arc.size());
    for (size_t k = 0; k < fr.size(); ++k) fr[k] = URec{0.0, (uint32_t)M.f_src[k] | (UNROLLED_NO_SLOT << 16), UNROLLED_NO_SLOT | (UNROLLED_NO_SLOT << 16)};
    for (size_t k = 0; k < br.size(); ++k) {
      uint32_t s0 = UNROLLED_NO_SLOT, s1 = UNROLLED_NO_SLOT, s2 = UNROLLED_NO_SLOT;
      if (M.b_arc[k] != 0xffffffffu) {
        s0 = arc_slot[(size_t)M.b_arc[k] * UNROLLED_MAX_CHAIN];
        s1 = arc_slot[(size_t)M.b_arc[k] * UNROLLED_MAX_CHAIN + 1];
        s2 = arc_slot[(size_t)M.b_arc[k] * UNROLLED_MAX_CHAIN + 2];
      }
      br[k] = URec{0.0, (uint32_t)M.b_dst[k] | (s2 << 16), s0 | (s1 << 16)};
    }
    HIPCHK(t->u_f_rec.upload(fr, s));
    HIPCHK(t->u_b_rec.upload(br, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(t->u_e_src.upload(M.e_src, s));
  HIPCHK(t->u_e_dst.upload(M.e_dst, s));
  HIPCHK(t->u_e_slot.upload(e_slot, s));
  HIPCHK(t->u_seq_off.upload(M.seq_off, s));
  HIPCHK(t->u_seq_sym.upload(M.seq_sym, s));
  HIPCHK(t->u_pair_id.upload(M.pair_id, s));
  HIPCHK(t->u_pair_weight.upload(M.pair_weight, s));
  HIPCHK(t->u_We.alloc(M.e_arc.size()));
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, t->device);
  // LDS holds only accumulators and scales: several workgroups share a CU, their waves hide the table latency
  const uint32_t per_wave = M.S <= 16 ? 4u : M.S <= 32 ? 2u : 1u;  // pairs side by side in one wavefront
  const int wg_per_cu = 2;
  t->u_n_wg = (uint32_t)std::min<uint64_t>((uint64_t)n_cu * wg_per_cu, (M.pair_id.size() + n_waves * per_wave - 1) / (n_waves * per_wave));
  if (wide) {
    t->u_n_wg = (uint32_t)std::min<uint64_t>((uint64_t)n_cu * 2, M.pair_id.size());
    HIPCHK(t->u_scratch.alloc(unrolled_wide_scratch_doubles(t->u_n_wg, M.S, M.max_len)));
  } else
    HIPCHK(t->u_scratch.alloc(unrolled_scratch_doubles(t->u_n_wg, n_waves, M.max_len)));
  t->u_n_slots = n_slots;
  HIPCHK(t->u_partial.alloc((size_t)t->u_n_wg * n_slots));
  if (t->cascade) {
    HIPCHK(t->u_param_uses.upload(uses, s));
    HIPCHK(t->u_param_wprior.upload(wprior, s));
    HIPCHK(t->u_slot_of.upload(slot_of, s));
    HIPCHK(t->u_em_param.alloc(n_slots));
    HIPCHK(t->u_best_param.alloc(n_slots));
  }
  HIPCHK(t->pair_logprob.alloc(t->corpus.n_pairs));
  {
    std::vector<double> pw(t->corpus.n_pairs);
    for (uint64_t p = 0; p < t->corpus.n_pairs; ++p)
      pw[p] = M.has_deriv[p] ? (t->corpus.weight.empty() ? 1.0 : t->corpus.weight[p]) : -1.0;
    HIPCHK(t->pair_w.upload(pw, s));
    HIPCHK(t->scalar_partial.alloc(3 * 256));
  }
  HIPCHK(launch_fill(t->pair_logprob.p, -std::numeric_limits<double>::infinity(), t->corpus.n_pairs, s));
  HIPCHK(hipStreamSynchronize(s));
  if (has_derivation) std::memcpy(has_derivation, M.has_deriv.data(), M.has_deriv.size());
  t->device_bytes = t->u_seq_sym.bytes() + t->u_seq_off.bytes() + t->u_pair_id.bytes() + t->u_pair_weight.bytes() +
                    t->u_partial.bytes() + t->u_scratch.bytes() + t->u_f_rec.bytes() + t->u_b_rec.bytes() + t->u_f_arc.bytes() + t->u_b_arc.bytes() +
                    t->pair_logprob.bytes() + t->pair_w.bytes();
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->n_pairs = t->corpus.n_pairs;
    stats->n_pairs_kept = M.pair_id.size();
    stats->explored_arcs = M.explored_arcs;
    stats->kept_states = M.lattice_states;
    stats->kept_arcs = M.lattice_arcs;
    stats->n_bundles = 0;  // nothing is laid out: the lattices are implicit
    stats->last_pair_explored_states = stats->last_pair_kept_states = stats->last_pair_kept_arcs = 0;  // not tracked per pair
    stats->n_windowed_pairs = 0;
    stats->max_levels = M.max_len + 1;
    stats->device_bytes = t->device_bytes;
    stats->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (t->cascade) {
    bool dense_built = false;
    int rc = dense_try_build(t, M, slot_of, n_slots, dense_built);
    if (rc) return rc;
    if (dense_built) t->device_bytes += t->d_vbuf.bytes() + t->d_zbuf.bytes() + t->d_sym.bytes() + t->d_partial.bytes();
    if (stats) stats->device_bytes = t->device_bytes;
  } else
    t->dense = false;
  // the bulk host arrays are on the device now
  std::vector<uint16_t>().swap(M.seq_sym);
  std::vector<uint64_t>().swap(M.seq_off);
  std::vector<double>().swap(M.pair_weight);
  t->unrolled = true;
  t->have_lattices = true;
  ++t->lattice_epoch;
  return CARMEL_HIP_OK;
}

// the E-step proper (weights are current in t->arc_logw): counts into counts_ptr()[0 .. n_slots), ln p per pair
int unrolled_estimate(carmel_hip_trainer* t, hipStream_t s) {
  const UnrolledModel& M = t->um;
  if (t->dense) {
    HIPCHK(launch_dense_tables(t->d_A.p, t->d_AT.p, t->d_B.p, t->d_SP, M.V, t->d_a_off.p, t->d_a_par.p, t->d_b_off.p, t->d_b_par.p,
                               t->d_a_has.p, t->d_b_has.p, t->param_logw_c.p, s));
    DenseArgs D;
    D.S = M.S;
    D.SP = t->d_SP;
    D.V = M.V;
    D.start = M.start;
    D.fin = M.fin;
    D.n_slots = t->u_n_slots;
    D.n_eps = (uint32_t)t->u_e_arc.n;
    D.debug = 0u;
    D.e_src = t->u_e_src.p;
    D.e_dst = t->u_e_dst.p;
    D.We = t->u_We.p;
    HIPCHK(launch_unrolled_weights(t->u_e_arc.p, t->arc_logw.p, t->u_We.p, 1, (uint32_t)t->u_e_arc.n, s));
    D.A = t->d_A.p;
    D.AT = t->d_AT.p;
    D.B = t->d_B.p;
    D.Bslot = t->d_Bslot.p;
    D.sym = t->d_sym.p;
    D.sym_off = t->d_sym_off.p;
    D.len = t->d_len.p;
    D.pair = t->d_pair.p;
    D.weight = t->d_weight.p;
    D.pair_logprob = t->pair_logprob.p;
    D.vbuf = t->d_vbuf.p;
    D.vbuf_off = t->d_vbuf_off.p;
    D.zbuf = t->d_zbuf.p;
    D.afbuf = t->d_afbuf.p;
    D.partial = t->d_partial.p;
    HIPCHK(launch_dense_sweep(D, t->d_groups, s));
    HIPCHK(launch_unrolled_reduce(t->d_partial.p, t->d_groups, t->u_n_slots, t->counts_ptr(), s));
    return CARMEL_HIP_OK;
  }
  HIPCHK(launch_unrolled_weights(t->u_f_arc.p, t->arc_logw.p, (double*)t->u_f_rec.p, 2, (uint32_t)t->u_f_arc.n, s));
  HIPCHK(launch_unrolled_weights(t->u_b_arc.p, t->arc_logw.p, (double*)t->u_b_rec.p, 2, (uint32_t)t->u_b_arc.n, s));
  HIPCHK(launch_unrolled_weights(t->u_e_arc.p, t->arc_logw.p, t->u_We.p, 1, (uint32_t)t->u_e_arc.n, s));
  UnrolledArgs A;
  A.S = M.S;
  A.V = M.V;
  A.start = M.start;
  A.fin = M.fin;
  A.n_eps = (uint32_t)t->u_e_arc.n;
  A.n_slots = t->u_n_slots;
  A.max_len = M.max_len;
  A.f_deg_u = M.f_deg_u;
  A.b_deg_u = M.b_deg_u;
  A.n_pairs = M.pair_id.size();
  A.f_off = t->u_f_off.p;
  A.f_rec = t->u_f_rec.p;
  A.b_off = t->u_b_off.p;
  A.b_rec = t->u_b_rec.p;
  A.e_src = t->u_e_src.p;
  A.e_dst = t->u_e_dst.p;
  A.We = t->u_We.p;
  A.e_slot = t->u_e_slot.p;
  A.seq_off = t->u_seq_off.p;
  A.seq_sym = t->u_seq_sym.p;
  A.pair_id = t->u_pair_id.p;
  A.pair_weight = t->u_pair_weight.p;
  A.pair_logprob = t->pair_logprob.p;
  A.partial = t->u_partial.p;
  A.alpha_scratch = t->u_scratch.p;
  A.debug_no_acc = 0u;
  HIPCHK(launch_unrolled_sweep(A, t->u_n_wg, t->counts_ptr(), s));
  return CARMEL_HIP_OK;
}
