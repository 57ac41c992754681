// engine_unrolled.cpp: the trainer's side of the unrolled sweep (unrolled.hpp): eligibility, upload, E-step.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "engine.hpp"
#include "unrolled_args.hpp"

// Tries to set the trainer up for the unrolled sweep.  Returns CARMEL_HIP_OK with t->unrolled set when the model and
// the corpus are eligible, CARMEL_HIP_OK with t->unrolled clear when they are not (the caller then builds explicit
// lattices), an error code on a HIP failure.
int unrolled_try_build(carmel_hip_trainer* t, int host_threads, uint8_t* has_derivation, carmel_hip_lattice_stats* stats) {
  t->unrolled = false;
  bool forced = false;  // CARMEL_HIP_UNROLLED: 0 = never, 1 = whenever eligible, unset = when it pays (density test below)
  if (const char* e = getenv("CARMEL_HIP_UNROLLED")) {
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
  const int wg_per_cu = getenv("CARMEL_HIP_UNROLLED_WGS_PER_CU") ? std::max(1, atoi(getenv("CARMEL_HIP_UNROLLED_WGS_PER_CU"))) : 2;
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
  A.debug_no_acc = getenv("CARMEL_HIP_UNROLLED_NOACC") ? (uint32_t)std::max(1, atoi(getenv("CARMEL_HIP_UNROLLED_NOACC"))) : 0u;  // 1: no adds, 2: forward only
  HIPCHK(launch_unrolled_sweep(A, t->u_n_wg, t->counts_ptr(), s));
  return CARMEL_HIP_OK;
}
