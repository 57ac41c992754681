// engine.cpp — implementation of the C-ABI in include/carmel_hip.h on top of kernels.hip / lattice.cpp.
// One trainer = one GPU, one HIP stream; all per-iteration state stays in HBM.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "engine.hpp"
#include "options.hpp"
#include "host/refhash.hpp"
#include "rng.hpp"
#include "unrolled_args.hpp"

int unrolled_try_build(carmel_hip_trainer* t, int host_threads, uint8_t* has_derivation, carmel_hip_lattice_stats* stats);
int gpu_build_lattices(carmel_hip_trainer* t, const BuildOptions& opt, uint8_t* has_derivation, carmel_hip_lattice_stats* stats,
                       bool& done);
int gpu_tables_for_host_layout(carmel_hip_trainer* t, const std::vector<uint32_t>& lane_arc, const std::vector<uint32_t>& wave_arc);
int carmel_hip_debug_lattice_fingerprint_impl(carmel_hip_trainer* t, uint64_t* out);
int build_run_tables(carmel_hip_trainer* t);
// exchange.cpp: the sharded count exchange of corpus-sharded EM
int exchange_weights_in(carmel_hip_trainer* t, ExchangePlan* xp, const TransArgs& T, bool need_x);
int exchange_counts_out(carmel_hip_trainer* t, ExchangePlan* xp, const TransArgs& T);
int exchange_counts_tail(carmel_hip_trainer* t, ExchangePlan* xp);
int exchange_maximize(carmel_hip_trainer* t, ExchangePlan* xp, double* max_change, int* handled);
int exchange_settle(carmel_hip_trainer* t, bool need_counts);
void exchange_drop(carmel_hip_trainer* t);
int unrolled_estimate(carmel_hip_trainer* t, hipStream_t s);

static thread_local std::string g_err;
namespace carmel_hip {
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace carmel_hip

// [0, n) cut over a few host threads (the record arrays of a large corpus are tens of millions of words)
template <class F>
static void host_parallel_for(size_t n, F f) {
  const size_t nt = n < (1u << 20) ? 1 : std::min<size_t>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
  if (nt <= 1) {
    f(0, n);
    return;
  }
  std::vector<std::thread> th;
  for (size_t k = 0; k < nt; ++k) th.emplace_back([&, k]() { f(n * k / nt, n * (k + 1) / nt); });
  for (auto& x : th) x.join();
}

extern "C" {

const char* carmel_hip_last_error(void) { return g_err.c_str(); }

int carmel_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int carmel_hip_create(carmel_hip_trainer** out, int device, uint32_t n_states, uint32_t final_state, uint64_t n_arcs,
                      const uint32_t* src, const uint32_t* dst, const uint32_t* in_sym, const uint32_t* out_sym,
                      const double* logw, const uint32_t* group) {
  if (!out || !src || !dst || !in_sym || !out_sym || !logw) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (final_state >= n_states) return fail(CARMEL_HIP_ERR_ARG, "final state out of range");
  if (n_arcs >= 0xfffffff0ull) return fail(CARMEL_HIP_ERR_ARG, "arc ids are 32-bit (derivations.h GraphArc data)");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(CARMEL_HIP_ERR_HIP, "no HIP device: the EM hot path has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(CARMEL_HIP_ERR_ARG, "bad device index");
  HIPCHK(hipSetDevice(device));
  carmel_hip_trainer* t = new carmel_hip_trainer();
  t->device = device;
  hipError_t e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete t;
    return fail(CARMEL_HIP_ERR_HIP, hipGetErrorString(e));
  }
  (void)hipEventCreate(&t->ev0);
  (void)hipEventCreate(&t->ev1);
  (void)hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking);
  (void)hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&t->ev_w, hipEventDisableTiming);
  if (hipHostMalloc((void**)&t->h_box, 2 * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) t->h_box = nullptr;
  if (t->h_box) t->h_box[0] = t->h_box[1] = 0;
  (void)hipStreamCreateWithFlags(&t->bstream, hipStreamNonBlocking);
  (void)hipEventCreateWithFlags(&t->ev_b0, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&t->ev_b1, hipEventDisableTiming);
  for (int k = 0; k < carmel_hip_trainer::N_CHUNK_STREAMS; ++k) {
    (void)hipStreamCreateWithFlags(&t->cstream[k], hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&t->cev[k], hipEventDisableTiming);
  }
  HostWfst& w = t->w;
  w.n_states = n_states;
  w.final_state = final_state;
  w.n_arcs = n_arcs;
  w.src.assign(src, src + n_arcs);
  w.dst.assign(dst, dst + n_arcs);
  w.in.assign(in_sym, in_sym + n_arcs);
  w.out.assign(out_sym, out_sym + n_arcs);
  if (group)
    w.group.assign(group, group + n_arcs);
  else
    w.group.assign(n_arcs, CARMEL_HIP_NO_GROUP);
  for (uint64_t k = 0; k < n_arcs; ++k) {
    if (src[k] >= n_states || dst[k] >= n_states) {
      delete t;
      return fail(CARMEL_HIP_ERR_ARG, "arc endpoint out of range");
    }
    if (k && src[k] < src[k - 1]) {
      delete t;
      return fail(CARMEL_HIP_ERR_ARG, "arcs must be in arc-id order (state-major)");
    }
  }
  w.build_index();
  std::vector<double> lw(logw, logw + n_arcs);
  e = t->arc_logw.upload(lw, t->stream);
  if (e == hipSuccess) e = t->arc_group.upload(w.group, t->stream);
  if (e == hipSuccess) e = t->counts.alloc(n_arcs + 4);
  if (e == hipSuccess) e = t->old_logw.alloc(n_arcs);
  if (e == hipSuccess) e = t->em_logw.alloc(n_arcs);
  if (e == hipSuccess) e = t->best_logw.alloc(n_arcs);
  if (e == hipSuccess) e = t->maxchg.alloc(1);
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
  if (e != hipSuccess) {
    delete t;
    return fail(CARMEL_HIP_ERR_HIP, hipGetErrorString(e));
  }
  *out = t;
  return CARMEL_HIP_OK;
}

int carmel_hip_destroy(carmel_hip_trainer* t) {
  if (t && t->xplan) exchange_drop(t);
  if (!t) return CARMEL_HIP_OK;
  (void)hipSetDevice(t->device);
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  if (t->matrix) matrix_release(t->matrix);
  t->matrix = nullptr;
  if (t->ev0) (void)hipEventDestroy(t->ev0);
  if (t->ev1) (void)hipEventDestroy(t->ev1);
  if (t->side) (void)hipStreamSynchronize(t->side);
  if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
  if (t->ev_join) (void)hipEventDestroy(t->ev_join);
  if (t->ev_w) (void)hipEventDestroy(t->ev_w);
  if (t->h_box) (void)hipHostFree(t->h_box);
  if (t->bstream) {
    (void)hipStreamSynchronize(t->bstream);
    (void)hipStreamDestroy(t->bstream);
  }
  if (t->ev_b0) (void)hipEventDestroy(t->ev_b0);
  if (t->ev_b1) (void)hipEventDestroy(t->ev_b1);
  for (int k = 0; k < carmel_hip_trainer::N_CHUNK_STREAMS; ++k) {
    if (t->cstream[k]) {
      (void)hipStreamSynchronize(t->cstream[k]);
      (void)hipStreamDestroy(t->cstream[k]);
    }
    if (t->cev[k]) (void)hipEventDestroy(t->cev[k]);
    if (k == 0) {
      for (hipEvent_t e : t->ev_piece)
        if (e) (void)hipEventDestroy(e);
      t->ev_piece.clear();
    }
  }
  hipStream_t s = t->stream, s2 = t->side;
  delete t;
  if (s) (void)hipStreamDestroy(s);
  if (s2) (void)hipStreamDestroy(s2);
  return CARMEL_HIP_OK;
}

int carmel_hip_set_corpus(carmel_hip_trainer* t, uint64_t n_pairs, const uint64_t* in_off, const uint32_t* in_sym,
                          const uint64_t* out_off, const uint32_t* out_sym, const double* pair_weight) {
  if (!t || !in_off || !out_off) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HostCorpus& c = t->corpus;
  c.n_pairs = n_pairs;
  c.in_off.assign(in_off, in_off + n_pairs + 1);
  c.out_off.assign(out_off, out_off + n_pairs + 1);
  c.in_sym.assign(in_sym, in_sym + in_off[n_pairs]);
  c.out_sym.assign(out_sym, out_sym + out_off[n_pairs]);
  if (pair_weight)
    c.weight.assign(pair_weight, pair_weight + n_pairs);
  else
    c.weight.assign(n_pairs, 1.0);
  t->have_corpus = true;
  t->have_lattices = false;
  return CARMEL_HIP_OK;
}

int carmel_hip_build_lattices(carmel_hip_trainer* t, int prune, int host_threads, uint8_t* has_derivation,
                              carmel_hip_lattice_stats* stats) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (!t->have_corpus) return fail(CARMEL_HIP_ERR_STATE, "set_corpus first");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) exchange_drop(t);  // the plan follows the lattices' transposition tables: plan again after a rebuild
  if (t->scalars_pending) {  // (the side stream may still be reading the arrays about to be rebuilt)
    (void)hipStreamSynchronize(t->side);
    t->scalars_pending = false;
  }
  t->build_prune = prune;
  t->build_threads = host_threads;
  if (t->matrix) {  // its tables follow the corpus: carmel_hip_set_matrix_fb again after a rebuild
    HIPCHK(hipStreamSynchronize(t->stream));
    matrix_release(t->matrix);
    t->matrix = nullptr;
  }
  {
    // one-tape models never store their lattices (unrolled.hpp)
    int rc = unrolled_try_build(t, host_threads, has_derivation, stats);
    if (rc) return rc;
    if (t->unrolled) return CARMEL_HIP_OK;
  }
  auto t0 = std::chrono::steady_clock::now();
  BuildOptions opt;
  opt.prune = prune != 0;
  opt.threads = host_threads;
  if (const char* e = lib_opt("lane_states")) opt.lane_states = (uint32_t)atoi(e);  // tuning / A-B runs
  if (const char* e = lib_opt("tile_sweep")) opt.tile_sweep = atoi(e) != 0;  // A/B: 0 = the five-kernel E-step's layout
  if (const char* e = lib_opt("lane_fused")) opt.lane_fused = atoi(e) != 0;  // A/B: 0 = the 16384-position tiles for lane corpora
  if (lib_opt("transpose") && atoi(lib_opt("transpose")) == 0) opt.lane_fused = false;  // (the gather formulation has no tiles)
  if (const char* e = lib_opt("lane_chunks")) opt.lane_chunks = (uint32_t)std::max(1, atoi(e));  // experiment: pieces per lane class
  if (const char* e = lib_opt("lane_window")) opt.lane_window = (uint32_t)std::max(0, atoi(e));          // 0: no windowed groups
  if (const char* e = lib_opt("lane_window_min")) opt.lane_window_min = (uint32_t)std::max(0, atoi(e));  // tests: window small lattices too
  if (const char* e = lib_opt("wave_ring")) opt.wave_ring = atoi(e) != 0;  // A/B: 0 = every value in LDS
  if (const char* e = lib_opt("wave_min_width")) opt.wave_min_width = opt.wave_lane_min_width = atof(e);  // tests: narrow lattices too
  {
    // lattice construction on the GPU (lattice_gpu.hip) when every lattice of the corpus is a one-per-lane case;
    // otherwise -- or with CARMEL_HIP_GPU_BUILD=0 -- the host builder below does the whole corpus
    const bool want_gpu = !(lib_opt("gpu_build") && atoi(lib_opt("gpu_build")) == 0) &&
                          !(lib_opt("transpose") && atoi(lib_opt("transpose")) == 0);
    // a probe: a few pairs spread over the corpus built on the host first (milliseconds).  If one of them is already not
    // a case for the GPU builder -- a cycle, more states than a lane takes, more arcs than its record buffers (256) -- the
    // corpus goes to the host builder without the device attempt (0.1 s on the tagging cascade x400 before it gives up)
    bool probe_ok = true;
    if (want_gpu && t->corpus.n_pairs > 4096) {
      const uint64_t np = t->corpus.n_pairs, n_probe = 64;
      for (uint64_t k = 0; k < n_probe && probe_ok; ++k) {
        const uint64_t p = k * (np / n_probe);
        PairLattice pl;
        bool hd = false;
        build_pair_lattice(t->w, t->corpus.in_sym.data() + t->corpus.in_off[p], (uint32_t)(t->corpus.in_off[p + 1] - t->corpus.in_off[p]),
                           t->corpus.out_sym.data() + t->corpus.out_off[p], (uint32_t)(t->corpus.out_off[p + 1] - t->corpus.out_off[p]),
                           opt.prune, pl, hd);
        // beyond the small capacities (256 explored states, 256 lattice arcs): the builder's large ones; beyond those, a cycle:
        // the host
        if (pl.explored_states > 200 || (hd && (pl.n_states > opt.lane_states || pl.edges.size() > 200))) opt.gpu_large_caps = true;
        if (pl.explored_states > 1024 || (hd && (pl.cyclic || pl.n_states > 1023 || pl.edges.size() > 1536))) probe_ok = false;
      }
    }
    if (want_gpu && probe_ok) {
      bool done = false;
      int rc = gpu_build_lattices(t, opt, has_derivation, stats, done);
      if (rc) return rc;
      if (!done && !opt.gpu_large_caps) {  // (small corpora are not probed: a pair beyond the small capacities may fit the large ones)
        opt.gpu_large_caps = true;
        rc = gpu_build_lattices(t, opt, has_derivation, stats, done);
        if (rc) return rc;
        if (!done) opt.gpu_large_caps = false;
      }
      if (done) {
        if (lib_opt("timing")) fprintf(stderr, "timing: lattices built on the GPU\n");
        rc = build_run_tables(t);
        if (stats) stats->device_bytes = t->device_bytes;  // (with the run tables / the tiles' arc ids)
        return rc;
      }
    }
  }
  std::string err;
  LatticeSet& L = t->lat;
  L = LatticeSet();
  // the host builder lays the lattices out (one-per-wavefront lattices, bundles: the cases the device builder leaves to it);
  // sorting the posterior slots by arc and the transposition tables -- two thirds of its time on the `long` workload, all of it
  // counting sorts -- are the device's (lattice_gpu.hip gpu_tables_for_host_layout; CARMEL_HIP_DEVICE_TABLES=0: the host's, A/B)
  // one-per-wavefront lattices over a WFST whose weights the chip's caches hold (128 MB of them: the last-level cache is 256 MB)
  // gather them from the table (sweep_wave_kernel<.., GW>): no pass writes them out in lattice order -- a fifth of the E-step on
  // the `long` workload.  CARMEL_HIP_WAVE_GATHER=0/1: never / whatever the table's size (A/B; the same sums in the same order)
  {
    const char* e = lib_opt("wave_gather");
    const bool want_t = !(lib_opt("transpose") && atoi(lib_opt("transpose")) == 0);
    // (128 MB since round 6, as for the tile passes: a mixed corpus' long lattices lie on a part of a larger table -- `mix`: 26 MB
    // of 78 --, and the last-level cache is 256 MB)
    opt.wave_gather = want_t && (e ? atoi(e) != 0 : t->w.n_arcs * sizeof(double) <= (128ull << 20));
  }
  opt.device_tables = !(lib_opt("transpose") && atoi(lib_opt("transpose")) == 0) &&
                      !(lib_opt("device_tables") && atoi(lib_opt("device_tables")) == 0);
  if (!build_lattices(t->w, t->corpus, opt, L, err)) return fail(CARMEL_HIP_ERR_ARG, err);
  const bool have_tables = !L.t_buckets.empty() || L.tables_deferred;
  if (has_derivation) std::memcpy(has_derivation, L.has_deriv.data(), L.has_deriv.size());
  hipStream_t s = t->stream;
  HIPCHK(t->bundles.upload(L.bundles, s));
  HIPCHK(t->in_arcs.upload(L.in_arcs, s));
  HIPCHK(t->out_arcs.upload(L.out_arcs, s));
  HIPCHK(t->in_off.upload(L.in_off, s));
  HIPCHK(t->out_off.upload(L.out_off, s));
  HIPCHK(t->level_off.upload(L.level_off, s));
  HIPCHK(t->pair_start.upload(L.pair_start, s));
  HIPCHK(t->pair_final.upload(L.pair_final, s));
  HIPCHK(t->pair_id.upload(L.pair_id, s));
  HIPCHK(t->pair_logw.upload(L.pair_logw, s));
  HIPCHK(t->lane_groups.upload(L.lane_groups, s));
  if (L.tile_sweep)
    HIPCHK(t->tile_group.upload(L.tile_group, s));
  else
    t->tile_group.release();
  HIPCHK(t->wave_descs.upload(L.waves, s));
  HIPCHK(t->wave_fwd.upload(L.wave_fwd, s));
  HIPCHK(t->wave_bwd.upload(L.wave_bwd, s));
  if (L.wave_gather && !L.waves.empty())
    HIPCHK(t->wave_bwd_arc.upload(L.wave_bwd_arc, s));
  else
    t->wave_bwd_arc.release();
  HIPCHK(t->wave_level_off.upload(L.wave_level_off, s));
  HIPCHK(t->wave_frow.upload(L.wave_frow, s));
  HIPCHK(t->wave_brow.upload(L.wave_brow, s));
  t->wave_slot_base = L.wave_slot_base;
  t->wave_records = L.wave_bwd.size();
  HIPCHK(t->wave_spill.alloc(L.wave_spill_states));
  {
    // the transposition path never looks at a forward record's arc id: it gets the flags words alone (half the bytes)
    const bool want_t = !(lib_opt("transpose") && atoi(lib_opt("transpose")) == 0);
    if (want_t && have_tables) {
      std::vector<uint32_t> fx(L.lane_fwd.size());
      host_parallel_for(fx.size(), [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) fx[k] = L.lane_fwd[k].x;
      });
      HIPCHK(t->lane_fwdx.upload(fx, s));
      HIPCHK(hipStreamSynchronize(s));
      t->lane_fwd.release();
    } else {
      HIPCHK(t->lane_fwd.upload(L.lane_fwd, s));
      t->lane_fwdx.release();
    }
  }
  {
    // the kernel needs only the destination/flags word of a backward record; the arc id (slot construction) stays here
    std::vector<uint32_t> bx(L.lane_bwd.size());
    host_parallel_for(bx.size(), [&](size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; ++k) bx[k] = L.lane_bwd[k].x;
    });
    HIPCHK(t->lane_bwd.upload(bx, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(t->lane_pair.upload(L.lane_pair, s));
  HIPCHK(t->lane_nstates.upload(L.lane_nstates, s));
  HIPCHK(t->lane_logw.upload(L.lane_logw, s));
  if (L.tile_sweep && t->lane_fwdx.n) {
    HIPCHK(t->lane_rec2.alloc(t->lane_bwd.n));
    HIPCHK(t->lane_chain.alloc(t->lane_groups.n));
    HIPCHK(t->tile_chain.alloc(t->tile_group.n - 1));
    HIPCHK(hipMemsetAsync(t->lane_rec2.p, 0, t->lane_rec2.bytes(), s));
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
  t->lane_records = L.wave_slot_base + L.wave_bwd.size();  // first bundle slot: [lane records | wave records | bundle arcs]
  // (the fused-lane layout's sweep hands its posteriors to the count pass itself: `post` is allocated on first use by the
  // A/B switch's three kernels, ensure_post)
  if (L.lane_fused)
    t->post.release();
  else
    HIPCHK(t->post.alloc(L.n_post));
  // (gathered weights: the wave records have no place in wcache, and the weight passes stop at the lane records)
  HIPCHK(t->wcache.alloc(t->wave_bwd_arc.n ? L.wave_slot_base : t->lane_records));
  if (!L.waves.empty() && !have_tables)
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "lattice set too large for the blocked transposition (2^32 items) with one-per-wavefront lattices");
  if (L.lane_spill_rows)
    HIPCHK(t->lane_spill.alloc(L.lane_spill_rows * 64));
  else
    t->lane_spill.release();
  if (L.tables_deferred) {
    // the arc of every lane record (the kernels' copy keeps the flags word only), then the device does the rest
    std::vector<uint32_t> la(L.lane_bwd.size());
    host_parallel_for(la.size(), [&](size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; ++k) la[k] = (L.lane_bwd[k].x & LANE_VALID) ? L.lane_bwd[k].y : 0xffffffffu;
    });
    t->hot_chunks.release();  // (the gather formulation is not offered on top of device-built tables)
    int rc = gpu_tables_for_host_layout(t, la, L.wave_bwd_arc);
    if (rc) return rc;
    t->use_transpose = true;
  } else {
    HIPCHK(t->arc_off.upload(L.arc_off, s));
    HIPCHK(t->slot_pos.upload(L.slot_pos, s));
    HIPCHK(t->hot_chunks.upload(L.hot_chunks, s));
    {
      const bool want = !(lib_opt("transpose") && atoi(lib_opt("transpose")) == 0);
      t->use_transpose = want && !L.t_buckets.empty();
      if (t->use_transpose) {
        HIPCHK(t->t_buckets.upload(L.t_buckets, s));
        HIPCHK(t->t_tile_base.upload(L.t_tile_base, s));
        HIPCHK(t->t_b_arc.upload(L.t_b_arc, s));
        HIPCHK(t->t_b_rank.upload(L.t_b_rank, s));
        HIPCHK(t->t_b_src.upload(L.t_b_src, s));
        HIPCHK(t->t_t_pos.upload(L.t_t_pos, s));
        HIPCHK(t->t_t_src.upload(L.t_t_src, s));
        HIPCHK(t->t_a_off.upload(L.t_a_off, s));
        HIPCHK(t->t_split_arcs.upload(L.t_split_arcs, s));
        HIPCHK(t->t_x.alloc(L.slot_pos.size()));
        HIPCHK(t->t_xc.alloc(L.slot_pos.size()));
        HIPCHK(hipStreamSynchronize(s));
      }
      std::vector<uint16_t>().swap(L.t_b_arc);
      std::vector<uint16_t>().swap(L.t_a_off);
      std::vector<uint16_t>().swap(L.t_b_rank);
      std::vector<uint16_t>().swap(L.t_t_pos);
      std::vector<uint32_t>().swap(L.t_b_src);
      std::vector<uint32_t>().swap(L.t_t_src);
    }
  }
  // (their sweep takes its weights from the transposition's weight pass or from the table: the gather formulation of the A/B
  // switch CARMEL_HIP_TRANSPOSE=0 has neither)
  if (!L.waves.empty() && !t->use_transpose)
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "one-per-wavefront lattices need the blocked transposition (CARMEL_HIP_TRANSPOSE=0 is for lane corpora)");
  HIPCHK(t->pair_logprob.alloc(t->corpus.n_pairs));
  {
    std::vector<double> pw(t->corpus.n_pairs);
    for (uint64_t p = 0; p < t->corpus.n_pairs; ++p)
      pw[p] = L.has_deriv[p] ? (t->corpus.weight.empty() ? 1.0 : t->corpus.weight[p]) : -1.0;
    HIPCHK(t->pair_w.upload(pw, s));
    HIPCHK(t->scalar_partial.alloc(3 * 256));
    HIPCHK(hipStreamSynchronize(s));
  }
  bool need_val = false, need_val2 = false;
  for (auto& lc : L.classes) {
    if (lc.serial || lc.max_states == 0) need_val = true;
    if (lc.serial) need_val2 = true;
  }
  HIPCHK(t->alpha_g.alloc(need_val ? L.in_off.size() : 0));
  HIPCHK(t->beta_g.alloc(need_val2 ? L.in_off.size() : 0));
  HIPCHK(launch_fill(t->pair_logprob.p, -std::numeric_limits<double>::infinity(), t->corpus.n_pairs, s));
  HIPCHK(hipStreamSynchronize(s));
  t->device_bytes = t->bundles.bytes() + t->in_arcs.bytes() + t->out_arcs.bytes() + t->in_off.bytes() +
                    t->out_off.bytes() + t->level_off.bytes() + t->pair_start.bytes() + t->pair_final.bytes() +
                    t->pair_id.bytes() + t->pair_logw.bytes() + t->pair_logprob.bytes() + t->alpha_g.bytes() +
                    t->beta_g.bytes() + t->lane_groups.bytes() + t->lane_fwd.bytes() + t->lane_fwdx.bytes() + t->lane_bwd.bytes() +
                    t->lane_pair.bytes() + t->lane_nstates.bytes() + t->lane_logw.bytes() + t->post.bytes() + t->wcache.bytes() +
                    t->arc_off.bytes() + t->slot_pos.bytes() + t->hot_chunks.bytes() + t->t_b_arc.bytes() + t->t_b_rank.bytes() +
                    t->t_t_pos.bytes() + t->t_b_src.bytes() + t->t_t_src.bytes() + t->t_x.bytes() + t->t_xc.bytes() +
                    t->wave_descs.bytes() + t->wave_fwd.bytes() + t->wave_bwd.bytes() + t->wave_bwd_arc.bytes() + t->wave_level_off.bytes() +
                    t->wave_frow.bytes() + t->wave_brow.bytes() + t->wave_spill.bytes();
  std::vector<uint64_t>().swap(L.arc_off);
  std::vector<uint64_t>().swap(L.slot_pos);
  std::vector<uint2_t>().swap(L.lane_fwd);
  std::vector<uint2_t>().swap(L.lane_bwd);
  std::vector<uint2_t>().swap(L.wave_fwd);
  std::vector<uint32_t>().swap(L.wave_bwd);
  std::vector<uint32_t>().swap(L.wave_bwd_arc);
  // free the bulk host arrays; keep descriptors + classes
  std::vector<uint2_t>().swap(L.in_arcs);
  std::vector<uint2_t>().swap(L.out_arcs);
  std::vector<uint32_t>().swap(L.in_off);
  std::vector<uint32_t>().swap(L.out_off);
  std::vector<uint32_t>().swap(L.level_off);
  if (lib_opt("timing")) {
    fprintf(stderr, "timing: layout lane_arcs=%llu lane_groups=%zu lane_pieces=%zu bundles=%zu bundle_classes=%zu total_arcs=%llu aligned=%d\n",
            (unsigned long long)L.lane_arcs, L.lane_groups.size(), L.lane_classes.size(), L.bundles.size(), L.classes.size(),
            (unsigned long long)L.total_arcs, (int)L.lane_tiles_aligned);
    for (auto& lc : L.lane_classes) fprintf(stderr, "timing:   lane piece groups=%u %s=%u tiles=%u\n", lc.count, lc.windowed ? "window" : "max_states", lc.max_states, lc.tile_count);
    for (auto& wc : L.wave_classes) fprintf(stderr, "timing:   wave class count=%u max_states=%u max_width=%u ring=%u\n", wc.count, wc.max_states, wc.max_width, wc.ring);
    for (auto& lc : L.classes) fprintf(stderr, "timing:   bundle class count=%u block=%u max_states=%u serial=%d\n", lc.count, lc.block, lc.max_states, (int)lc.serial);
  }
  t->have_lattices = true;
  ++t->lattice_epoch;
  {
    int rc = build_run_tables(t);
    if (rc) return rc;
  }
  if (stats) {
    stats->n_pairs = t->corpus.n_pairs;
    stats->n_pairs_kept = L.n_kept;
    stats->explored_states = L.explored_states;
    stats->explored_arcs = L.explored_arcs;
    stats->kept_states = L.total_states;
    stats->kept_arcs = L.total_arcs;
    stats->n_cyclic_pairs = L.n_cyclic;
    stats->n_bundles = L.bundles.size() + L.lane_groups.size() + L.waves.size();
    stats->max_levels = L.max_levels;
    stats->device_bytes = t->device_bytes;
    stats->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->last_pair_explored_states = L.last_pre_states;
    stats->last_pair_kept_states = L.last_post_states;
    stats->last_pair_kept_arcs = L.last_post_arcs;
    stats->n_windowed_pairs = 0;
    for (auto& g : L.lane_groups)
      if (g.window) stats->n_windowed_pairs += g.n_lanes;
  }
  return CARMEL_HIP_OK;
}

// tied parameters: dense index per (cascade member, tie id) -- WFST::normalize keeps its tie tables per transducer
static int build_ties(carmel_hip_trainer* t, uint64_t n, const uint32_t* member, const uint32_t* group) {
  std::vector<uint32_t> tie_of(n, 0xffffffffu);
  std::unordered_map<uint64_t, uint32_t> ids;
  t->any_locked = false;
  t->h_tie_member.clear();
  for (uint64_t k = 0; k < n; ++k) {
    if (group[k] == CARMEL_HIP_LOCKED_GROUP) t->any_locked = true;
    if (group[k] == CARMEL_HIP_NO_GROUP || group[k] == CARMEL_HIP_LOCKED_GROUP) continue;
    const uint64_t key = ((uint64_t)(member ? member[k] : 0) << 32) | group[k];
    auto it = ids.find(key);
    if (it == ids.end()) {
      it = ids.emplace(key, (uint32_t)ids.size()).first;
      t->h_tie_member.push_back(member ? member[k] : 0);
    }
    tie_of[k] = it->second;
  }
  t->n_ties = ids.size();
  if (t->n_ties) {
    HIPCHK(t->tie_of.upload(tie_of, t->stream));
    HIPCHK(t->tie_tab.alloc(4 * t->n_ties));
    HIPCHK(hipStreamSynchronize(t->stream));
  } else {
    t->tie_of.release();
    t->tie_tab.release();
  }
  return CARMEL_HIP_OK;
}

// norm groups over a parameter table given (member, src state, input symbol, method)
static int build_norm_groups(carmel_hip_trainer* t, uint64_t n, const uint32_t* member, const uint32_t* src,
                             const uint32_t* in, const std::vector<int>& method, const std::vector<double>& addc,
                             const uint32_t* group) {
  std::vector<uint32_t> norm_of(n);
  std::vector<double> add;
  std::unordered_map<uint64_t, uint32_t> ids;
  ids.reserve(n);
  t->h_group_member.clear();
  t->h_group_src.clear();
  t->h_group_joint.clear();
  t->any_digamma = false;
  t->dig_alpha.release();
  t->tie_alpha.release();
  // key: member | state | (input symbol or ~0 for JOINT).  States and symbols are 32-bit, members few: two maps
  // deep would be simpler but slower; mix into 64 bits + verify by construction (member < 2^8, state < 2^32,
  // symbol < 2^24 are checked).
  for (uint64_t k = 0; k < n; ++k) {
    uint32_t m = member ? member[k] : 0;
    int g = method[m];
    if (g == CARMEL_HIP_NORM_NONE) {
      norm_of[k] = 0xffffffffu;
      continue;
    }
    uint64_t sym = (g == CARMEL_HIP_NORM_JOINT) ? 0xffffffull : (uint64_t)in[k];
    if (m >= 256 || sym > 0xffffffull) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "norm-group key out of range");
    uint64_t key = ((uint64_t)m << 56) | (sym << 32) | src[k];
    auto it = ids.find(key);
    if (it == ids.end()) {
      uint32_t id = (uint32_t)add.size();
      ids.emplace(key, id);
      add.push_back(addc[m]);
      t->h_group_member.push_back(m);
      t->h_group_src.push_back(src[k]);
      t->h_group_joint.push_back(g == CARMEL_HIP_NORM_JOINT ? 1 : 0);
      norm_of[k] = id;
    } else
      norm_of[k] = it->second;
  }
  {
    // The reference numbers a CONDITIONAL member's groups state after state, and within a state in the order its walk over the
    // state's symbol index visits the input symbols (fst.h:1362-1446; host/refhash.hpp replays the index: a table made for the
    // state's arc count, one insert per arc in list order).  The ids above are first-seen ids -- what the M-step kernels use --;
    // whoever has to talk about groups in the reference's numbering (the sampler's prior-scale groups) gets the rank here.
    t->h_group_ref_rank.assign(add.size(), 0u);
    std::vector<uint32_t> next_rank;  // per member
    uint64_t k = 0;
    std::vector<uint32_t> syms;
    while (k < n) {
      const uint32_t m = member ? member[k] : 0;
      uint64_t e = k;
      while (e < n && (member ? member[e] : 0) == m && src[e] == src[k]) ++e;  // the arcs of one state (the tables are state-major)
      if (method[m] == CARMEL_HIP_NORM_CONDITIONAL) {
        if (next_rank.size() <= m) next_rank.resize((size_t)m + 1, 0u);
        syms.assign(in + k, in + e);
        for (uint32_t sym : carmel_host::conditional_group_order(syms)) {
          const uint64_t key = ((uint64_t)m << 56) | ((uint64_t)sym << 32) | src[k];
          t->h_group_ref_rank[ids[key]] = next_rank[m]++;
        }
      }
      k = e;
    }
  }
  t->n_norm_groups = add.size();
  t->h_norm_of = norm_of;
  t->all_grouped = true;
  for (uint64_t k = 0; k < n; ++k)
    if (norm_of[k] == 0xffffffffu) t->all_grouped = false;
  {
    // how far apart do the members of one group lie?  (state-major arc tables: less than an out-degree)
    std::vector<uint64_t> first(add.size(), ~0ull);
    uint64_t span = 0;
    for (uint64_t k = 0; k < n; ++k) {
      const uint32_t g = norm_of[k];
      if (g == 0xffffffffu) continue;
      if (first[g] == ~0ull) first[g] = k;
      span = std::max(span, k - first[g]);
    }
    t->norm_span = (span >= 1 && span <= 64) ? (uint32_t)span : 0u;
    if (span == 0 && !add.empty()) t->norm_span = 1;  // all groups are singletons
  }
  t->h_group_add = add;
  {
    // members of every group, contiguous (counting sort by group id); big groups listed separately
    std::vector<uint64_t> off(add.size() + 1, 0), perm, big;
    for (uint64_t k = 0; k < n; ++k)
      if (norm_of[k] != 0xffffffffu) off[norm_of[k] + 1]++;
    for (size_t g = 0; g < add.size(); ++g) off[g + 1] += off[g];
    perm.resize(off[add.size()]);
    std::vector<uint64_t> cur(off.begin(), off.end() - 1);
    for (uint64_t k = 0; k < n; ++k)
      if (norm_of[k] != 0xffffffffu) perm[cur[norm_of[k]]++] = k;
    for (size_t g = 0; g < add.size(); ++g)
      if (off[g + 1] - off[g] > MSTEP_BIG_GROUP) big.push_back(g);
    HIPCHK(t->group_off.upload(off, t->stream));
    HIPCHK(t->norm_perm.upload(perm, t->stream));
    t->h_group_off = off;
    t->h_norm_perm = perm;
    HIPCHK(t->big_groups.upload(big, t->stream));
  }
  HIPCHK(t->norm_of.upload(norm_of, t->stream));
  {
    std::vector<uint16_t> code(t->norm_span ? n : 0);
    for (uint64_t k = 0; k < code.size(); ++k)
      code[k] = norm_of[k] == 0xffffffffu ? (uint16_t)0xffffu
                                          : (uint16_t)((norm_of[k] & 0x3fffu) | (group[k] == CARMEL_HIP_LOCKED_GROUP ? 0x4000u : 0u));
    HIPCHK(t->norm_code16.upload(code, t->stream));
    // members of every parameter's group as offsets -15 .. +16 (bit offset + 15), unlocked and locked apart
    std::vector<uint32_t> mask, lmask;
    if (t->norm_span && t->norm_span <= 15) {
      mask.assign(n, 0u);
      lmask.assign(n, 0u);
      const std::vector<uint64_t>& off = t->h_group_off;
      const std::vector<uint64_t>& perm = t->h_norm_perm;
      for (size_t g = 0; g + 1 < off.size(); ++g)
        for (uint64_t a = off[g]; a < off[g + 1]; ++a)
          for (uint64_t b = off[g]; b < off[g + 1]; ++b) {
            const int64_t d = (int64_t)perm[b] - (int64_t)perm[a];  // within +-15 by the span check
            (group[perm[b]] == CARMEL_HIP_LOCKED_GROUP ? lmask : mask)[perm[a]] |= 1u << (uint32_t)(d + 15);
          }
    }
    HIPCHK(t->norm_mask32.upload(mask, t->stream));
    HIPCHK(t->norm_lockmask32.upload(lmask, t->stream));
    // spans of 16 .. 31 (config 2: 20 arcs per state, groups up to 19 apart): the same with 64-bit masks, bit offset + 31
    std::vector<unsigned long long> mask64, lmask64;
    if (t->norm_span > 15 && t->norm_span <= 31) {
      mask64.assign(n, 0ull);
      lmask64.assign(n, 0ull);
      const std::vector<uint64_t>& off = t->h_group_off;
      const std::vector<uint64_t>& perm = t->h_norm_perm;
      for (size_t g = 0; g + 1 < off.size(); ++g)
        for (uint64_t a = off[g]; a < off[g + 1]; ++a)
          for (uint64_t b = off[g]; b < off[g + 1]; ++b) {
            const int64_t d = (int64_t)perm[b] - (int64_t)perm[a];
            (group[perm[b]] == CARMEL_HIP_LOCKED_GROUP ? lmask64 : mask64)[perm[a]] |= 1ull << (uint32_t)(d + 31);
          }
    }
    HIPCHK(t->norm_mask64.upload(mask64, t->stream));
    HIPCHK(t->norm_lockmask64.upload(lmask64, t->stream));
  }
  HIPCHK(t->add_count.upload(add, t->stream));
  HIPCHK(t->gscale.alloc(add.size()));
  t->any_add_count = false;
  for (double a : add)
    if (a != 0.0) t->any_add_count = true;
  HIPCHK(hipStreamSynchronize(t->stream));
  t->have_norm = true;
  return CARMEL_HIP_OK;
}

int carmel_hip_set_norm(carmel_hip_trainer* t, int norm_group_by, double add_count) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (t->cascade) return fail(CARMEL_HIP_ERR_STATE, "cascade: per-member methods come from set_cascade");
  HIPCHK(hipSetDevice(t->device));
  {
    int rc = build_ties(t, t->w.n_arcs, nullptr, t->w.group.data());
    if (rc) return rc;
  }
  t->norm_group_by = norm_group_by;
  t->norm_add_count = add_count;
  std::vector<int> m(1, norm_group_by);
  std::vector<double> a(1, add_count);
  return build_norm_groups(t, t->w.n_arcs, nullptr, t->w.src.data(), t->w.in.data(), m, a, t->w.group.data());
}

int carmel_hip_set_prior(carmel_hip_trainer* t, double smooth_floor, int weight_is_prior_count) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  t->smooth_floor = smooth_floor;
  uint64_t n = t->w.n_arcs;  // the prior belongs to the (composed) arc table (derivations.h:96-101)
  std::vector<double> pr(n, smooth_floor > 0 ? smooth_floor : 0.0);
  t->h_arc_prior_w.clear();
  t->arc_prior_w.release();
  if (weight_is_prior_count) {
    std::vector<double> lw(n);
    HIPCHK(hipMemcpyAsync(lw.data(), t->arc_logw.p, n * sizeof(double), hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    for (uint64_t k = 0; k < n; ++k) pr[k] += std::exp(lw[k]);
    if (t->cascade) {  // a cascade's M-step adds the scalar -f per composed arc; -U adds the arc's own initial weight
      t->h_arc_prior_w.resize(n);
      for (uint64_t k = 0; k < n; ++k) t->h_arc_prior_w[k] = std::exp(lw[k]);
      HIPCHK(t->arc_prior_w.upload(t->h_arc_prior_w, t->stream));
    }
  }
  t->prior_nonzero = false;
  for (double x : pr)
    if (x != 0.0) {
      t->prior_nonzero = true;
      break;
    }
  HIPCHK(t->prior.upload(pr, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  t->have_prior = true;
  return CARMEL_HIP_OK;
}

int carmel_hip_set_cascade(carmel_hip_trainer* t, uint64_t n_params, const double* param_logw,
                           const uint32_t* param_group, const uint32_t* param_member, const uint32_t* param_src,
                           const uint32_t* param_in, uint32_t n_members, const int* member_norm,
                           const double* member_add_count, uint64_t n_chains, const uint64_t* chain_off,
                           const uint64_t* chain_param) {
  if (!t || !param_logw || !param_group || !param_member || !param_src || !param_in || !member_norm || !chain_off)
    return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  for (uint64_t k = 0; k < t->w.n_arcs; ++k)
    if (t->w.group[k] >= n_chains) return fail(CARMEL_HIP_ERR_ARG, "composed arc refers to a chain id out of range");
  for (uint64_t j = 0; j < chain_off[n_chains]; ++j)
    if (chain_param[j] >= n_params) return fail(CARMEL_HIP_ERR_ARG, "chain refers to a parameter out of range");
  t->cascade = true;
  t->h_param_group.assign(param_group, param_group + n_params);
  t->h_chain_off.assign(chain_off, chain_off + n_chains + 1);
  t->h_chain_param.assign(chain_param, chain_param + chain_off[n_chains]);
  t->n_params = n_params;
  t->n_chains = n_chains;
  hipStream_t s = t->stream;
  HIPCHK(t->param_logw_c.upload(std::vector<double>(param_logw, param_logw + n_params), s));
  HIPCHK(t->param_group_c.upload(std::vector<uint32_t>(param_group, param_group + n_params), s));
  HIPCHK(t->param_counts_c.alloc(n_params));
  HIPCHK(t->chain_off.upload(std::vector<uint64_t>(chain_off, chain_off + n_chains + 1), s));
  HIPCHK(t->chain_param.upload(std::vector<uint64_t>(chain_param, chain_param + chain_off[n_chains]), s));
  HIPCHK(t->old_logw.alloc(std::max<uint64_t>(n_params, t->w.n_arcs)));
  HIPCHK(t->em_logw.alloc(t->w.n_arcs));    // for a cascade: composed-arc counts of the previous iteration
  HIPCHK(t->best_logw.alloc(t->w.n_arcs));  // ... and the best of those (train.cc:123-130, 449-455)
  std::vector<int> m(member_norm, member_norm + n_members);
  std::vector<double> a(n_members, 0.0);
  if (member_add_count) a.assign(member_add_count, member_add_count + n_members);
  for (uint64_t p = 0; p < n_params; ++p)
    if (param_member[p] >= n_members) return fail(CARMEL_HIP_ERR_ARG, "param_member out of range");
  int rc = build_norm_groups(t, n_params, param_member, param_src, param_in, m, a, param_group);
  if (rc) return rc;
  rc = build_ties(t, n_params, param_member, param_group);
  if (rc) return rc;
  // composed weights from the chains (cascade.update)
  HIPCHK(launch_chain_update(t->arc_logw.p, t->arc_group.p, t->chain_off.p, t->chain_param.p, t->param_logw_c.p,
                             t->w.n_arcs, s));
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}

// the argument block of one M-step pass over the parameters (shared with the sharded M-step of exchange.cpp)
int mstep_args(carmel_hip_trainer* t, int use_counts, int save_old, MstepArgs& M) {
  hipStream_t s = t->stream;
  M.block_first = 0;
  M.n_ranges = 0;
  M.logw = t->params();
  M.lw_src = nullptr;  // the one-pass kernel needs no current weight besides each thread's own
  M.code16 = t->norm_code16.p;
  M.mask32 = t->norm_mask32.n ? t->norm_mask32.p : nullptr;
  M.lockmask32 = t->norm_lockmask32.n ? t->norm_lockmask32.p : nullptr;
  M.mask64 = t->norm_mask64.n ? t->norm_mask64.p : nullptr;
  M.lockmask64 = t->norm_lockmask64.n ? t->norm_lockmask64.p : nullptr;
  if (t->norm_span && !t->any_digamma && !t->n_ties && (!use_counts || t->any_locked)) {
    // the one-pass kernel reads the weights of its halo, which its neighbours are rewriting: give it a snapshot
    if (t->mstep_snap.n != t->np()) HIPCHK(t->mstep_snap.alloc(t->np()));
    HIPCHK(hipMemcpyAsync(t->mstep_snap.p, M.logw, t->np() * sizeof(double), hipMemcpyDeviceToDevice, s));
    M.lw_src = t->mstep_snap.p;
  }
  M.old_logw = t->old_logw.p;
  M.counts = t->pcounts();
  M.prior = (!t->cascade && t->have_prior && t->prior_nonzero) ? t->prior.p : nullptr;
  M.group = t->pgroup();
  M.norm_of = t->norm_of.p;
  M.add_count = t->any_add_count ? t->add_count.p : nullptr;
  M.group_off = t->group_off.p;
  M.norm_perm = t->norm_perm.p;
  M.big_groups = t->big_groups.p;
  M.n_groups = t->n_norm_groups;
  M.n_big = t->big_groups.n;
  if (t->max_partial.n != MSTEP_PARTIALS + t->big_groups.n) {
    HIPCHK(t->max_partial.alloc(MSTEP_PARTIALS + t->big_groups.n));
    HIPCHK(hipMemsetAsync(t->max_partial.p, 0, t->max_partial.bytes(), s));  // once: mstep_max_final_kernel clears what it reads
    t->mstep_stream_work = true;  // (an M-step launched on another stream must wait for this: exchange.cpp, the direct form)
  }
  M.max_partial = t->max_partial.p;
  M.gscale = t->gscale.p;
  M.tie_of = t->n_ties ? t->tie_of.p : nullptr;
  M.tie_tab = t->tie_tab.p;
  if (t->n_ties && t->glocked.n != t->n_norm_groups) HIPCHK(t->glocked.alloc(t->n_norm_groups));
  M.glocked = t->glocked.p;
  M.n_ties = t->n_ties;
  M.all_grouped = t->all_grouped ? 1 : 0;
  M.window_span = t->any_digamma ? 0u : t->norm_span;  // the one-pass kernel knows the linear scale only
  M.dig_alpha = t->any_digamma ? t->dig_alpha.p : nullptr;
  M.tie_alpha = (t->any_digamma && t->n_ties) ? t->tie_alpha.p : nullptr;
  M.max_change_bits = t->maxchg.p;
  const bool mailbox = !(lib_opt("mailbox") && !atoi(lib_opt("mailbox")));
  M.box = mailbox ? t->h_box : nullptr;  // (every M-step's last kernel leaves its result there; carmel_hip_maximize waits for its own)
  M.box_seq = M.box ? ++t->box_seq : 0;
  M.n = t->np();
  M.save_old = save_old;
  return CARMEL_HIP_OK;
}
static int run_mstep(carmel_hip_trainer* t, int use_counts, int save_old) {
  MstepArgs M;
  int rc = mstep_args(t, use_counts, save_old, M);
  if (rc) return rc;
  HIPCHK(launch_mstep(M, use_counts, t->stream));
  return CARMEL_HIP_OK;
}

int carmel_hip_normalize(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (!t->have_norm) return fail(CARMEL_HIP_ERR_STATE, "set_norm / set_cascade first");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  int rc = run_mstep(t, 0, 2);
  if (rc) return rc;
  if (t->cascade)
    HIPCHK(launch_chain_update(t->arc_logw.p, t->arc_group.p, t->chain_off.p, t->chain_param.p, t->param_logw_c.p,
                               t->w.n_arcs, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}

int carmel_hip_set_weights(carmel_hip_trainer* t, const double* logw) {
  if (!t || !logw) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  HIPCHK(hipMemcpyAsync(t->params(), logw, t->np() * sizeof(double), hipMemcpyHostToDevice, t->stream));
  if (t->cascade)
    HIPCHK(launch_chain_update(t->arc_logw.p, t->arc_group.p, t->chain_off.p, t->chain_param.p, t->param_logw_c.p,
                               t->w.n_arcs, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}
int carmel_hip_get_weights(carmel_hip_trainer* t, double* logw) {
  if (!t || !logw) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  HIPCHK(hipMemcpyAsync(logw, t->params(), t->np() * sizeof(double), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}
int carmel_hip_get_arc_weights(carmel_hip_trainer* t, double* logw) {
  if (!t || !logw) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  HIPCHK(hipMemcpyAsync(logw, t->arc_logw.p, t->w.n_arcs * sizeof(double), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}

void trans_args(carmel_hip_trainer* t, TransArgs& T) {
  T.buckets = t->t_buckets.p;
  T.tile_base = t->t_tile_base.p;
  T.b_arc = t->t_b_arc.p;
  T.b_rank = t->t_b_rank.p;
  T.b_src = t->t_b_src.p;
  T.t_pos = t->t_t_pos.p;
  T.t_src = t->t_t_src.p;
  T.a_off = t->t_a_off.p;
  T.tr_off = t->tr_off.p;
  T.tr_rel = t->tr_rel.p;
  T.tr_src = t->tr_src.p;
  T.br_off = t->br_off.p;
  T.br_rel = t->br_rel.p;
  T.br_src = t->br_src.p;
  T.use_runs = t->use_runs ? 1u : 0u;
  {
    // which pass of a direction does the random access: the first one (a scattered write nobody waits for) where a
    // (tile, bucket) cell holds a run of items -- the corpora that get run-length indices --, the second one (a gather)
    // where cells hold an item or two and a scattered write would touch a line per item.  CARMEL_HIP_TRANS_SCATTER is the
    // A/B switch (bit-identical results).
    const char* e = lib_opt("trans_scatter");
    T.scatter = e ? (uint32_t)atoi(e) : (t->use_runs ? 3u : 0u);
    if (t->lat.lane_fused) T.scatter &= ~2u;  // the sweep writes XC in tile-major item order; the bucket pass gathers
  }
  T.x = t->t_x.p;
  T.xc = t->t_xc.p;
  T.logw = t->arc_logw.p;
  T.wcache = t->wcache.p;
  T.post = t->post.p;
  T.counts = t->counts_ptr();
  T.n_wcache = t->wcache.n;
  T.n_post = t->post.n;
  T.n_buckets = (uint32_t)t->t_buckets.n;
  T.n_tiles = t->t_tile_base.n ? (uint32_t)(t->t_tile_base.n - 1) : 0u;
  T.tile = t->lat.tile;
  T.bucket = t->lat.bucket;
  T.tile_first = T.tile_count = 0;
  T.bucket_first = 0;
  T.bucket_count = T.n_buckets;
  T.n_wtiles = 0;
  T.slack_bytes = (uint32_t)DEVBUF_SLACK;  // x, xc, t_pos, t_src are DevBufs
}

// CARMEL_HIP_POISON=2 (debugging): LDS keeps what the last workgroup on the CU left there -- zeros on an idle box, somebody's
// data after somebody's job.  Filling every CU's LDS with 0xff bytes before an E-step / M-step makes a kernel that reads LDS it
// never wrote fail on every run.
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned* sink) {
  extern __shared__ unsigned lds_words[];
  const unsigned n = 160 * 1024 / 4 - 64;
  for (unsigned i = threadIdx.x; i < n; i += 256) lds_words[i] = 0xffffffffu;
  __syncthreads();
  if (lds_words[(threadIdx.x * 977u) % n] != 0xffffffffu) *sink = 1;  // (keeps the stores)
}
static int lds_poison(carmel_hip_trainer* t) {
  const int mode = lib_opt("poison") ? atoi(lib_opt("poison")) : 0;
  if (mode < 2) return CARMEL_HIP_OK;
  static bool attr = false;
  const int bytes = 160 * 1024 - 256;
  if (!attr) {
    HIPCHK(hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr = true;
  }
  hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), bytes, t->stream, (unsigned*)t->maxchg.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}

// the trainer's stream takes delivery of the corpus scalars (a9) the side stream computed behind the last count pass
int scalars_join(carmel_hip_trainer* t) {
  if (!t->scalars_pending) return CARMEL_HIP_OK;
  if (hipEventQuery(t->ev_join) != hipSuccess) HIPCHK(hipStreamWaitEvent(t->stream, t->ev_join, 0));
  t->scalars_pending = false;
  return CARMEL_HIP_OK;
}

// one 8-byte value from the device to the host at the end of everything enqueued on s so far, and the host waiting for it:
// a one-thread kernel stores the value and then a sequence number into pinned coherent memory (release, system scope) and
// the host spins on the sequence number -- a third of the round trip of hipMemcpyAsync to pageable memory +
// hipStreamSynchronize (measured on config 2, whose iteration is mostly such round trips)
__global__ void publish_kernel(const unsigned long long* src, unsigned long long* box, unsigned long long seq) {
  __hip_atomic_store(box, *src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(box + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// enqueue only: the value at dev goes to the mailbox behind what is on s so far; fetch_u64(.., published = true) waits for it later
int publish_u64(carmel_hip_trainer* t, const unsigned long long* dev, hipStream_t s) {
  const bool off = lib_opt("mailbox") && !atoi(lib_opt("mailbox"));
  if (!t->h_box || off) return CARMEL_HIP_OK;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, s, dev, t->h_box, ++t->box_seq);
  HIPCHK(hipGetLastError());
  return CARMEL_HIP_OK;
}
// published: the last kernel on s has already stored the value under sequence number t->box_seq (mstep_max_final_kernel)
int fetch_u64(carmel_hip_trainer* t, const unsigned long long* dev, unsigned long long* out, hipStream_t s, bool published) {
  const bool off = lib_opt("mailbox") && !atoi(lib_opt("mailbox"));
  if (!t->h_box || off) {
    HIPCHK(hipMemcpyAsync(out, dev, sizeof *out, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CARMEL_HIP_OK;
  }
  const unsigned long long seq = published ? t->box_seq : ++t->box_seq;
  if (!published) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, s, dev, t->h_box, seq);
    HIPCHK(hipGetLastError());
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spin = 0;; ++spin) {
    if (__atomic_load_n(t->h_box + 1, __ATOMIC_ACQUIRE) == seq) break;
    if ((spin & 0xfff) == 0xfff) {  // something went wrong on the stream, or the value is very late: ask the runtime
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.002) {
        HIPCHK(hipStreamSynchronize(s));
        if (__atomic_load_n(t->h_box + 1, __ATOMIC_ACQUIRE) != seq) return fail(CARMEL_HIP_ERR_HIP, "the M-step's result never reached the host");
        break;
      }
    }
  }
  *out = __atomic_load_n(t->h_box, __ATOMIC_RELAXED);
  return CARMEL_HIP_OK;
}

// Enqueues one E-step on the trainer's stream(s).  timed: bracket it with ev0 / ev1 (not inside a graph capture).
static int estimate_enqueue(carmel_hip_trainer* t, bool timed) {
  hipStream_t s = t->stream;
  {
    int prc = lds_poison(t);
    if (prc) return prc;
  }
  {  // (the last E-step's scalars read pair_logprob[], which this one rewrites)
    int rc = scalars_join(t);
    if (rc) return rc;
  }
  if (t->cascade)  // cascade.update(): composed weights from the chains
    HIPCHK(launch_chain_update(t->arc_logw.p, t->arc_group.p, t->chain_off.p, t->chain_param.p, t->param_logw_c.p,
                               t->w.n_arcs, s));
  if (t->matrix) {  // carmel --matrix-fb: no lattice is swept
    if (timed) HIPCHK(hipEventRecord(t->ev0, s));
    int rc = matrix_estimate(t, t->matrix, s);
    if (rc) return rc;
    HIPCHK(launch_scalars(t->pair_logprob.p, t->pair_w.p, t->corpus.n_pairs, t->scalar_partial.p,
                          t->counts_ptr() + t->w.n_arcs, s));
    if (timed) HIPCHK(hipEventRecord(t->ev1, s));
    return CARMEL_HIP_OK;
  }
  if (t->unrolled) {
    if (timed) HIPCHK(hipEventRecord(t->ev0, s));
    int rc = unrolled_estimate(t, s);
    if (rc) return rc;
    HIPCHK(launch_scalars(t->pair_logprob.p, t->pair_w.p, t->corpus.n_pairs, t->scalar_partial.p,
                          t->counts_ptr() + t->w.n_arcs, s));
    if (timed) HIPCHK(hipEventRecord(t->ev1, s));
    return CARMEL_HIP_OK;
  }
  SweepArgs A;
  A.bundles = t->bundles.p;
  A.in_arcs = (const uint2*)t->in_arcs.p;
  A.out_arcs = (const uint2*)t->out_arcs.p;
  A.in_off = t->in_off.p;
  A.out_off = t->out_off.p;
  A.level_off = t->level_off.p;
  A.pair_start = t->pair_start.p;
  A.pair_final = t->pair_final.p;
  A.pair_id = t->pair_id.p;
  A.pair_logw = t->pair_logw.p;
  A.logw = t->arc_logw.p;
  A.post = t->post.p + t->lane_records;
  A.scalars = t->counts_ptr() + t->w.n_arcs;
  A.pair_logprob = t->pair_logprob.p;
  A.val_g = t->alpha_g.p;
  A.val2_g = t->beta_g.p;
  A.first_bundle = 0;
  LaneArgs LA;
  LA.groups = t->lane_groups.p;
  LA.fwd = (const uint2*)t->lane_fwd.p;
  LA.fwdx = t->lane_fwdx.p;
  LA.bwd = t->lane_bwd.p;
  LA.rec2 = t->lane_rec2.p;
  LA.chain = t->lane_chain.p;
  LA.tile_chain = t->tile_chain.p;
  LA.lane_pair = t->lane_pair.p;
  LA.lane_nstates = t->lane_nstates.p;
  LA.lane_logw = t->lane_logw.p;
  LA.logw = t->arc_logw.p;
  LA.post = t->post.p;
  LA.wcache = t->wcache.p;
  LA.scalars = t->counts_ptr() + t->w.n_arcs;
  LA.pair_logprob = t->pair_logprob.p;
  LA.spill = t->lane_spill.p;
  LA.first_group = 0;
  LA.lds_rows = 0;
  LA.xc_tile_base = nullptr;
  LA.xc_t_pos = nullptr;
  LA.xc = nullptr;
  LA.trace = nullptr;
  const char* trace_path = lib_opt("lane_trace");  // experiment: per-wave cycle stamps of the last E-step
  static DevBuf<unsigned long long> trace_buf;
  if (trace_path) {
    if (trace_buf.n != t->lane_groups.n * 16) {
      HIPCHK(trace_buf.alloc(t->lane_groups.n * 16));
      HIPCHK(hipMemset(trace_buf.p, 0, trace_buf.bytes()));
    }
    LA.trace = trace_buf.p;
  }
  WaveArgs WA;
  WA.descs = t->wave_descs.p;
  WA.fwd = (const uint2*)t->wave_fwd.p;
  WA.bwd = t->wave_bwd.p;
  WA.level_off = t->wave_level_off.p;
  WA.frow = t->wave_frow.p;
  WA.brow = t->wave_brow.p;
  WA.wcache = t->wcache.p + t->wave_slot_base;
  if (t->wave_bwd_arc.n) {
    WA.wcache = nullptr;
    WA.logw = t->arc_logw.p;
    WA.bwd_arc = t->wave_bwd_arc.p;
    WA.n_arcs = (uint32_t)t->w.n_arcs;
  }
  WA.post = t->post.p + t->wave_slot_base;
  WA.pair_logprob = t->pair_logprob.p;
  WA.spill = t->wave_spill.p;
  WA.first = 0;
  WA.max_states = WA.max_width = 0;
  TransArgs T;
  trans_args(t, T);
  // the wave sweeps' posteriors straight to XC (wave_xc_idx, build_run_tables): no `post`, no tile pass over the wave tiles
  const bool wave_xc = t->use_transpose && t->wave_xc_idx.n && !T.use_runs && !(T.scatter & 2u);
  if (wave_xc) {
    WA.xc = T.xc;
    WA.xc_idx = t->wave_xc_idx.p;
  }
  LA.pre_weights = t->use_transpose ? 1u : 0u;
  // fused-lane layout: the lane sweep's backward pass sends a tile's posteriors to XC itself (sweep_lane_kernel<.., XC>);
  // CARMEL_HIP_LANE_FUSED_KERNEL=0: sweep -> post -> trans_c_tile on the same layout (the same bits in XC)
  const bool lane_fused = t->use_transpose && t->lat.lane_fused &&
                          !(lib_opt("lane_fused_kernel") && atoi(lib_opt("lane_fused_kernel")) == 0);
  if (lane_fused) {
    LA.xc_tile_base = T.tile_base;
    LA.xc_t_pos = T.t_pos;
    LA.xc = T.xc;
  } else if (t->lat.lane_fused && !t->post.n && t->lat.n_post) {
    HIPCHK(t->post.alloc(t->lat.n_post));
    T.post = LA.post = t->post.p;
    T.n_post = t->post.n;
  }
  ExchangePlan* const xp = (t->xplan && exchange_is_sharded(t->xplan) && t->use_transpose) ? t->xplan : nullptr;
  if (timed) HIPCHK(hipEventRecord(t->ev0, s));
  const uint32_t lane_tiles = (uint32_t)((t->wcache.n + t->lat.tile - 1) / t->lat.tile);
  // a corpus of plain lane lattices laid out for it: weights in, sweeps and posteriors out of a tile in one kernel
  // (CARMEL_HIP_TILE_SWEEP_KERNEL=0: the three kernels on the same layout, bit-identical)
  const bool tile_kernel_off = lib_opt("tile_sweep_kernel") && atoi(lib_opt("tile_sweep_kernel")) == 0;
  const bool tile_sweep = t->use_transpose && t->lat.tile_sweep && t->tile_group.n && !tile_kernel_off &&
                          ((T.scatter & 3u) == 0u || ((T.scatter & 3u) == 3u && T.use_runs));
  bool tile_sweep_done = false;
  // the tile sweep clears, on its way in, the counts the count pass adds up with atomics (the arcs whose items lie in several
  // buckets): one launch less between the sweep and the count pass (the exchange clears its own, chunk by chunk)
  // the tiles' weights straight from the table (t_t_arc, build_run_tables): the tile kernels read `x[t_src[i]]`, whatever the two are
  const bool tile_gather = t->use_transpose && t->t_t_arc.n && !T.use_runs && !(T.scatter & 1u);
  TransArgs TW = T;
  if (tile_gather) {
    TW.x = const_cast<double*>(T.logw);
    TW.t_src = t->t_t_arc.p;
  }
  TransArgs TZ = TW;
  if (!xp) {
    TZ.zero_list = t->t_split_arcs.p;
    TZ.n_zero = (uint32_t)t->t_split_arcs.n;
  }
  const bool side_by_side = t->use_transpose && t->lat.lane_classes.size() > 1 && t->lat.lane_tiles_aligned && t->lat.wave_classes.empty();
  // the bundle sweeps need nothing from the transposition: beside the lane work, on a stream of their own
  const bool bundles_beside = side_by_side && !t->lat.classes.empty();
  if (xp) {  // the weights arrive arc range by arc range (all-gather of the sharded M-step): exchange.cpp
    int rc = exchange_weights_in(t, xp, T, t->wcache.n && !tile_gather);
    if (rc) return rc;
  } else if (t->use_transpose && t->wcache.n && !tile_gather)  // (nothing but gathering sweeps / tiles: no weight goes through X)
    HIPCHK(launch_trans_w_bucket(T, s));
  if (bundles_beside) {  // (after the bucket pass: its workgroups need a CU's LDS nearly whole)
    HIPCHK(hipEventRecord(t->ev_b0, s));
    HIPCHK(hipStreamWaitEvent(t->bstream, t->ev_b0, 0));
    for (auto& lc : t->lat.classes) HIPCHK(launch_sweep(A, lc, t->bstream));
    HIPCHK(hipEventRecord(t->ev_b1, t->bstream));
  }
  if (side_by_side && t->lat.lane_fused) {
    // Fused-lane layout: the pieces' weights go to lattice order on one stream, piece after piece, and the pieces' sweeps follow
    // on another, each behind its own piece's weights -- while piece k is swept (a stream of records, weights and posteriors:
    // bound by HBM bandwidth) the weights of piece k + 1 are gathered out of X (bound by the rate of scattered line requests):
    // two kernels that want different things of the memory system, side by side.  (Measured on c4a: every piece on a stream
    // of its own makes the pieces' tile kernels run beside EACH OTHER, then their sweeps beside each other -- nothing gained.)
    // (the sweeps stay on the trainer's stream, the weights go ahead on the side stream: two streams that are known to sit on
    // different hardware queues -- two of the chunk streams were seen to share one, which serialises them)
    HIPCHK(hipEventRecord(t->ev_w, s));
    hipStream_t ps = t->side, ss = s;
    HIPCHK(hipStreamWaitEvent(ps, t->ev_w, 0));
    const size_t np = t->lat.lane_classes.size();
    while (t->ev_piece.size() < np) {
      hipEvent_t e = nullptr;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      t->ev_piece.push_back(e);
    }
    for (size_t k = 0; k < np; ++k) {
      const auto& lc = t->lat.lane_classes[k];
      HIPCHK(launch_trans_w_tiles(TW, lc.tile_first, lc.tile_count, ps));
      HIPCHK(hipEventRecord(t->ev_piece[k], ps));
      HIPCHK(hipStreamWaitEvent(ss, t->ev_piece[k], 0));
      HIPCHK(launch_lane_sweep(LA, lc, ss, lane_fused));
      if (!lane_fused) HIPCHK(launch_trans_c_tiles(T, lc.tile_first, lc.tile_count, ss));
    }
  } else if (side_by_side) {
    // chunk k: weights of its tiles to lattice order -> its sweep -> its posteriors out to X, on stream k mod 4.  The
    // kernels of different chunks overlap: a workgroup of the tile passes holds a whole CU's LDS and alternates between
    // a load phase and a store phase, the sweep is a stream of many light waves -- side by side they fill each other's
    // idle memory cycles.
    HIPCHK(hipEventRecord(t->ev_w, s));
    const int NS = carmel_hip_trainer::N_CHUNK_STREAMS;
    bool used[carmel_hip_trainer::N_CHUNK_STREAMS] = {false, false, false, false};
    size_t k = 0;
    for (auto& lc : t->lat.lane_classes) {
      hipStream_t cs = t->cstream[k % NS];
      if (!used[k % NS]) HIPCHK(hipStreamWaitEvent(cs, t->ev_w, 0));
      used[k % NS] = true;
      HIPCHK(launch_trans_w_tiles(TW, lc.tile_first, lc.tile_count, cs));
      HIPCHK(launch_lane_sweep(LA, lc, cs, lane_fused));
      if (!lane_fused) HIPCHK(launch_trans_c_tiles(T, lc.tile_first, lc.tile_count, cs));
      ++k;
    }
    for (int q = 0; q < NS; ++q)
      if (used[q]) {
        HIPCHK(hipEventRecord(t->cev[q], t->cstream[q]));
        HIPCHK(hipStreamWaitEvent(s, t->cev[q], 0));
      }
  } else if (tile_sweep && launch_tile_sweep(TZ, LA, t->tile_group.p, 0, lane_tiles, s) == hipSuccess) {
    // (a refused launch -- the device would not take the kernel's LDS attribute -- falls through to the three kernels, which
    // work on the tile-sweep layout too: tile_sweep_done below stays false)
    tile_sweep_done = true;
  } else {
    if (tile_sweep) (void)hipGetLastError();
    if (t->use_transpose) HIPCHK(launch_trans_w_tiles(TW, 0, lane_tiles, s));
    // the one-per-wavefront lattices: every class is one launch of single-wave workgroups with its own LDS size -- side by
    // side on the chunk streams (a class alone rarely fills the chip), the lane waves beside them on the main stream
    if (t->lat.wave_classes.size() > 1 || (!t->lat.wave_classes.empty() && !t->lat.lane_classes.empty())) {
      HIPCHK(hipEventRecord(t->ev_w, s));
      const int NS = carmel_hip_trainer::N_CHUNK_STREAMS;
      bool used[carmel_hip_trainer::N_CHUNK_STREAMS] = {false, false, false, false};
      size_t k = 0;
      for (auto& wc : t->lat.wave_classes) {
        hipStream_t cs = t->cstream[k % NS];
        if (!used[k % NS]) HIPCHK(hipStreamWaitEvent(cs, t->ev_w, 0));
        used[k % NS] = true;
        HIPCHK(launch_wave_sweep(WA, wc, cs));
        ++k;
      }
      for (auto& lc : t->lat.lane_classes) HIPCHK(launch_lane_sweep(LA, lc, s, lane_fused));
      for (int q = 0; q < NS; ++q)
        if (used[q]) {
          HIPCHK(hipEventRecord(t->cev[q], t->cstream[q]));
          HIPCHK(hipStreamWaitEvent(s, t->cev[q], 0));
        }
    } else {
      for (auto& wc : t->lat.wave_classes) HIPCHK(launch_wave_sweep(WA, wc, s));
      for (auto& lc : t->lat.lane_classes) HIPCHK(launch_lane_sweep(LA, lc, s, lane_fused));
    }
  }
  if (bundles_beside)
    HIPCHK(hipStreamWaitEvent(s, t->ev_b1, 0));
  else
    for (auto& lc : t->lat.classes) HIPCHK(launch_sweep(A, lc, s));
  ReduceArgs R;
  R.arc_off = t->arc_off.p;
  R.slot_pos = t->slot_pos.p;
  R.hot_chunks = t->hot_chunks.p;
  R.post = t->post.p;
  R.counts = t->counts_ptr();
  R.n_arcs = t->w.n_arcs;
  R.n_hot_chunks = t->hot_chunks.n / 3;
  // the corpus scalars (a9) only need pair_logprob[]: on the side stream.  Under an exchange plan they are forked HERE, beside the
  // count pass -- the exchange's last group carries them and must not wait for them behind it; otherwise behind the count pass
  // (below), beside whatever comes next.
  auto fork_scalars = [&](hipEvent_t fork) -> int {
    HIPCHK(hipEventRecord(fork, s));
    HIPCHK(hipStreamWaitEvent(t->side, fork, 0));
    HIPCHK(launch_scalars(t->pair_logprob.p, t->pair_w.p, t->corpus.n_pairs, t->scalar_partial.p,
                          t->counts_ptr() + t->w.n_arcs, t->side));
    HIPCHK(hipEventRecord(t->ev_join, t->side));
    t->scalars_pending = true;
    return CARMEL_HIP_OK;
  };
  if (xp) {
    int rc = fork_scalars(t->ev_fork);
    if (rc) return rc;
  }
  if (t->use_transpose) {
    // posteriors of the tiles not yet sent out: all of them, or (side by side) the bundle positions after the lane records
    const uint32_t first = (side_by_side || tile_sweep_done || lane_fused) ? lane_tiles : 0u;
    const uint32_t end = wave_xc ? std::min(T.n_tiles, (uint32_t)(t->wave_slot_base / t->lat.tile)) : T.n_tiles;
    HIPCHK(launch_trans_c_tiles(T, first, end > first ? end - first : 0u, s));
    if (xp) {  // the counts leave arc range by arc range, each into its reduce-scatter while the next is being summed
      int rc = exchange_counts_out(t, xp, T);
      if (rc) return rc;
    } else if (tile_sweep_done)
      HIPCHK(launch_trans_c_bucket_range(T, 0, T.n_buckets, s));  // (cleared by the sweep)
    else
      HIPCHK(launch_trans_c_bucket(T, t->t_split_arcs.p, (uint32_t)t->t_split_arcs.n, s));
  } else
    HIPCHK(launch_count_reduce(R, s));
  // ... BEHIND the count pass -- beside the M-step, or whatever the caller enqueues next -- and the trainer's stream joins them when
  // somebody needs them (scalars_join).  Forked between the sweep and the count pass (rounds 2-5) the event cost the E-step a
  // bubble of 8 us, and the join another before the M-step (tools/r5_timeline.sh).  One event at the end of the count pass: the
  // E-step's closing time stamp is also what the side stream waits for.
  if (xp) {
    if (timed) HIPCHK(hipEventRecord(t->ev1, s));
  } else {
    int rc = fork_scalars(timed ? t->ev1 : t->ev_fork);
    if (rc) return rc;
  }
  // somebody reads the scalars straight off the stream: the exchange's tail, or a caller who owns the count buffer and orders
  // work of their own behind the E-step (carmel_hip_stream)
  if (xp || t->ext_counts) {
    int rc = scalars_join(t);
    if (rc) return rc;
  }
  if (xp) {
    int rc = exchange_counts_tail(t, xp);
    if (rc) return rc;
  }
  if (timed && trace_path && trace_buf.n) {
    std::vector<unsigned long long> h(trace_buf.n);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(h.data(), trace_buf.p, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "wb")) {
      fwrite(h.data(), 8, h.size(), f);
      fclose(f);
    }
  }
  return CARMEL_HIP_OK;
}

// (The E-step is a fixed chain of a dozen small launches with unchanging arguments, so it could be captured once into a
// hipGraph and replayed.  Measured on MI355X / ROCm 7.2 in rounds 1, 3 and 4: the replay is SLOWER than the eager launches --
// config 2 0.155 vs 0.132 ms per iteration, config 4 0.763 vs 0.743 -- so the capture path is gone.)
int carmel_hip_estimate_async(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (!t->have_lattices) return fail(CARMEL_HIP_ERR_STATE, "build_lattices first");
  HIPCHK(hipSetDevice(t->device));
  return estimate_enqueue(t, true);
}

int carmel_hip_set_layout_policy(carmel_hip_trainer* t, int allow_unrolled) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  t->allow_unrolled = allow_unrolled != 0;
  return CARMEL_HIP_OK;
}
int carmel_hip_set_matrix_fb(carmel_hip_trainer* t, int on) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  if (t->matrix) {
    HIPCHK(hipStreamSynchronize(t->stream));
    matrix_release(t->matrix);
    t->matrix = nullptr;
  }
  if (!on) return CARMEL_HIP_OK;
  if (!t->have_lattices) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_set_matrix_fb: build_lattices first (it finds the pairs without a derivation)");
  if (t->xplan) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_set_matrix_fb: drop the exchange plan first (the matrix E-step keeps the one all-reduce)");
  if (t->unrolled) {
    // the unrolled / dense sweep of a one-tape cascade keeps its counts per parameter slot; the matrix walk produces them
    // per composed arc, as explicit lattices do: rebuild with explicit lattices so that maximize / save_best / get_counts
    // read the layout they are given
    const bool keep = t->allow_unrolled;
    t->allow_unrolled = false;
    const int rc = carmel_hip_build_lattices(t, t->build_prune, t->build_threads, nullptr, nullptr);
    t->allow_unrolled = keep;
    if (rc) return rc;
  }
  return matrix_setup(t, &t->matrix);
}
int carmel_hip_lattice_tile_sweep(carmel_hip_trainer* t) {
  if (!t || !t->have_lattices || t->unrolled || !t->lat.tile_sweep || !t->tile_group.n) return 0;
  return (int)(t->tile_group.n - 1);
}
int carmel_hip_lattice_fused_lanes(carmel_hip_trainer* t) {
  if (!t || !t->have_lattices || t->unrolled || !t->use_transpose || !t->lat.lane_fused) return 0;
  return (int)((t->wcache.n + t->lat.tile - 1) / t->lat.tile);
}
int carmel_hip_lattice_weight_source(carmel_hip_trainer* t) {
  if (!t || !t->have_lattices || t->unrolled || !t->use_transpose) return 0;
  return (t->t_t_arc.n ? 1 : 0) | (t->wave_bwd_arc.n ? 2 : 0) | (t->wave_xc_idx.n ? 4 : 0);
}
int carmel_hip_lattice_layout(carmel_hip_trainer* t) {
  if (!t || !t->have_lattices) return -1;
  return t->unrolled ? (t->dense ? 2 : 1) : 0;
}

int carmel_hip_read_scalars(carmel_hip_trainer* t, carmel_hip_estimate_result* res) {
  if (!t || !res) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {  // a planned exchange delivers the corpus-wide scalars with the counts
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  int jrc = scalars_join(t);
  if (jrc) return jrc;
  double sc[4];
  HIPCHK(hipMemcpyAsync(sc, t->counts_ptr() + t->w.n_arcs, sizeof sc, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  res->sum_logprob = sc[0];
  res->sum_weighted_logprob = sc[1];
  res->n_pairs = (uint64_t)(sc[2] + 0.5);
  return CARMEL_HIP_OK;
}

int carmel_hip_estimate_finish(carmel_hip_trainer* t, carmel_hip_estimate_result* res, double* per_pair_logprob) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  carmel_hip_estimate_result r;
  std::memset(&r, 0, sizeof r);
  int rc = carmel_hip_read_scalars(t, &r);
  if (rc) return rc;
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, t->ev0, t->ev1));
  r.kernel_ms = ms;
  if (per_pair_logprob) {
    HIPCHK(hipMemcpyAsync(per_pair_logprob, t->pair_logprob.p, t->corpus.n_pairs * sizeof(double),
                          hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
  }
  if (res) *res = r;
  if ((t->unrolled ? t->um.pair_id.size() : t->lat.n_kept) == 0)
    return fail(CARMEL_HIP_ERR_NO_DERIV, "No training example had a derivation - aborting training.");
  return CARMEL_HIP_OK;
}

int carmel_hip_estimate(carmel_hip_trainer* t, carmel_hip_estimate_result* res, double* per_pair_logprob) {
  int rc = carmel_hip_estimate_async(t);
  if (rc) return rc;
  return carmel_hip_estimate_finish(t, res, per_pair_logprob);
}

void* carmel_hip_counts_dev(carmel_hip_trainer* t) { return t ? (void*)t->counts_ptr() : nullptr; }
int carmel_hip_use_external_counts(carmel_hip_trainer* t, void* dev_ptr) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  // a sharded plan reduce-scatters the trainer's OWN count buffer chunk by chunk: it cannot be pointed elsewhere under it
  if (dev_ptr && t->xplan && exchange_is_sharded(t->xplan))
    return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_use_external_counts: a sharded exchange is planned -- carmel_hip_exchange_clear (collective) "
                                      "or destroy / abort the communicator first");
  t->ext_counts = (double*)dev_ptr;
  return CARMEL_HIP_OK;
}
int carmel_hip_last_sweep_ms(carmel_hip_trainer* t, double* ms) {
  if (!t || !ms) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  HIPCHK(hipEventSynchronize(t->ev1));
  float f = 0;
  HIPCHK(hipEventElapsedTime(&f, t->ev0, t->ev1));
  *ms = f;
  return CARMEL_HIP_OK;
}
int carmel_hip_synchronize(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  int jrc = scalars_join(t);
  if (jrc) return jrc;
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}
uint64_t carmel_hip_counts_len(carmel_hip_trainer* t) { return t ? t->w.n_arcs + 4 : 0; }
void* carmel_hip_stream(carmel_hip_trainer* t) { return t ? (void*)t->stream : nullptr; }

int carmel_hip_get_counts(carmel_hip_trainer* t, double* counts) {
  if (!t || !counts) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  if (t->unrolled && t->cascade) {
    // The unrolled / dense sweep of a cascade accumulates per PARAMETER; the reference always has arc_counts::counts per
    // composed arc (train.h:28-40, derivations.h:432-449).  They depend on the composed arcs' weights and the corpus only,
    // so they come from one on-demand pass over EXPLICIT lattices: a second trainer over the same composed transducer
    // with its weights as they stand now, explicit layout, one E-step.  (Counts of the weights the last estimate saw: the
    // trainer's weights change only in maximize.)
    const double need = (double)t->um.lattice_arcs * 64.0;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    if (t->um.lattice_arcs >= (1ull << 31) || need > 0.5 * (double)free_b)
      return fail(CARMEL_HIP_ERR_UNSUPPORTED, "composed-arc counts under the unrolled sweep need explicit lattices: " +
                                                  std::to_string((unsigned long long)(need / 1e9)) + " GB for this corpus");
    std::vector<double> aw(t->w.n_arcs);
    HIPCHK(hipMemcpyAsync(aw.data(), t->arc_logw.p, aw.size() * sizeof(double), hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    carmel_hip_trainer* x = nullptr;
    int rc = carmel_hip_create(&x, t->device, t->w.n_states, t->w.final_state, t->w.n_arcs, t->w.src.data(), t->w.dst.data(),
                               t->w.in.data(), t->w.out.data(), aw.data(), nullptr);
    if (rc) return rc;
    struct Guard {
      carmel_hip_trainer* x;
      ~Guard() { carmel_hip_destroy(x); }
    } guard{x};
    x->allow_unrolled = false;
    const HostCorpus& c = t->corpus;
    rc = carmel_hip_set_corpus(x, c.n_pairs, c.in_off.data(), c.in_sym.data(), c.out_off.data(), c.out_sym.data(),
                               c.weight.empty() ? nullptr : c.weight.data());
    if (!rc) rc = carmel_hip_build_lattices(x, 1, 0, nullptr, nullptr);
    carmel_hip_estimate_result er;
    if (!rc) rc = carmel_hip_estimate(x, &er, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(counts, x->counts_ptr(), t->w.n_arcs * sizeof(double), hipMemcpyDeviceToHost, x->stream));
    HIPCHK(hipStreamSynchronize(x->stream));
    return CARMEL_HIP_OK;
  }
  if (t->xplan) {
    int xrc = exchange_settle(t, true);
    if (xrc) return xrc;
  }
  HIPCHK(hipMemcpyAsync(counts, t->counts_ptr(), t->w.n_arcs * sizeof(double), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}
int carmel_hip_set_counts(carmel_hip_trainer* t, const double* counts) {
  if (!t || !counts) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  HIPCHK(hipMemcpyAsync(t->counts_ptr(), counts, t->w.n_arcs * sizeof(double), hipMemcpyHostToDevice, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}

int carmel_hip_accumulate_counts(carmel_hip_trainer* t, int op) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_accumulate_counts: drop the exchange plan first (its pieces follow one lattice set's buckets)");
  if (t->unrolled && op != 0)
    return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_accumulate_counts: explicit lattices only (carmel_hip_set_layout_policy(t, 0) before build_lattices)");
  const uint64_t n = t->w.n_arcs + 4;
  hipStream_t s = t->stream;
  {
    int jrc = scalars_join(t);
    if (jrc) return jrc;
  }
  if (op == 0) {
    if (t->counts_acc.n != n) HIPCHK(t->counts_acc.alloc(n));
    HIPCHK(hipMemsetAsync(t->counts_acc.p, 0, n * sizeof(double), s));
  } else if (op == 1 || op == 2) {
    if (t->counts_acc.n != n) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_accumulate_counts: clear the accumulator first (op 0)");
    if (op == 1)
      HIPCHK(launch_add(t->counts_acc.p, t->counts_ptr(), n, s));
    else
      HIPCHK(hipMemcpyAsync(t->counts_ptr(), t->counts_acc.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  } else
    return fail(CARMEL_HIP_ERR_ARG, "carmel_hip_accumulate_counts: op is 0 (clear), 1 (add) or 2 (write back)");
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}

static int cascade_param_counts(carmel_hip_trainer* t, hipStream_t s);
int carmel_hip_maximize(carmel_hip_trainer* t, double delta_scale, double* max_change) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (!t->have_norm) return fail(CARMEL_HIP_ERR_STATE, "set_norm / set_cascade first");
  HIPCHK(hipSetDevice(t->device));
  hipStream_t s = t->stream;
  {
    int prc = lds_poison(t);
    if (prc) return prc;
  }
  if (t->xplan) {
    // corpus-sharded EM with a planned exchange: every rank normalises its own arc ranges and the weights are gathered
    // (exchange.cpp); an over-relaxed step needs every count everywhere, so the count pieces are gathered first
    if (delta_scale <= 1.0) {
      int handled = 0;
      int xrc = exchange_maximize(t, t->xplan, max_change, &handled);
      if (xrc || handled) return xrc;
    }
    int xrc = exchange_settle(t, true);
    if (xrc) return xrc;
  }
  if (t->cascade) {
    // distribute_counts (cascade.h:318-325): parameter counts = sum over composed arcs using it of
    // (composed count + composed prior)
    int rcc = cascade_param_counts(t, s);
    if (rcc) return rcc;
  }
  // the weights before the update are kept only when the over-relaxation step needs them (train.cc:157-171)
  int rc = run_mstep(t, 1, (!t->cascade && delta_scale > 1.0) ? 1 : 2);
  if (rc) return rc;
  double result = 10.0;  // train.cc:922
  if (!t->cascade) {
    if (delta_scale > 1.0) {
      HIPCHK(launch_overrelax(t->arc_logw.p, t->old_logw.p, t->em_logw.p, t->arc_group.p, delta_scale, t->w.n_arcs, s));
      rc = run_mstep(t, 0, 0);  // x.normalize(methods[0]) on the overrelaxed weights, scratch kept
      if (rc) return rc;
      HIPCHK(hipMemsetAsync(t->maxchg.p, 0, sizeof(unsigned long long), s));
      HIPCHK(launch_max_change(t->arc_logw.p, t->old_logw.p, t->arc_group.p, t->maxchg.p, t->w.n_arcs, s));
      t->em_valid = true;
    } else {
      t->em_valid = false;  // the EM update IS the weight vector: nothing to keep apart (train.cc:157-171 only acts for rate > 1)
    }
    unsigned long long bits = 0;
    int frc = fetch_u64(t, t->maxchg.p, &bits, s, !(delta_scale > 1.0));  // (an over-relaxed step ends in max_change_kernel)
    if (frc) return frc;
    double d;
    std::memcpy(&d, &bits, sizeof d);
    result = d;
  } else {
    HIPCHK(hipStreamSynchronize(s));
  }
  if (max_change) *max_change = result;
  return CARMEL_HIP_OK;
}

// distribute_counts on the current counts (shared by maximize and fractional_counts)
static int cascade_param_counts(carmel_hip_trainer* t, hipStream_t s) {
  if (t->unrolled) {  // the sweep accumulated per parameter already
    HIPCHK(launch_unrolled_param_counts(t->param_counts_c.p, t->counts_ptr(), t->u_param_uses.p,
                                        t->smooth_floor > 0 ? t->smooth_floor : 0.0, t->u_param_wprior.p, t->param_group_c.p,
                                        t->u_slot_of.p, (uint32_t)t->n_params, s));
  } else {
    HIPCHK(hipMemsetAsync(t->param_counts_c.p, 0, t->param_counts_c.bytes(), s));
    HIPCHK(launch_chain_scatter(t->param_counts_c.p, t->counts_ptr(), t->smooth_floor > 0 ? t->smooth_floor : 0.0,
                                t->arc_prior_w.p, t->arc_group.p, t->chain_off.p, t->chain_param.p, t->param_group_c.p,
                                t->w.n_arcs, s));
  }
  return CARMEL_HIP_OK;
}

int carmel_hip_fractional_counts(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, true);
    if (xrc) return xrc;
  }
  hipStream_t s = t->stream;
  if (t->cascade) {
    int rc = cascade_param_counts(t, s);
    if (rc) return rc;
    HIPCHK(launch_counts_to_logw(t->param_logw_c.p, t->param_counts_c.p, nullptr, t->param_group_c.p, t->n_params, s));
  } else {
    HIPCHK(launch_counts_to_logw(t->arc_logw.p, t->counts_ptr(), (t->have_prior && t->prior_nonzero) ? t->prior.p : nullptr,
                                 t->arc_group.p, t->w.n_arcs, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}

int carmel_hip_set_digamma(carmel_hip_trainer* t, uint32_t n_members, const double* alpha, const uint8_t* enabled) {
  if (!t || !alpha || !enabled) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (!t->have_norm) return fail(CARMEL_HIP_ERR_STATE, "set_norm / set_cascade first");
  HIPCHK(hipSetDevice(t->device));
  const double nan = std::numeric_limits<double>::quiet_NaN();
  std::vector<double> ga(t->h_group_member.size(), nan), ta(t->h_tie_member.size(), nan);
  bool any = false;
  for (size_t g = 0; g < ga.size(); ++g) {
    const uint32_t m = t->h_group_member[g];
    if (m < n_members && enabled[m]) {
      ga[g] = alpha[m];
      any = true;
    }
  }
  for (size_t k = 0; k < ta.size(); ++k) {
    const uint32_t m = t->h_tie_member[k];
    if (m < n_members && enabled[m]) ta[k] = alpha[m];
  }
  t->any_digamma = any;
  if (any) {
    HIPCHK(t->dig_alpha.upload(ga, t->stream));
    HIPCHK(t->tie_alpha.upload(ta, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
  } else {
    t->dig_alpha.release();
    t->tie_alpha.release();
  }
  return CARMEL_HIP_OK;
}

int carmel_hip_debug_lattice_fingerprint(carmel_hip_trainer* t, uint64_t* out16) {
  if (!t || !out16) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (!t->have_lattices) return fail(CARMEL_HIP_ERR_STATE, "build_lattices first");
  HIPCHK(hipSetDevice(t->device));
  return carmel_hip_debug_lattice_fingerprint_impl(t, out16);
}

int carmel_hip_random_restart(carmel_hip_trainer* t, uint64_t seed, uint32_t restart) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (!t->have_norm) return fail(CARMEL_HIP_ERR_STATE, "set_norm / set_cascade first");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  const uint64_t n = t->np();
  std::vector<double> lw(n);
  HIPCHK(hipMemcpyAsync(lw.data(), t->params(), n * sizeof(double), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  const std::vector<uint32_t>& grp = t->cascade ? t->h_param_group : t->w.group;
  for (uint64_t p = 0; p < n; ++p) {
    if (grp[p] == CARMEL_HIP_LOCKED_GROUP) continue;        // randomSet leaves locked arcs alone (fst.h:976)
    if (t->h_norm_of[p] == 0xffffffffu) continue;           // members normalised by NONE are not randomised (cascade.h:398-401)
    lw[p] = std::log(1.0 - gibbs_uniform(seed, restart, (uint32_t)p, 0));
  }
  int rc = carmel_hip_set_weights(t, lw.data());
  if (rc) return rc;
  return carmel_hip_normalize(t);
}

int carmel_hip_keep_em_weights(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (t->cascade) return fail(CARMEL_HIP_ERR_STATE, "over-relaxed EM is off for cascades (train.cc:545-549)");
  if (!t->em_valid) return CARMEL_HIP_OK;  // last step was a plain EM update already
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  HIPCHK(hipMemcpyAsync(t->arc_logw.p, t->em_logw.p, t->w.n_arcs * sizeof(double), hipMemcpyDeviceToDevice, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return CARMEL_HIP_OK;
}

// single transducer: best_weight <- weight (train.cc:184-186).  cascade: best <- em_weight, i.e. the composed
// COUNTS saved before this estimate (train.cc:123-130)
int carmel_hip_save_counts(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  // for_arcs::save_counts: em_weight <- weight() of the composed arc, which after the previous maximize's
  // prep_new_weights holds (count + prior) — here: the counts buffer of the previous estimate
  if (t->unrolled && t->cascade)  // parameter space: the sweep never had composed-arc counts
    HIPCHK(hipMemcpyAsync(t->u_em_param.p, t->counts_ptr(), t->u_n_slots * sizeof(double), hipMemcpyDeviceToDevice, t->stream));
  else
    HIPCHK(hipMemcpyAsync(t->em_logw.p, t->counts_ptr(), t->w.n_arcs * sizeof(double), hipMemcpyDeviceToDevice, t->stream));
  return CARMEL_HIP_OK;
}
int carmel_hip_save_best(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  if (t->unrolled && t->cascade) {
    HIPCHK(hipMemcpyAsync(t->u_best_param.p, t->u_em_param.p, t->u_n_slots * sizeof(double), hipMemcpyDeviceToDevice, t->stream));
    return CARMEL_HIP_OK;
  }
  const double* from = t->cascade ? t->em_logw.p : t->arc_logw.p;
  HIPCHK(hipMemcpyAsync(t->best_logw.p, from, t->w.n_arcs * sizeof(double), hipMemcpyDeviceToDevice, t->stream));
  return CARMEL_HIP_OK;
}
int carmel_hip_load_best(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan) {
    int xrc = exchange_settle(t, false);
    if (xrc) return xrc;
  }
  hipStream_t s = t->stream;
  if (!t->cascade) {
    HIPCHK(hipMemcpyAsync(t->arc_logw.p, t->best_logw.p, t->w.n_arcs * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return CARMEL_HIP_OK;
  }
  // load_best + use_counts_final (train.cc:673-674, cascade.h:358-364): best composed counts -> parameters
  if (t->unrolled) {
    HIPCHK(launch_unrolled_param_counts(t->param_counts_c.p, t->u_best_param.p, t->u_param_uses.p,
                                        t->smooth_floor > 0 ? t->smooth_floor : 0.0, t->u_param_wprior.p, t->param_group_c.p,
                                        t->u_slot_of.p, (uint32_t)t->n_params, s));
  } else {
    HIPCHK(hipMemcpyAsync(t->counts_ptr(), t->best_logw.p, t->w.n_arcs * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemsetAsync(t->param_counts_c.p, 0, t->param_counts_c.bytes(), s));
    // the saved value is the composed COUNT of the best iteration; the composed prior is added here, as maximize does
    HIPCHK(launch_chain_scatter(t->param_counts_c.p, t->counts_ptr(), t->smooth_floor > 0 ? t->smooth_floor : 0.0,
                                t->arc_prior_w.p, t->arc_group.p, t->chain_off.p, t->chain_param.p, t->param_group_c.p,
                                t->w.n_arcs, s));
  }
  int rc = run_mstep(t, 1, 2);
  if (rc) return rc;
  HIPCHK(launch_chain_update(t->arc_logw.p, t->arc_group.p, t->chain_off.p, t->chain_param.p, t->param_logw_c.p,
                             t->w.n_arcs, s));
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}

}  // extern "C"
