/* carmel_hip.h — C-ABI of the MI355X-native EM / Gibbs hot path for carmel.
 *
 * The reference (graehl/carmel) has no FFI: its seams for this path are C++ member functions
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces; INTEGRATION.md shows
 * the few lines a carmel maintainer adds in train.cc / gibbs.cc to call them.
 *
 * Conventions: plain pointers and sizes only; every array is caller-owned host memory unless the name says
 * `_dev`; all functions return 0 on success or a negative code, with text in carmel_hip_last_error() (no
 * exceptions cross this boundary — the reference throws std::runtime_error, carmel.cc:1558-1561); one handle
 * per GPU; a handle is not thread-safe (like the reference, cascade.h:16).
 *
 * Arc numbering is the reference's: arc id = position in `for s in states: for a in states[s].arcs`
 * (derivations.h:86-101, fst.h:1331-1334); all per-arc arrays are in that order.  Weights are natural logs
 * (logweight<double>, weight.h:132-135; zero = -inf).  Expected counts cross the boundary in the LINEAR
 * domain (f64): the reference accumulates them with log-add (derivations.h:445), here they are plain sums so
 * that corpus shards can be combined with one all-reduce.
 */
#ifndef CARMEL_HIP_H
#define CARMEL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CARMEL_HIP_OK 0
#define CARMEL_HIP_ERR_ARG -1
#define CARMEL_HIP_ERR_HIP -2      /* a HIP runtime call failed / no device */
#define CARMEL_HIP_ERR_STATE -3    /* call order violated (e.g. estimate before build_lattices) */
#define CARMEL_HIP_ERR_NO_DERIV -4 /* no training example had a derivation (train.cc:241-252) */
#define CARMEL_HIP_ERR_UNSUPPORTED -5

#define CARMEL_HIP_NO_GROUP 0xFFFFFFFFu /* FSTArc::no_group  (arc.h:49) */
#define CARMEL_HIP_LOCKED_GROUP 0u      /* FSTArc::locked_group (arc.h:50) */

/* WFST::norm_group_by (fst.h) */
#define CARMEL_HIP_NORM_CONDITIONAL 0
#define CARMEL_HIP_NORM_JOINT 1
#define CARMEL_HIP_NORM_NONE 2

typedef struct carmel_hip_trainer carmel_hip_trainer;

const char* carmel_hip_last_error(void);
/* The library's switches (round 6; csrc/options.hpp): formulation choices that leave the results the same -- each exists because a
 * test holds the two forms together --, layout limits the tests force onto small cases, and traces.  Process-wide; `key` is one of
 * carmel_hip_option_name(0 .. carmel_hip_option_count() - 1) (e.g. "tile_sweep", "gibbs_lane", "timing"), `value` a string as the
 * former environment variable CARMEL_HIP_<KEY> took it, NULL to unset; an unknown key is CARMEL_HIP_ERR_ARG.  Options that shape
 * a lattice layout are read by carmel_hip_build_lattices / carmel_hip_gibbs_create / carmel_hip_forests_create, the others by the
 * call they steer.  The library does not read the environment: the front ends (carmel, forest-em, bench.py) translate
 * CARMEL_HIP_<KEY>=v into carmel_hip_set_option("<key>", "v") and CARMEL_TIMING into "timing" for the tools that drive them. */
int carmel_hip_set_option(const char* key, const char* value);
const char* carmel_hip_get_option(const char* key);
int carmel_hip_option_count(void);
const char* carmel_hip_option_name(int i);
int carmel_hip_device_count(void);

/* Replaces: arcs_table<arc_counts>(WFST&, per_arc_prior, global_prior)  derivations.h:79-101, train.h:28-40,
 * i.e. the forward_backward constructor's view of the (composed) transducer, train.cc:367-411.
 * `group` may be NULL (all arcs normal).  For a real cascade `group[k]` is the chain id of composed arc k
 * (cascade.h:18-21) and carmel_hip_set_cascade must follow. */
int carmel_hip_create(carmel_hip_trainer** out, int device, uint32_t n_states, uint32_t final_state, uint64_t n_arcs,
                      const uint32_t* src, const uint32_t* dst, const uint32_t* in_sym, const uint32_t* out_sym,
                      const double* logw, const uint32_t* group);
int carmel_hip_destroy(carmel_hip_trainer* t);

/* Replaces: training_corpus + IOSymSeq (train.h:80-189) as filled by WFST::read_training_corpus
 * (train.cc:985-1025).  Pair p's input symbols are in_sym[in_off[p] .. in_off[p+1]).  pair_weight NULL = 1. */
int carmel_hip_set_corpus(carmel_hip_trainer* t, uint64_t n_pairs, const uint64_t* in_off, const uint32_t* in_sym,
                          const uint64_t* out_off, const uint32_t* out_sym, const double* pair_weight);

typedef struct carmel_hip_lattice_stats {
  uint64_t n_pairs;          /* pairs given */
  uint64_t n_pairs_kept;     /* pairs with >= 1 derivation (cached_derivs.h:87-98) */
  uint64_t explored_states;  /* derivations::statistics pre  (derivations.h:191-247) */
  uint64_t explored_arcs;
  uint64_t kept_states;      /* ... post: states/arcs on some start->goal path */
  uint64_t kept_arcs;
  uint64_t n_cyclic_pairs;   /* lattices with a cycle (derivations.h:726-728 warns); swept in the reference's order */
  uint64_t n_bundles;        /* workgroup-sized batches of lattices laid out in HBM */
  uint64_t max_levels;
  uint64_t device_bytes;     /* HBM held by lattices */
  double build_seconds;
  /* derivations::statistics exactly as the reference keeps it (derivations.h:197-210, 617-618, 687), for its
   * "Pre pruning / Post pruning" log lines: pre.arcs accumulates over all pairs (= explored_arcs above), but pre.states,
   * post.states and post.arcs are ASSIGNED per pair -- what is logged is the last pair's (post: the last pair that has a
   * derivation).  commands.trace:6984-6986 pins them on the tagging cascade: (100 states, 182891 arcs) -> (75, 164). */
  uint64_t last_pair_explored_states, last_pair_kept_states, last_pair_kept_arcs;
  uint64_t n_windowed_pairs; /* lattices swept one per lane through a ring of LDS rows (lattice.hpp, LaneGroup::window) */
} carmel_hip_lattice_stats;

/* Replaces: cached_derivs::cache_derivations (cached_derivs.h:104-138) -> derivations::init_and_compute
 * (derivations.h:471-513, 640-704) + prune (:572-629): builds every pair's derivation lattice once, drops
 * pairs without a derivation, and lays the lattices out in HBM as batched CSR.  has_derivation[n_pairs] and
 * stats may be NULL.  host_threads <= 0: all cores. */
int carmel_hip_build_lattices(carmel_hip_trainer* t, int prune, int host_threads, uint8_t* has_derivation,
                              carmel_hip_lattice_stats* stats);

/* Test hook: 16 checksums of the lattice image in device memory (lane groups and record streams, transposition tables,
 * pair weights) -- the GPU lattice builder (csrc/lattice_gpu.hip; CARMEL_HIP_GPU_BUILD=0 switches it off) must leave
 * the very bytes the host builder leaves. */
int carmel_hip_debug_lattice_fingerprint(carmel_hip_trainer* t, uint64_t* out16);
/* > 0: the explicit lattices are laid out for the tile sweep (csrc/lattice.hpp, TILE_SWEEP_TILE: a corpus of small plain lane
 * lattices; the E-step is bucket pass, tile_sweep_kernel, bucket pass) -- the number of tiles; 0 otherwise.
 * CARMEL_HIP_TILE_SWEEP=0 at build time keeps the five-kernel layout, CARMEL_HIP_TILE_SWEEP_KERNEL=0 runs the three middle
 * kernels on the tile-sweep layout (A/B switches: same counts). */
int carmel_hip_lattice_tile_sweep(carmel_hip_trainer* t);
/* > 0: the explicit lattices are laid out as fused lanes (csrc/lattice.hpp, LANE_FUSED_TILE: a corpus of one-per-lane lattices
 * the tile sweep does not take -- windowed groups, lattices above 48 arcs; every lane group starts on a tile of its own (LANE_FUSED_TILE positions)
 * and the lane sweep's backward pass hands its posteriors to the count pass itself, derivations.h:432-449 without a stored
 * posterior array) -- the number of lane tiles; 0 otherwise.  CARMEL_HIP_LANE_FUSED=0 at build time keeps the 16384-position
 * tiles, CARMEL_HIP_LANE_FUSED_KERNEL=0 runs sweep -> post -> trans_c_tile on the fused layout (A/B switches: same counts). */
int carmel_hip_lattice_fused_lanes(carmel_hip_trainer* t);
/* where the E-step's sweeps get an arc's weight from (derivations.h:395-417 reads the arc's own weight through a pointer): bit 0
 * = the tile passes fetch a tile's weights from the WFST's table through the arc id of every item (no bucket pass, no X:
 * WFSTs of four and more items an arc whose table is at most 128 MB; CARMEL_HIP_TILE_GATHER=0/1), bit 1 = the
 * one-per-wavefront sweeps gather theirs from the table through the records' arc ids (tables of at most 64 MB;
 * CARMEL_HIP_WAVE_GATHER=0/1); neither: every weight goes through the blocked transposition.  Bit 2: on the way out, the
 * one-per-wavefront sweeps send an arc's posterior (derivations.h:432-449) straight to its item's place in the count pass's
 * input -- no posterior array, no tile pass over their tiles (corpora of such lattices alone whose rows' items are neighbours
 * there; CARMEL_HIP_WAVE_XC=0/1).  A/B switches: the same values at the same places of the same sums. */
int carmel_hip_lattice_weight_source(carmel_hip_trainer* t);
/* how the derivation lattices are held: 0 = explicit (lane groups / bundles in HBM), 1 = unrolled over string positions
 * (one-tape models, never stored), 2 = unrolled in the rank-1 dense form (LM o channel cascades, dense.hpp); -1: none built */
int carmel_hip_lattice_layout(carmel_hip_trainer* t);

/* Replaces: WFST::NormalizeMethod for a single (non-cascade) transducer — carmel -n/-j/-u and --priors
 * (carmel.cc:488-499); used by carmel_hip_maximize / carmel_hip_normalize. */
int carmel_hip_set_norm(carmel_hip_trainer* t, int norm_group_by, double add_count);

/* Replaces: per-arc prior of arcs_table (derivations.h:99-100): prior = smooth_floor (+ current arc weight when
 * weight_is_prior_count, carmel -U), captured at call time like the arcs_table constructor does. */
int carmel_hip_set_prior(carmel_hip_trainer* t, double smooth_floor, int weight_is_prior_count);

/* Replaces: cascade_parameters chains (cascade.h:233, 489-599) for --train-cascade.  Parameters are the arcs
 * of the original transducers, concatenated member by member in visit order.  chain c lists
 * chain_param[chain_off[c] .. chain_off[c+1]).  param_member[p] = index of the member transducer;
 * member_norm[m] / member_add_count[m] = its NormalizeMethod; param_src / param_in give (state, input symbol)
 * for norm-group formation; param_group as FSTArc::groupId (0 = locked). */
int carmel_hip_set_cascade(carmel_hip_trainer* t, uint64_t n_params, const double* param_logw,
                           const uint32_t* param_group, const uint32_t* param_member, const uint32_t* param_src,
                           const uint32_t* param_in, uint32_t n_members, const int* member_norm,
                           const double* member_add_count, uint64_t n_chains, const uint64_t* chain_off,
                           const uint64_t* chain_param);

/* Replaces: WFST::normalize(method) (fst.cc:86-244) applied to the current weights (cascade: every member,
 * cascade.h:402-405, then cascade.update()).  WFST::train calls it once before iteration 1 (train.cc:509). */
int carmel_hip_normalize(carmel_hip_trainer* t);

/* Set / get the current arc weights (ln).  For a cascade these are the PARAMETER weights (n_params). */
int carmel_hip_set_weights(carmel_hip_trainer* t, const double* logw);
int carmel_hip_get_weights(carmel_hip_trainer* t, double* logw);
/* composed-arc weights after cascade.update() (n_arcs); same as get_weights for a single transducer */
int carmel_hip_get_arc_weights(carmel_hip_trainer* t, double* logw);

typedef struct carmel_hip_estimate_result {
  double sum_logprob;          /* ln of unweighted_corpus_prob  (train.cc:330) */
  double sum_weighted_logprob; /* ln of weighted_corpus_prob    (train.cc:331) */
  uint64_t n_pairs;            /* pairs swept (those with a derivation) */
  double kernel_ms;            /* HIP-event time of the sweep kernels of this call, on the trainer's stream */
} carmel_hip_estimate_result;

/* Replaces: forward_backward::estimate (train.cc:763-773) = cascade.update (cascade.h:466-479) + clear counts +
 * for every pair derivations::collect_counts (derivations.h:400-449: forward sweep, backward sweep, count
 * accumulation).  Runs on the GPU over this trainer's corpus shard; counts stay on the device
 * (carmel_hip_counts_dev) until carmel_hip_get_counts / carmel_hip_maximize.
 * per_pair_logprob (n_pairs given to set_corpus; -inf for dropped pairs) may be NULL.
 * carmel_hip_estimate_async only enqueues (for overlap with an all-reduce on another stream); _finish syncs and
 * fills the result. */
int carmel_hip_estimate(carmel_hip_trainer* t, carmel_hip_estimate_result* res, double* per_pair_logprob);
int carmel_hip_estimate_async(carmel_hip_trainer* t);
int carmel_hip_estimate_finish(carmel_hip_trainer* t, carmel_hip_estimate_result* res, double* per_pair_logprob);

/* Device buffer of n_arcs + 4 doubles: linear expected counts per (composed) arc followed by
 * {sum_logprob, sum_weighted_logprob, n_pairs, 0}.  This is the ONE buffer a data-parallel caller sums across
 * ranks between estimate and maximize (RCCL allReduce(sum) over xGMI); see bench.py / INTEGRATION.md. */
void* carmel_hip_counts_dev(carmel_hip_trainer* t);
uint64_t carmel_hip_counts_len(carmel_hip_trainer* t);
void* carmel_hip_stream(carmel_hip_trainer* t); /* hipStream_t the trainer enqueues on */
/* Let the caller own that buffer (e.g. a torch tensor handed to torch.distributed): dev_ptr must hold
 * n_arcs + 4 doubles on this trainer's device and outlive the trainer; NULL switches back. */
int carmel_hip_use_external_counts(carmel_hip_trainer* t, void* dev_ptr);
int carmel_hip_synchronize(carmel_hip_trainer* t); /* wait for everything enqueued on the trainer's stream */
/* HIP-event time (ms) of the sweep kernels of the most recent estimate, measured on the trainer's stream */
int carmel_hip_last_sweep_ms(carmel_hip_trainer* t, double* ms);
/* after an external all-reduce of counts_dev: re-read the three scalars into res */
int carmel_hip_read_scalars(carmel_hip_trainer* t, carmel_hip_estimate_result* res);
int carmel_hip_get_counts(carmel_hip_trainer* t, double* counts /* n_arcs, linear */);
int carmel_hip_set_counts(carmel_hip_trainer* t, const double* counts /* n_arcs, linear */);
/* Corpora whose derivation lattices are not kept resident all at once (the reference without -? rebuilds every pair's derivations
 * in every iteration, cached_derivs.h:60-101, and spills its cache to disk when it outgrows memory, fst.h:1057-1076,
 * --disk-cache-derivations): the caller walks the corpus in shards -- carmel_hip_set_corpus + carmel_hip_build_lattices +
 * carmel_hip_estimate per shard -- and adds the shards' count buffers (n_arcs counts + the four corpus scalars) up on the
 * device: op 0 clears the accumulator, op 1 adds the trainer's count buffer to it (after a shard's estimate), op 2 writes the
 * sums back into the count buffer, where carmel_hip_read_scalars / _get_counts / _maximize find the whole corpus.  Explicit
 * lattices only (carmel_hip_set_layout_policy(t, 0): the unrolled layouts keep per-parameter sums). */
int carmel_hip_accumulate_counts(carmel_hip_trainer* t, int op);

/* Replaces: forward_backward::maximize (train.cc:893-923): prep_new_weights (:134-153), cascade.use_counts
 * (distribute_counts cascade.h:318-325 + normalize fst.cc:86-244), overrelax (:157-171, delta_scale > 1 only for
 * a single transducer) and max_change (:173-182).  *max_change gets max |new - old| in the real domain
 * (10 for a real cascade, train.cc:922). */
int carmel_hip_maximize(carmel_hip_trainer* t, double delta_scale, double* max_change);

/* Replaces: the `max_iter == 0` branch of WFST::train (train.cc:520-531: for_arcs::prep_new_weights(1.0) followed by
 * cascade.distribute_counts(), no normalisation) -- after an estimate, every unlocked parameter's weight becomes its
 * unnormalised fractional count (+ prior); locked parameters keep their weights.  `carmel -t -M 0`. */
int carmel_hip_fractional_counts(carmel_hip_trainer* t);

/* Replaces: WFST::NormalizeMethod::scale = mean_field_scale (graehl/shared/mean_field_scale.hpp:40-52; carmel --digamma=a,,b
 * per cascade member, carmel.cc:495, and `-+ a` for a single transducer, carmel.cc:1009-1013): for member m with
 * enabled[m] != 0 the normalisation of carmel_hip_normalize / carmel_hip_maximize uses exp(digamma(x + alpha[m])) in place
 * of x for the numerator and for the group sum (fst.cc:189, 217-221).  Call after carmel_hip_set_norm / _set_cascade
 * (which reset it); a single transducer is member 0. */
int carmel_hip_set_digamma(carmel_hip_trainer* t, uint32_t n_members, const double* alpha, const uint8_t* enabled);

/* Replaces cascade_parameters::random_restart (cascade.h:398-411: WFST::randomSet on every member not normalised by
 * NONE, then normalize): every unlocked parameter p gets the weight 1 - u(seed, restart, p) in (0, 1] from the
 * library's counter-based generator (carmel_hip_gibbs_uniform(seed, restart, p, 0)), then the model is normalised.
 * The reference draws from Boost's lagged_fibonacci607: restart sequences are not comparable with it (unpinned). */
int carmel_hip_random_restart(carmel_hip_trainer* t, uint64_t seed, uint32_t restart);
/* Replaces for_arcs::keep_em_weight (train.cc:188-190, used at :639-643): after an over-relaxed step (delta_scale > 1)
 * failed to improve, the weights go back to the plain EM update of that step.  Valid after a carmel_hip_maximize call
 * with delta_scale > 1 on a single transducer. */
int carmel_hip_keep_em_weights(carmel_hip_trainer* t);
/* Replaces: for_arcs::save_best / save_best_counts / use_best_weight (train.cc:123-198, 449-457) and
 * cascade.use_counts_final (cascade.h:358-364): device-side snapshots so WFST::train's "keep the weights that
 * produced the best estimate" needs no host copies. */
int carmel_hip_save_counts(carmel_hip_trainer* t); /* for_arcs::save_counts: em_weight <- weight (cascade) */
int carmel_hip_save_best(carmel_hip_trainer* t);
int carmel_hip_load_best(carmel_hip_trainer* t);

/* ---- corpus-sharded EM over the GPUs of one node (RCCL over xGMI) ----
 * No reference counterpart (carmel is single-process); SURVEY.md 8(e): every rank holds the whole model and a shard of
 * the training pairs, and the one exchange per iteration is the sum of counts[n_arcs + 4] (expected counts + the corpus
 * scalars of train.cc:326-332) between estimate and maximize.  One process per GPU; rank 0 makes the 128-byte id
 * (carmel_hip_comm_unique_id) and hands it to the others by any means (the carmel front end uses a pipe, bench.py
 * torch.distributed).  carmel_hip_allreduce_counts ENQUEUES the all-reduce on the trainer's stream, in place on
 * carmel_hip_counts_dev: estimate_async -> allreduce_counts -> maximize needs no host synchronisation in between
 * (carmel_hip_read_scalars afterwards gives the corpus-wide scalars).  RCCL is loaded at run time (dlopen); without it
 * these calls return CARMEL_HIP_ERR_UNSUPPORTED.  carmel_hip_comm_allreduce_host sums (or maximises) a few host doubles
 * across ranks -- corpus statistics, timings. */
typedef struct carmel_hip_comm carmel_hip_comm;
int carmel_hip_comm_unique_id(void* id128);
int carmel_hip_comm_create(carmel_hip_comm** out, int device, int rank, int world, const void* id128);
/* Destroying (or aborting) a communicator first drops the exchange plan of every trainer planned on it
 * (carmel_hip_exchange_plan): trainer and communicator may be destroyed in either order, and a trainer that outlives its
 * communicator simply runs unplanned (local counts, replicated M-step). */
int carmel_hip_comm_destroy(carmel_hip_comm* c);
/* after a collective failed on some rank: drop whatever is still enqueued instead of waiting for it (ncclCommAbort) */
int carmel_hip_comm_abort(carmel_hip_comm* c);
int carmel_hip_comm_rank(carmel_hip_comm* c);
int carmel_hip_comm_world(carmel_hip_comm* c);
const char* carmel_hip_comm_transport_name(carmel_hip_comm* c);
int carmel_hip_allreduce_counts(carmel_hip_trainer* t, carmel_hip_comm* c);
int carmel_hip_comm_allreduce_host(carmel_hip_comm* c, double* v, uint32_t n, int op_max);

/* A communicator over a transport of the caller's own (MPI, sockets, a test harness) instead of RCCL: three collectives on
 * f64 DEVICE buffers.  `stream` is a hipStream_t: the operation must be ordered after the work already enqueued on it and
 * be complete -- or enqueued on it -- when the call returns (a transport that works through the host synchronises the
 * stream).  allreduce: buf[0 .. n) := sum (op 0) or max (op 1) over the ranks.  reduce_scatter: buf holds world * count
 * doubles; afterwards this rank's piece buf[rank * count ..) holds the sum over the ranks of that piece (the other pieces are
 * unspecified).  all_gather: every rank's piece is copied to all ranks.  reduce_scatter / all_gather may be NULL (the
 * library then uses allreduce: same sums, more traffic).  All return 0 on success.  destroy may be NULL. */
typedef struct carmel_hip_transport {
  void* ctx;
  int (*allreduce)(void* ctx, double* dev_buf, uint64_t n, int op, void* stream);
  int (*reduce_scatter)(void* ctx, double* dev_buf, uint64_t count, void* stream);
  int (*all_gather)(void* ctx, double* dev_buf, uint64_t count, void* stream);
  void (*destroy)(void* ctx);
  const char* name;
} carmel_hip_transport;
int carmel_hip_comm_create_custom(carmel_hip_comm** out, int device, int rank, int world, const carmel_hip_transport* tr);
/* ... and, optionally, point-to-point transfers for the exchange's DIRECT form (below): one call = one group of sends and
 * receives on device buffers, all of them complete (or enqueued on `stream`) when it returns; between a pair of ranks the
 * k-th send of the one meets the k-th receive of the other and their lengths agree; no operation names the caller itself.
 * (RCCL: ncclGroupStart, ncclSend / ncclRecv, ncclGroupEnd.)  A custom transport without it keeps the collective form. */
typedef struct carmel_hip_p2p {
  int32_t peer;     /* the other rank */
  int32_t send;     /* 1: dev_buf[0 .. n) goes to peer; 0: it is filled by peer */
  double* dev_buf;
  uint64_t n;       /* doubles */
} carmel_hip_p2p;
typedef int (*carmel_hip_sendrecv_fn)(void* ctx, const carmel_hip_p2p* ops, uint32_t n_ops, void* stream);
int carmel_hip_comm_set_sendrecv(carmel_hip_comm* c, carmel_hip_sendrecv_fn fn);
/* collective: every rank sends every rank n doubles (0: 1024) of a pattern naming sender and receiver through one such group
 * and checks what arrives -- the transfers of the direct form on this transport, before a training run depends on them
 * (over RCCL a rank also sends to itself, so a world of one exercises ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd). */
int carmel_hip_comm_selftest(carmel_hip_comm* c, uint32_t n);

/* The per-iteration exchange, planned once after carmel_hip_build_lattices (collective: every rank calls it; it also
 * checks that the ranks hold their lattices in the same layout).  With a plan an iteration is still
 *     carmel_hip_estimate_async -> carmel_hip_allreduce_counts -> carmel_hip_maximize,
 * but for a single transducer under JOINT / CONDITIONAL normalisation it runs SHARDED: the arc table in n_chunks chunks
 * (0: 4; at most 16) of `world` pieces; the count pass hands each chunk to a reduce-scatter on the communicator's own stream while the
 * next chunk is still being summed; carmel_hip_maximize normalises this rank's pieces only and all-gathers the weights chunk
 * by chunk into the next count pass (csrc/exchange.cpp).  The sharded exchange has two forms.  DIRECT
 * (the default where the transport has point-to-point transfers -- RCCL has): every rank sends each peer that peer's piece
 * of the chunk, with the few arcs either side that a straddling norm group needs, in ONE group of sends and receives per
 * chunk, and adds up what arrives in rank order; the weights travel back the same way; the corpus scalars ride with the last
 * chunk of counts, the largest weight change with the first chunk of weights: 2 x n_chunks groups per iteration and no
 * small collective.  COLLECTIVES: ncclReduceScatter / ncclAllGather per chunk, one small all-reduce for the arcs at piece
 * boundaries and the scalars, one for the largest change.  `form`: 0 = choose (direct, else collectives, else all-reduce),
 * 1 = the one all-reduce of counts[n_arcs + 4] and the replicated M-step, 2 = collectives, 3 = direct (refused when the
 * transport cannot).  Other models (cascades, unrolled / dense layouts, tied groups) keep the all-reduce whatever is asked.
 * The results are the same either way up to the order of the sums.  carmel_hip_exchange_info says which form was planned
 * (*sharded: 0 = the one all-reduce, 1 = sharded over the collectives, 2 = sharded, direct) and what one iteration moves per rank; carmel_hip_exchange_measure times the exchange of one iteration on its own (all
 * its collectives back to back, nothing to hide behind; collective); carmel_hip_exchange_clear drops the plan.
 * COLLECTIVE while a sharded plan holds reduced pieces (between carmel_hip_allreduce_counts and the next count pass):
 * whatever needs the WHOLE count vector all-gathers it first -- carmel_hip_get_counts, carmel_hip_fractional_counts,
 * carmel_hip_exchange_clear, carmel_hip_maximize with delta_scale > 1.  Every rank must make these calls together (a
 * rank-0-only carmel_hip_get_counts would wait for its peers forever); with the all-reduce form they are local.
 * carmel_hip_use_external_counts is refused under a sharded plan; matrix mode (carmel_hip_set_matrix_fb) plans the
 * all-reduce form. */
int carmel_hip_exchange_plan(carmel_hip_trainer* t, carmel_hip_comm* c, uint32_t n_chunks, int form);
int carmel_hip_exchange_info(carmel_hip_trainer* t, int* sharded, uint32_t* n_chunks, uint64_t* bytes_reduce_scatter,
                             uint64_t* bytes_all_gather, uint64_t* bytes_all_reduce);
int carmel_hip_exchange_measure(carmel_hip_trainer* t, uint32_t reps, double* ms_per_exchange);
int carmel_hip_exchange_clear(carmel_hip_trainer* t);
/* allow_unrolled = 0: carmel_hip_build_lattices keeps explicit lattices even for one-tape models (what every rank must do
 * when the ranks' shards would choose different layouts); 1 (default): the builder decides per shard */
int carmel_hip_set_layout_policy(carmel_hip_trainer* t, int allow_unrolled);
/* `carmel --matrix-fb` (carmel.cc:238; forward_backward::matrix_compute / estimate_matrix / matrix_count, train.cc:698-745,
 * 776-860, 288-296): on != 0 makes carmel_hip_estimate* run the dense (input position x output position x state)
 * forward/backward -- one workgroup per pair, anti-diagonal by anti-diagonal, *e*:*e* arcs by levels of the epsilon graph
 * (csrc/matrix_fb.hip) -- instead of sweeping derivation lattices; counts, ln p per pair and the corpus scalars come out
 * as usual, so maximize / cascades / over-relaxation are unchanged.  After carmel_hip_build_lattices (which drops the pairs
 * without a derivation).  CARMEL_HIP_ERR_UNSUPPORTED when the *e*:*e* arcs form a cycle or the longest pair's matrices do
 * not fit in device memory; 0 switches back to the lattices. */
int carmel_hip_set_matrix_fb(carmel_hip_trainer* t, int on);

/* ---- blocked Gibbs sampling of derivations: `carmel --crp` ----
 * Replaces: WFST::train_gibbs / carmel_gibbs (gibbs.cc:15-41, 386-430) + gibbs_base::run_starts
 * (gibbs.hpp:803-914) + derivations::random_path (derivations.h:345-375).  The sampler is created from a trainer
 * that already has its transducer, normalisation (the per-member --priors are the Dirichlet alphas) and corpus;
 * parameters, norm groups and prior pseudo-counts alpha*p0*|group| are derived as add_gibbs_params does
 * (gibbs.cc:114-186). */
typedef struct carmel_hip_gibbs carmel_hip_gibbs;
typedef struct carmel_hip_gibbs_opts {
  uint32_t iter;      /* resampling sweeps after the initial sample (-M / --crp=N) */
  uint32_t burnin;    /* --burnin: sweeps before time-averaging starts */
  uint64_t seed;      /* of the counter-based generator carmel_hip_gibbs_uniform */
  int mode;           /* 0 = exact: blocks strictly in order (the reference's chain); 1 = parallel stale-count sweep */
  int uniform_p0, dirichlet_p0, final_counts, exclude_prior; /* --uniform-p0 --dirichlet-p0 --final-counts
                                                                 --crp-exclude-prior (gibbs_opts.hpp:31-268) */
  double min_prior;   /* replaces non-positive --priors (gibbs.cc:390-397); 0 => 0.01 */
  double high_temp, low_temp; /* --high-temp / --low-temp (gibbs_opts.hpp:50-53, 206-211): choices are made with
                                 probabilities raised to 1/temperature, the temperature running from high_temp at
                                 sweep 0 to low_temp at sweep `iter`; 0 => 1 (no annealing) */
  int expectation;    /* --expectation (gibbs_opts.hpp:125,166; derivations.h:381-398): instead of one sampled
                         derivation a block contributes the posterior of every lattice arc ("online EM"); mode 0
                         only; iter_logprob is then the ln probability of all derivations */
  uint32_t restarts;  /* --crp-restarts=N: N further runs from the priors, each with its own draws (the uniforms of run
                         r, sweep i are those of sweep r * (iter + 1) + i); the run that is best by gibbs_stats::better
                         (gibbs_opts.hpp:270-316: product of the sweep probabilities from burn-in on) gives the final
                         weights and sample (gibbs_base::run_starts, gibbs.hpp:880-914).  carmel_hip_gibbs_run then
                         writes (restarts + 1) * (iter + 1) values into iter_logprob / iter_cheap_logprob */
  int argmax_final, argmax_sum; /* --crp-argmax-final / --crp-argmax-sum: compare runs by their last sweep / by the sum */
  int include_self;   /* --include-self (gibbs_opts.hpp:40-41, 162; gibbs.hpp:851-870): a block's previous counts stay in the
                         counts while its proposal is formed and leave just before the new ones are added (with
                         --expectation: "incremental EM"); mode 1: no counterfactual subtraction of the block's own uses */
  int random_start;   /* --random-start (gibbs_opts.hpp:127-128, 167; gibbs.hpp:816, 860-864, 296-301): --expectation only --
                         the initial sweep's fractional counts are scaled entry by entry (one entry per lattice arc and chain
                         element, in the reference's listing order) by carmel_hip_gibbs_uniform(seed, sweep, block, entry)
                         and that sweep's probability is logged as 0.  Restart runs (run > 0) do this whatever the flag says,
                         as the reference does */
} carmel_hip_gibbs_opts;
int carmel_hip_gibbs_create(carmel_hip_gibbs** out, carmel_hip_trainer* t, const carmel_hip_gibbs_opts* opts);
int carmel_hip_gibbs_destroy(carmel_hip_gibbs* g);
uint32_t carmel_hip_gibbs_n_blocks(carmel_hip_gibbs* g);
/* derivations::global_stats of the sampler's cached lattices (cached_derivs.h:137): see carmel_hip_lattice_stats */
int carmel_hip_gibbs_lattice_stats(carmel_hip_gibbs* g, carmel_hip_lattice_stats* stats);
uint32_t carmel_hip_gibbs_max_sample(carmel_hip_gibbs* g);
/* Runs iter+1 sweeps; iter_logprob[i] = ln of the "cache-model prob" log line of sweep i (gibbs.hpp:712-742; mode
 * 1: the proposal prob), iter_cheap_logprob[i] = ln of the proposal ("--sample-prob") probability; either may be
 * NULL.  On return the trainer's parameters hold the time-averaged probabilities (probs_to_cascade,
 * gibbs.cc:66-76): read them with carmel_hip_get_weights. */
int carmel_hip_gibbs_run(carmel_hip_gibbs* g, double* iter_logprob, double* iter_cheap_logprob);
/* ... and a third series (exact mode only; may be NULL): ln of the product over blocks of the proposal probability of the
 * block's new sample evaluated AFTER that sample was added back to the counts -- the "overestimate" the comment at
 * gibbs.hpp:866 describes.  It is what the older carmel binary that recorded carmel-tutorial/commands.trace logged as
 * "sample prob" (its 6001 values for `--crp -M 6000` on the tagging cascade, trace lines 6989-12990, are this repo's only
 * reference-held datum for the sampler); today's carmel logs the cache-model probability instead. */
int carmel_hip_gibbs_run_ex(carmel_hip_gibbs* g, double* iter_logprob, double* iter_cheap_logprob, double* iter_after_logprob);
/* The runs of --crp-restarts as replicas (gibbs.hpp:880-914: every run starts from the priors, the uniforms of run r are
 * those of sweeps r * (iter + 1) ...: the runs are independent): a sampler with run share (first, stride) executes the runs
 * r with r % stride == first and keeps the best of them; carmel_hip_gibbs_best_run / _best_stats ({ln allprob, ln finalprob,
 * ln sumprob} of the kept run, gibbs_opts.hpp:270-316) let the caller pick the overall winner -- the better by
 * gibbs_stats::better, the earlier run on a tie, which is what the sequential loop keeps (carmel --crp --gpus=N). */
int carmel_hip_gibbs_set_run_share(carmel_hip_gibbs* g, uint32_t first, uint32_t stride);
int carmel_hip_gibbs_best_stats(carmel_hip_gibbs* g, double* out3, int* ran_any);

/* Replaces: prior-scale inference, gibbs_base::propose_new_priors (gibbs.hpp:404-553; carmel --prior-inference-stddev=s
 * [--prior-inference-global | -local] [--prior-inference-restart-fresh] [--prior-inference-start= --prior-inference-end=]
 * [--prior-groupby=012..]): after every inferring sweep (gibbs.hpp:559-563) each scale group's prior pseudo-counts are
 * proposed to be multiplied by a factor drawn from N(1, stddev) truncated to > 0, and the proposal is accepted with
 * probability p2 / p1 * q(old|new) / q(new|old), p1 / p2 = the cache-model probability of the whole current sample under
 * the old / new priors.  member_priorgroup[m]: 0 = never scaled, 1 = one scale for the whole member transducer (default),
 * 2 = one per norm group.  member_n_states[m] (may be null: the highest source state + 1): a JOINT member has one
 * norm group per state in the reference, arcs or not (fst.h:1362-1445), and every group draws a scale that enters the
 * acceptance ratio, so the count of states decides how many scales there are.  Exact mode only.  Call between create and run.  The uniforms are
 * carmel_hip_gibbs_uniform(seed, sweep, 0xfffffffe, k) for scale group k (1-based) and (seed, sweep, 0xffffffff, 0) for
 * the acceptance.  carmel_hip_gibbs_prior_trace: per sweep {proposed, accepted, ln p1, ln p2, a2, p_accept} and the
 * cumulative scale per scale group (--prior-inference-show). */
int carmel_hip_gibbs_set_prior_inference(carmel_hip_gibbs* g, double stddev, int global, int local, int restart_fresh,
                                         uint32_t start, uint32_t end, const int* member_priorgroup,
                                         const uint32_t* member_n_states, uint32_t n_members);
int carmel_hip_gibbs_prior_trace(carmel_hip_gibbs* g, double* out6, uint32_t n_sweeps, double* cumulative, uint32_t n_cumulative);
uint32_t carmel_hip_gibbs_n_prior_scales(carmel_hip_gibbs* g);

/* Replaces: gibbs_base::maybe_print_periodic (gibbs.hpp:959-968; carmel --print-every=N): after every sweep i of a run with
 * i % every == 0 the sampler calls fn(ctx, run, i, time) on the calling thread, the stream idle; inside the call
 * carmel_hip_gibbs_get_sample gives the sample as it stands and carmel_hip_gibbs_current_probs the proposal probability
 * (gibbs.hpp:163-170) of every parameter from the counts as they stand -- what print_path prints on the arcs
 * (gibbs.cc:272-286).  every = 0 or fn = NULL: none. */
typedef void (*carmel_hip_gibbs_observer_fn)(void* ctx, uint32_t run, uint32_t iter, double time);
int carmel_hip_gibbs_set_observer(carmel_hip_gibbs* g, uint32_t every, carmel_hip_gibbs_observer_fn fn, void* ctx);
int carmel_hip_gibbs_current_probs(carmel_hip_gibbs* g, double* prob);
/* --print-counts-* / --print-norms-* (gibbs.hpp:970-1078): per parameter (the trainer's parameter order) the count as it stands,
 * its time-weighted sum, the time it is summed up to (gibbs_param::sumcount: delta_sum.hpp) and the prior pseudo-count; any
 * pointer may be NULL.  Inside an observer call or after the run.  _final_counts: the kept run's counts as
 * finalize_cumulative_counts left them (gibbs.hpp:626-638). */
int carmel_hip_gibbs_get_state(carmel_hip_gibbs* g, double* x, double* sum, double* tmax, double* prior, double* last_touch);
int carmel_hip_gibbs_final_counts(carmel_hip_gibbs* g, double* x);
/* the current sample of one block: parameter ids in path order (sample[b].id, gibbs.hpp:285-338) */
int carmel_hip_gibbs_get_sample(carmel_hip_gibbs* g, uint32_t block, uint32_t* ids, uint32_t* n);
/* --init-em (gibbs.cc:386-430, 306-383 p_init): ln weights of the composed arcs (carmel_hip_get_arc_weights after an EM
 * run) from which the FIRST sweep of the first run draws its sample, in place of the proposal from the prior counts;
 * NULL clears them */
int carmel_hip_gibbs_set_init_weights(carmel_hip_gibbs* g, const double* arc_logw);
/* --crp-restarts: which run (0-based) was kept */
uint32_t carmel_hip_gibbs_best_run(carmel_hip_gibbs* g);
/* the uniform the sampler uses at (sweep, block, step of the walk): lets a checker replay the same choices */
double carmel_hip_gibbs_uniform(uint64_t seed, uint32_t iter, uint32_t block, uint32_t step);
/* the exponent 1/temperature sweep `sweep` of `iter` applies to every choice (gibbs.hpp:838-839 over
 * gibbs_opts::temperature(), gibbs_opts.hpp:206-211) */
double carmel_hip_gibbs_power(double high_temp, double low_temp, uint32_t iter, uint32_t sweep);

/* ---- forest-em: packed AND/OR derivation forests (forest-em/forest.hpp, forest-em.hpp) ----
 * Forests arrive as the reference's own node arrays: forest f owns nodes [node_off[f], node_off[f+1]) in preorder;
 * label = rule id (0 = OR), ref >= 0 marks a back-reference to an earlier node of the same forest (#k), next = one
 * past the node's subtree (ForestNode::next as an index relative to the forest).  Rule ids are 1-based; n_rules =
 * max id + 1.  Normalisation groups (`-n ((1 2 3) (5 8))`) as CSR over rule ids. */
typedef struct carmel_hip_forests carmel_hip_forests;
int carmel_hip_forests_create(carmel_hip_forests** out, int device, uint64_t n_forests, const uint64_t* node_off,
                              const uint32_t* label, const int32_t* ref, const uint32_t* next, uint32_t n_rules,
                              const double* rule_logw, uint64_t n_groups, const uint64_t* group_off,
                              const uint32_t* group_rule);
int carmel_hip_forests_destroy(carmel_hip_forests* f);
/* Replaces FForests::estimate (forest-em.hpp:561-578): inside (forest.hpp:636-697), normalised outside (:439-491)
 * and count[rule] += inside*norm_outside over AND nodes (:417-438) for every forest.  avg_logprob = mean ln
 * inside[root] over forests with non-zero probability; n_zero = the others. */
int carmel_hip_forests_estimate(carmel_hip_forests* f, double prior_count, double* avg_logprob, uint64_t* n_zero,
                                double* per_forest_logprob);
int carmel_hip_forests_get_counts(carmel_hip_forests* f, double prior_count, double* counts /* n_rules, linear */);
/* Replaces FForests::maximize (forest-em.hpp:626-655) -> NormalizeGroups::operator() (normalize.hpp:123-164) */
int carmel_hip_forests_maximize(carmel_hip_forests* f, double prior_count, double add_k, int zero_zerocounts,
                                double* max_delta);
int carmel_hip_forests_get_weights(carmel_hip_forests* f, double* rule_logw);
int carmel_hip_forests_set_weights(carmel_hip_forests* f, const double* rule_logw);
/* Replaces FForests::run_gibbs (forest-em.hpp:714-766) + gibbs_base::run (gibbs.hpp:803-877) + choose_random
 * (forest.hpp:725-758).  alpha = --alpha (gibbs_opts.hpp:229).  Rule weights become the time-averaged
 * probabilities (from_gibbs). */
int carmel_hip_forests_gibbs(carmel_hip_forests* f, const carmel_hip_gibbs_opts* opts, double alpha,
                             double* iter_logprob, double* iter_cheap_logprob);
/* opts->restarts > 0 (forest-em --crp-restarts; FForests::run_gibbs goes through gibbs_base::run_starts, forest-em.hpp:718,
 * gibbs.hpp:880-914, like carmel's sampler): restarts + 1 independent runs, each from the priors with the uniforms of its own
 * sweeps (run r, sweep i: those of sweep r * (iter + 1) + i); the run that is better by gibbs_stats::better (argmax_final /
 * argmax_sum choose the statistic) gives the weights and the sample.  The runs go SIDE BY SIDE on the device, one wavefront
 * each, up to 64 at a time; iter_logprob / iter_cheap_logprob then hold (restarts + 1) * (iter + 1) values, run-major.  For
 * the exact chain on the device only: mode 0, temperature 1, no locked parameter, no prior inference
 * (CARMEL_HIP_ERR_UNSUPPORTED otherwise).  carmel_hip_forests_best_run: which run (0-based) was kept. */
uint32_t carmel_hip_forests_best_run(carmel_hip_forests* f);
/* per rule id the kept run's count as finalize_cumulative_counts left it (gibbs.hpp:626-638): forest-em --print-counts-* */
int carmel_hip_forests_final_counts(carmel_hip_forests* f, double* x /* n_rules */);
/* Replaces: prior-scale inference in forest-em's sampler (forest-em.hpp:723-734 to_gibbs + gibbs.hpp:404-563; forest-em
 * --prior-inference-stddev / -global / -local / -start / -end): as carmel_hip_gibbs_set_prior_inference, with forest-em's
 * scale groups -- one per norm group, shifted by one as the reference registers them (the last norm group is never scaled,
 * the first factor scales nobody; --prior-inference-local scales every group).  Exact mode (mode 0) only.  Call before
 * carmel_hip_forests_gibbs; _prior_trace afterwards: per sweep {proposed, accepted, ln p1, ln p2, a2, p_accept}. */
int carmel_hip_forests_set_prior_inference(carmel_hip_forests* F, double stddev, int global, int local, uint32_t start,
                                           uint32_t end);
int carmel_hip_forests_prior_trace(carmel_hip_forests* F, double* out6, uint32_t n_sweeps, double* cumulative,
                                   uint32_t n_cumulative, uint32_t* n_scales);

/* --alpha=FILE of forest-em (gibbs_opts.hpp:98-99; forest-em.hpp:681-709): a prior strength per parameter, indexed by
 * rule id like the weights (entry 0 unused); a negative entry locks the parameter (it keeps its probability and leaves
 * its normalisation group's counts); rules beyond n use the scalar alpha.  NULL / n = 0 clears. */
int carmel_hip_forests_set_alphas(carmel_hip_forests* f, const double* alpha_per_rule, uint32_t n);
int carmel_hip_forests_get_sample(carmel_hip_forests* f, uint64_t forest, uint32_t* rules, uint32_t* n);
uint32_t carmel_hip_forests_max_sample(carmel_hip_forests* f);
/* Replaces FForest::compute_viterbi + write_viterbi (forest.hpp:507-632; forest-em -v / --outviterbi-file, FForests::
 * operator() forest-em.hpp:546-550): with the current weights, the max-product inside of every forest -- an OR node keeps
 * its first best child, a later child must be strictly better (forest.hpp:547) -- in best_logprob[n_forests] (natural log),
 * and the best derivation of each forest walked from its root.  carmel_hip_forests_get_viterbi then returns a forest's
 * derivation in pre-order as {rule id, number of children} per AND node -- what write_viterbi_rec prints as
 * "(rule child child)" / "rule" (at most carmel_hip_forests_max_sample entries). */
int carmel_hip_forests_viterbi(carmel_hip_forests* f, double* best_logprob);
int carmel_hip_forests_get_viterbi(carmel_hip_forests* f, uint64_t forest, uint32_t* rules, uint32_t* arity, uint32_t* n);

/* ---- composition on the GPU ----
 * Replaces: the product construction of WFST::set_compose with the default 3-state epsilon filter (compose.cc:163-498):
 * composite states (qa, qb, filter), each expanded in the reference's own emission order (which of its three code paths
 * runs depends on the operand states' sizes and carmel -T = index_threshold, compose.cc:330-498), arc weight wa * wb.
 * Operands are CSR (a_off[q] .. a_off[q + 1] are state q's arcs in list order, state 0 the start), a2b / b2a map A's
 * output symbol ids to B's input symbol ids and back (0xffffffff: no such symbol).  The result lists the composite
 * states in DISCOVERY order of the device's frontier expansion (state 0 = (0, 0, 0)) with their arcs in emission order;
 * arc_dst are those temporary state ids, arc_ka / arc_kb the position (within its state) of the A / B arc the composed
 * arc was built from (0xffffffff: none) -- what cascade_parameters::record / record1 / record2 are given
 * (cascade.h:507-599).  The reference's state NUMBERING (LIFO work list, compose.cc:193, 326-328) and chain ids are
 * sequential definitions: host/compose.hpp derives them from this output in one pass (Composer::run_device). */
typedef struct carmel_hip_composition carmel_hip_composition;
int carmel_hip_compose(carmel_hip_composition** out, int device, uint32_t a_states, const uint64_t* a_off, const uint32_t* a_in,
                       const uint32_t* a_out, const uint32_t* a_dst, const double* a_logw, uint32_t b_states,
                       const uint64_t* b_off, const uint32_t* b_in, const uint32_t* b_out, const uint32_t* b_dst,
                       const double* b_logw, const uint32_t* a2b, uint32_t n_a2b, const uint32_t* b2a, uint32_t n_b2a,
                       uint32_t index_threshold);
uint64_t carmel_hip_composition_states(carmel_hip_composition* c);
uint64_t carmel_hip_composition_arcs(carmel_hip_composition* c);
double carmel_hip_composition_seconds(carmel_hip_composition* c);
int carmel_hip_composition_export(carmel_hip_composition* c, uint64_t* state_off, uint32_t* state_qa, uint32_t* state_qb,
                                  uint8_t* state_filter, uint32_t* arc_in, uint32_t* arc_out, uint32_t* arc_dst,
                                  double* arc_logw, uint32_t* arc_ka, uint32_t* arc_kb);
int carmel_hip_composition_free(carmel_hip_composition* c);

/* ---- host-only inspection (no GPU needed): the lattice image carmel_hip_build_lattices uploads ----
 * Used by the CPU test-suite to check lattice construction and layout against the oracle. */
typedef struct carmel_hip_host_lattices carmel_hip_host_lattices;
int carmel_hip_host_build(carmel_hip_host_lattices** out, uint32_t n_states, uint32_t final_state, uint64_t n_arcs,
                          const uint32_t* src, const uint32_t* dst, const uint32_t* in_sym, const uint32_t* out_sym,
                          uint64_t n_pairs, const uint64_t* in_off, const uint32_t* cin, const uint64_t* out_off,
                          const uint32_t* cout, const double* pair_weight, int prune, int threads,
                          uint32_t small_pairs, uint32_t small_states, int lane_states);
void carmel_hip_host_dims(carmel_hip_host_lattices* h, uint64_t* dims19);
void carmel_hip_host_export_lanes(carmel_hip_host_lattices* h, void* groups32, uint32_t* fwd, uint32_t* bwd,
                                  uint32_t* lane_pair, uint32_t* lane_nstates, double* lane_logw, uint32_t* classes3);
void carmel_hip_host_export(carmel_hip_host_lattices* h, void* bundles64, uint32_t* in_arcs, uint32_t* out_arcs,
                            uint32_t* in_off, uint32_t* out_off, uint32_t* level_off, uint32_t* pair_start,
                            uint32_t* pair_final, uint32_t* pair_id, double* pair_logw, uint32_t* classes5,
                            uint8_t* has_deriv);
/* the blocked arc-order <-> lattice-order transposition tables the E-step uses instead of random gathers; dims6 =
 * n_items, n_buckets, n_tiles, n_split_arcs, n_post, n_arcs; null pointers are skipped */
void carmel_hip_host_transpose(carmel_hip_host_lattices* h, uint64_t* dims6, void* buckets24, uint64_t* tile_base,
                               uint16_t* b_arc, uint16_t* b_rank, uint32_t* b_src, uint16_t* t_pos, uint32_t* t_src,
                               uint32_t* split_arcs, uint64_t* arc_off, uint64_t* slot_pos);
/* the tile-sweep layout (csrc/lattice.hpp, LatticeSet::tile_sweep): info2 = positions per tile of the transposition, entries of
 * tile_group (0: the corpus is not laid out for the one-kernel tile sweep); tile_group (may be null) = first lane group of every
 * lane tile + the number of groups */
void carmel_hip_host_tile_sweep(carmel_hip_host_lattices* h, uint32_t* info2, uint32_t* tile_group);
/* the one-lattice-per-wavefront layout (csrc/lattice.hpp, WaveDesc); dims6 = n_waves, forward records, backward records,
 * level entries, n_classes, first wave slot; null pointers are skipped */
void carmel_hip_host_export_waves(carmel_hip_host_lattices* h, uint64_t* dims6, void* descs64, uint32_t* fwd, uint32_t* bwd,
                                  uint32_t* bwd_arc, uint32_t* level_off, uint32_t* frow, uint32_t* brow, uint32_t* classes4);
void carmel_hip_host_free(carmel_hip_host_lattices* h);

#ifdef __cplusplus
}
#endif
#endif
