// exchange.cpp — the per-iteration exchange of corpus-sharded EM: what travels between the GPUs, and when.
//
// The reference is single-process; the data-parallel form keeps the model on every GPU, gives each a shard of the training
// pairs, and has to make forward_backward::maximize (train.cc:893-923) see the counts of ALL pairs.  Two forms:
//
//  * plain: one all-reduce (sum) of counts[n_arcs + 4] between the count pass and a replicated M-step.  2 x 8 B x n_arcs
//    cross every GPU's links per iteration, nothing overlaps.  Used for cascades (the parameters are few), the unrolled /
//    dense layouts (the buffer holds per-parameter sums) and models whose normalisation groups are not local.
//
//  * sharded (a single transducer under JOINT / CONDITIONAL normalisation, i.e. norm groups inside a window of consecutive
//    arcs -- the 10^7-arc case the exchange matters for): the arc table is cut into K chunks [A_k, A_k+1), every chunk into
//    N equal pieces, rank r owns piece r of every chunk.
//      - counts out: trans_c_bucket produces counts arc range by arc range; as soon as the buckets below A_k+1 are summed
//        the chunk goes into a REDUCE-SCATTER on the communicator's own stream, while the next chunk is still being summed
//        (8 B x n_arcs x (N-1)/N per GPU -- half of what the all-reduce moves at this point);
//      - the M-step runs on this rank's pieces only (mstep_window_kernel over block ranges: 1/N of the work).  A norm group
//        may straddle a piece boundary: the few arcs within `norm_span` of a boundary, the arcs after the last whole
//        chunk and the four corpus scalars travel in ONE small all-reduce beside the reduce-scatters and are written back
//        over whatever the reduce-scatter left there, so every rank sums a straddling group from the same numbers;
//      - weights in: the pieces are ALL-GATHERed chunk by chunk (8 B x n_arcs x (N-1)/N per GPU), and the next iteration's
//        trans_w_bucket launches wait chunk by chunk: the weights of the first arc ranges are on their way into lattice
//        order while the last pieces are still on the links.
//    The largest weight change is a one-double all-reduce (max).
//
// Per-link arithmetic for 8 GPUs is in DESIGN.md section 5.
#include <algorithm>
#include <chrono>
#include <cstring>
#include "comm.hpp"

static const int XCH_MAX_CHUNKS = MSTEP_MAX_RANGES - 1;

struct ExchangePlan {
  carmel_hip_comm* comm = nullptr;
  bool sharded = false;
  uint32_t K = 0, N = 1, rank = 0;
  uint64_t n_arcs = 0;
  std::vector<uint64_t> A;           // K + 1 chunk boundaries, multiples of N * 256; A[K] <= n_arcs
  std::vector<uint32_t> cb_end;      // counts out: after buckets [0, cb_end[k]) every arc below A[k + 1] is summed
  std::vector<uint32_t> wb_end;      // weights in: buckets [0, wb_end[k]) read no arc at or above A[k + 1]
  uint32_t n_buckets = 0;
  DevBuf<uint32_t> halo_idx;         // arcs within norm_span of a piece boundary, then [A[K], n_arcs + 4)
  DevBuf<double> small;              // their values: the one small all-reduce
  std::vector<uint32_t> halo_end;    // halo entries [0, halo_end[k]) lie below A[k + 1] (gathered before chunk k's reduce-scatter)
  uint32_t n_small = 0;
  hipEvent_t ev_chunk[XCH_MAX_CHUNKS] = {}, ev_ag[XCH_MAX_CHUNKS] = {};
  hipEvent_t ev_tail = nullptr, ev_rs_done = nullptr, ev_m_done = nullptr, ev_ag_done = nullptr, ev_max = nullptr;
  hipEvent_t tx0 = nullptr, tx1 = nullptr;  // timing of the stand-alone exchange (carmel_hip_exchange_measure)
  bool counts_sharded = false;  // the count buffer holds reduced values on this rank's pieces (+ halo, tail) only
  bool counts_pending = false;  // reduce-scatters enqueued, the main stream has not yet waited for them
  bool ag_pending = false;      // all-gathers of the weights enqueued, the main stream has not yet waited for them
  unsigned long long* h_max = nullptr;  // pinned
  uint64_t bytes_rs = 0, bytes_ag = 0, bytes_small = 0;
};

static void plan_free(ExchangePlan* xp) {
  for (auto& e : xp->ev_chunk)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : xp->ev_ag)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {xp->ev_tail, xp->ev_rs_done, xp->ev_m_done, xp->ev_ag_done, xp->ev_max, xp->tx0, xp->tx1})
    if (e) (void)hipEventDestroy(e);
  if (xp->h_max) (void)hipHostFree(xp->h_max);
  delete xp;
}

bool exchange_is_sharded(const ExchangePlan* xp) { return xp && xp->sharded; }

void exchange_drop(carmel_hip_trainer* t) {
  if (!t->xplan) return;
  (void)hipSetDevice(t->device);
  if (carmel_hip_comm* c = t->xplan->comm) {  // (null: the communicator went first -- exchange_comm_gone)
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    c->planned.erase(std::remove(c->planned.begin(), c->planned.end(), t), c->planned.end());
  }
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  plan_free(t->xplan);
  t->xplan = nullptr;
}

void exchange_comm_gone(carmel_hip_comm* c) {
  const std::vector<carmel_hip_trainer*> ts = c->planned;  // (exchange_drop edits the list)
  for (carmel_hip_trainer* t : ts) exchange_drop(t);
  c->planned.clear();
}

// ---- weights in: trans_w_bucket chunk by chunk behind the all-gathers of the previous M-step ----
int exchange_weights_in(carmel_hip_trainer* t, ExchangePlan* xp, const TransArgs& T) {
  hipStream_t s = t->stream;
  uint32_t done = 0;
  for (uint32_t k = 0; k < xp->K; ++k) {
    if (xp->ag_pending) HIPCHK(hipStreamWaitEvent(s, xp->ev_ag[k], 0));
    if (xp->wb_end[k] > done) HIPCHK(launch_trans_w_bucket_range(T, done, xp->wb_end[k] - done, s));
    done = std::max(done, xp->wb_end[k]);
  }
  if (xp->ag_pending) HIPCHK(hipStreamWaitEvent(s, xp->ev_ag_done, 0));
  xp->ag_pending = false;
  if (xp->n_buckets > done) HIPCHK(launch_trans_w_bucket_range(T, done, xp->n_buckets - done, s));
  return CARMEL_HIP_OK;
}

// ---- counts out: trans_c_bucket chunk by chunk, each chunk into its reduce-scatter on the communicator's stream ----
int exchange_counts_out(carmel_hip_trainer* t, ExchangePlan* xp, const TransArgs& T) {
  hipStream_t s = t->stream, x = xp->comm->xstream;
  double* counts = t->counts_ptr();
  HIPCHK(launch_zero_list(counts, t->t_split_arcs.p, (uint32_t)t->t_split_arcs.n, s));
  // the trainer's stream first, all of it (enqueueing a collective costs the host several kernel launches' worth of time:
  // interleaved, the count pass would wait for the host), then the collectives on the communicator's stream, each behind
  // its chunk's event
  uint32_t done = 0, hdone = 0;
  for (uint32_t k = 0; k < xp->K; ++k) {
    if (xp->cb_end[k] > done) HIPCHK(launch_trans_c_bucket_range(T, done, xp->cb_end[k] - done, s));
    done = std::max(done, xp->cb_end[k]);
    // this rank's own (unreduced) values of the boundary arcs of the chunk, before the reduce-scatter overwrites any
    if (xp->halo_end[k] > hdone)
      HIPCHK(launch_gather_idx(xp->small.p + hdone, counts, xp->halo_idx.p + hdone, xp->halo_end[k] - hdone, s));
    hdone = std::max(hdone, xp->halo_end[k]);
    HIPCHK(hipEventRecord(xp->ev_chunk[k], s));
  }
  if (xp->n_buckets > done) HIPCHK(launch_trans_c_bucket_range(T, done, xp->n_buckets - done, s));
  for (uint32_t k = 0; k < xp->K; ++k) {
    HIPCHK(hipStreamWaitEvent(x, xp->ev_chunk[k], 0));
    int rc = comm_reduce_scatter(xp->comm, counts + xp->A[k], (size_t)((xp->A[k + 1] - xp->A[k]) / xp->N), x);
    if (rc) return rc;
  }
  xp->counts_pending = true;
  return CARMEL_HIP_OK;
}

// ... and, once the corpus scalars have joined the stream, the arcs after the last chunk + the scalars + the boundary
// arcs in one small all-reduce
int exchange_counts_tail(carmel_hip_trainer* t, ExchangePlan* xp) {
  hipStream_t s = t->stream, x = xp->comm->xstream;
  const uint32_t hdone = xp->K ? xp->halo_end[xp->K - 1] : 0u;
  if (xp->n_small > hdone)
    HIPCHK(launch_gather_idx(xp->small.p + hdone, t->counts_ptr(), xp->halo_idx.p + hdone, xp->n_small - hdone, s));
  HIPCHK(hipEventRecord(xp->ev_tail, s));
  HIPCHK(hipStreamWaitEvent(x, xp->ev_tail, 0));
  int rc = comm_allreduce(xp->comm, xp->small.p, xp->n_small, false, x);
  if (rc) return rc;
  HIPCHK(hipEventRecord(xp->ev_rs_done, x));
  return CARMEL_HIP_OK;
}

// the main stream takes delivery: waits for the reduce-scatters and writes the all-reduced boundary values in place
static int exchange_counts_arrive(carmel_hip_trainer* t, ExchangePlan* xp) {
  if (!xp->counts_pending) return CARMEL_HIP_OK;
  hipStream_t s = t->stream;
  HIPCHK(hipStreamWaitEvent(s, xp->ev_rs_done, 0));
  HIPCHK(launch_scatter_idx(t->counts_ptr(), xp->small.p, xp->halo_idx.p, xp->n_small, s));
  xp->counts_pending = false;
  xp->counts_sharded = xp->N > 1;
  return CARMEL_HIP_OK;
}

// make the trainer's buffers whole again for whoever reads them next: weights (wait for the all-gathers), and -- if
// need_counts -- the count vector (all-gather of the reduced pieces)
int exchange_settle(carmel_hip_trainer* t, bool need_counts) {
  ExchangePlan* xp = t->xplan;
  if (!xp || !xp->sharded) return CARMEL_HIP_OK;
  hipStream_t s = t->stream, x = xp->comm->xstream;
  if (xp->ag_pending) {
    HIPCHK(hipStreamWaitEvent(s, xp->ev_ag_done, 0));
    xp->ag_pending = false;
  }
  int rc = exchange_counts_arrive(t, xp);
  if (rc) return rc;
  if (need_counts && xp->counts_sharded) {
    HIPCHK(hipEventRecord(xp->ev_tail, s));
    HIPCHK(hipStreamWaitEvent(x, xp->ev_tail, 0));
    for (uint32_t k = 0; k < xp->K; ++k) {
      rc = comm_all_gather(xp->comm, t->counts_ptr() + xp->A[k], (size_t)((xp->A[k + 1] - xp->A[k]) / xp->N), x);
      if (rc) return rc;
    }
    HIPCHK(hipEventRecord(xp->ev_rs_done, x));
    HIPCHK(hipStreamWaitEvent(s, xp->ev_rs_done, 0));
    // (the boundary arcs were all-reduced in another order than the reduce-scatter: keep ONE value per arc everywhere)
    HIPCHK(launch_scatter_idx(t->counts_ptr(), xp->small.p, xp->halo_idx.p, xp->n_small, s));
    xp->counts_sharded = false;
  }
  return CARMEL_HIP_OK;
}

// ---- the sharded M-step + the all-gather of the weights ----
int exchange_maximize(carmel_hip_trainer* t, ExchangePlan* xp, double* max_change, int* handled) {
  *handled = 0;
  if (!xp->sharded) return CARMEL_HIP_OK;
  int rc = exchange_counts_arrive(t, xp);
  if (rc) return rc;
  if (!xp->counts_sharded && xp->N > 1) return CARMEL_HIP_OK;  // counts are whole (e.g. set by the caller): replicated M-step
  hipStream_t s = t->stream, x = xp->comm->xstream;
  MstepArgs M;
  rc = mstep_args(t, 1, 2, M);
  if (rc) return rc;
  // one launch over this rank's K pieces and the arcs after the last whole chunk (every rank has THEIR counts from the small
  // all-reduce and normalises them itself)
  M.n_ranges = 0;
  uint32_t cum = 0;
  auto add = [&](uint64_t first_block, uint64_t n_blocks) {
    if (!n_blocks) return;
    M.range_first[M.n_ranges] = (uint32_t)first_block;
    M.range_cum[M.n_ranges] = cum;
    cum += (uint32_t)n_blocks;
    ++M.n_ranges;
  };
  for (uint32_t k = 0; k < xp->K; ++k) {
    const uint64_t P = (xp->A[k + 1] - xp->A[k]) / xp->N, p0 = xp->A[k] + (uint64_t)xp->rank * P;
    add(p0 / 256, P / 256);
  }
  const uint64_t tail0 = xp->A[xp->K];
  if (xp->n_arcs > tail0) add(tail0 / 256, (xp->n_arcs - tail0 + 255) / 256);
  M.range_cum[M.n_ranges] = cum;
  HIPCHK(launch_mstep_window_range(M, 1, 0, cum, s));
  HIPCHK(launch_mstep_max_final(M, s));
  HIPCHK(hipEventRecord(xp->ev_m_done, s));
  HIPCHK(hipStreamWaitEvent(x, xp->ev_m_done, 0));
  rc = comm_allreduce(xp->comm, (double*)t->maxchg.p, 1, true, x);  // non-negative doubles: max of the values
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(xp->h_max, t->maxchg.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, x));
  HIPCHK(hipEventRecord(xp->ev_max, x));
  for (uint32_t k = 0; k < xp->K; ++k) {
    rc = comm_all_gather(xp->comm, t->arc_logw.p + xp->A[k], (size_t)((xp->A[k + 1] - xp->A[k]) / xp->N), x);
    if (rc) return rc;
    HIPCHK(hipEventRecord(xp->ev_ag[k], x));
  }
  HIPCHK(hipEventRecord(xp->ev_ag_done, x));
  xp->ag_pending = true;
  xp->counts_sharded = false;  // consumed
  t->em_valid = false;
  HIPCHK(hipEventSynchronize(xp->ev_max));
  double d;
  std::memcpy(&d, xp->h_max, sizeof d);
  if (max_change) *max_change = d;
  *handled = 1;
  return CARMEL_HIP_OK;
}

extern "C" {

int carmel_hip_exchange_plan(carmel_hip_trainer* t, carmel_hip_comm* c, uint32_t n_chunks, int force_allreduce) {
  if (!t || !c) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (t->device != c->device) return fail(CARMEL_HIP_ERR_ARG, "trainer and communicator live on different devices");
  if (!t->have_lattices) return fail(CARMEL_HIP_ERR_STATE, "build_lattices first");
  HIPCHK(hipSetDevice(t->device));
  exchange_drop(t);
  ExchangePlan* xp = new ExchangePlan();
  xp->comm = c;
  xp->N = (uint32_t)c->world;
  xp->rank = (uint32_t)c->rank;
  xp->n_arcs = t->w.n_arcs;
  // every rank must hold the same layout, or the buffers being summed mean different things (per-parameter sums of the
  // unrolled sweep against per-arc counts: round-2 advisor finding) -- and must take the same form of the exchange
  const uint32_t span = t->norm_span;
  bool can = !force_allreduce && !t->cascade && !t->unrolled && t->use_transpose && span > 0 && span <= 64 && !t->any_digamma &&
             !t->n_ties && t->have_norm && t->w.n_arcs >= (uint64_t)xp->N * 256 * 2 && !t->ext_counts &&
             !t->matrix;  // (--matrix-fb's E-step leaves before the count pass the reduce-scatters hang on: plain all-reduce)
  {
    double v[4] = {(double)carmel_hip_lattice_layout(t), -(double)carmel_hip_lattice_layout(t), can ? 1.0 : 0.0, can ? 0.0 : 1.0};
    int rc = carmel_hip_comm_allreduce_host(c, v, 4, 1);
    if (rc) {
      delete xp;
      return rc;
    }
    if (v[0] != -v[1]) {
      delete xp;
      return fail(CARMEL_HIP_ERR_STATE, "the ranks hold their lattices in different layouts (explicit / unrolled / dense): rebuild every "
                                        "rank with carmel_hip_set_layout_policy(t, 0) -- explicit lattices -- before planning the exchange");
    }
    can = v[2] == 1.0 && v[3] == 0.0;  // every rank can
  }
  xp->sharded = can;
  if (can) {
    const uint64_t M = t->w.n_arcs, gran = (uint64_t)xp->N * 256;
    uint32_t K = n_chunks ? n_chunks : 4u;
    K = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(K, XCH_MAX_CHUNKS), std::max<uint64_t>(1, M / gran));
    xp->K = K;
    xp->A.assign(K + 1, 0);
    for (uint32_t k = 1; k <= K; ++k) xp->A[k] = (M / gran) * k / K * gran;
    // bucket launch groups (the buckets tile the arc table in order; a hub arc's buckets share one arc)
    std::vector<TransBucket> B(t->t_buckets.n);  // (from the device: the GPU lattice builder leaves no host copy)
    HIPCHK(hipMemcpy(B.data(), t->t_buckets.p, B.size() * sizeof(TransBucket), hipMemcpyDeviceToHost));
    xp->n_buckets = (uint32_t)B.size();
    xp->cb_end.assign(K, 0);
    xp->wb_end.assign(K, 0);
    for (uint32_t k = 0; k < K; ++k) {
      uint32_t b = k ? xp->cb_end[k - 1] : 0u;
      while (b < B.size() && B[b].arc_lo < xp->A[k + 1]) ++b;  // touches an arc below A[k + 1]
      xp->cb_end[k] = b;
      uint32_t wbe = k ? xp->wb_end[k - 1] : 0u;
      while (wbe < B.size() && (uint64_t)B[wbe].arc_lo + B[wbe].n_arcs <= xp->A[k + 1]) ++wbe;  // reads nothing at or above A[k + 1]
      xp->wb_end[k] = wbe;
    }
    // the boundary arcs: within `span` of a piece boundary (both sides), chunk by chunk; then the tail and the scalars
    std::vector<uint32_t> idx;
    xp->halo_end.assign(K, 0);
    uint64_t last = 0;  // first index not yet listed
    auto add_range = [&](uint64_t lo, uint64_t hi) {  // [lo, hi) clipped, no duplicates (ranges arrive in ascending order)
      lo = std::max(lo, last);
      for (uint64_t a = lo; a < hi && a < M; ++a) idx.push_back((uint32_t)a);
      last = std::max(last, std::min(hi, M));
    };
    for (uint32_t k = 0; k < K; ++k) {
      // chunk k's own arcs near any of its N + 1 piece boundaries: they are read (this rank's unreduced values) before
      // chunk k's reduce-scatter may overwrite them.  The other side of the boundaries A[k] / A[k + 1] belongs to the
      // neighbouring chunk's list (or to the tail).
      const uint64_t P = (xp->A[k + 1] - xp->A[k]) / xp->N;
      for (uint32_t r = 0; r <= xp->N; ++r) {
        const uint64_t b = xp->A[k] + (uint64_t)r * P;
        add_range(std::max<uint64_t>(b > span ? b - span : 0, xp->A[k]), std::min<uint64_t>(b + span, xp->A[k + 1]));
      }
      xp->halo_end[k] = (uint32_t)idx.size();
    }
    add_range(xp->A[K], M);
    for (uint32_t q = 0; q < 4; ++q) idx.push_back((uint32_t)(M + q));
    xp->n_small = (uint32_t)idx.size();
    HIPCHK(xp->halo_idx.upload(idx, t->stream));
    HIPCHK(xp->small.alloc(idx.size()));
    HIPCHK(hipStreamSynchronize(t->stream));
    for (uint32_t k = 0; k < K; ++k) {
      HIPCHK(hipEventCreateWithFlags(&xp->ev_chunk[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&xp->ev_ag[k], hipEventDisableTiming));
    }
    for (hipEvent_t* e : {&xp->ev_tail, &xp->ev_rs_done, &xp->ev_m_done, &xp->ev_ag_done, &xp->ev_max})
      HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&xp->tx0));
    HIPCHK(hipEventCreate(&xp->tx1));
    HIPCHK(hipHostMalloc((void**)&xp->h_max, sizeof(unsigned long long), hipHostMallocDefault));
    const uint64_t per = (xp->A[K] / xp->N) * (xp->N - 1) * 8;  // what a rank sends (and receives) in one pass over its pieces
    xp->bytes_rs = per;
    xp->bytes_ag = per;
    xp->bytes_small = (uint64_t)xp->n_small * 8;
  } else {
    xp->bytes_small = (t->w.n_arcs + 4) * 8;
  }
  t->xplan = xp;
  c->planned.push_back(t);
  return CARMEL_HIP_OK;
}

int carmel_hip_exchange_info(carmel_hip_trainer* t, int* sharded, uint32_t* n_chunks, uint64_t* bytes_reduce_scatter,
                             uint64_t* bytes_all_gather, uint64_t* bytes_all_reduce) {
  if (!t || !t->xplan) return fail(CARMEL_HIP_ERR_STATE, "no exchange planned");
  const ExchangePlan* xp = t->xplan;
  if (sharded) *sharded = xp->sharded ? 1 : 0;
  if (n_chunks) *n_chunks = xp->K;
  if (bytes_reduce_scatter) *bytes_reduce_scatter = xp->bytes_rs;
  if (bytes_all_gather) *bytes_all_gather = xp->bytes_ag;
  if (bytes_all_reduce) *bytes_all_reduce = xp->bytes_small;
  return CARMEL_HIP_OK;
}

int carmel_hip_allreduce_counts(carmel_hip_trainer* t, carmel_hip_comm* c) {
  if (!t || !c) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (t->device != c->device) return fail(CARMEL_HIP_ERR_ARG, "trainer and communicator live on different devices");
  HIPCHK(hipSetDevice(t->device));
  if (t->xplan && t->xplan->sharded && t->xplan->comm == c) return exchange_counts_arrive(t, t->xplan);  // enqueued by the count pass
  // the unrolled cascade sweep keeps per-PARAMETER sums in the same buffer: its first u_n_slots entries are what counts
  // there (everything is a sum over pairs either way, so the reduction is the same plain sum); the scalars follow at
  // n_arcs.  Reducing the whole buffer keeps one collective per iteration.
  return comm_allreduce(c, t->counts_ptr(), t->w.n_arcs + 4, false, t->stream);
}

// The exchange of one iteration on its own -- every reduce-scatter, the small all-reduce, every all-gather, back to back
// on the communicator's stream with nothing to wait for --, timed with HIP events on that stream: what the exchange
// costs when none of it is hidden (bench.py: exchange_ms; the exposed part is the step time minus a step without it).
// The count buffer and the weights are restored afterwards.  Collective: every rank calls it.
int carmel_hip_exchange_measure(carmel_hip_trainer* t, uint32_t reps, double* ms_per_exchange) {
  if (!t || !ms_per_exchange) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (!t->xplan) return fail(CARMEL_HIP_ERR_STATE, "no exchange planned");
  ExchangePlan* xp = t->xplan;
  HIPCHK(hipSetDevice(t->device));
  int rc = exchange_settle(t, false);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(t->stream));
  hipStream_t x = xp->comm->xstream;
  if (!reps) reps = 5;
  const size_t n = t->w.n_arcs + 4;
  DevBuf<double> keep_c, keep_w;
  HIPCHK(keep_c.alloc(n));
  HIPCHK(keep_w.alloc(t->w.n_arcs));
  HIPCHK(hipMemcpyAsync(keep_c.p, t->counts_ptr(), n * 8, hipMemcpyDeviceToDevice, x));
  HIPCHK(hipMemcpyAsync(keep_w.p, t->arc_logw.p, t->w.n_arcs * 8, hipMemcpyDeviceToDevice, x));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  float total = 0;
  for (uint32_t r = 0; r <= reps; ++r) {  // (the first pass is a warm-up)
    HIPCHK(hipEventRecord(e0, x));
    if (xp->sharded) {
      for (uint32_t k = 0; k < xp->K && !rc; ++k)
        rc = comm_reduce_scatter(xp->comm, t->counts_ptr() + xp->A[k], (size_t)((xp->A[k + 1] - xp->A[k]) / xp->N), x);
      if (!rc) rc = comm_allreduce(xp->comm, xp->small.p, xp->n_small, false, x);
      if (!rc) rc = comm_allreduce(xp->comm, (double*)t->maxchg.p, 1, true, x);
      for (uint32_t k = 0; k < xp->K && !rc; ++k)
        rc = comm_all_gather(xp->comm, t->arc_logw.p + xp->A[k], (size_t)((xp->A[k + 1] - xp->A[k]) / xp->N), x);
    } else
      rc = comm_allreduce(xp->comm, t->counts_ptr(), n, false, x);
    if (rc) break;
    HIPCHK(hipEventRecord(e1, x));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (r) total += ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(t->counts_ptr(), keep_c.p, n * 8, hipMemcpyDeviceToDevice, x));
  HIPCHK(hipMemcpyAsync(t->arc_logw.p, keep_w.p, t->w.n_arcs * 8, hipMemcpyDeviceToDevice, x));
  HIPCHK(hipStreamSynchronize(x));
  *ms_per_exchange = total / reps;
  return CARMEL_HIP_OK;
}

int carmel_hip_exchange_clear(carmel_hip_trainer* t) {
  if (!t) return fail(CARMEL_HIP_ERR_ARG, "null trainer");
  if (t->xplan) {
    int rc = exchange_settle(t, true);
    if (rc) return rc;
    exchange_drop(t);
  }
  return CARMEL_HIP_OK;
}

}  // extern "C"
