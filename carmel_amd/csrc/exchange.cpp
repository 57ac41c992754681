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
//  * sharded, DIRECT (the same chunks and pieces; the default where the transport has point-to-point transfers): a ring
//    reduce-scatter over seven point-to-point xGMI links is bound by one link; here piece j goes STRAIGHT to rank j.
//      - counts out: as soon as a chunk is summed, every rank sends each peer that peer's piece of it -- extended by
//        `norm_span` arcs either side, so that a norm group straddling a piece boundary is whole on both sides -- in one
//        group of sends / receives; what arrives is added up in RANK ORDER (xchg_sum_kernel) into a buffer of its own
//        (`red`: the unreduced values stay where the later sends read them).  Both owners of a straddling group add the same
//        numbers in the same order: no small all-reduce.  The arcs after the last whole chunk and the four corpus scalars,
//        contiguous in the count buffer, go to everybody with the last chunk.
//      - the M-step on this rank's pieces reads `red`;
//      - weights in: piece by piece to every peer, one group per chunk; the local largest change rides with the first.
//    2 K groups per iteration, nothing small in between; with one rank nothing is enqueued at all.
//
// Per-link arithmetic for 8 GPUs is in DESIGN.md section 5.
#include <algorithm>
#include <chrono>
#include <cstring>
#include "comm.hpp"
#include "options.hpp"

static const int XCH_MAX_CHUNKS = MSTEP_MAX_RANGES - 1;

struct ExchangePlan {
  carmel_hip_comm* comm = nullptr;
  bool sharded = false;
  bool m_ready = false;
  bool direct = false;               // the sharded exchange's direct form (point-to-point groups) instead of the collectives
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
  // ---- direct form ----
  uint32_t span = 0;
  std::vector<uint32_t> cbx_end;     // after buckets [0, cbx_end[k]) every arc below A[k + 1] + span is summed
  uint64_t tail_lo = 0;              // [tail_lo, n_arcs + 4): the last span arcs of the chunks, the arcs after them, the scalars
  uint64_t stride = 0, tail_n = 0;   // staging: peer slot q' (q' = q below rank ? q : q - 1) at stage[q' * stride], its tail at
  DevBuf<double> red, stage, fin;    //   stage[(N - 1) * stride + q' * tail_n]; fin: the ranks' largest weight changes
  std::vector<std::vector<carmel_hip_p2p>> ops_x, ops_g;  // per chunk: counts out / weights in
  // ---- direct form, TOUCHED arcs only (round 6).  A rank's lattices lie on a part of the arc table -- an eighth of config 4's
  // corpus touches a quarter of its arcs -- and the rest of its count vector is zero in every iteration.  Which arcs a rank
  // touches is fixed with its lattices: the ranks tell each other ONCE, at plan time, which arcs of every piece they will send
  // (index lists), and per iteration a piece whose sender touches less than half of it travels as the VALUES of those arcs
  // alone, packed.  The owner adds, per arc and in rank order, the values of the ranks that sent one: the sum the dense form
  // makes, without its zeros -- the same bits (x + 0.0 = x).
  bool sparse = false;
  std::vector<uint64_t> s_off;       // per chunk: its first entry in s_idx / s_pack (K + 1)
  DevBuf<uint32_t> s_idx;            // the arcs this rank sends packed, chunk after chunk, peer after peer
  DevBuf<double> s_pack;             // their counts, gathered before the chunk's group
  DevBuf<uint32_t> r_pos;            // per chunk and sender slot: where, in the sender's packed piece, the value of arc
                                     // ext_lo(k, rank) + i lies (0xffffffff: the sender does not touch it); [K][N - 1][stride]
  std::vector<uint32_t> r_dense;     // per chunk: bit s = sender slot s sends its piece whole
  uint64_t bytes_rs_dense = 0;       // what the counts' way out would move without this
  std::vector<carmel_hip_p2p> ops_cg;                     // the reduced count pieces to everybody (exchange_settle)
  uint64_t piece(uint32_t k) const { return (A[k + 1] - A[k]) / N; }
  uint64_t ext_lo(uint32_t k, uint32_t j) const {
    const uint64_t b = A[k] + (uint64_t)j * piece(k);
    return b > span ? b - span : 0;
  }
  uint64_t ext_hi(uint32_t k, uint32_t j) const { return std::min<uint64_t>(A[k] + (uint64_t)(j + 1) * piece(k) + span, n_arcs); }
  const double* reduced(const carmel_hip_trainer* t) const;
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

// ---- direct form: the sums of what the peers sent ----
// out[i] = v_0[i] + v_1[i] + ... in rank order, v_rank = own (this rank's unreduced values), the others from their staging
// slots: the order every rank uses, so two ranks that both need an arc (a norm group across a piece boundary) hold the same bits
__global__ __launch_bounds__(256) void xchg_sum_kernel(double* __restrict__ out, const double* __restrict__ own,
                                                       const double* __restrict__ stage, uint64_t n, uint64_t stride, uint32_t N,
                                                       uint32_t rank) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    double v = 0.0;
    for (uint32_t q = 0; q < N; ++q) {
      const double x = q == rank ? own[i] : stage[(uint64_t)(q < rank ? q : q - 1) * stride + i];
      v = q ? v + x : x;
    }
    out[i] = v;
  }
}
// ... the same sums where some senders sent the values of their touched arcs only: sender slot s of chunk k is dense (bit s of
// `dense`: its piece lies in its staging slot as above) or packed (pos[s * stride + i] = where arc i's value lies in the slot, or none)
__global__ __launch_bounds__(256) void xchg_sum_sparse_kernel(double* __restrict__ out, const double* __restrict__ own,
                                                              const double* __restrict__ stage, const uint32_t* __restrict__ pos, uint64_t n,
                                                              uint64_t stride, uint32_t N, uint32_t rank, uint32_t dense) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    double v = 0.0;
    for (uint32_t q = 0; q < N; ++q) {
      double x;
      if (q == rank)
        x = own[i];
      else {
        const uint32_t sl = q < rank ? q : q - 1;
        if ((dense >> sl) & 1u)
          x = stage[(uint64_t)sl * stride + i];
        else {
          const uint32_t p = pos[(uint64_t)sl * stride + i];
          if (p == 0xffffffffu) continue;  // (the dense form adds this rank's zero here: v + 0.0 = v, and 0.0 if v is none yet)
          x = stage[(uint64_t)sl * stride + p];
        }
      }
      v = q ? v + x : x;
    }
    out[i] = v;
  }
}
__global__ __launch_bounds__(256) void xchg_pack_kernel(double* __restrict__ out, const double* __restrict__ counts, const uint32_t* __restrict__ idx,
                                                        uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = counts[idx[i]];
}
// plan time: pos[(uint32_t)rel[j]] = j for the n arcs a sender listed (their places relative to the receiver's extended piece, as doubles)
__global__ __launch_bounds__(256) void xchg_pos_kernel(uint32_t* __restrict__ pos, const double* __restrict__ rel, uint64_t n) {
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) pos[(uint32_t)rel[j]] = (uint32_t)j;
}
// the largest weight change over the ranks (non-negative doubles order like their bit patterns)
__global__ void xchg_max_kernel(unsigned long long* bits, const unsigned long long* fin, uint32_t N, uint32_t rank) {
  unsigned long long m = *bits;
  for (uint32_t q = 0; q < N; ++q)
    if (q != rank && fin[q] > m) m = fin[q];
  *bits = m;
}
static hipError_t launch_xchg_sum(double* out, const double* own, const double* stage, uint64_t n, uint64_t stride, uint32_t N,
                                  uint32_t rank, hipStream_t s) {
  if (!n) return hipSuccess;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(xchg_sum_kernel, dim3(grid), dim3(256), 0, s, out, own, stage, n, stride, N, rank);
  return hipGetLastError();
}
const double* ExchangePlan::reduced(const carmel_hip_trainer* t) const {
  return (direct && N > 1) ? red.p : const_cast<carmel_hip_trainer*>(t)->counts_ptr();
}

// chunk k's counts to their owners and the owners' sums, on the communicator's stream (which already waits for the chunk)
static int direct_counts_chunk(carmel_hip_trainer* t, ExchangePlan* xp, uint32_t k) {
  if (xp->N == 1) return CARMEL_HIP_OK;
  hipStream_t x = xp->comm->xstream;
  const double* counts = t->counts_ptr();
  if (xp->sparse && xp->s_off[k + 1] > xp->s_off[k]) {  // the touched arcs' counts, packed peer after peer
    const uint64_t n = xp->s_off[k + 1] - xp->s_off[k];
    hipLaunchKernelGGL(xchg_pack_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, x, xp->s_pack.p + xp->s_off[k], counts,
                       xp->s_idx.p + xp->s_off[k], n);
    HIPCHK(hipGetLastError());
  }
  int rc = comm_p2p(xp->comm, xp->ops_x[k].data(), (uint32_t)xp->ops_x[k].size(), x);
  if (rc) return rc;
  const uint64_t lo = xp->ext_lo(k, xp->rank), hi = xp->ext_hi(k, xp->rank);
  if (xp->sparse) {
    hipLaunchKernelGGL(xchg_sum_sparse_kernel, dim3((unsigned)std::min<uint64_t>((hi - lo + 255) / 256, 2048)), dim3(256), 0, x, xp->red.p + lo, counts + lo,
                       xp->stage.p, xp->r_pos.p + (uint64_t)k * (xp->N - 1) * xp->stride, hi - lo, xp->stride, xp->N, xp->rank, xp->r_dense[k]);
    HIPCHK(hipGetLastError());
  } else
    HIPCHK(launch_xchg_sum(xp->red.p + lo, counts + lo, xp->stage.p, hi - lo, xp->stride, xp->N, xp->rank, x));
  if (k + 1 == xp->K) {  // the tail and the scalars: everybody's, summed by everybody
    HIPCHK(launch_xchg_sum(xp->red.p + xp->tail_lo, counts + xp->tail_lo, xp->stage.p + (uint64_t)(xp->N - 1) * xp->stride, xp->tail_n,
                           xp->tail_n, xp->N, xp->rank, x));
    HIPCHK(hipMemcpyAsync(t->counts_ptr() + xp->n_arcs, xp->red.p + xp->n_arcs, 4 * sizeof(double), hipMemcpyDeviceToDevice, x));
  }
  return CARMEL_HIP_OK;
}

// ---- weights in: trans_w_bucket chunk by chunk behind the all-gathers of the previous M-step ----
// (need_x false: every sweep fetches its weights from the table itself -- the trainer's stream waits for the chunks and no bucket
// pass runs)
int exchange_weights_in(carmel_hip_trainer* t, ExchangePlan* xp, const TransArgs& T, bool need_x) {
  hipStream_t s = t->stream;
  uint32_t done = 0;
  for (uint32_t k = 0; k < xp->K; ++k) {
    // (a chunk that has already arrived needs no barrier packet in the queue: small models, one rank)
    if (xp->ag_pending && hipEventQuery(xp->ev_ag[k]) != hipSuccess) HIPCHK(hipStreamWaitEvent(s, xp->ev_ag[k], 0));
    if (need_x && xp->wb_end[k] > done) HIPCHK(launch_trans_w_bucket_range(T, done, xp->wb_end[k] - done, s));
    done = std::max(done, xp->wb_end[k]);
  }
  if (xp->ag_pending && hipEventQuery(xp->ev_ag_done) != hipSuccess) HIPCHK(hipStreamWaitEvent(s, xp->ev_ag_done, 0));
  xp->ag_pending = false;
  if (need_x && xp->n_buckets > done) HIPCHK(launch_trans_w_bucket_range(T, done, xp->n_buckets - done, s));
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
  if (xp->direct) {
    for (uint32_t k = 0; k < xp->K; ++k) {
      if (xp->cbx_end[k] > done) HIPCHK(launch_trans_c_bucket_range(T, done, xp->cbx_end[k] - done, s));
      done = std::max(done, xp->cbx_end[k]);
      if (k + 1 < xp->K) HIPCHK(hipEventRecord(xp->ev_chunk[k], s));
    }
    if (xp->n_buckets > done) HIPCHK(launch_trans_c_bucket_range(T, done, xp->n_buckets - done, s));
    for (uint32_t k = 0; k + 1 < xp->K; ++k) {  // (the last chunk carries the scalars: exchange_counts_tail)
      HIPCHK(hipStreamWaitEvent(x, xp->ev_chunk[k], 0));
      int rc = direct_counts_chunk(t, xp, k);
      if (rc) return rc;
    }
    xp->counts_pending = true;
    return CARMEL_HIP_OK;
  }
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
  if (xp->direct) {
    HIPCHK(hipEventRecord(xp->ev_tail, s));
    HIPCHK(hipStreamWaitEvent(x, xp->ev_tail, 0));
    int rc = direct_counts_chunk(t, xp, xp->K - 1);
    if (rc) return rc;
    HIPCHK(hipEventRecord(xp->ev_rs_done, x));
    return CARMEL_HIP_OK;
  }
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
  if (!xp->direct) HIPCHK(launch_scatter_idx(t->counts_ptr(), xp->small.p, xp->halo_idx.p, xp->n_small, s));
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
    if (xp->direct) {  // the owners' sums to everybody (one value per arc everywhere), the tail's from this rank's own sum
      rc = comm_p2p(xp->comm, xp->ops_cg.data(), (uint32_t)xp->ops_cg.size(), x);
      if (rc) return rc;
      for (uint32_t k = 0; k < xp->K; ++k) {
        const uint64_t p0 = xp->A[k] + (uint64_t)xp->rank * xp->piece(k);
        HIPCHK(hipMemcpyAsync(t->counts_ptr() + p0, xp->red.p + p0, xp->piece(k) * sizeof(double), hipMemcpyDeviceToDevice, x));
      }
      if (xp->n_arcs > xp->A[xp->K])
        HIPCHK(hipMemcpyAsync(t->counts_ptr() + xp->A[xp->K], xp->red.p + xp->A[xp->K], (xp->n_arcs - xp->A[xp->K]) * sizeof(double),
                              hipMemcpyDeviceToDevice, x));
      HIPCHK(hipEventRecord(xp->ev_rs_done, x));
      HIPCHK(hipStreamWaitEvent(s, xp->ev_rs_done, 0));
      xp->counts_sharded = false;
      return CARMEL_HIP_OK;
    }
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
  bool direct_mailbox = false;
  MstepArgs M;
  rc = mstep_args(t, 1, 2, M);
  if (rc) return rc;
  M.counts = xp->reduced(t);
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
  if (xp->direct) {
    // the M-step on the communicator's stream, behind the sums it reads and ahead of the groups that carry its weights: one
    // hop between the streams per iteration (count pass -> exchange) instead of three
    // mstep_args may leave work on the trainer's stream on ANY call -- a snapshot copy (lw_src), first-time scratch, max_partial
    // re-allocated and cleared after a new set_norm / set_prior (mstep_stream_work: round-5 advisor) --: the M-step waits for it
    // whenever there is some, and only then (an event pair every iteration cost 0.03 ms of the one-rank loopback)
    if (M.lw_src || !xp->m_ready || t->mstep_stream_work) {
      HIPCHK(hipEventRecord(xp->ev_m_done, s));
      HIPCHK(hipStreamWaitEvent(x, xp->ev_m_done, 0));
      xp->m_ready = true;
      t->mstep_stream_work = false;
    }
    HIPCHK(launch_mstep_window_range(M, 1, 0, cum, x));
    HIPCHK(launch_mstep_max_final(M, x));
  } else {
    HIPCHK(launch_mstep_window_range(M, 1, 0, cum, s));
    HIPCHK(launch_mstep_max_final(M, s));
    HIPCHK(hipEventRecord(xp->ev_m_done, s));
    HIPCHK(hipStreamWaitEvent(x, xp->ev_m_done, 0));
  }
  if (xp->direct) {
    // the pieces straight to every peer, one group per chunk; the first carries this rank's largest change to everybody,
    // so the host has max |dw| while the other chunks are still on the links
    for (uint32_t k = 0; k < xp->K; ++k) {
      rc = comm_p2p(xp->comm, xp->ops_g[k].data(), (uint32_t)xp->ops_g[k].size(), x);
      if (rc) return rc;
      if (k == 0) {
        if (xp->N > 1) hipLaunchKernelGGL(xchg_max_kernel, dim3(1), dim3(1), 0, x, t->maxchg.p, (const unsigned long long*)xp->fin.p, xp->N, xp->rank);
        // to the host through the trainer's mailbox (with one rank the M-step's own last kernel has already put it there)
        if (xp->N > 1) {
          int prc = publish_u64(t, t->maxchg.p, x);
          if (prc) return prc;
        }
        direct_mailbox = true;
      }
      HIPCHK(hipEventRecord(xp->ev_ag[k], x));
    }
  } else {
    rc = comm_allreduce(xp->comm, (double*)t->maxchg.p, 1, true, x);  // non-negative doubles: max of the values
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(xp->h_max, t->maxchg.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, x));
    HIPCHK(hipEventRecord(xp->ev_max, x));
    for (uint32_t k = 0; k < xp->K; ++k) {
      rc = comm_all_gather(xp->comm, t->arc_logw.p + xp->A[k], (size_t)((xp->A[k + 1] - xp->A[k]) / xp->N), x);
      if (rc) return rc;
      HIPCHK(hipEventRecord(xp->ev_ag[k], x));
    }
  }
  HIPCHK(hipEventRecord(xp->ev_ag_done, x));
  xp->ag_pending = true;
  xp->counts_sharded = false;  // consumed
  t->em_valid = false;
  double d;
  if (direct_mailbox) {  // (the host has the value while the later chunks are still on the links)
    unsigned long long bits = 0;
    rc = fetch_u64(t, t->maxchg.p, &bits, x, true);
    if (rc) return rc;
    std::memcpy(&d, &bits, sizeof d);
  } else {
    HIPCHK(hipEventSynchronize(xp->ev_max));
    std::memcpy(&d, xp->h_max, sizeof d);
  }
  if (max_change) *max_change = d;
  *handled = 1;
  return CARMEL_HIP_OK;
}

// ---- the direct form's counts as touched arcs only (ExchangePlan::sparse); collective ----
// uses[a] > 0: some item of this rank's lattices lies on arc a -- the count pass over posteriors of 1 (every layout ends in
// trans_c_bucket reading XC, one entry per item)
static int arc_uses(carmel_hip_trainer* t, std::vector<double>& uses) {
  hipStream_t s = t->stream;
  HIPCHK(hipStreamSynchronize(s));
  TransArgs T;
  trans_args(t, T);
  DevBuf<double> tmp;
  HIPCHK(tmp.alloc(t->w.n_arcs + 4));
  HIPCHK(hipMemsetAsync(tmp.p, 0, tmp.bytes(), s));
  HIPCHK(launch_fill(t->t_xc.p, 1.0, t->t_xc.n, s));
  T.counts = tmp.p;
  if (T.n_buckets) HIPCHK(launch_trans_c_bucket_range(T, 0, T.n_buckets, s));
  uses.resize(t->w.n_arcs);
  HIPCHK(hipMemcpyAsync(uses.data(), tmp.p, uses.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CARMEL_HIP_OK;
}
static int plan_sparse(carmel_hip_trainer* t, ExchangePlan* xp) {
  const uint32_t N = xp->N, K = xp->K, me = xp->rank;
  carmel_hip_comm* c = xp->comm;
  hipStream_t x = c->xstream;
  std::vector<double> uses;
  int rc = arc_uses(t, uses);
  if (rc) return rc;
  // what I would send packed: per chunk and peer, the touched arcs of the peer's extended piece
  std::vector<std::vector<std::vector<uint32_t>>> mine(K, std::vector<std::vector<uint32_t>>(N));
  std::vector<double> cnt((size_t)N * K * N, 0.0);  // [sender][chunk][receiver]
  for (uint32_t k = 0; k < K; ++k)
    for (uint32_t q = 0; q < N; ++q) {
      if (q == me) continue;
      for (uint64_t a = xp->ext_lo(k, q), hi = xp->ext_hi(k, q); a < hi; ++a)
        if (uses[a] > 0.0) mine[k][q].push_back((uint32_t)a);
      cnt[((size_t)me * K + k) * N + q] = (double)mine[k][q].size();
    }
  rc = carmel_hip_comm_allreduce_host(c, cnt.data(), (uint32_t)cnt.size(), 0);
  if (rc) return rc;
  auto n_of = [&](uint32_t sender, uint32_t k, uint32_t receiver) { return (uint64_t)cnt[((size_t)sender * K + k) * N + receiver]; };
  auto dense = [&](uint32_t sender, uint32_t k, uint32_t receiver) {  // (both ends decide from the same numbers)
    return 2 * n_of(sender, k, receiver) > xp->ext_hi(k, receiver) - xp->ext_lo(k, receiver);
  };
  // my packed sends, chunk after chunk
  xp->s_off.assign(K + 1, 0);
  std::vector<uint32_t> idx;
  std::vector<std::vector<uint64_t>> off_kq(K, std::vector<uint64_t>(N, 0));
  for (uint32_t k = 0; k < K; ++k) {
    xp->s_off[k] = idx.size();
    for (uint32_t q = 0; q < N; ++q) {
      if (q == me || dense(me, k, q)) continue;
      off_kq[k][q] = idx.size();
      idx.insert(idx.end(), mine[k][q].begin(), mine[k][q].end());
    }
  }
  xp->s_off[K] = idx.size();
  if (idx.empty()) idx.push_back(0u);
  HIPCHK(xp->s_idx.upload(idx, x));
  HIPCHK(xp->s_pack.alloc(idx.size()));
  HIPCHK(xp->r_pos.alloc((uint64_t)K * (N - 1) * xp->stride));
  HIPCHK(hipMemsetAsync(xp->r_pos.p, 0xff, xp->r_pos.bytes(), x));
  xp->r_dense.assign(K, 0u);
  // the index lists travel once, as doubles (places relative to the receiver's extended piece), through the same groups
  for (uint32_t k = 0; k < K; ++k) {
    std::vector<double> out_rel;
    std::vector<uint64_t> out_at(N, 0), in_at(N, 0);
    uint64_t in_total = 0;
    for (uint32_t q = 0; q < N; ++q) {
      if (q == me) continue;
      if (!dense(me, k, q)) {
        out_at[q] = out_rel.size();
        const uint64_t lo = xp->ext_lo(k, q);
        for (uint32_t a : mine[k][q]) out_rel.push_back((double)(a - lo));
      }
      if (dense(q, k, me))
        xp->r_dense[k] |= 1u << (q < me ? q : q - 1);
      else {
        in_at[q] = in_total;
        in_total += n_of(q, k, me);
      }
    }
    DevBuf<double> d_out, d_in;
    if (out_rel.empty()) out_rel.push_back(0.0);
    HIPCHK(d_out.upload(out_rel, x));
    HIPCHK(d_in.alloc(std::max<uint64_t>(in_total, 1)));
    std::vector<carmel_hip_p2p> ops;
    for (uint32_t q = 0; q < N; ++q) {
      if (q == me) continue;
      if (!dense(me, k, q) && n_of(me, k, q)) ops.push_back({(int32_t)q, 1, d_out.p + out_at[q], n_of(me, k, q)});
      if (!dense(q, k, me) && n_of(q, k, me)) ops.push_back({(int32_t)q, 0, d_in.p + in_at[q], n_of(q, k, me)});
    }
    rc = comm_p2p(c, ops.data(), (uint32_t)ops.size(), x);
    if (rc) return rc;
    for (uint32_t q = 0; q < N; ++q) {
      if (q == me || dense(q, k, me) || !n_of(q, k, me)) continue;
      const uint32_t sl = q < me ? q : q - 1;
      const uint64_t n = n_of(q, k, me);
      hipLaunchKernelGGL(xchg_pos_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1024)), dim3(256), 0, x,
                         xp->r_pos.p + ((uint64_t)k * (N - 1) + sl) * xp->stride, d_in.p + in_at[q], n);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(x));
  }
  // the groups of an iteration: packed pieces where the sender is sparse (ops_x was built dense, in the order send / receive [/
  // tail send / tail receive] per peer)
  xp->bytes_rs = 0;
  for (uint32_t k = 0; k < K; ++k) {
    size_t o = 0;
    for (uint32_t q = 0; q < N; ++q) {
      if (q == me) continue;
      carmel_hip_p2p& snd = xp->ops_x[k][o];
      carmel_hip_p2p& rcv = xp->ops_x[k][o + 1];
      if (!dense(me, k, q)) {
        snd.dev_buf = xp->s_pack.p + off_kq[k][q];
        snd.n = n_of(me, k, q);
      }
      if (!dense(q, k, me)) rcv.n = n_of(q, k, me);
      xp->bytes_rs += snd.n * 8;
      o += 2;
      if (k + 1 == K) {
        xp->bytes_rs += xp->tail_n * 8;
        o += 2;
      }
    }
    // (an empty transfer is no transfer: both ends know)
    auto& v = xp->ops_x[k];
    v.erase(std::remove_if(v.begin(), v.end(), [](const carmel_hip_p2p& p) { return p.n == 0; }), v.end());
  }
  xp->sparse = true;
  return CARMEL_HIP_OK;
}

extern "C" {

int carmel_hip_exchange_plan(carmel_hip_trainer* t, carmel_hip_comm* c, uint32_t n_chunks, int form) {
  if (form < 0 || form > 3) return fail(CARMEL_HIP_ERR_ARG, "carmel_hip_exchange_plan: form is 0 (choose), 1 (all-reduce), 2 (collectives) or 3 (direct)");
  if (!t || !c) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (form == 3 && !comm_has_p2p(c))
    return fail(CARMEL_HIP_ERR_UNSUPPORTED, "carmel_hip_exchange_plan: the transport has no point-to-point transfers (carmel_hip_comm_set_sendrecv)");
  const bool force_allreduce = form == 1;
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
  xp->direct = can && form != 2 && comm_has_p2p(c);
  xp->span = span;
  if (can) {
    const uint64_t M = t->w.n_arcs, gran = (uint64_t)xp->N * 256;
    // chunks: every extra chunk costs the two bucket passes a ramp and a tail (measured on config 4, one rank: +26 us at 2
    // chunks, +58 at 3, +97 at 4, +170 at 8) and hides (K - 1) / K of them behind the links: 2 for the
    // direct form, whose transfers are short; the ring collectives keep their 4
    uint32_t K = n_chunks ? n_chunks : (xp->direct ? 2u : 4u);
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
    if (xp->direct) {
      xp->cbx_end.assign(K, 0);
      for (uint32_t k = 0; k < K; ++k) {
        uint32_t b = k ? xp->cbx_end[k - 1] : 0u;
        while (b < B.size() && B[b].arc_lo < xp->A[k + 1] + span) ++b;  // touches an arc a piece of chunk k is sent with
        xp->cbx_end[k] = b;
      }
      xp->tail_lo = xp->A[K] > span ? xp->A[K] - span : 0;
      xp->tail_n = M + 4 - xp->tail_lo;
      const uint32_t N = xp->N, me = xp->rank;
      for (uint32_t k = 0; k < K; ++k) xp->stride = std::max(xp->stride, xp->ext_hi(k, me) - xp->ext_lo(k, me));
      if (N > 1) {
        HIPCHK(xp->red.alloc(M + 4));
        HIPCHK(xp->stage.alloc((uint64_t)(N - 1) * (xp->stride + xp->tail_n)));
        HIPCHK(xp->fin.alloc(N));
      }
      double* counts = t->counts_ptr();
      xp->ops_x.assign(K, {});
      xp->ops_g.assign(K, {});
      for (uint32_t k = 0; k < K; ++k) {
        const uint64_t P = xp->piece(k);
        for (uint32_t q = 0; q < N; ++q) {
          if (q == me) continue;
          const uint32_t slot = q < me ? q : q - 1;
          // counts out: the peer's piece of the chunk with its halo; mine (with my halo) from the peer
          xp->ops_x[k].push_back({(int32_t)q, 1, counts + xp->ext_lo(k, q), xp->ext_hi(k, q) - xp->ext_lo(k, q)});
          xp->ops_x[k].push_back({(int32_t)q, 0, xp->stage.p + (uint64_t)slot * xp->stride, xp->ext_hi(k, me) - xp->ext_lo(k, me)});
          xp->bytes_rs += (xp->ext_hi(k, q) - xp->ext_lo(k, q)) * 8;
          if (k + 1 == K) {
            xp->ops_x[k].push_back({(int32_t)q, 1, counts + xp->tail_lo, xp->tail_n});
            xp->ops_x[k].push_back({(int32_t)q, 0, xp->stage.p + (uint64_t)(N - 1) * xp->stride + (uint64_t)slot * xp->tail_n, xp->tail_n});
            xp->bytes_rs += xp->tail_n * 8;
          }
          // weights in: my piece to the peer, the peer's piece into place
          xp->ops_g[k].push_back({(int32_t)q, 1, t->arc_logw.p + xp->A[k] + (uint64_t)me * P, P});
          xp->ops_g[k].push_back({(int32_t)q, 0, t->arc_logw.p + xp->A[k] + (uint64_t)q * P, P});
          xp->bytes_ag += P * 8;
          if (k == 0) {
            xp->ops_g[k].push_back({(int32_t)q, 1, (double*)t->maxchg.p, 1});
            xp->ops_g[k].push_back({(int32_t)q, 0, xp->fin.p + q, 1});
            xp->bytes_ag += 8;
          }
          // the reduced count pieces to everybody, when the whole vector is asked for
          xp->ops_cg.push_back({(int32_t)q, 1, xp->red.p + xp->A[k] + (uint64_t)me * P, P});
          xp->ops_cg.push_back({(int32_t)q, 0, counts + xp->A[k] + (uint64_t)q * P, P});
        }
      }
      xp->bytes_rs_dense = xp->bytes_rs;
      if (N > 1 && N <= 32 && !lib_opt_off("exchange_sparse")) {
        int src = plan_sparse(t, xp);
        if (src) {
          plan_free(xp);
          return src;
        }
      }
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
    // the events that order this device's two streams release to the device only (a system-scope release writes the caches
    // back: measured, 15 us per chunk boundary); ev_max is the one the host reads pinned memory behind
    const unsigned dev_only = hipEventDisableTiming | hipEventReleaseToDevice;
    for (uint32_t k = 0; k < K; ++k) {
      HIPCHK(hipEventCreateWithFlags(&xp->ev_chunk[k], dev_only));
      HIPCHK(hipEventCreateWithFlags(&xp->ev_ag[k], dev_only));
    }
    for (hipEvent_t* e : {&xp->ev_tail, &xp->ev_rs_done, &xp->ev_m_done, &xp->ev_ag_done})
      HIPCHK(hipEventCreateWithFlags(e, dev_only));
    HIPCHK(hipEventCreateWithFlags(&xp->ev_max, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&xp->tx0));
    HIPCHK(hipEventCreate(&xp->tx1));
    HIPCHK(hipHostMalloc((void**)&xp->h_max, sizeof(unsigned long long), hipHostMallocDefault));
    if (!xp->direct) {
      const uint64_t per = (xp->A[K] / xp->N) * (xp->N - 1) * 8;  // what a rank sends (and receives) in one pass over its pieces
      xp->bytes_rs = per;
      xp->bytes_ag = per;
      xp->bytes_small = (uint64_t)xp->n_small * 8;
    }
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
  if (sharded) *sharded = xp->sharded ? (xp->direct ? 2 : 1) : 0;
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
  {
    int jrc = scalars_join(t);  // (the scalars travel with the counts)
    if (jrc) return jrc;
  }
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
    if (xp->sharded && xp->direct) {
      for (uint32_t k = 0; k < xp->K && !rc; ++k) rc = comm_p2p(xp->comm, xp->ops_x[k].data(), (uint32_t)xp->ops_x[k].size(), x);
      for (uint32_t k = 0; k < xp->K && !rc; ++k) rc = comm_p2p(xp->comm, xp->ops_g[k].data(), (uint32_t)xp->ops_g[k].size(), x);
    } else if (xp->sharded) {
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
