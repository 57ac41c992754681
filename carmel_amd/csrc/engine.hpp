// engine.hpp — trainer state shared by engine.cpp (EM entry points) and gibbs.hip (sampler entry points).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/carmel_hip.h"
#include "options.hpp"
#include "kernels.hpp"
#include "unrolled_args.hpp"

namespace carmel_hip {
int fail(int code, const std::string& msg);  // records carmel_hip_last_error() text, returns code
}
using carmel_hip::fail;
using namespace carmel_hip;

#define HIPCHK(x)                                                                                             \
  do {                                                                                                        \
    hipError_t e_ = (x);                                                                                      \
    if (e_ != hipSuccess)                                                                                     \
      return fail(CARMEL_HIP_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                        \
  } while (0)

// Every DevBuf is over-allocated by this many bytes: tile_sweep_kernel (and the fused lane sweep) request a tile's stretch of
// X / t_pos / t_src as whole rounds of loads without clamping the last tile's to its items -- what lies behind them is read
// and ignored, not faulted on.  trans_args() hands the figure to the launchers (TransArgs::slack_bytes), which refuse a
// buffer set that does not promise it.
static const size_t DEVBUF_SLACK = 131072;
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (!count) return hipSuccess;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T) + DEVBUF_SLACK);
    // CARMEL_HIP_POISON=1 (debugging): fresh device memory is whatever the last process left there, usually zeros on an idle box;
    // a pattern of 0xff bytes (a NaN to every double, an out-of-range index to every integer) makes a read of something never
    // written show up on every run instead of on the one after somebody else's job
    const bool poison = lib_opt("poison") && atoi(lib_opt("poison"));
    if (poison && e == hipSuccess) {
      e = hipMemset(p, 0xff, count * sizeof(T) + DEVBUF_SLACK);
      if (e == hipSuccess) e = hipDeviceSynchronize();  // (before anything on another stream writes there)
    }
    return e;
  }
  hipError_t upload(const std::vector<T>& v, hipStream_t s) {
    hipError_t e = alloc(v.size());
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
  }
  size_t bytes() const { return n * sizeof(T); }
};

struct ExchangePlan;  // exchange.cpp
struct carmel_hip_trainer {
  ExchangePlan* xplan = nullptr;  // the count exchange of corpus-sharded EM, when a communicator was planned in
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t side = nullptr;          // small independent kernels of the E-step run beside the main chain
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // the chunks of a lane class flow through weights-in / sweep / posteriors-out side by side on these streams
  static const int N_CHUNK_STREAMS = 4;
  hipStream_t cstream[N_CHUNK_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t cev[N_CHUNK_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> ev_piece;  // fused-lane layout: "the weights of lane piece k are in lattice order" (created on demand)
  hipEvent_t ev_w = nullptr;
  hipStream_t bstream = nullptr;  // the bundle sweeps (they gather their weights themselves) run beside the lane pieces
  hipEvent_t ev_b0 = nullptr, ev_b1 = nullptr;
  HostWfst w;
  HostCorpus corpus;
  bool have_corpus = false, have_lattices = false, cascade = false;
  LatticeSet lat;  // host copy of descriptors (classes); bulk arrays are freed after upload
  // device: model
  DevBuf<double> arc_logw;      // composed-arc weights (single transducer: THE parameters)
  DevBuf<double> counts;        // n_arcs + 4
  DevBuf<uint32_t> arc_group;   // groupId (single) / chain id (cascade)
  // device: parameters (cascade only; single transducer aliases the arc arrays)
  DevBuf<double> param_logw_c, param_counts_c;
  DevBuf<uint32_t> param_group_c;
  DevBuf<uint64_t> chain_off, chain_param;
  uint64_t n_params = 0, n_chains = 0;
  // M-step state over parameters
  DevBuf<double> old_logw, em_logw, best_logw, prior;
  DevBuf<double> mstep_snap;  // immutable copy of the weights for the one-pass M-step (see MstepArgs::lw_src)
  bool any_locked = false;    // some parameter is locked (group 0): its weight enters its group's sum
  DevBuf<uint32_t> norm_of;
  DevBuf<uint16_t> norm_code16;  // MstepArgs::code16 (one-pass M-step)
  DevBuf<uint32_t> norm_mask32, norm_lockmask32;  // MstepArgs::mask32 / lockmask32
  DevBuf<unsigned long long> norm_mask64, norm_lockmask64;  // ... for spans of 16 .. 31
  // tied arcs (!N, fst.cc:107-152): dense tie index per parameter (0xffffffff = not tied) and the per-tie tables
  DevBuf<uint32_t> tie_of;
  DevBuf<double> glocked;  // per norm group scratch
  DevBuf<double> tie_tab;  // 4 * n_ties: arc total, state total, max locked sum, resulting weight (linear)
  uint64_t n_ties = 0;
  DevBuf<uint64_t> group_off, norm_perm, big_groups;
  DevBuf<double> add_count, gscale;
  DevBuf<double> dig_alpha, tie_alpha;  // --digamma: per norm group / per tie (NaN = linear); empty when unused
  bool any_digamma = false;
  std::vector<uint32_t> h_group_member, h_tie_member;  // member transducer of every norm group / tie
  std::vector<uint32_t> h_group_src;    // source state (in its member) of every norm group
  std::vector<uint8_t> h_group_joint;   // 1 if the group's member normalises JOINT (one group per state)
  std::vector<uint32_t> h_group_ref_rank;  // CONDITIONAL groups: the group's place among its member's groups in the order the
                                           // reference enumerates them (NormGroupIter over State::index: host/refhash.hpp)
  DevBuf<double> arc_prior_w;           // cascade + carmel -U: initial weight of every composed arc (added to -f)
  std::vector<double> h_arc_prior_w;
  DevBuf<double> u_param_wprior;        // ... summed per parameter for the unrolled sweep
  bool any_add_count = false;
  DevBuf<unsigned long long> maxchg;
  // the M-step's largest change reaches the host through a pinned, coherent mailbox a one-thread kernel writes (value, then a
  // sequence number, released to the system): no copy command, no stream synchronisation on the way back (engine.cpp publish_*)
  unsigned long long* h_box = nullptr;
  unsigned long long box_seq = 0;
  // the corpus scalars run on the side stream behind the count pass; the trainer's stream joins them only when somebody
  // needs them (scalars_join): an iteration that goes straight on to the M-step never waits for them
  bool scalars_pending = false;
  uint64_t n_norm_groups = 0;
  bool have_norm = false, have_prior = false, prior_nonzero = false;
  int norm_group_by = CARMEL_HIP_NORM_CONDITIONAL;
  double norm_add_count = 0, smooth_floor = 0;
  // device: lattices
  DevBuf<BundleDesc> bundles;
  DevBuf<uint2_t> in_arcs, out_arcs;
  DevBuf<uint32_t> in_off, out_off, level_off, pair_start, pair_final, pair_id;
  DevBuf<double> pair_logw, pair_logprob, alpha_g, beta_g;
  DevBuf<WaveDesc> wave_descs;  // one lattice per wavefront (lattice.hpp)
  DevBuf<uint2_t> wave_fwd;
  DevBuf<uint32_t> wave_xc_idx;   // the item (place in XC) of every wave position: the sweep writes XC itself (build_run_tables)
  DevBuf<uint32_t> wave_bwd_arc;  // gathered weights (LatticeSet::wave_gather): the arc id of every backward record
  DevBuf<uint32_t> wave_bwd, wave_level_off, wave_frow, wave_brow;
  uint64_t wave_slot_base = 0, wave_records = 0;
  DevBuf<double> wave_spill;
  DevBuf<LaneGroup> lane_groups;
  DevBuf<uint32_t> tile_group;  // LatticeSet::tile_group (tile sweep)
  DevBuf<uint32_t> lane_rec2;   // ... and its packed records (LaneArgs::rec2)
  DevBuf<uint32_t> lane_chain;  // ... per group: a group of single paths (LaneArgs::chain)
  DevBuf<uint32_t> tile_chain;  // ... per tile: all of them (LaneArgs::tile_chain)
  DevBuf<uint2_t> lane_fwd;
  // blocked transposition tables (TransBucket, lattice.hpp); empty => gather / count_reduce path
  DevBuf<TransBucket> t_buckets;
  DevBuf<uint64_t> t_tile_base;
  DevBuf<uint16_t> t_b_arc, t_b_rank, t_t_pos, t_a_off;
  DevBuf<uint32_t> t_b_src, t_t_src, t_split_arcs;
  DevBuf<uint32_t> t_t_arc;  // the WFST arc of every tile-major item: the tile passes fetch weights from the table (build_run_tables)
  DevBuf<double> t_x, t_xc;
  DevBuf<uint32_t> tr_off, tr_src, br_off, br_src;  // run-length form of t_t_src / t_b_src (TransArgs)
  DevBuf<uint16_t> tr_rel, br_rel;
  bool use_runs = false;
  DevBuf<unsigned long long> max_partial;  // M-step scratch
  uint32_t norm_span = 0;                  // max over norm groups of (last member - first member); 0 = unknown / too wide
  bool all_grouped = true;                 // every parameter is in a norm group
  DevBuf<double> pair_w;          // pair weight by pair id, < 0 for pairs dropped at build_lattices
  DevBuf<double> scalar_partial;  // scratch of the corpus-scalar reduction
  // one-tape models: lattices are never stored, the sweep walks positions (unrolled.hpp)
  bool unrolled = false;
  bool allow_unrolled = true;  // carmel_hip_set_layout_policy
  UnrolledModel um;  // host tables (bulk arrays are freed after upload)
  DevBuf<uint32_t> u_f_off, u_b_off, u_f_arc, u_b_arc, u_e_arc, u_pair_id;
  DevBuf<uint16_t> u_e_src, u_e_dst;
  DevBuf<uint32_t> u_slot_of;  // cascade: parameter -> accumulator slot of the unrolled sweep (0xffffffff: locked)
  DevBuf<URec> u_f_rec, u_b_rec;  // packed tables (unrolled_args.hpp)
  DevBuf<uint16_t> u_e_slot, u_seq_sym;
  DevBuf<uint64_t> u_seq_off;
  DevBuf<double> u_We, u_pair_weight, u_partial, u_scratch;
  DevBuf<double> u_param_uses;          // cascade: composed arcs whose chain holds the parameter (for the -f prior)
  DevBuf<double> u_em_param, u_best_param;  // cascade: parameter-space images of em_weight / best_weight
  uint32_t u_n_slots = 0, u_n_wg = 0;
  // rank-1 dense form of the unrolled sweep (dense.hpp): weight(s -> s', c) = A[s][s'] * B[c][s']
  bool dense = false;
  uint32_t d_SP = 0, d_groups = 0;
  DevBuf<double> d_A, d_AT, d_B, d_vbuf, d_zbuf, d_afbuf, d_weight, d_partial;
  DevBuf<uint32_t> d_a_off, d_a_par, d_b_off, d_b_par, d_len, d_pair;
  DevBuf<uint8_t> d_a_has, d_b_has;
  DevBuf<uint16_t> d_Bslot, d_sym;
  DevBuf<uint64_t> d_sym_off, d_vbuf_off;
  // the E-step as a replayed hipGraph (engine.cpp: carmel_hip_estimate_async)
  uint64_t lattice_epoch = 0;
  bool use_transpose = false;
  bool em_valid = false;  // em_logw holds the plain EM update of the last (over-relaxed) maximize
  bool mstep_stream_work = false;  // mstep_args left work on the trainer's stream (scratch re-allocated and cleared after a new
                                   // set_norm / set_prior) that an M-step on another stream has not yet waited for
  DevBuf<uint32_t> lane_bwd;  // destination | flags words only
  DevBuf<uint32_t> lane_fwdx; // source | backward position | flags words only (transposition path)
  DevBuf<uint32_t> lane_pair, lane_nstates;
  DevBuf<double> lane_logw, post, wcache, lane_spill;
  DevBuf<double> counts_acc;  // carmel_hip_accumulate_counts: the count buffers of a corpus walked shard by shard, summed
  DevBuf<uint64_t> arc_off, slot_pos, hot_chunks;
  uint64_t lane_records = 0;
  uint64_t device_bytes = 0;
  // host copies the Gibbs sampler set-up needs (gibbs.hip)
  std::vector<uint32_t> h_norm_of, h_param_group;
  std::vector<uint64_t> h_group_off, h_norm_perm;  // members of every norm group (host copy, for the M-step masks)
  std::vector<double> h_group_add;
  std::vector<uint64_t> h_chain_off, h_chain_param;

  int build_prune = 1, build_threads = 0;  // the arguments of the last carmel_hip_build_lattices
  void* matrix = nullptr;  // carmel_hip_set_matrix_fb: device tables of the dense-matrix E-step (matrix_fb.hip)
  double* ext_counts = nullptr;  // caller-owned n_arcs + 4 doubles (carmel_hip_use_external_counts)
  double* counts_ptr() { return ext_counts ? ext_counts : counts.p; }
  double* params() { return cascade ? param_logw_c.p : arc_logw.p; }
  double* pcounts() { return cascade ? param_counts_c.p : counts_ptr(); }
  uint32_t* pgroup() { return cascade ? param_group_c.p : arc_group.p; }
  uint64_t np() const { return cascade ? n_params : w.n_arcs; }
};

// engine.cpp pieces the exchange shares
extern "C" int mstep_args(carmel_hip_trainer* t, int use_counts, int save_old, MstepArgs& M);
extern "C" void trans_args(carmel_hip_trainer* t, TransArgs& T);
bool exchange_is_sharded(const ExchangePlan* xp);  // exchange.cpp
// engine.cpp: one 8-byte value to the host at the end of what is enqueued on s (the pinned mailbox; published: the last kernel on
// s was an M-step's mstep_max_final_kernel, which has already stored it)
extern "C" int publish_u64(carmel_hip_trainer* t, const unsigned long long* dev, hipStream_t s);
extern "C" int fetch_u64(carmel_hip_trainer* t, const unsigned long long* dev, unsigned long long* out, hipStream_t s, bool published);
extern "C" int scalars_join(carmel_hip_trainer* t);  // engine.cpp: before anything reads counts[n_arcs .. n_arcs + 4) on the trainer's stream
namespace carmel_hip {  // matrix_fb.hip
int matrix_setup(carmel_hip_trainer* t, void** out);
int matrix_estimate(carmel_hip_trainer* t, void* state, hipStream_t s);
void matrix_release(void* state);
}
