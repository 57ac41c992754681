// kernels.hpp — argument blocks and launch entry points shared by kernels.hip (device code) and engine.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lattice.hpp"

namespace carmel_hip {

// Expected counts are produced in two phases so that no sweep kernel issues random atomics (f64 atomic adds to a
// table much larger than L2 run at ~5 G/s on MI355X — 10x slower than everything else in the E-step):
//   1. the sweeps write one posterior per lattice arc into `post` at the arc's SLOT (coalesced stores);
//   2. count_reduce sums, for every WFST arc, the posteriors of the slots that use it (slots pre-sorted by arc id
//      on the host — the topology never changes between iterations).
struct LaneArgs {
  const LaneGroup* groups;
  const uint2* fwd;      // forward records {flags, arc id} (gather path)
  const uint32_t* fwdx;  // forward records, flags word only (pre-distributed weights)
  const uint32_t* bwd;   // backward records: destination | flags only (the arc id stays on the host)
  const uint32_t* rec2;  // tile sweep: forward and backward record of a position packed into one word (launch_pack_tile_records)
  const uint32_t* chain; // ... per group: bit 0 = every lattice of the group is a single path; from bit 8: its lanes' most padding rows
  const uint32_t* tile_chain;  // ... per tile: 1 = all of its groups are
  const uint32_t* lane_pair;
  const uint32_t* lane_nstates;
  const double* lane_logw;
  const double* logw;
  double* post;          // slot = record position in the backward stream
  double* wcache;        // weight of the arc at BACKWARD-stream position k (written forward, streamed backward)
  double* scalars;       // {sum ln p, sum weight*ln p, n pairs, -}
  double* pair_logprob;
  double* spill;         // windowed groups: forward values, one column of 64-lane rows per group (LaneGroup::spill_row)
  uint32_t first_group;
  uint32_t pre_weights;  // wcache already holds every arc's weight at its backward position (blocked transposition)
  unsigned long long* trace;  // experiment: per-block {t_start, t_mid, t_end, hw id} (CARMEL_HIP_LANE_TRACE)
  // fused-lane layout (LatticeSet::lane_fused), sweep_lane_kernel<.., XC>: the backward pass stages the posteriors of a tile of
  // LANE_FUSED_TILE positions in LDS and writes the tile's items out in tile-major order itself (what trans_c_tile does from `post`)
  uint32_t lds_rows;             // value rows of this launch (LaneClass::max_states): the stage lies behind them
  const uint64_t* xc_tile_base;  // TransArgs::tile_base
  const uint16_t* xc_t_pos;      // TransArgs::t_pos
  double* xc;                    // TransArgs::xc
};

struct SweepArgs {
  const BundleDesc* bundles;
  const uint2* in_arcs;
  const uint2* out_arcs;
  const uint32_t* in_off;
  const uint32_t* out_off;
  const uint32_t* level_off;
  const uint32_t* pair_start;
  const uint32_t* pair_final;
  const uint32_t* pair_id;
  const double* pair_logw;
  const double* logw;     // per WFST arc
  double* post;           // slot = post_base + out_base + position in out_arcs
  double* scalars;
  double* pair_logprob;   // per corpus pair
  double* val_g;          // global scratch for alpha/beta of bundles too large for LDS and of serial sweeps
  double* val2_g;
  uint32_t first_bundle;
};

// one lattice per wavefront (WaveDesc, lattice.hpp)
struct WaveArgs {
  const WaveDesc* descs;
  const uint2* fwd;
  const uint32_t* bwd;
  const uint32_t* level_off;
  const uint32_t* frow;
  const uint32_t* brow;
  const double* wcache;   // at the first wave slot: the weight of the arc at backward position k
  const double* logw = nullptr;       // gathered weights (bwd_arc set): the WFST's table, a forward record's y = the arc id
  const uint32_t* bwd_arc = nullptr;  // ... and the arc id of every backward record (0xffffffff: padding)
  uint32_t n_arcs = 0;
  double* xc = nullptr;               // posteriors straight to the count pass (xc_idx set): XC, tile-major item order
  const uint32_t* xc_idx = nullptr;   // ... the item of every backward position (0xffffffff: none), at the first wave slot
  double* post;           // at the first wave slot: posteriors, same positions
  double* pair_logprob;
  double* spill;          // ring lattices: parked forward values (WaveDesc::spill_base)
  uint32_t first;         // first descriptor of this launch
  uint32_t max_states, max_width;  // LDS of this launch: (max_states + 2 * max_width) doubles
};

struct ReduceArgs {
  const uint64_t* arc_off;    // arc a's slots: slot_pos[arc_off[a] .. arc_off[a+1])
  const uint64_t* slot_pos;   // position in post[] of each slot, grouped by arc
  const uint64_t* hot_chunks; // (arc, first slot, end slot) triples covering arcs with more than COUNT_HOT slots
  const double* post;
  double* counts;             // per WFST arc, linear; every entry is written
  uint64_t n_arcs, n_hot_chunks;
};

// blocked transposition (see TransBucket, lattice.hpp)
struct TransArgs {
  const TransBucket* buckets;
  const uint64_t* tile_base;
  const uint16_t* b_arc;
  const uint16_t* b_rank;
  const uint32_t* b_src;
  const uint16_t* t_pos;
  const uint32_t* t_src;
  const uint16_t* a_off;  // per arc: first item inside its bucket
  double* x;              // intermediate of the weights direction, one f64 per item (bucket-major index)
  double* xc;             // intermediate of the counts direction (tile-major index): a buffer of its own, so that one
                          // chunk's posteriors can leave while another chunk's weights are still being read
  const double* logw;     // per arc
  double* wcache;         // per lane position (n_wcache entries)
  const double* post;     // per position (n_post entries)
  double* counts;         // per arc
  uint64_t n_wcache, n_post;
  uint32_t n_buckets, n_tiles;
  uint32_t n_wtiles;      // tiles that cover wcache (set by launch_transpose_weights)
  // run-length form of t_src / b_src: inside one (bucket, tile) cell both indices advance together, so a run of items
  // is {first item relative to its tile / bucket, source index of that item}: ~0.5 B per item instead of 4
  const uint32_t* tr_off;   // per tile: its runs are [tr_off[t], tr_off[t + 1])
  const uint16_t* tr_rel;
  const uint32_t* tr_src;
  const uint32_t* br_off;   // per bucket
  const uint16_t* br_rel;
  const uint32_t* br_src;
  uint32_t use_runs;        // every tile and bucket has at most TRANS_RUN_CAP runs (they are staged in LDS)
  uint32_t scatter;         // bit 0 (weights) / bit 1 (counts): the first pass of that direction writes its items where the
                            // second pass reads them as one sequential stretch (x tile-major / xc bucket-major), instead of
                            // writing sequentially and leaving the gather to the second pass
  uint32_t tile;            // positions per tile (LatticeSet::tile)
  uint32_t bucket;          // items a bucket holds at most (LatticeSet::bucket: TRANS_BUCKET or half of it)
  uint32_t tile_first, tile_count;  // the tile range of this launch (a chunk of a lane class, or the bundle tiles)
  uint32_t bucket_first, bucket_count;  // the bucket range of this launch of a bucket pass (arc-range chunks of the exchange)
  uint32_t slack_bytes;     // readable bytes behind x / xc / t_pos / t_src (DEVBUF_SLACK when they are DevBufs, engine.hpp): the
                            // persistent tile kernels read whole rounds past a tile's last item
  // tile_sweep_kernel, when asked: counts[zero_list[0 .. n_zero)] := 0 on its way in -- the arcs whose items lie in several
  // buckets, which the count pass adds up with atomics (instead of a zero_list_kernel launch between the sweep and the count pass)
  const uint32_t* zero_list = nullptr;
  uint32_t n_zero = 0;
};

#define TRANS_RUN_CAP 4096
#define TRANS_RUN_LDS (TRANS_RUN_CAP * 4 + 4096)  // r_src + the run-start mask and its prefix counts (RunLds, kernels.hip)
#define MSTEP_BIG_GROUP 48  // a group above this size is summed by a workgroup of its own (one thread walking 500 members is a
                           // chain of 500 dependent loads: 0.22 ms on the tagging lexicon's per-tag groups)
#define MSTEP_PARTIALS 2048
#define MSTEP_MAX_RANGES 17
struct MstepArgs {
  double* logw;             // parameters (ln), updated in place
  const double* lw_src;     // mstep_window_kernel: where the ln weights of the tile AND its halo are read -- `logw` itself
                            // only when no thread needs a neighbour's weight (counts in use, no locked member);
                            // otherwise a snapshot taken before the launch, because neighbouring workgroups overwrite
                            // logw in place with no ordering between them
  double* old_logw;         // scratch (arc_counts::scratch)
  const double* counts;     // linear expected counts per parameter
  const double* prior;      // linear prior count per parameter (may be null => 0)
  const uint32_t* group;    // FSTArc::groupId per parameter (0 = locked)
  const uint32_t* norm_of;  // norm-group id per parameter; 0xffffffff = member normalised by NONE
  const double* add_count;  // per norm group (--priors of the member it belongs to); null when all zero
  const uint64_t* group_off;   // norm group g's members: norm_perm[group_off[g] .. group_off[g+1])
  const uint64_t* norm_perm;   // parameter ids sorted by norm group
  const uint64_t* big_groups;  // groups with more than MSTEP_BIG_GROUP members (one workgroup each)
  uint64_t n_groups, n_big;
  double* gscale;           // per norm group: ln((1 - sum of locked arcs) / sum of normal arcs), -inf if nothing left
  unsigned long long* max_partial;  // MSTEP_PARTIALS + n_big entries of scratch
  unsigned long long* max_change_bits;
  int all_grouped;          // every parameter belongs to a norm group
  uint32_t window_span;     // > 0: every group's members lie within this many consecutive parameters (<= 64)
  double* glocked;          // per norm group scratch (ties only): sum of its locked members
  const uint32_t* tie_of;   // dense tie index per parameter, 0xffffffff = not tied; null when the model has no ties
  double* tie_tab;          // [4][n_ties]: arc total, state total, max locked sum, weight (linear)
  uint64_t n_ties;
  uint64_t n;
  uint32_t block_first;     // mstep_window_kernel: first 256-parameter block of this launch
  // ... or (n_ranges > 0) the launch covers several block ranges -- the sharded M-step of the multi-GPU exchange runs over
  // this rank's arc ranges only: workgroup b belongs to range r with range_cum[r] <= b < range_cum[r + 1] and handles
  // block range_first[r] + (b - range_cum[r])
  uint32_t n_ranges;
  uint32_t range_first[MSTEP_MAX_RANGES];
  uint32_t range_cum[MSTEP_MAX_RANGES + 1];
  int save_old;             // 1: old_logw <- the weights before this pass, |change| against them; 0: keep old_logw from the
                            // previous pass and compare against it (second normalise after overrelax); 2: old_logw is not
                            // needed afterwards (no over-relaxation): do not write it, compare against the weight read
  const uint32_t* mask32;   // mstep_window_kernel, span <= 15: per parameter, which of the parameters at offsets -15 .. +16 are the
                            // unlocked members of its norm group (bit offset + 15; its own bit included when it is unlocked) --
                            // the topology is static, so the kernel adds up exactly its group's members instead of scanning
                            // and comparing the whole window (the scan, not the 26 bytes per parameter, was what the kernel
                            // spent its time on); lockmask32: the locked members.  Null when span > 15.
  const uint32_t* lockmask32;
  const unsigned long long* mask64;  // the same for spans of 16 .. 31: offsets -31 .. +32, bit offset + 31
  const unsigned long long* lockmask64;
  const uint16_t* code16;   // mstep_window_kernel: per parameter, norm-group id mod 2^14 | 0x4000 if locked; 0xffff = no norm
                            // group.  Ids are handed out in first-seen order, so inside a window of < 2^14 parameters equal
                            // low bits mean equal groups: 2 bytes per parameter instead of norm_of + group (8)
  // --digamma (mean_field_scale.hpp:40-52): per norm group / per tie the alpha of exp(digamma(x + alpha)) that replaces
  // x in the numerator and the denominator of the normalisation; NaN = the usual linear scale.  Null when no member
  // asked for it (the one-pass window kernel only handles the linear scale).
  const double* dig_alpha;
  const double* tie_alpha;
  // mstep_max_final_kernel also stores the result and then box_seq into this pinned, coherent host mailbox (release, system
  // scope) when box != null: the host spins on the sequence number instead of a copy command and a stream synchronisation
  unsigned long long* box = nullptr;
  unsigned long long box_seq = 0;
};

// fused: the XC form of the backward pass (LaneArgs::xc_* set; the groups lie on LANE_FUSED_TILE boundaries)
hipError_t launch_lane_sweep(const LaneArgs& A, const LatticeSet::LaneClass& lc, hipStream_t stream, bool fused = false);
hipError_t launch_sweep(const SweepArgs& A, const LatticeSet::LaunchClass& lc, hipStream_t stream);
hipError_t launch_wave_sweep(const WaveArgs& A, const LatticeSet::WaveClass& wc, hipStream_t stream);
// the two passes of either direction as separate launches: the bucket passes cover the whole model, the tile passes a
// range of tiles (so that chunks of the corpus can flow through weights-in / sweep / posteriors-out side by side)
hipError_t launch_trans_w_bucket(const TransArgs& T, hipStream_t stream);
// ... over buckets [first, first + count) only
hipError_t launch_trans_w_bucket_range(const TransArgs& T, uint32_t first, uint32_t count, hipStream_t stream);
hipError_t launch_trans_c_bucket_range(const TransArgs& T, uint32_t first, uint32_t count, hipStream_t stream);
// LatticeSet::tile_sweep: weights in, the lane sweeps and posteriors out of tiles [tile_first, tile_first + tile_count) in one kernel
hipError_t launch_tile_sweep(const TransArgs& T, const LaneArgs& A, const uint32_t* tile_group, uint32_t tile_first, uint32_t tile_count,
                             hipStream_t stream);
hipError_t launch_pack_tile_records(const LaneGroup* groups, uint32_t n_groups, const uint32_t* lane_nstates, const uint32_t* fwdx,
                                    const uint32_t* bwd, uint32_t* out, uint32_t* chain, const uint32_t* tile_group, uint32_t n_tiles,
                                    uint32_t* tile_chain, hipStream_t stream);
hipError_t launch_zero_list(double* p, const uint32_t* idx, uint32_t n, hipStream_t stream);
// small[k] = src[idx[k]] / dst[idx[k]] = small[k]   (halo values of the exchange)
hipError_t launch_gather_idx(double* small, const double* src, const uint32_t* idx, uint32_t n, hipStream_t stream);
hipError_t launch_scatter_idx(double* dst, const double* small, const uint32_t* idx, uint32_t n, hipStream_t stream);
// the one-pass M-step over parameters [256 * block_first, 256 * (block_first + n_blocks)) only
hipError_t launch_mstep_window_range(const MstepArgs& M, int use_counts, uint32_t block_first, uint32_t n_blocks, hipStream_t s);
hipError_t launch_mstep_max_final(const MstepArgs& M, hipStream_t s);
hipError_t launch_trans_w_tiles(const TransArgs& T, uint32_t tile_first, uint32_t tile_count, hipStream_t stream);
hipError_t launch_trans_c_tiles(const TransArgs& T, uint32_t tile_first, uint32_t tile_count, hipStream_t stream);
hipError_t launch_trans_c_bucket(const TransArgs& T, const uint32_t* split_arcs, uint32_t n_split, hipStream_t stream);
// partial: 3 * 256 doubles of scratch
hipError_t launch_scalars(const double* pair_logprob, const double* pair_w, uint64_t n_pairs, double* partial,
                          double* scalars, hipStream_t s);
hipError_t launch_count_reduce(const ReduceArgs& R, hipStream_t stream);
hipError_t launch_fill(double* p, double v, uint64_t n, hipStream_t s);
hipError_t launch_add(double* dst, const double* src, uint64_t n, hipStream_t s);  // dst += src
hipError_t launch_mstep(const MstepArgs& M, int use_counts, hipStream_t s);
hipError_t launch_overrelax(double* logw, const double* old_logw, double* em_logw, const uint32_t* group, double rate,
                            uint64_t n, hipStream_t s);
hipError_t launch_max_change(const double* logw, const double* old_logw, const uint32_t* group,
                             unsigned long long* bits, uint64_t n, hipStream_t s);
hipError_t launch_chain_update(double* arc_logw, const uint32_t* arc_chain, const uint64_t* chain_off,
                               const uint64_t* chain_param, const double* param_logw, uint64_t n_arcs, hipStream_t s);
// arc_prior_w: optional per-arc addition to the prior (carmel -U on a cascade: the composed arc's initial weight)
hipError_t launch_chain_scatter(double* param_counts, const double* arc_counts, double arc_prior, const double* arc_prior_w,
                                const uint32_t* arc_chain, const uint64_t* chain_off, const uint64_t* chain_param,
                                const uint32_t* param_group, uint64_t n_arcs, hipStream_t s);
// the max_iter == 0 branch of WFST::train (train.cc:527-529): logw[k] = ln(counts[k] (+ prior[k])) for unlocked k
hipError_t launch_counts_to_logw(double* logw, const double* counts, const double* prior, const uint32_t* group, uint64_t n,
                                 hipStream_t s);

}  // namespace carmel_hip
