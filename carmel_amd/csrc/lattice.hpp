// lattice.hpp — host-side construction of per-pair derivation lattices and their batched-CSR layout for HBM.
//
// What it computes is what the reference's derivations::compute + prune compute
// (/root/reference/carmel/src/derivations.h:479-513, 640-704, 572-629): the product graph of
// (input position, WFST state, output position) restricted to states that lie on a start->goal path, each
// lattice arc carrying the id of the WFST arc it uses.  How it computes it is different by design: no
// recursion, no per-node malloc — an explicit-stack forward exploration over a flat (state,in,out)-sorted arc
// index, a reverse sweep for co-reachability, a Kahn pass that assigns every state its longest-path LEVEL,
// and then packing of many lattices into workgroup-sized BUNDLES whose states are stored level-major so the
// GPU sweeps are level-synchronous gathers with no atomics on the forward/backward values.
#pragma once
#include <cstdint>
#include <vector>
#include <string>

namespace carmel_hip {

struct HostWfst {
  uint32_t n_states = 0, final_state = 0;
  uint64_t n_arcs = 0;
  std::vector<uint32_t> src, dst, in, out, group;
  // (src, in, out)-sorted index: idx_off[s]..idx_off[s+1] are positions into idx_key/idx_arc
  std::vector<uint64_t> idx_off;
  std::vector<uint64_t> idx_key;  // (in << 32) | out, ascending within a state
  std::vector<uint32_t> idx_arc;  // arc id; ties keep arc-id order
  void build_index();
};

struct HostCorpus {
  uint64_t n_pairs = 0;
  std::vector<uint64_t> in_off, out_off;
  std::vector<uint32_t> in_sym, out_sym;
  std::vector<double> weight;
};

// One pair's pruned lattice, states numbered arbitrarily, with levels.
struct PairLattice {
  uint32_t n_states = 0, n_levels = 0;
  uint32_t start = 0, fin = 0;
  bool cyclic = false;
  std::vector<uint32_t> level;  // per state (acyclic) — for cyclic lattices: position in the reference's sweep order
  struct E {
    uint32_t src, dst, arc;
  };
  std::vector<E> edges;
  uint64_t explored_states = 0, explored_arcs = 0;
  uint32_t span = 0;  // max over the arcs of (dst - src) in the (level, id) numbering of the lane layout; 0 = not computed
};

// A bundle = lattices swept together by one workgroup.  All indices below are bundle-local.
struct BundleDesc {      // mirrored on the device (keep POD, 64 bytes)
  uint64_t in_base;      // into in_arcs[]   ({src_state, arc_id} sorted by dst state, level-major)
  uint64_t out_base;     // into out_arcs[]  ({dst_state, arc_id} sorted by src state)
  uint64_t off_base;     // into in_off[] / out_off[] (n_states + 1 entries per bundle) and into alpha scratch
  uint32_t n_states;
  uint32_t n_levels;
  uint32_t level_base;   // into level_off[] (n_levels + 1 entries per bundle)
  uint32_t pair_base;    // into pair_start[] / pair_final[] / pair_id[] / pair_logw[]
  uint32_t n_pairs;
  uint32_t flags;        // bit0: cyclic -> serial in-order sweep (reference order)
  uint64_t n_arcs;
  uint64_t pad;
};
static_assert(sizeof(BundleDesc) == 64, "BundleDesc layout");

struct uint2_t {
  uint32_t x, y;
};

// Lane groups: 64 small acyclic lattices swept by ONE wavefront, one lattice per lane.  Each lattice is flattened
// into two record streams (forward: in-arcs grouped by destination in topological order; backward: out-arcs
// grouped by source in reverse topological order) and the 64 streams are interleaved record by record, so every
// wave-wide load is one coalesced 512-byte row and the topology needs no offsets, no levels and no barriers.
// record.x = local state index (bits 0..9) | [forward records: backward position of the same arc, bits 10..29]
//            | [backward records: the arc's SOURCE state, bits 10..19] | LANE_VALID | LANE_LAST (last arc of its state);
//            record.y = WFST arc id
// Windowed groups (LaneGroup::window = W, a power of two): every arc of every lattice in the group spans fewer than W states
// of the topological numbering, so the sweep keeps only a ring of W values per lane in LDS (state s at row s mod W) and
// parks the forward values in a global column (lane_spill) for the backward pass.  LDS per wave: 512 B * W instead of
// 512 B * states -- what decides how many waves a CU holds for lattices of a hundred states.
static const uint32_t LANE_CHUNK = 4;  // a group's row count (maxlen) is a multiple of this: the kernel consumes whole chunks
static const uint32_t LANE_LAST = 0x80000000u;
static const uint32_t LANE_VALID = 0x40000000u;
static const uint32_t LANE_STATE_MASK = 0x3ffu;
static const uint32_t LANE_POS_SHIFT = 10;
static const uint32_t LANE_POS_MAX = (1u << 20) - 1;
// Blocked transposition.  Both directions of the E-step's data exchange -- weights from arc order to lattice order,
// posteriors from lattice order back to arc order -- are the same static sparse permutation of ~one item per lattice
// arc.  Done as random 8-byte accesses each item moves a whole 128-byte line; done in two LDS-blocked passes it is
// sequential traffic:
//   buckets = contiguous arc ranges with <= TRANS_BUCKET items and arcs (a bucket's weights / its items fit in LDS),
//   tiles   = TRANS_TILE consecutive slot positions (a tile of wcache / post fits in LDS),
//   the intermediate array X holds the items bucket-major, each bucket's items sorted by position, so the items of
//   (bucket b, tile t) are one contiguous run in both orders.
// An arc with more than TRANS_HEAVY items gets single-arc buckets of its own (TRANS_SINGLE), several of them
// (TRANS_SPLIT) when it has more items than a bucket holds.
#ifndef TRANS_K
#define TRANS_K 16  // rounds of 1024 threads per tile / bucket
#endif
#ifndef TRANS_KT
#define TRANS_KT TRANS_K  // ... per tile
#endif
#ifndef TRANS_KB
#define TRANS_KB TRANS_K  // ... per bucket
#endif
static const uint32_t TRANS_TILE = TRANS_KT * 1024, TRANS_BUCKET = TRANS_KB * 1024, TRANS_HEAVY = 2048, TRANS_SPLIT = 1u, TRANS_SINGLE = 2u;
// Tile sweep (LatticeSet::tile_sweep): a corpus made of plain lane lattices only is laid out in tiles of TILE_SWEEP_TILE positions
// that no lane group straddles, so that one workgroup can take a tile's weights in, sweep its groups out of LDS and send the
// posteriors out (tile_sweep_kernel) -- LDS of a tile: its positions (8 + 4 B each) + TILE_SWEEP_ALPHA_ROWS rows of 64 forward /
// backward values for its groups (LaneGroup::spill_row = a group's first row; a tile closes when either runs out, or
// at TILE_SWEEP_GROUPS groups).
// Lattices of at most TILE_SWEEP_ROWS arcs (a packed record has six bits for a row) and 256 states (eight for a state).
static const uint32_t TILE_SWEEP_TILE = 8192, TILE_SWEEP_ALPHA_ROWS = 126, TILE_SWEEP_ROWS = 48, TILE_SWEEP_GROUPS = 16;
static const uint32_t TILE_SWEEP_LDS = TILE_SWEEP_TILE * 12 + TILE_SWEEP_ALPHA_ROWS * 64 * 8;  // weights + records + values
// Fused lanes (LatticeSet::lane_fused): a corpus whose lattices all go one per lane but are too long for the tile sweep (windowed
// groups, plain lattices above TILE_SWEEP_ROWS arcs) is laid out with every lane group's streams starting on a tile of
// LANE_FUSED_TILE positions = LANE_FUSED_ROWS rows of its 64 lanes: the wavefront that sweeps a group then owns whole tiles of
// the blocked transposition, and its backward pass hands the posteriors of a tile to the count pass itself (out of a 16 KB LDS
// stage, in tile-major item order: sweep_lane_kernel<.., XCOUT>) -- `post` is never written and trans_c_tile never runs.
#ifndef LANE_FUSED_ROWS_N
#define LANE_FUSED_ROWS_N 16
#endif
static const uint32_t LANE_FUSED_ROWS = LANE_FUSED_ROWS_N, LANE_FUSED_TILE = LANE_FUSED_ROWS * 64;
struct TransBucket {      // mirrored on the device, 24 bytes
  uint64_t item_base;     // first item: bucket-major index J == index into slot_pos[] (arc-sorted order)
  uint32_t n_items, arc_lo, n_arcs, flags;
};
struct LaneGroup {        // mirrored on the device, 32 bytes
  uint64_t stream_base;   // into lane_fwd[] / lane_bwd[] (maxlen * 64 records each)
  uint32_t maxlen;        // records per lane (shorter lattices are padded with invalid records)
  uint32_t n_lanes;       // lattices in this group (<= 64)
  uint32_t pair_base;     // into lane_pair[] / lane_nstates[] / lane_logw[]
  uint32_t max_states;    // LDS rows the group needs: its largest lattice, or the window
  uint32_t window;        // 0: the whole column lives in LDS; W: ring of W rows + lane_spill
  uint32_t spill_row;     // windowed: first row (of 64 doubles) of the group's columns in lane_spill; tile sweep: first row
                          // of the group's values in its tile's LDS
};
static_assert(sizeof(LaneGroup) == 32, "LaneGroup layout");

// Wave lattices: one lattice per WAVEFRONT, its 64 lanes over the ARCS of a level -- for lattices that are too large for a
// lane (or too few to fill the chip one per lane) and wide enough to feed 64 lanes.  States are numbered level-major
// (level = longest path from the start: the start is state 0, the goal the last state).  Two record streams, both cut
// into ROWS of 64 records that never straddle a level (padded with invalid records), so that a row is one coalesced load
// and all of its arcs are independent:
//   forward : the in-arcs of the states of level 1, 2, ... (by destination); record {x, y}:
//             x = source state (bits 0..15) | destination - first state of its level (bits 16..29) | WAVE_VALID,
//             y = the arc's position in the backward stream (relative to the lattice) = where its weight lies in wcache;
//   backward: the out-arcs of the states of level L-2, L-3, ... 0 (by source); record
//             x = destination state (bits 0..15) | source - first state of its level (bits 16..29) | WAVE_VALID.
// A backward record's position is also the arc's posterior slot and the place of its weight in wcache (the blocked
// transposition delivers the weights there), so the backward pass streams rows and the forward pass gathers inside the
// lattice's own stretch of wcache.  The sweep keeps all forward / backward values of the lattice in LDS (8 B per state)
// and adds up a level's log-sums with LDS atomics (max, then sum of exp): no segment bookkeeping, any arc order.
static const uint32_t WAVE_VALID = 0x80000000u;
static const uint32_t WAVE_MAX_STATES = 16384, WAVE_MAX_WIDTH = 1024;
// Ring form (WaveDesc::ring): when no arc spans more than R states of the level-major numbering and no level is wider than
// a wavefront, the sweep keeps only a ring of R values in LDS -- a kilobyte instead of 8 B per state, so a CU holds its full
// complement of waves (the sweep of one lattice is a chain of dependent LDS round trips: what hides it is other lattices)
// -- and parks the forward values in a global column for the posteriors of the backward pass (+16 B per state).
static const uint32_t WAVE_RING_MAX = 1024, WAVE_RING_WIDTH = 64;
struct WaveDesc {         // mirrored on the device, 64 bytes
  uint64_t fwd_base;      // into wave_fwd[]
  uint64_t bwd_base;      // into wave_bwd[]; slot of a backward record = wave_slot_base + bwd_base + its position
  uint32_t n_states, n_levels;
  uint32_t level_base;    // into wave_level_off[] / wave_frow[] / wave_brow[] (n_levels + 1 entries each)
  uint32_t pair;          // corpus pair
  uint32_t max_width;     // states of its widest level
  uint32_t ring;          // 0: every state's value lives in LDS; R (a power of two): a ring of R values, state s at s mod R
  double logw;            // ln(pair weight)
  uint64_t n_arcs;
  uint64_t spill_base;    // ring lattices: first state in the global column of parked forward values (wave_spill)
};
static_assert(sizeof(WaveDesc) == 64, "WaveDesc layout");

struct LatticeSet {
  std::vector<WaveDesc> waves;
  std::vector<uint2_t> wave_fwd;
  std::vector<uint32_t> wave_bwd;
  std::vector<uint32_t> wave_bwd_arc;    // host only: WFST arc id of every backward record (0xffffffff: padding)
  std::vector<uint32_t> wave_level_off;  // per lattice n_levels + 1: first state of level l
  std::vector<uint32_t> wave_frow;       // ... first forward row (of 64 records) of destination level l
  std::vector<uint32_t> wave_brow;       // ... first backward row of step k (source level n_levels - 1 - k)
  struct WaveClass {
    uint32_t first, count, max_states, max_width;
    uint32_t ring = 0;  // > 0: every member sweeps through a ring of this many values (max_states is not used then)
  };
  uint64_t wave_spill_states = 0;        // doubles of the parked forward values of the ring lattices
  std::vector<WaveClass> wave_classes;   // launches, largest lattices first
  uint64_t wave_slot_base = 0;           // position of the first wave slot in post[] / wcache[] (a tile boundary)
  uint64_t wave_states = 0, wave_arcs = 0;
  std::vector<BundleDesc> bundles;
  std::vector<uint2_t> in_arcs, out_arcs;
  std::vector<uint32_t> in_off, out_off;  // bundle-relative arc offsets
  std::vector<uint32_t> level_off;
  std::vector<uint32_t> pair_start, pair_final, pair_id;
  std::vector<double> pair_logw;  // ln(pair weight)
  // launch classes: bundle index ranges sorted by LDS need
  struct LaunchClass {
    uint32_t first, count;   // bundles[first .. first+count)
    uint32_t block;          // threads per workgroup
    uint32_t max_states;     // LDS doubles needed (0 => values live in global scratch)
    bool serial;             // cyclic bundles
  };
  std::vector<LaunchClass> classes;
  // lane groups (see LaneGroup)
  std::vector<LaneGroup> lane_groups;
  std::vector<uint2_t> lane_fwd, lane_bwd;
  std::vector<uint32_t> lane_pair, lane_nstates;
  std::vector<double> lane_logw;
  // a launch of the lane sweep: groups [first, first + count) sharing one LDS size.  A class is cut into CHUNKS whose
  // record streams start on a tile boundary of the blocked transposition (padding rows in between), so that the
  // weights-in / sweep / posteriors-out kernels of different chunks can run side by side on separate streams:
  // tiles [tile_first, tile_first + tile_count) cover exactly this chunk's records.
  struct LaneClass {
    uint32_t first, count, max_states;
    uint32_t tile_first = 0, tile_count = 0;
    bool windowed = false;
  };
  uint64_t lane_spill_rows = 0;  // rows of 64 doubles for the forward values of the windowed groups
  std::vector<LaneClass> lane_classes;
  bool lane_tiles_aligned = false;  // the pieces' tile ranges are disjoint (required for launching the tile passes per piece)
  uint32_t tile = TRANS_TILE;       // positions per tile of the blocked transposition: TRANS_TILE, LANE_FUSED_TILE, or TILE_SWEEP_TILE when ...
  bool tile_sweep = false;          // ... no lane group straddles a tile and a tile's groups fit one workgroup's LDS (see TILE_SWEEP_TILE)
  bool wave_gather = false;         // a wave lattice's forward records carry the WFST arc id in y (BuildOptions::wave_gather)
  bool tables_deferred = false;     // the host builder stopped after the layout: slots by arc and the transposition tables are
                                    // built on the device from the records (BuildOptions::device_tables)
  bool lane_fused = false;          // tile == LANE_FUSED_TILE and every lane group starts on a tile (see LANE_FUSED_TILE)
  uint32_t bucket = TRANS_BUCKET;   // items (and arcs) a bucket of the blocked transposition holds at most: TRANS_BUCKET, or half of
                                    // it under the fused-lane layout (two bucket workgroups to a CU: measured on c4a, whose
                                    // (bucket, tile) cells are single runs of ~6 items either way, trans_c_bucket 1229 -> 1000 us)
  std::vector<uint32_t> tile_group; // tile sweep: per lane tile, its first lane group (+ one entry: the number of groups)
  uint64_t lane_states = 0, lane_arcs = 0;  // real (unpadded) totals in lane groups
  // posterior slots: one per lattice arc.  Lane records use their position in lane_bwd[]; bundle out-arcs use
  // lane_bwd.size() + position in out_arcs[].  slot_arc / slot_pos list every slot sorted by WFST arc id, which is
  // what lets the expected counts be a segmented sum instead of random atomics.
  std::vector<uint32_t> state_orig;  // BuildOptions::keep_state_ids: per bundle state (indexed like out_off), the state's id in
                                     // its pair's own lattice = the reference's numbering (derivations.h: creation order)
  std::vector<uint64_t> arc_off;   // n_arcs + 1
  std::vector<uint64_t> slot_pos;  // grouped by arc id
  std::vector<uint64_t> hot_chunks;  // (arc, first, end) triples: arcs with more than 64 slots, cut into 4096-slot chunks
  uint64_t n_post = 0;  // size of the posterior array (lane records incl. padding + bundle arcs)
  // Blocked transposition between WFST-arc order and slot (position) order -- see TransBucket below.
  std::vector<TransBucket> t_buckets;
  std::vector<uint64_t> t_tile_base;   // n_tiles + 1: first tile-major item index of every tile
  std::vector<uint16_t> t_b_arc;       // [J] arc - bucket.arc_lo
  std::vector<uint16_t> t_b_rank;      // [J] rank of the item in arc-sorted order within its bucket
  std::vector<uint32_t> t_b_src;       // [J] tile-major index I of the same item
  std::vector<uint16_t> t_t_pos;       // [I] position - tile * TRANS_TILE
  std::vector<uint32_t> t_t_src;       // [I] bucket-major index J of the same item
  std::vector<uint16_t> t_a_off;       // [arc] offset of the arc's first item inside its bucket (arc-sorted order)
  std::vector<uint32_t> t_split_arcs;  // arcs cut over several buckets (their counts are accumulated atomically)
  uint64_t total_states = 0, total_arcs = 0, max_levels = 0, n_cyclic = 0;
  uint64_t explored_states = 0, explored_arcs = 0;
  // derivations::statistics as the reference keeps it (derivations.h:197-210, 617-618): pre.states is ASSIGNED per pair
  // (the last pair's), post.states / post.arcs are the last pair's that has a derivation
  uint64_t last_pre_states = 0, last_post_states = 0, last_post_arcs = 0;
  std::vector<uint8_t> has_deriv;
  uint64_t n_kept = 0;
};

struct BuildOptions {
  bool prune = true;
  int threads = 0;
  uint32_t small_pairs = 64;       // lattices per small bundle (one wavefront wide)
  uint32_t small_states = 2048;    // state cap of a small bundle (16 KiB of f64 in LDS)
  uint32_t lds_states_max = 16384; // one array of f64 in LDS: 128 KiB
  uint32_t lane_states = 96;       // lattices up to this many states go one-per-lane (0 disables lane groups)
  uint32_t lane_window = 64;       // lattices whose arcs span fewer states than this go one-per-lane with a ring of that
                                   // many LDS rows whatever their size (up to 1023 states); 0 disables windowed groups
  uint32_t lane_window_min = 40;   // ... but only lattices above this many states: below ~20 KB of LDS per wave the
                                   // occupancy is not what bounds the sweep (and the GPU builder covers those)
  bool keep_state_ids = false;     // also fill LatticeSet::state_orig (the sampler's --expectation entry order needs them)
  bool gpu_large_caps = false;     // GPU builder: the larger per-pair capacities (lattices of up to 1 023 states; set by the probe)
  bool wave_ring = true;           // ring form of the wave sweep where the lattice allows it (WaveDesc::ring)
  bool wave = true;                // one-lattice-per-wavefront layout for large / few-and-wide lattices (WaveDesc)
  double wave_min_width = 4.0;     // ... for lattices that no lane takes: at least this many arcs per level on average
                                   // (narrower ones stay in bundles: 64 of them side by side feed a wave better)
  double wave_lane_min_width = 16.0;    // ... for lattices a WINDOWED lane would take: this wide, and only when the corpus
  uint64_t wave_lane_threshold = 262144;  // has fewer lane-sized lattices than this (4 waves per SIMD of one-per-lane work)
  uint32_t wave_lane_arcs = 2048;       // ... or the lattice has more arcs than this, however large the corpus: a lane group's time is
                                        // its longest lane's stream, and a few groups of 8 000-arc lattices beside thousands of
                                        // 300-arc ones are a tail of a dozen wavefronts the chip waits for (`mix`, round 6: 5.4 ms)
  bool tile_sweep = true;          // lay a corpus of plain lane lattices out for the one-kernel tile sweep (LatticeSet::tile_sweep)
  bool wave_gather = false;        // one-per-wavefront lattices: forward record y = the WFST arc id, not the arc's backward position
                                   // (the sweep gathers its weights from the table; no weight is laid out in lattice order)
  bool device_tables = false;      // stop after the layout: the engine builds arc_off / slot_pos and the transposition tables on
                                   // the device (lattice_gpu.hip gpu_tables_for_host_layout) -- the same bytes, a fraction of the time
  bool lane_fused = true;          // lay a corpus of lane lattices the tile sweep does not take out for the fused backward pass
                                   // (LatticeSet::lane_fused)
  uint32_t lane_chunks = 1;        // chunks per lane class (see LatticeSet::LaneClass); 1 = one launch per class (default:
                                   // measured on config 4, four chunks on four streams overlap their kernels but finish no
                                   // sooner -- the E-step is bound by its total HBM traffic -- and cost 46 us of extra tails)
};

// Builds every pair's lattice (parallel over pairs) and packs them.  Returns false + err on failure.
bool build_lattices(const HostWfst& w, const HostCorpus& c, const BuildOptions& opt, LatticeSet& out, std::string& err);

// launch classes, pieces and stream bases of the lane groups (shared by the host builder and the GPU builder)
// only_lanes: the corpus has no lattice outside the lane groups (what the tile sweep's layout requires)
uint64_t assign_lane_classes(LatticeSet& out, const BuildOptions& opt, bool only_lanes);

// single pair (exposed for tests)
void build_pair_lattice(const HostWfst& w, const uint32_t* in, uint32_t n_in, const uint32_t* out, uint32_t n_out,
                        bool prune, PairLattice& lat, bool& has_deriv);

}  // namespace carmel_hip
