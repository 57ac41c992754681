// host_api.cpp — host-only inspection entry points (no GPU needed): build the lattices exactly as
// carmel_hip_build_lattices does and hand the batched-CSR image back, so the layout can be checked against the
// oracle on machines without a device.  Nothing here computes forward/backward values.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include "../../include/carmel_hip.h"
#include "options.hpp"
#include "lattice.hpp"

using namespace carmel_hip;

struct carmel_hip_host_lattices {
  HostWfst w;
  HostCorpus c;
  LatticeSet L;
  std::string err;
};

extern "C" {

int carmel_hip_host_build(carmel_hip_host_lattices** out, uint32_t n_states, uint32_t final_state, uint64_t n_arcs,
                          const uint32_t* src, const uint32_t* dst, const uint32_t* in_sym, const uint32_t* out_sym,
                          uint64_t n_pairs, const uint64_t* in_off, const uint32_t* cin, const uint64_t* out_off,
                          const uint32_t* cout, const double* pair_weight, int prune, int threads,
                          uint32_t small_pairs, uint32_t small_states, int lane_states) {
  if (!out) return CARMEL_HIP_ERR_ARG;
  carmel_hip_host_lattices* h = new carmel_hip_host_lattices();
  h->w.n_states = n_states;
  h->w.final_state = final_state;
  h->w.n_arcs = n_arcs;
  h->w.src.assign(src, src + n_arcs);
  h->w.dst.assign(dst, dst + n_arcs);
  h->w.in.assign(in_sym, in_sym + n_arcs);
  h->w.out.assign(out_sym, out_sym + n_arcs);
  h->w.build_index();
  h->c.n_pairs = n_pairs;
  h->c.in_off.assign(in_off, in_off + n_pairs + 1);
  h->c.out_off.assign(out_off, out_off + n_pairs + 1);
  h->c.in_sym.assign(cin, cin + in_off[n_pairs]);
  h->c.out_sym.assign(cout, cout + out_off[n_pairs]);
  if (pair_weight)
    h->c.weight.assign(pair_weight, pair_weight + n_pairs);
  else
    h->c.weight.assign(n_pairs, 1.0);
  BuildOptions opt;
  opt.prune = prune != 0;
  opt.threads = threads;
  if (small_pairs) opt.small_pairs = small_pairs;
  if (small_states) opt.small_states = small_states;
  if (lane_states >= 0) opt.lane_states = (uint32_t)lane_states;
  if (const char* e = lib_opt("lane_window")) opt.lane_window = (uint32_t)std::max(0, atoi(e));  // as engine.cpp
  if (const char* e = lib_opt("tile_sweep")) opt.tile_sweep = atoi(e) != 0;
  if (const char* e = lib_opt("lane_fused")) opt.lane_fused = atoi(e) != 0;
  if (const char* e = lib_opt("lane_window_min")) opt.lane_window_min = (uint32_t)std::max(0, atoi(e));
  // lane_states = 0 asks for the plain inspection form -- every lattice a bundle (with small_pairs = 1: one lattice each),
  // what the front end's --fem-forest export walks -- unless a test forces the one-per-wavefront layout explicitly
  if (lane_states == 0 && !lib_opt("wave_min_width")) opt.wave = false;
  if (const char* e = lib_opt("wave_ring")) opt.wave_ring = atoi(e) != 0;
  if (const char* e = lib_opt("wave_min_width")) opt.wave_min_width = opt.wave_lane_min_width = atof(e);
  if (!build_lattices(h->w, h->c, opt, h->L, h->err)) {
    delete h;
    return CARMEL_HIP_ERR_ARG;
  }
  *out = h;
  return CARMEL_HIP_OK;
}

// dims[0..15] = n_bundles, n_off (states + bundles), n_arcs, n_level_off, n_pair_slots, n_classes, n_kept,
//              n_cyclic, explored_states, explored_arcs, n_lane_groups, n_lane_records, n_lane_slots,
//              n_lane_classes, total_states, total_arcs, last pair: explored states, kept states, kept arcs
void carmel_hip_host_dims(carmel_hip_host_lattices* h, uint64_t* dims) {
  dims[0] = h->L.bundles.size();
  dims[1] = h->L.in_off.size();
  dims[2] = h->L.in_arcs.size();
  dims[3] = h->L.level_off.size();
  dims[4] = h->L.pair_id.size();
  dims[5] = h->L.classes.size();
  dims[6] = h->L.n_kept;
  dims[7] = h->L.n_cyclic;
  dims[8] = h->L.explored_states;
  dims[9] = h->L.explored_arcs;
  dims[10] = h->L.lane_groups.size();
  dims[11] = h->L.lane_fwd.size();
  dims[12] = h->L.lane_pair.size();
  dims[13] = h->L.lane_classes.size();
  dims[14] = h->L.total_states;
  dims[15] = h->L.total_arcs;
  dims[16] = h->L.last_pre_states;
  dims[17] = h->L.last_post_states;
  dims[18] = h->L.last_post_arcs;
}

// lane groups (see LaneGroup in lattice.hpp): groups as raw 32-byte records, streams as (x, y) u32 pairs
void carmel_hip_host_export_lanes(carmel_hip_host_lattices* h, void* groups32, uint32_t* fwd, uint32_t* bwd,
                                  uint32_t* lane_pair, uint32_t* lane_nstates, double* lane_logw, uint32_t* classes3) {
  const LatticeSet& L = h->L;
  std::memcpy(groups32, L.lane_groups.data(), L.lane_groups.size() * sizeof(LaneGroup));
  std::memcpy(fwd, L.lane_fwd.data(), L.lane_fwd.size() * sizeof(uint2_t));
  std::memcpy(bwd, L.lane_bwd.data(), L.lane_bwd.size() * sizeof(uint2_t));
  std::memcpy(lane_pair, L.lane_pair.data(), L.lane_pair.size() * 4);
  std::memcpy(lane_nstates, L.lane_nstates.data(), L.lane_nstates.size() * 4);
  std::memcpy(lane_logw, L.lane_logw.data(), L.lane_logw.size() * 8);
  for (size_t k = 0; k < L.lane_classes.size(); ++k) {
    classes3[3 * k + 0] = L.lane_classes[k].first;
    classes3[3 * k + 1] = L.lane_classes[k].count;
    classes3[3 * k + 2] = L.lane_classes[k].max_states;
  }
}

// blocked transposition tables (see TransBucket in lattice.hpp).  dims6 = n_items, n_buckets, n_tiles, n_split_arcs,
// n_post, n_arcs; a null pointer skips that array.  buckets24 = raw 24-byte TransBucket records.
void carmel_hip_host_transpose(carmel_hip_host_lattices* h, uint64_t* dims6, void* buckets24, uint64_t* tile_base,
                               uint16_t* b_arc, uint16_t* b_rank, uint32_t* b_src, uint16_t* t_pos, uint32_t* t_src,
                               uint32_t* split_arcs, uint64_t* arc_off, uint64_t* slot_pos) {
  const LatticeSet& L = h->L;
  if (dims6) {
    dims6[0] = L.slot_pos.size();
    dims6[1] = L.t_buckets.size();
    dims6[2] = L.t_tile_base.empty() ? 0 : L.t_tile_base.size() - 1;
    dims6[3] = L.t_split_arcs.size();
    dims6[4] = L.n_post;
    dims6[5] = L.arc_off.empty() ? 0 : L.arc_off.size() - 1;
  }
  if (buckets24) std::memcpy(buckets24, L.t_buckets.data(), L.t_buckets.size() * sizeof(TransBucket));
  if (tile_base) std::memcpy(tile_base, L.t_tile_base.data(), L.t_tile_base.size() * 8);
  if (b_arc) std::memcpy(b_arc, L.t_b_arc.data(), L.t_b_arc.size() * 2);
  if (b_rank) std::memcpy(b_rank, L.t_b_rank.data(), L.t_b_rank.size() * 2);
  if (b_src) std::memcpy(b_src, L.t_b_src.data(), L.t_b_src.size() * 4);
  if (t_pos) std::memcpy(t_pos, L.t_t_pos.data(), L.t_t_pos.size() * 2);
  if (t_src) std::memcpy(t_src, L.t_t_src.data(), L.t_t_src.size() * 4);
  if (split_arcs) std::memcpy(split_arcs, L.t_split_arcs.data(), L.t_split_arcs.size() * 4);
  if (arc_off) std::memcpy(arc_off, L.arc_off.data(), L.arc_off.size() * 8);
  if (slot_pos) std::memcpy(slot_pos, L.slot_pos.data(), L.slot_pos.size() * 8);
}

void carmel_hip_host_tile_sweep(carmel_hip_host_lattices* h, uint32_t* info2, uint32_t* tile_group) {
  const LatticeSet& L = h->L;
  if (info2) {
    info2[0] = L.tile;
    info2[1] = L.tile_sweep ? (uint32_t)L.tile_group.size() : 0u;
  }
  if (tile_group && L.tile_sweep) std::memcpy(tile_group, L.tile_group.data(), L.tile_group.size() * 4);
}

void carmel_hip_host_export(carmel_hip_host_lattices* h, void* bundles64, uint32_t* in_arcs, uint32_t* out_arcs,
                            uint32_t* in_off, uint32_t* out_off, uint32_t* level_off, uint32_t* pair_start,
                            uint32_t* pair_final, uint32_t* pair_id, double* pair_logw, uint32_t* classes5,
                            uint8_t* has_deriv) {
  const LatticeSet& L = h->L;
  std::memcpy(bundles64, L.bundles.data(), L.bundles.size() * sizeof(BundleDesc));
  std::memcpy(in_arcs, L.in_arcs.data(), L.in_arcs.size() * sizeof(uint2_t));
  std::memcpy(out_arcs, L.out_arcs.data(), L.out_arcs.size() * sizeof(uint2_t));
  std::memcpy(in_off, L.in_off.data(), L.in_off.size() * 4);
  std::memcpy(out_off, L.out_off.data(), L.out_off.size() * 4);
  std::memcpy(level_off, L.level_off.data(), L.level_off.size() * 4);
  std::memcpy(pair_start, L.pair_start.data(), L.pair_start.size() * 4);
  std::memcpy(pair_final, L.pair_final.data(), L.pair_final.size() * 4);
  std::memcpy(pair_id, L.pair_id.data(), L.pair_id.size() * 4);
  std::memcpy(pair_logw, L.pair_logw.data(), L.pair_logw.size() * 8);
  for (size_t k = 0; k < L.classes.size(); ++k) {
    classes5[5 * k + 0] = L.classes[k].first;
    classes5[5 * k + 1] = L.classes[k].count;
    classes5[5 * k + 2] = L.classes[k].block;
    classes5[5 * k + 3] = L.classes[k].max_states;
    classes5[5 * k + 4] = L.classes[k].serial ? 1u : 0u;
  }
  if (has_deriv) std::memcpy(has_deriv, L.has_deriv.data(), L.has_deriv.size());
}

// one-per-wavefront lattices (WaveDesc, lattice.hpp).  dims6 = n_waves, forward records, backward records, level
// entries, n_classes, wave_slot_base; null pointers are skipped.  descs64 = raw 64-byte WaveDesc records; fwd as (x, y)
// u32 pairs; bwd_arc = the WFST arc of every backward record (0xffffffff: padding); classes4 = first, count, max_states,
// max_width per launch class.
void carmel_hip_host_export_waves(carmel_hip_host_lattices* h, uint64_t* dims6, void* descs64, uint32_t* fwd, uint32_t* bwd,
                                  uint32_t* bwd_arc, uint32_t* level_off, uint32_t* frow, uint32_t* brow, uint32_t* classes4) {
  const LatticeSet& L = h->L;
  if (dims6) {
    dims6[0] = L.waves.size();
    dims6[1] = L.wave_fwd.size();
    dims6[2] = L.wave_bwd.size();
    dims6[3] = L.wave_level_off.size();
    dims6[4] = L.wave_classes.size();
    dims6[5] = L.wave_slot_base;
  }
  if (descs64) std::memcpy(descs64, L.waves.data(), L.waves.size() * sizeof(WaveDesc));
  if (fwd) std::memcpy(fwd, L.wave_fwd.data(), L.wave_fwd.size() * sizeof(uint2_t));
  if (bwd) std::memcpy(bwd, L.wave_bwd.data(), L.wave_bwd.size() * 4);
  if (bwd_arc) std::memcpy(bwd_arc, L.wave_bwd_arc.data(), L.wave_bwd_arc.size() * 4);
  if (level_off) std::memcpy(level_off, L.wave_level_off.data(), L.wave_level_off.size() * 4);
  if (frow) std::memcpy(frow, L.wave_frow.data(), L.wave_frow.size() * 4);
  if (brow) std::memcpy(brow, L.wave_brow.data(), L.wave_brow.size() * 4);
  if (classes4)
    for (size_t k = 0; k < L.wave_classes.size(); ++k) {
      classes4[4 * k + 0] = L.wave_classes[k].first;
      classes4[4 * k + 1] = L.wave_classes[k].count;
      classes4[4 * k + 2] = L.wave_classes[k].max_states;
      classes4[4 * k + 3] = L.wave_classes[k].max_width | (L.wave_classes[k].ring << 16);
    }
}

void carmel_hip_host_free(carmel_hip_host_lattices* h) { delete h; }

}  // extern "C"
