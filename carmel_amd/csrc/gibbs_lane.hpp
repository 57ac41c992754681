// gibbs_lane.hpp — `carmel --crp --crp-parallel`: the stale-count sweep with ONE BLOCK PER LANE (gibbs_lane.hip).
// The wavefront-per-block kernels of gibbs_exact.hip issue every instruction for the five arcs of a lattice level: 8 % of the
// lanes work, and the sweep is bound by instruction issue (profiles/r6_crp_phases.txt).  Here 64 trellis lattices share a
// wavefront the way the EM path's lane groups do (lattice.hpp): each lane streams its own lattice's arcs, in the order of the
// backward sweep, out of record streams interleaved row by row (every load of a row is coalesced), keeps two levels of backward
// values in its own LDS column, and walks its own path afterwards -- no LDS atomics, no barriers between levels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "gibbs_exact.hpp"
#include "lattice.hpp"

namespace carmel_hip {

#define GL_NONE 0xffffffffu
// GlRecB::ctrl
#define GL_DST(c) ((c) & 0xffu)          // the destination's place in its level
#define GL_SRC(c) (((c) >> 8) & 0xffu)   // the source's place in its level
#define GL_VALID 0x10000u
#define GL_STATE_LAST 0x20000u           // the last arc of its source state (in list order: the walk's order of subtraction)
#define GL_LEVEL_LAST 0x40000u           // the last arc of its level
#define GL_HAS0 0x80000u                 // the arc has a first / second parameter ...
#define GL_FIX0 0x100000u                // ... of fixed probability (no norm group: its prior is its probability)
#define GL_HAS1 0x200000u
#define GL_FIX1 0x400000u
#define GL_NPAR(c) ((((c) >> 19) & 1u) + (((c) >> 21) & 1u))

struct GlGroup {  // 64 blocks, one per lane
  uint64_t rec_base;    // row r of lane i: rec_base + r * 64 + i  (recA, recB, arc_id, the scratch of shares)
  uint64_t samp_base;   // path entry k of lane i: samp_base + k * 64 + i
  uint32_t rows, path;  // the most arcs / path arcs of a lane
  uint32_t pad[2];
};
struct GlLane {  // a lane of a group
  uint32_t block;    // its block in corpus order (GL_NONE: an empty lane)
  uint32_t n_arcs;
  uint32_t start;    // the start state's first row << 8 | its out-degree
  uint32_t path;     // arcs of a path: levels - 1
  double wt;         // the pair's weight
  uint64_t sample_off;  // where the block's sample lives in the sampler's own buffers (gibbs.hip)
};

struct GlClass {  // a launch: groups [first, first + count) and their LDS need per lane
  uint32_t first = 0, count = 0;
  uint32_t W = 0, LP = 0, LN = 0;  // widest level, most local parameters / norm groups of a block (LP, LN multiples of 4)
};

struct GlHost {  // the layout, built once per sampler (host)
  std::vector<GlGroup> groups;
  std::vector<GlLane> lanes;
  std::vector<uint32_t> recA, recB;  // four words per record
  std::vector<uint32_t> arc_id;
  std::vector<GlClass> classes;
  std::vector<uint8_t> taken;        // per block: laid out here
  uint64_t n_rec = 0, n_samp = 0;
};
// which blocks go one per lane (trellis lattices whose levels, local parameters and path fit the LDS budget) and their streams
void gibbs_lane_build(const LatticeSet& L, const std::vector<uint32_t>& block_bundle, const std::vector<GxBlock>& gb,
                      const std::vector<uint64_t>& chain_off, const std::vector<uint32_t>& chain_param,
                      const std::vector<uint32_t>& p_norm, GlHost& out);

struct GlArgs {
  const GlGroup* groups;
  const GlLane* lanes;
  const uint4* recA;       // {parameter 0, parameter 1, norm group 0, norm group 1} (global numbers; GL_NONE: none)
  const uint4* recB;       // {ctrl, local parameter 0 | 1 << 16, local norm group 0 | 1 << 16 (0xffff: nothing to correct), the
                           //  destination's first row << 8 | its out-degree}
  const uint32_t* arc_id;  // the composed arc (read on the sweeps that sample from --init-em weights only)
  double2* sw;             // scratch per row: {the arc's share of its state's total; at a state's LAST row: that total}
  double* wq;              // ... and the arc's proposal weight (read for the chosen arcs only)
  const uint4* samp_old;   // the previous sweep's paths: {row, recB.y, recB.z, place of its first parameter in the block's sample}
  uint4* samp_new;
  const double* p_x;       // the snapshot the sweep samples against
  const double* normsum;
  const double* p_prior;
  const double* init_logw;
  double* iter_out;
  unsigned long long* phase_clk;
  uint64_t seed;
  uint32_t iter;
  uint32_t first_group;
  uint32_t W, LP, LN;
  int have_old;            // take the block's previous sample out of the counts (it exists, and no --include-self)
};
size_t gibbs_lane_lds_bytes(uint32_t W, uint32_t LP, uint32_t LN);
hipError_t launch_gibbs_lane(const GlArgs& A, uint32_t n_groups, hipStream_t s);
// counts of the new paths: new_x / new_norm (set to the priors by the caller) += weight per use, through per-workgroup LDS tables
hipError_t launch_gibbs_lane_recount(const GlGroup* groups, const GlLane* lanes, const uint4* recA, const uint4* samp, uint32_t n_groups,
                                     double* new_x, double* new_norm, hipStream_t s);
// the paths in the sampler's own format: parameter ids and norm groups in chain order at the block's place, its length
hipError_t launch_gibbs_lane_materialize(const GlGroup* groups, const GlLane* lanes, const uint4* recA, const uint4* samp, uint32_t n_groups,
                                         uint32_t* ids, uint32_t* nrm, uint32_t* len, hipStream_t s);

}  // namespace carmel_hip
