// options.cpp — see options.hpp
#include "options.hpp"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include "../../include/carmel_hip.h"

namespace carmel_hip {
namespace {
// every key the library reads: what it switches is documented where it is read (grep for lib_opt and the key) and in DESIGN.md section 7
const char* const kKeys[] = {
    "timing",  // phase times of the lattice build / layout and the samplers on stderr (the front ends' CARMEL_TIMING)
    // E-step layouts and formulations (engine.cpp, lattice_gpu.hip, host_api.cpp)
    "lane_states", "tile_sweep", "lane_fused", "transpose", "lane_chunks", "lane_window", "lane_window_min", "wave_ring", "wave_min_width",
    "gpu_build", "wave_gather", "wave_xc", "tile_gather", "device_tables", "trans_scatter", "trans_runs", "lane_fused_kernel",
    "tile_sweep_kernel", "mailbox", "poison", "lane_trace", "exchange_sparse",
    // one-tape cascades (engine_unrolled.cpp, unrolled.cpp)
    "dense", "unrolled", "unrolled_ragged",
    // the samplers (gibbs.hip, forest.hip)
    "gibbs_chains", "gibbs_own_cap", "gibbs_workgroup", "gibbs_clk", "gibbs_reg", "gibbs_lane", "forest_sweep", "forest_ldswalk", "forest_multi",
    "forest_nohash", "forest_trace", "forest_exact_host", "forest_exact_clk", "forest_logdomain", "forest_gcol", "forest_gather"};
constexpr int kN = (int)(sizeof kKeys / sizeof kKeys[0]);
const char* volatile g_val[kN];  // interned strings (a replaced value is not freed: a reader may still hold it)
std::mutex g_mu;
int find(const char* key) {
  for (int i = 0; i < kN; ++i)
    if (!std::strcmp(kKeys[i], key)) return i;
  return -1;
}
}  // namespace
const char* lib_opt(const char* key) {
  const int i = find(key);
  return i < 0 ? nullptr : g_val[i];
}
bool lib_opt_off(const char* key) {
  const char* v = lib_opt(key);
  return v && std::atoi(v) == 0;
}
}  // namespace carmel_hip

extern "C" {
int carmel_hip_set_option(const char* key, const char* value) {
  using namespace carmel_hip;
  if (!key) return CARMEL_HIP_ERR_ARG;
  const int i = find(key);
  if (i < 0) return CARMEL_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!value)
    g_val[i] = nullptr;
  else {
    char* c = (char*)std::malloc(std::strlen(value) + 1);
    if (!c) return CARMEL_HIP_ERR_HIP;
    std::strcpy(c, value);
    g_val[i] = c;
  }
  return CARMEL_HIP_OK;
}
const char* carmel_hip_get_option(const char* key) { return key ? carmel_hip::lib_opt(key) : nullptr; }
int carmel_hip_option_count(void) { return carmel_hip::kN; }
const char* carmel_hip_option_name(int i) { return i >= 0 && i < carmel_hip::kN ? carmel_hip::kKeys[i] : nullptr; }
}
