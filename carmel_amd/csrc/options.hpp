// options.hpp — the library's switches (carmel_hip_set_option, include/carmel_hip.h).  Formulation choices that leave results the
// same (A/B: a test holds the two forms together), layout limits the tests force onto small cases, traces.  Until round 5 these
// were ~70 getenv reads spread over the sources; a library takes its options through its ABI.  The ENVIRONMENT is read by the
// front ends only (carmel, forest-em, bench.py: every CARMEL_HIP_<KEY> becomes set_option("<key>"), CARMEL_TIMING "timing").
#pragma once
namespace carmel_hip {
// key: the lower-case name (e.g. "tile_sweep").  nullptr when unset -- what getenv() returned
const char* lib_opt(const char* key);
inline bool lib_opt_set(const char* key) { return lib_opt(key) != nullptr; }
// set and equal to zero ("0"): the usual way to switch a default formulation off
bool lib_opt_off(const char* key);
}  // namespace carmel_hip
