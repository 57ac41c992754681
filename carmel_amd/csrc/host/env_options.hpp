// env_options.hpp — the front ends read the environment, the library does not (include/carmel_hip.h, carmel_hip_set_option):
// CARMEL_HIP_<KEY>=value becomes the option "<key>", CARMEL_TIMING the option "timing".  Called once, before the first library call.
#pragma once
#include <cctype>
#include <cstring>
#include <iostream>
#include <string>
#include "../../../include/carmel_hip.h"
extern char** environ;
namespace carmel_host {
inline void import_env_options() {
  for (char** e = environ; e && *e; ++e) {
    const char* s = *e;
    const char* eq = std::strchr(s, '=');
    if (!eq) continue;
    std::string name(s, eq - s), key;
    if (name == "CARMEL_TIMING")
      key = "timing";
    else if (name.compare(0, 11, "CARMEL_HIP_") == 0 && name != "CARMEL_HIP_LIB")
      for (size_t i = 11; i < name.size(); ++i) key += (char)std::tolower((unsigned char)name[i]);
    if (key.empty()) continue;
    if (carmel_hip_set_option(key.c_str(), eq + 1) != CARMEL_HIP_OK)
      std::cerr << "warning: " << name << " names no option of libcarmel_hip (carmel_hip_option_name lists them); ignored\n";
  }
}
}  // namespace carmel_host
