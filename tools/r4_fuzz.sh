#!/bin/bash
# round 4: fuzzing the new chains (forest_exact.hip, gibbs_exact.hip)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FUZZ_VERBOSE=1 FUZZ_ONLY="${1:-exact chain}" timeout 1500 python tools/fuzz_gpu.py ${2:-4000} ${3:-150} > gpurun_out/r4_fuzz.log 2>&1
tail -6 gpurun_out/r4_fuzz.log
