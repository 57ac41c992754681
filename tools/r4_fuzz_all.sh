#!/bin/bash
# round 4: the whole fuzzing campaign (every randomised parity test, seeds beyond the suite's)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 3000 python tools/fuzz_gpu.py ${1:-5000} ${2:-40} > gpurun_out/r4_fuzz_all.log 2>&1
echo "rc=$?" >> gpurun_out/r4_fuzz_all.log
tail -16 gpurun_out/r4_fuzz_all.log
