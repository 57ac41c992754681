#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FUZZ_ONLY="${ONLY:-tile sweep}" timeout 1500 python tools/fuzz_gpu.py ${1:-7000} ${2:-150} > gpurun_out/r4_fuzz_ts.log 2>&1
echo "rc=$?" >> gpurun_out/r4_fuzz_ts.log
tail -16 gpurun_out/r4_fuzz_ts.log
