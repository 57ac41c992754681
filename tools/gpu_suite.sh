#!/bin/bash
# the whole GPU suite as the driver runs it, with the slowest tests listed (gpurun_out/gpu_suite.log)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=45 ) > gpurun_out/gpu_suite.log 2>&1
echo "rc=$?" >> gpurun_out/gpu_suite.log
tail -70 gpurun_out/gpu_suite.log
