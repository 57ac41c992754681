#!/bin/bash
# the driver's bench command, with the complete objects kept beside the compact line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python bench.py --full-out gpurun_out/bench_full.json ) > gpurun_out/bench_stdout.log 2> gpurun_out/bench_stderr.log
grep '^{"metric' gpurun_out/bench_stdout.log | tail -1 > gpurun_out/bench_line.json
cat gpurun_out/bench_line.json
tail -4 gpurun_out/bench_stderr.log
