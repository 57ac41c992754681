#!/bin/bash
# round 4, the records of the final tree, second half (the PMC files of the first half committed under profiles/): kernel
# statistics of the two tile-sweep workloads, the default bench run, the phase stamps of the tile sweep, the whole GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for c in c4 c2; do bash tools/kstats.sh $c 2>&1 | head -6 | cut -c1-150; done
bash tools/bench_full.sh 2>&1 | tail -6
bash tools/ts_trace.sh 2>&1 | head -12 > gpurun_out/ts_trace.txt; cat gpurun_out/ts_trace.txt
bash tools/gpu_suite.sh 2>&1 | tail -8
