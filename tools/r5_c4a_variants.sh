#!/bin/bash
# round 5: c4a under the switches of the fused-lane layout (per-kernel averages of each)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { echo "== $1"; shift; env "$@" bash tools/kstats.sh c4a 2>&1 | grep -E "^[0-9]|trans_|sweep_" | cut -c1-170; }
run "default (log sweep, XC, 16-row tiles)" X=1
run "run-length indices" CARMEL_HIP_TRANS_RUNS=1
run "8192-item buckets" CARMEL_HIP_LIB=$ROOT/build/ab_kb8/libcarmel_hip.so
run "8192-item buckets, 2 chunks" CARMEL_HIP_LIB=$ROOT/build/ab_kb8/libcarmel_hip.so CARMEL_HIP_LANE_CHUNKS=2
run "linear sweep" CARMEL_HIP_LANE_LINEAR=1
