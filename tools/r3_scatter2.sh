#!/bin/bash
# scatter with per-item instead of run-length indices (c4)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_scatter2
mkdir -p $O
cd $R
bash tools/ab_env.sh c4 "X=1" "CARMEL_HIP_TRANS_RUNS=0 CARMEL_HIP_TRANS_SCATTER=3" "CARMEL_HIP_TRANS_RUNS=0 CARMEL_HIP_TRANS_SCATTER=0" "X=2" 2>&1 | tee $O/ab_c4.txt
bash tools/ab_env.sh amb "X=1" "CARMEL_HIP_TRANS_SCATTER=3" 2>&1 | tee $O/ab_amb.txt
