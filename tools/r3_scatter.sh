#!/bin/bash
# A/B of TransArgs::scatter (which pass of a transposition direction does the random access): parity test, then kernel
# times of c4 / c4a / long / c2 under CARMEL_HIP_TRANS_SCATTER = 0 .. 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_scatter
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scattering_first_pass" > $O/test.log 2>&1
tail -3 $O/test.log
for cfg in c4 c4a long c2; do
  bash tools/ab_env.sh $cfg "CARMEL_HIP_TRANS_SCATTER=0" "CARMEL_HIP_TRANS_SCATTER=1" "CARMEL_HIP_TRANS_SCATTER=2" "CARMEL_HIP_TRANS_SCATTER=3" 2>&1 | tee $O/ab_$cfg.txt
done
