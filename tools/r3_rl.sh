#!/bin/bash
# run-length transposition indices: parity test, then c4 kernel stats with them off / on
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_rl
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "run_length or full_size" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for m in 0 1; do
  echo "== CARMEL_HIP_TRANS_RUNS=$m"
  export CARMEL_HIP_TRANS_RUNS=$m
  bash tools/kstats.sh c4 2>&1 | head -9
  cp $R/gpurun_out/c4_kernel_stats.csv $O/c4_kernel_stats_runs$m.csv
done
unset CARMEL_HIP_TRANS_RUNS
echo "== default, c2"
bash tools/kstats.sh c2 2>&1 | head -10
