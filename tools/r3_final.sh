#!/bin/bash
# round-3 closing pass: the whole GPU test suite, then tools/r3_measure.sh (full bench line, kernel stats, PMC traffic)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_final
mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -q -m gpu -x ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
bash tools/r3_measure.sh 2>&1 | tee $O/measure.log
