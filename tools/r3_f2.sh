#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_f2
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_forest_gpu.py -q > $O/pytest_forest.log 2>&1; tail -3 $O/pytest_forest.log
timeout 900 python3 -m pytest tests/test_bench_workloads_gpu.py -q -k "marginals_against" > $O/pytest_bw.log 2>&1; tail -12 $O/pytest_bw.log
