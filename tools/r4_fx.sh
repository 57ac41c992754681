#!/bin/bash
# round 4: the device exact forest chain -- parity tests, then timing with the phase counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_forest_gpu.py tests/test_bench_workloads_gpu.py::test_c5_slice_forest_em_and_exact_chain_match_the_oracle tests/test_forest_cli_gpu.py -x -q -m gpu > gpurun_out/fx_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/fx_tests.log
tail -5 gpurun_out/fx_tests.log
CARMEL_HIP_FOREST_EXACT_CLK=1 timeout 600 python tools/fx_time.py 100000 3 > gpurun_out/fx_time.log 2>&1
cat gpurun_out/fx_time.log
