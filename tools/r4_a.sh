#!/bin/bash
# round 4: the crp pin (CLI), communicator lifetime, bench --config crp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cli_gpu.py::test_crp_tagging_bookkeeping_against_the_reference_output tests/test_multirank_gpu.py::test_communicator_and_trainer_may_go_in_either_order -x -q -m gpu > gpurun_out/a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/a_tests.log
tail -15 gpurun_out/a_tests.log
timeout 900 python bench.py --config crp --no-secondary --full-out gpurun_out/bench_crp_full.json > gpurun_out/bench_crp.log 2>&1
tail -3 gpurun_out/bench_crp.log
