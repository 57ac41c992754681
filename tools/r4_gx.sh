#!/bin/bash
# round 4: the lattice sampler's exact chain on one wavefront -- parity tests, then the crp workload
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gibbs_gpu.py tests/test_cli_gpu.py tests/test_multirank_gpu.py::test_bench_two_ranks_strong_scaling_is_the_one_rank_run -x -q -m gpu -k "gibbs or crp or bench_two" --durations=8 > gpurun_out/gx_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/gx_tests.log
tail -25 gpurun_out/gx_tests.log
CARMEL_HIP_GIBBS_CLK=1 timeout 900 python bench.py --config crp --no-secondary --full-out gpurun_out/bench_crp_full.json > gpurun_out/bench_crp.log 2>&1
grep '^{"metric' gpurun_out/bench_crp.log | tail -1
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_crp_full.json"))
print(d["exact"])
PY
