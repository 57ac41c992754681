#!/bin/bash
# round 4: the whole fuzzing campaign (every randomised parity test, seeds beyond the suite's) + the entry point's smoke test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_smoke.log 2>&1; tail -2 gpurun_out/r4_smoke.log
timeout 3000 python tools/fuzz_gpu.py ${1:-5000} ${2:-40} > gpurun_out/r4_fuzz_all.log 2>&1
echo "rc=$?" >> gpurun_out/r4_fuzz_all.log
tail -18 gpurun_out/r4_fuzz_all.log
