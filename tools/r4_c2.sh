#!/bin/bash
# round 4: config 2 under the A/B switches that change its launch structure
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  echo "== $*"
  ( export "$@" X_=1; timeout 600 python bench.py --config c2 --no-secondary --no-cpu-baseline --steps 200 --warmup 10 --full-out gpurun_out/c2_ab.json > gpurun_out/c2_ab.log 2>&1 )
  grep "^{\"metric" gpurun_out/c2_ab.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])" || tail -5 gpurun_out/c2_ab.log
}
run
run CARMEL_HIP_TRANSPOSE=0
run CARMEL_HIP_GRAPH=1
run CARMEL_HIP_TRANSPOSE=0 CARMEL_HIP_GRAPH=1
