#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== workgroup kernel"
CARMEL_HIP_GIBBS_WORKGROUP=1 FUZZ_VERBOSE=1 FUZZ_ONLY="gibbs exact chain" timeout 600 python tools/fuzz_gpu.py 4000 60 2>&1 | tail -3
echo "== wave kernel, seeds one by one from 4030"
for s in 4036 4037 4038 4039 4040; do FUZZ_ONLY="gibbs exact chain" timeout 100 python tools/fuzz_gpu.py $s 1 2>&1 | tail -2; done
