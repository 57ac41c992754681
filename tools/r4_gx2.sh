#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
G=tests/golden
CARMEL_HIP_GIBBS_CLK=1 CARMEL_TIMING=1 CARMEL_TRAINED_DIR=/tmp timeout 300 carmel_amd/bin/carmel --crp -M 200 -R 7 $G/tagging.data $G/tagging.fsa $G/tagging.fst 2>&1 | grep -E "carmel_hip\]|timing: gibbs" 
