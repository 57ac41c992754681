#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gibbs_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "gibbs or crp" 2>&1 | tail -3
G=tests/golden
CARMEL_HIP_GIBBS_CLK=1 CARMEL_TIMING=1 CARMEL_TRAINED_DIR=/tmp timeout 300 carmel_amd/bin/carmel --crp -M 200 -R 7 $G/tagging.data $G/tagging.fsa $G/tagging.fst 2>&1 | grep -E "carmel_hip\]|timing: gibbs"
python - <<'PY' > gpurun_out/dbg_cli.log 2>&1
import sys, os, pathlib, tempfile, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_cli_gpu as C
for name, fn, seed in (("one-tape", C.test_random_one_tape_cascades, 5038),):
    with tempfile.TemporaryDirectory() as d:
        try:
            fn(pathlib.Path(d), seed)
            print(name, seed, "ok")
        except BaseException as e:
            print(name, seed, "FAIL")
            traceback.print_exc()
PY
tail -c 3000 gpurun_out/dbg_cli.log
