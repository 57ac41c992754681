#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_lattice_gpu.py -x -q -m gpu 2>&1 | tail -3
FUZZ_ONLY="tile sweep" timeout 600 python tools/fuzz_gpu.py 9000 120 2>&1 | tail -3
bash tools/kstats.sh c4 2>&1 | grep -E "^[0-9]|trans_|tile_sweep"
