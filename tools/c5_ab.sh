#!/bin/bash
# config 5: parity tests of the forest sampler, the bench line (100 sweeps), per-wave phases and the kernel timeline of a sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-c5ab}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_forest_gpu.py tests/test_forest_cli_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
timeout 300 python bench.py --config c5 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
CARMEL_HIP_FOREST_TRACE=/tmp/ft.bin timeout 300 python bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/forest_trace.py /tmp/ft.bin > $O/trace.txt 2>&1
bash tools/forest_class_times.sh > $O/class_times.txt 2>&1
