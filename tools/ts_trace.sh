#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
CARMEL_HIP_LANE_TRACE=$R/gpurun_out/ts_trace.bin python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-exchange-loopback --full-out gpurun_out/ts_trace_bench.json | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'])"
python tools/ts_trace.py gpurun_out/ts_trace.bin
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
