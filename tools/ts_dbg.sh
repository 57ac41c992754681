#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
CARMEL_TIMING=1 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback --full-out gpurun_out/x.json 2>&1 | grep -i "tile sweep\|single"
bash tools/ts_trace.sh
