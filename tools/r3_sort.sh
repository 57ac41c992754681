#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_sort
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_forest_gpu.py -x -q > $O/pytest_forest.log 2>&1; tail -3 $O/pytest_forest.log
for v in 1 0; do
CARMEL_HIP_LANE_SORT=$v timeout 600 python3 bench.py --config c4a --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > $O/c4a_$v.json 2> $O/c4a_$v.err
python3 - <<PY
import json
d=json.loads([l for l in open("$O/c4a_$v.json") if l.startswith("{")][-1])
print("sort=$v", "ms/step %.3f kernel_ms %.3f frac %.4f build %.1fs" % (d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"], d["lattice_build_s"]))
PY
done
CARMEL_HIP_LANE_SORT=1 bash tools/kstats.sh c4a 2>&1 | head -9
