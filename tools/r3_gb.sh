#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_gb
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_lattice_gpu.py -x -q --durations=6 > $O/pytest.log 2>&1; tail -30 $O/pytest.log
for CFG in amb c4a; do
CARMEL_TIMING=1 timeout 900 python3 bench.py --config $CFG --no-secondary --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_$CFG.json 2> $O/bench_$CFG.err; echo "bench $CFG rc=$?"
grep "gpu lattice build\|lattices built\|lattice build:" $O/bench_$CFG.err | head -20
python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench_$CFG.json") if l.startswith("{")][-1])
print("$CFG", "ms/step %.3f frac %.4f build %.3fs" % (d["ms_per_step"], d["roofline"]["frac"], d["lattice_build_s"]))
PY
done
