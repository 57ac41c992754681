#!/bin/bash
# exchange: multi-rank tests + fem export test + bench loopback line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_xch
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_multirank_gpu.py tests/test_cli_gpu.py::test_fem_export_bridges_to_forest_em -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 900 python3 bench.py --no-secondary --no-cpu-baseline --steps 10 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench rc=$?"; tail -3 $O/bench_c4.err
python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench_c4.json") if l.startswith("{")][-1])
print("c4", d["ms_per_step"], d["roofline"]["frac"], d.get("exchange"))
PY
