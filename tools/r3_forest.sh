#!/bin/bash
# forest sampler: tests, bench (several-lanes sampler on / off), kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_forest
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_forest_gpu.py tests/test_forest_cli_gpu.py -q --durations=5 > $O/pytest_forest.log 2>&1; echo "pytest rc=$?" >> $O/pytest_forest.log
tail -15 $O/pytest_forest.log
timeout 1500 python3 -m pytest tests/test_bench_workloads_gpu.py -q -k "c5 or marginals_against_the_enumerated_stationary_distribution" --durations=5 > $O/pytest_bw.log 2>&1; echo "pytest rc=$?" >> $O/pytest_bw.log
tail -15 $O/pytest_bw.log
for m in 1 0; do
CARMEL_HIP_FOREST_MULTI=$m timeout 600 python3 bench.py --config c5 --no-cpu-baseline --no-secondary > $O/bench_c5_$m.json 2> $O/bench_c5_$m.err; echo "bench c5 multi=$m rc=$?"
python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench_c5_$m.json") if l.startswith("{")][-1])
print("c5 multi=$m ms/sweep %.4f frac %.4f exact %s" % (d["ms_per_step"], d["roofline"]["frac"], d.get("exact")))
PY
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kf && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-secondary --steps 200 > /tmp/kf.log 2>&1
f=$(find /tmp/kf -name '*kernel_stats.csv' | head -1); cp $f $O/c5_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("   %-70s calls %6s avg %9.1f us"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
