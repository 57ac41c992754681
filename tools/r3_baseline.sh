#!/bin/bash
# round-3 first GPU pass: the new parity tests, the full bench line (headline + secondaries), kernel stats of c4a / long
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_base
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_bench_workloads_gpu.py -x -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
tail -15 $O/pytest_new.log
timeout 1500 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench_full.json") if l.startswith("{")][-1])
print("c4", d["ms_per_step"], d["roofline"]["frac"], d.get("parity_checked_pairs"), d["bench_wall_s"])
for k,v in d.get("secondary",{}).items():
    print(k, v.get("error") or (v["ms_per_step"], v["roofline"]["frac"], v.get("parity_checked_pairs"), v.get("lattice_build_s"), v.get("exact")), v["wall_s"])
PY
for CFG in c4a long; do
  bash tools/kstats.sh $CFG 2>&1 | tail -25
  cp $R/gpurun_out/${CFG}_kernel_stats.csv $O/ 2>/dev/null
done
