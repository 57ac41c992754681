#!/bin/bash
# kernel-level profile of bench_forest.py (config 5)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -- python3 $R/bench_forest.py --sweeps 60 --em-iters 3 > /tmp/fprof.log 2>&1
grep '^{' /tmp/fprof.log | tail -1
f=$(find /tmp/fprof -name '*kernel_stats.csv' | head -1)
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/forest_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'forest' in r['Name']: print("   %-40s calls %5s avg %9.1f us  total %9.1f ms"%(r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
