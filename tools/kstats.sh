#!/bin/bash
# per-kernel averages of one bench.py configuration: bash tools/kstats.sh c2
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-exchange-loopback --full-out $R/gpurun_out/${CFG}_bench_kstats.json > /tmp/ks.log 2>&1
grep '^{' /tmp/ks.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['config']['workload'])"
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/${CFG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('carmel_hip::', '').replace('void ', '')
    print("   %-100s calls %5s avg %9.1f us"%(n[:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
