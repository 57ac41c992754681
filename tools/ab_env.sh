#!/bin/bash
# A/B of environment switches: bash tools/ab_env.sh c4 "" "CARMEL_HIP_TRANS_PIPE=1" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
n=0
for v in "$@"; do
  n=$((n+1))
  rm -rf /tmp/prof$n
  ( for kv in $v; do export $kv; done
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$n -- python3 $R/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-exchange-loopback > /tmp/log$n 2>&1 )
  grep '^{' /tmp/log$n | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] $CFG ms/step %.4f kernel_ms %.4f frac %.4f lnp %r' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('ln_corpus_prob')))" || tail -5 /tmp/log$n
  f=$(find /tmp/prof$n -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'carmel' in r['Name'] and float(r['AverageNs']) > 15000 and int(r['Calls']) > 5: print("   %-60s %5s x %8.1f us"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
