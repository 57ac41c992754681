#!/bin/bash
# A/B of library variants built by tools/build_variant.sh: bash tools/ab_variants.sh c4 "" _t8 _t8w4
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export CARMEL_HIP_LIB=$R/carmel_amd/libcarmel_hip$v.so
  rm -rf /tmp/prof$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$v -- python3 $R/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-exchange-loopback > /tmp/log$v 2>&1
  grep '^{' /tmp/log$v | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v] $CFG ms/step %.4f kernel_ms %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  f=$(find /tmp/prof$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'carmel' in r['Name'] and float(r['AverageNs']) > 15000 and int(r['Calls']) > 5: print("   %-60s %5s x %8.1f us"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
