#!/bin/bash
# A/B of transposition tile sizes: builds named libcarmel_hip_k<K>.so (-DTRANS_K=K) side by side with the default
# usage (through gpurun): bash tools/ab_trans_k.sh "" _k10 _k8
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export CARMEL_HIP_LIB=$R/carmel_amd/libcarmel_hip$v.so
  rm -rf /tmp/prof$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/log$v 2>&1
  grep '^{' /tmp/log$v | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v]', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  f=$(find /tmp/prof$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'carmel' in r['Name']: print("   %-50s %8.1f us"%(r['Name'][:50], float(r['AverageNs'])/1e3))
PY
done
