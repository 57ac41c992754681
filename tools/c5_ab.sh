#!/bin/bash
# c5 parallel sweep with environment switches: bash tools/c5_ab.sh "" "CARMEL_HIP_FOREST_COUNT=0" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "$@"; do
  ( for kv in $v; do export $kv; done
    timeout 600 python3 bench.py --config c5 --no-cpu-baseline --no-secondary --steps 400 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] c5 ms/sweep %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" )
done
