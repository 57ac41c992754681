#!/bin/bash
# c5 as a secondary behind other workloads (bench.py's default run) against c5 on its own, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
a=$(timeout 900 python3 bench.py --no-cpu-baseline --secondary c2,c3,c5 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['secondary']['c5']['ms_per_step'])")
b=$(timeout 600 python3 bench.py --config c5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
echo "c5 ms per sweep: behind c4, c2, c3 in one process $a, on its own $b"
