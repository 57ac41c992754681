#!/bin/bash
# per-launch durations of the forest sampler's kernels in one sweep (config 5): which launch class is the critical path?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fct
rocprofv3 --kernel-trace --output-format csv -d /tmp/fct -- python3 $R/bench_forest.py --sweeps 12 --em-iters 1 > /tmp/fct.log 2>&1
f=$(find /tmp/fct -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'forest' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = None
# the last sweep: from the last proposal kernel on
props = [int(r['Start_Timestamp']) for r in rows if 'proposal' in r['Kernel_Name']]
if len(props) > 2:
    d = sorted(b - a for a, b in zip(props, props[1:]))
    print("sweep period (proposal start to proposal start): median %.1f us, min %.1f us over %d sweeps" % (d[len(d) // 2] / 1e3, d[0] / 1e3, len(d)))
idx = max(i for i, r in enumerate(rows) if 'proposal' in r['Kernel_Name'])
base = int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']) - base, int(r['End_Timestamp']) - base
    print("%-46s grid %7s  start %8.1f us  end %8.1f us  (%7.1f us)" % (r['Kernel_Name'].split('(')[0][-46:], r.get('Grid_Size', r.get('Grid_Size_X', '?')), s / 1e3, e / 1e3, (e - s) / 1e3))
PY
