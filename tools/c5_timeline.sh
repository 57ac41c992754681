#!/bin/bash
# kernel timeline of one parallel sweep of config 5 (the last sweep of a short bench run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5t
rocprofv3 --kernel-trace --output-format csv -d /tmp/c5t -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-secondary --steps 40 --warmup 5 > /tmp/c5t.log 2>&1
f=$(find /tmp/c5t -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ms = [i for i, r in enumerate(rows) if 'forest_commit' in r['Kernel_Name']]
# the last sweep that is followed by another sweep
cand = [k for k in range(len(ms) - 1) if ms[k + 1] - ms[k] < 40]
a, b = ms[cand[-2]], ms[cand[-2] + 1]
base = int(rows[a]['End_Timestamp'])
print("sweep period: %.1f us" % ((int(rows[b]['End_Timestamp']) - base) / 1e3))
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']) - base, int(r['End_Timestamp']) - base
    print("%-44s grid %7s lds %6s  start %7.1f  end %7.1f  (%6.1f us)" % (r['Kernel_Name'].split('(')[0][-44:], r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('LDS_Block_Size', '?'), s / 1e3, e / 1e3, (e - s) / 1e3))
PY
