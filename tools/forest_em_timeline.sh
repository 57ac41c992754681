#!/bin/bash
# kernel timeline of one EM iteration of forest-em on config 5's forests (the last iteration of the run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fet
rocprofv3 --kernel-trace --output-format csv -d /tmp/fet -- python3 $R/bench_forest.py --sweeps 1 --em-iters 12 > /tmp/fet.log 2>&1
f=$(find /tmp/fet -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ms = [i for i, r in enumerate(rows) if 'forest_mstep' in r['Kernel_Name']]
a, b = ms[-2], ms[-1]
base = int(rows[a]['End_Timestamp'])
print("EM iteration period: %.1f us" % ((int(rows[b]['End_Timestamp']) - base) / 1e3))
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']) - base, int(r['End_Timestamp']) - base
    print("%-50s start %8.1f us  end %8.1f us  (%7.1f us)" % (r['Kernel_Name'].split('(')[0][-50:], s / 1e3, e / 1e3, (e - s) / 1e3))
PY
