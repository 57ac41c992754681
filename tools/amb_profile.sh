#!/bin/bash
# per-kernel averages of the ambiguous workload (bench.py --config amb): the tagging cascade with its corpus repeated.
# The corpus is written first; rocprofv3 then runs the carmel front end itself (no interpreter in between).
# usage (through gpurun): bash tools/amb_profile.sh [reps] [extra env assignments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${1:-100}
cd /tmp && export TMPDIR=/tmp
python3 - "$R" "$REPS" <<'PY'
import sys
r, reps = sys.argv[1], int(sys.argv[2])
open("/tmp/amb_corpus", "w").write(open(r + "/tests/golden/tagging.data").read() * reps)
PY
rm -rf /tmp/ambp
export CARMEL_TIMING=1 CARMEL_TRAINED_DIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ambp -- $R/carmel_amd/bin/carmel --train-cascade -HJ -M 8 -X 1.1 -e 0 /tmp/amb_corpus $R/tests/golden/tagging.fsa $R/tests/golden/tagging.fst > /tmp/amb.out 2> /tmp/amb.err
grep "timing:   \|timing: layout" /tmp/amb.err; grep "timing: i=" /tmp/amb.err | tail -2
f=$(find /tmp/ambp -name '*kernel_stats.csv' | head -1)
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/amb_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-70s calls %5s avg %9.1f us  total %9.1f ms"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
