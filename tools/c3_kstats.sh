#!/bin/bash
# per-kernel averages of config 3 (cipher cascade through the front end): bash tools/c3_kstats.sh [lines] [ENV=VALUE ...]
N=${1:-200000}; shift
for kv in "$@"; do export "$kv"; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/c3_$N
mkdir -p $D
if [ ! -f $D/corpus ]; then
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from carmel_amd import synth
lm, ch, co = synth.cipher_files($N)
open("$D/lm.wfsa", "w").write(lm); open("$D/ch.fst", "w").write(ch); open("$D/corpus", "w").write(co)
PY
fi
export CARMEL_TRAINED_DIR=$D
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c3ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3ks -- $ROOT/carmel_amd/bin/carmel --train-cascade --normby=NC -HJ -M 6 $D/corpus $D/lm.wfsa $D/ch.fst > /dev/null 2>&1
f=$(find /tmp/c3ks -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for i,r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<5: print("   %-70s calls %5s avg %9.1f us"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
