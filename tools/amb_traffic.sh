#!/bin/bash
# HBM counters of the ambiguous workload (bench.py --config amb: the tagging cascade, corpus x 400) through the front end itself:
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes -> gpurun_out/pmc_traffic_amb.json (committed as profiles/pmc_traffic_amb.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${1:-400}
cd /tmp && export TMPDIR=/tmp
python3 - "$R" "$REPS" <<'PY'
import sys
r, reps = sys.argv[1], int(sys.argv[2])
open("/tmp/amb_corpus", "w").write(open(r + "/tests/golden/tagging.data").read() * reps)
PY
rm -rf /tmp/ambt; mkdir -p /tmp/ambt
export CARMEL_TRAINED_DIR=/tmp
CMD="$R/carmel_amd/bin/carmel --train-cascade -HJ -M 5 -X 1.1 -e 0 /tmp/amb_corpus $R/tests/golden/tagging.fsa $R/tests/golden/tagging.fst"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ambt/fetch -- $CMD > /tmp/ambt/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ambt/write -- $CMD > /tmp/ambt/write.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/ambt amb > $R/gpurun_out/pmc_traffic_amb.json
head -c 600 $R/gpurun_out/pmc_traffic_amb.json
