#!/bin/bash
# rocprofv3 kernel stats + HBM counters for config 3 (cipher cascade, 200 000 lines) through the front end
N=${1:-200000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/c3_$N
mkdir -p $D $ROOT/gpurun_out/c3prof
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from carmel_amd import synth
lm, ch, co = synth.cipher_files($N)
open("$D/lm.wfsa", "w").write(lm); open("$D/ch.fst", "w").write(ch); open("$D/corpus", "w").write(co)
PY
export CARMEL_TRAINED_DIR=$D
cd /tmp && export TMPDIR=/tmp
ARGS="--train-cascade --normby=NC -HJ -M 6 $D/corpus $D/lm.wfsa $D/ch.fst"
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/c3prof/stats -- $ROOT/carmel_amd/bin/carmel $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/c3prof/fetch -- $ROOT/carmel_amd/bin/carmel $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/c3prof/write -- $ROOT/carmel_amd/bin/carmel $ARGS > /dev/null 2>&1
cut -c1-150 $ROOT/gpurun_out/c3prof/stats/*/*kernel_stats.csv | head -6
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/c3prof c3 > $ROOT/gpurun_out/c3prof/pmc_summary.json
python3 -c "
import json; d=json.load(open('$ROOT/gpurun_out/c3prof/pmc_summary.json'))
for k,v in d['kernels'].items():
    if 'unrolled' in k: print(k[:60], v)"
