#!/bin/bash
# kernel statistics of `carmel --crp --crp-parallel` on the tagging cascade x100 (bench.py --config crp's parallel leg)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/crp_prof
python - <<'PY'
import os
g=lambda n: os.path.join("tests","golden",n)
open("/tmp/crp_corpus","w").write(open(g("tagging.data")).read()*100)
PY
cd /tmp && export TMPDIR=/tmp
CARMEL_TRAINED_DIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/crp_prof -o crp -- $ROOT/carmel_amd/bin/carmel --crp --crp-parallel -M 40 -R 7 /tmp/crp_corpus $ROOT/tests/golden/tagging.fsa $ROOT/tests/golden/tagging.fst > $ROOT/gpurun_out/crp_prof/run.log 2>&1
find $ROOT/gpurun_out/crp_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/crp_kernel_stats.csv
head -8 $ROOT/gpurun_out/crp_kernel_stats.csv | cut -c1-200
