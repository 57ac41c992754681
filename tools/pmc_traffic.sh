#!/bin/bash
# Collect HBM traffic counters for bench.py's workload on the GPU box: two separate --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output under gpurun_out/pmc/.
# usage (from the repo root, through gpurun): bash tools/pmc_traffic.sh c4
CFG=${1:-c4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$CFG
rm -rf $OUT; mkdir -p $OUT/fetch $OUT/write
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback > $OUT/write.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT $CFG > $OUT/summary.json
cat $OUT/summary.json
