#!/bin/bash
# round 5, experiment: which pass of the c4a transposition does the random access (CARMEL_HIP_TRANS_SCATTER=0..3), per kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_scatter
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sc in 0 1 2 3; do
  export CARMEL_HIP_TRANS_SCATTER=$sc
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$sc -- python3 $ROOT/bench.py --config c4a --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-exchange-loopback > $OUT/s$sc.log 2>&1
  f=$(find $OUT/s$sc -name '*kernel_stats.csv' | head -1)
  echo "== scatter=$sc"; tail -1 $OUT/s$sc.log | cut -c1-400
  python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("trans_","sweep_lane","mstep_window")):
        print("  %-70s calls %4s avg %9.1f us" % (n[:70], r["Calls"], float(r["AverageNs"])/1e3))
P
done
