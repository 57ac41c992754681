#!/bin/bash
# Calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on known byte counts in this path's own access patterns
# (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count in your own access pattern"):
#   stream_bench: one-wave blocks streaming private regions, 4/8/16 B per lane (512 MiB per launch)
#   gather_bench: 22.5M random 8-byte gathers over an 80 MB table (+ 90 MB index read, 180 MB result write)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_cal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/stream_$c -- $ROOT/tools/stream_bench > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/gather_$c -- $ROOT/tools/gather_bench > /dev/null 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
for d in sorted(glob.glob(sys.argv[1] + "/*_SIZE")):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    print("==", d.split("/")[-1])
    for k, (n, v) in acc.items():
        print("  %-72s launches %3d  avg per launch %.1f KB" % (k, n, v / n))
PY
