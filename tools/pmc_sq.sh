#!/bin/bash
# SQ counters for the E-step kernels (own pass, no tracing): where do the wave cycles go?
CFG=${1:-c4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$CFG
mkdir -p $OUT/a $OUT/b
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "carmel" not in k: continue
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k in acc:
    print(k)
    for c, (n, v) in sorted(acc[k].items()):
        print("   %-24s launches %3d  avg %.4g" % (c, n, v / n))
PY
