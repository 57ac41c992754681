#!/bin/bash
# start / end of every kernel of one E-step (do the streams overlap?  where are the gaps?)
# usage: bash tools/timeline.sh [chunks] [config]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CH=${1:-2}
CFG=${2:-c4a}
OUT=/tmp/r5_tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CARMEL_HIP_LANE_CHUNKS=$CH
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback > $OUT/log 2>&1
f=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last whole iteration: from the kernel behind the last M-step but one (mstep_max_final closes an M-step)
idx=[i for i,r in enumerate(rows) if "mstep_max_final" in r["Kernel_Name"]]
i0 = idx[-2] + 1 if len(idx) > 1 else 0
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0+28]:
    n=r["Kernel_Name"].replace("carmel_hip::","").replace("void ","")[:60]
    print("%-62s q=%s start %8.1f end %8.1f us" % (n, r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3))
P
