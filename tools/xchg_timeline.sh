#!/bin/bash
# kernel timeline of one iteration with the exchange planned over a loopback communicator (world 1): what the
# chunking itself costs.  usage: bash tools/xchg_timeline.sh [chunks] [form]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-4}
FORM=${2:-sharded}
OUT=/tmp/r5_xtl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --exchange-chunks $K --exchange $FORM > $OUT/log 2>&1
f=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the loopback legs come last: the with-exchange steps are those whose trans_w_bucket launches come in ranges; take the
# third-last mstep_window of the run backwards until 2 iterations are on the page
ms=[i for i,r in enumerate(rows) if "mstep_window" in r["Kernel_Name"]]
# with_x runs before without_x (n_x each): pick an iteration in the middle of the with_x leg
n_x=(len(ms)-4)//2 if len(ms)>8 else 1
i_end=ms[4+ n_x//2] if len(ms)>8 else ms[-1]
i0=ms[4 + n_x//2 - 1]+1
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i_end+4]:
    n=r["Kernel_Name"].replace("carmel_hip::","").replace("void ","")[:60]
    print("%-62s q=%s start %8.1f end %8.1f us" % (n, r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3))
P
