#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_c2
mkdir -p $O
cd $R
run() {
  tag=$1; shift
  env "$@" timeout 300 python3 bench.py --config c2 --no-secondary --no-cpu-baseline --no-exchange-loopback --steps 50 --warmup 5 > $O/c2_$tag.json 2> $O/c2_$tag.err
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/c2_$tag.json") if l.startswith("{")][-1])
    print("$tag", "ms/step %.4f kernel_ms %.4f frac %.4f" % (d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("$tag FAILED", e, open("$O/c2_$tag.err").read()[-500:])
PY
}
run base X=1
run gather CARMEL_HIP_TRANSPOSE=0
run graph CARMEL_HIP_GRAPH=1
run gather_graph CARMEL_HIP_TRANSPOSE=0 CARMEL_HIP_GRAPH=1
CARMEL_HIP_TRANSPOSE=0 bash tools/kstats.sh c2 2>&1 | head -14
