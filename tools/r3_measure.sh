#!/bin/bash
# round-3 measurement pass: the full default bench line (headline + secondaries), kernel stats and PMC traffic of c4 / c4a / long
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_measure
mkdir -p $O
cd $R
timeout 1700 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench_full.json") if l.startswith("{")][-1])
print("c4 ms %.4f frac %.4f parity %s traffic %s wall %.0fs exchange %s" % (d["ms_per_step"], d["roofline"]["frac"], d.get("parity_checked_pairs"), d["roofline"]["traffic"], d["bench_wall_s"], {k: d["exchange"][k] for k in ("exchange_ms","exposed_exchange_ms")} if "exchange" in d else None))
for k,v in d.get("secondary",{}).items():
    print(k, v.get("error") or ("ms %.4f frac %.4f parity %s build %s exact %s cpu %s" % (v["ms_per_step"], v["roofline"]["frac"], v.get("parity_checked_pairs"), v.get("lattice_build_s"), v.get("exact"), (v.get("cpu_baseline") or {}).get("value"))), "wall %.0fs" % v["wall_s"])
PY
for CFG in c4 c4a long c2; do
  bash tools/kstats.sh $CFG 2>&1 | head -14
  cp $R/gpurun_out/${CFG}_kernel_stats.csv $O/ 2>/dev/null
done
for CFG in c4 c4a long; do
  bash tools/pmc_traffic.sh $CFG > /dev/null 2>&1
  cp $R/gpurun_out/pmc_$CFG/summary.json $O/pmc_traffic_$CFG.json 2>/dev/null
  python3 - <<PY
import json
try:
    d=json.load(open("$O/pmc_traffic_$CFG.json")); n=d.get("estep_count",1)
    tot=0
    for k,v in d["kernels"].items():
        if v["fetch_kb_per_launch"] is None: continue
        b=(v["launches"]/n)*(2*v["fetch_kb_per_launch"]+(v["write_kb_per_launch"] or 0))*1024
        if b>5e6: print("  %-50s %.3f GB/step"%(k[:50], b/1e9))
        tot+=b
    print("$CFG total %.3f GB per step (all kernels)"%(tot/1e9))
except Exception as e: print("$CFG pmc failed", e)
PY
done
# c5: kernel stats of the parallel sweep
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kf && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-secondary --steps 200 > /tmp/kf.log 2>&1
f=$(find /tmp/kf -name '*kernel_stats.csv' | head -1); cp $f $O/c5_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("   %-70s calls %6s avg %9.1f us"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
