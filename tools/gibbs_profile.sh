#!/bin/bash
# Kernel statistics and HBM counters of the two samplers' sweeps (bench.py --config crp's parallel leg through the front
# end, bench.py --config c5): rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes.
# usage (through gpurun, from the repo root): bash tools/gibbs_profile.sh [tag]     -> gpurun_out/<tag>_{crp,c5}_*
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
python3 - <<PY
import os
g = lambda n: os.path.join("$R", "tests", "golden", n)
open("/tmp/crp_corpus", "w").write(open(g("tagging.data")).read() * 100)
PY
cd /tmp && export TMPDIR=/tmp
CRP="$R/carmel_amd/bin/carmel --crp --crp-parallel -M 40 -R 7 /tmp/crp_corpus $R/tests/golden/tagging.fsa $R/tests/golden/tagging.fst"
C5="python3 $R/bench.py --config c5 --steps 60 --warmup 3 --no-cpu-baseline --no-secondary --no-exchange-loopback"
export CARMEL_TRAINED_DIR=/tmp
for W in crp c5; do
  if [ $W = crp ]; then CMD=$CRP; else CMD=$C5; fi
  rm -rf /tmp/gp_$W; mkdir -p /tmp/gp_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp_$W/ks -- $CMD > /tmp/gp_$W/ks.log 2>&1
  f=$(find /tmp/gp_$W/ks -name '*kernel_stats.csv' | head -1); cp $f $O/${TAG}_${W}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/gp_$W/fetch -- $CMD > /tmp/gp_$W/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/gp_$W/write -- $CMD > /tmp/gp_$W/write.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/gp_$W $W > $O/${TAG}_pmc_traffic_$W.json
  echo "== $W"; grep '^{' /tmp/gp_$W/ks.log | tail -1 | cut -c1-600; grep 'timing: gibbs' /tmp/gp_$W/ks.log
  python3 - "$O/${TAG}_${W}_kernel_stats.csv" "$O/${TAG}_pmc_traffic_$W.json" <<'PY'
import csv, json, sys
pm = json.load(open(sys.argv[2]))["kernels"]
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('carmel_hip::', '').replace('void ', '')
    k = next((v for kk, v in pm.items() if kk.replace('carmel_hip::', '').replace('void ', '') == n), None)
    fb = "fetch %8.1f MB write %8.1f MB" % (2 * (k["fetch_kb_per_launch"] or 0) / 1024, (k["write_kb_per_launch"] or 0) / 1024) if k else ""
    if float(r['Percentage']) > 0.3:
        print("   %-70s calls %5s avg %9.1f us  %5s%%  %s" % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage'], fb))
PY
done
