#!/bin/bash
# round 5, the records of the final tree: per-kernel statistics of the EM workloads (the same command as the bench line's), PMC
# traffic of c4 / c2 / c4a / long / c5 (amb and crp run the front end as a child process: no --pmc pass for those), the default bench run with its complete objects.  Outputs under gpurun_out/ (copied to profiles/).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for c in c4 c2 c4a long; do echo "== kstats $c"; bash tools/kstats.sh $c 2>&1 | head -9 | cut -c1-170; cp gpurun_out/${c}_kernel_stats.csv gpurun_out/r5_${c}_kernel_stats.csv; done
echo "== kstats amb"; bash tools/kstats.sh amb 2>&1 | head -9 | cut -c1-170
for c in c4 c2 c4a long c5; do echo "== pmc $c"; bash tools/pmc_traffic.sh $c > gpurun_out/pmc_$c.log 2>&1; tail -3 gpurun_out/pmc_$c.log | cut -c1-200; cp gpurun_out/pmc_$c/summary.json gpurun_out/pmc_traffic_$c.json; done
