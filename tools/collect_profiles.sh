#!/bin/bash
# Everything a round commits under profiles/ from ONE tree, in one gpurun call: kernel statistics and PMC traffic of the EM
# workloads, the samplers' (tools/gibbs_profile.sh), then the default bench run (the line the driver records + the full record).
# usage (through gpurun, from the repo root): bash tools/collect_profiles.sh r6     -> gpurun_out/<tag>_*, gpurun_out/pmc_*/summary.json
TAG=${1:-rN}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for c in c4 c4a c2 long mix; do
  bash tools/kstats.sh $c > gpurun_out/${TAG}_${c}_kstats.txt 2>&1
  cp gpurun_out/${c}_kernel_stats.csv gpurun_out/${TAG}_${c}_kernel_stats.csv
  bash tools/pmc_traffic.sh $c > /dev/null 2>&1
  cp gpurun_out/pmc_$c/summary.json gpurun_out/pmc_traffic_$c.json
done
bash tools/amb_traffic.sh > /dev/null 2>&1
bash tools/gibbs_profile.sh $TAG > gpurun_out/${TAG}_gibbs_profile.txt 2>&1
cp gpurun_out/${TAG}_pmc_traffic_crp.json gpurun_out/pmc_traffic_crp.json
cp gpurun_out/${TAG}_pmc_traffic_c5.json gpurun_out/pmc_traffic_c5.json
head -40 gpurun_out/${TAG}_gibbs_profile.txt
