#!/bin/bash
# experiment: LDS own-sample table capacity (slots per lane) vs parallel Gibbs sweep rate on config 5
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for cap in 256 128 64 32; do for stk in 32 8; do
  echo "own_cap=$cap stack=$stk: $(CARMEL_HIP_FOREST_OWNCAP=$cap CARMEL_HIP_FOREST_STACK=$stk timeout 300 python3 bench_forest.py --sweeps 60 --em-iters 2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gibbs_sweeps_per_s'], d['gibbs_last_cheap_logprob'])")"
done; done
