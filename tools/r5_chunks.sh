#!/bin/bash
# round 5, experiment: pieces per lane class (CARMEL_HIP_LANE_CHUNKS) under the fused-lane layout's two-stream pipeline, c4a and amb
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for cfg in c4a amb; do for ch in 1 2 4 8; do
  echo "== $cfg chunks=$ch"
  CARMEL_HIP_LANE_CHUNKS=$ch python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-exchange-loopback --full-out /tmp/x.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
done; done
