#!/bin/bash
# round 5, experiment: pieces per lane class on separate streams (CARMEL_HIP_LANE_CHUNKS), c4a, with and without the fused backward pass
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for fk in 1 0; do for ch in 1 2 4 8; do
  echo "== fused_kernel=$fk chunks=$ch"
  CARMEL_HIP_LANE_FUSED_KERNEL=$fk CARMEL_HIP_LANE_CHUNKS=$ch python3 bench.py --config c4a --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-exchange-loopback --full-out /tmp/x.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
done; done
