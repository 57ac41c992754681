#!/bin/bash
# A/B of the tiles' weights fetched from the table (CARMEL_HIP_TILE_GATHER): bash tools/r5_tile_gather.sh c2 c4a amb
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "$@"; do
  for g in 1 0; do
    CARMEL_HIP_TILE_GATHER=$g python3 $R/bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-exchange-loopback 2>/dev/null | grep '^{' | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg gather=$g', 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'lnp', d.get('ln_corpus_prob_last'))"
  done
done
