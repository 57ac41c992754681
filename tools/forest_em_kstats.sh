#!/bin/bash
# per-kernel time of forest-em's EM iterations on config 5's forests
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fek
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fek -- python3 $R/bench_forest.py --sweeps 1 --em-iters 20 > /tmp/fek.log 2>&1
f=$(find /tmp/fek -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4 "$f" | head -12
