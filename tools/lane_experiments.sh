#!/bin/bash
# timing experiments on the lane sweep (not parity-valid runs): E-step ms under ablations / occupancy changes
run() { env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'estep_ms', round(d['roofline']['kernel_ms'],4))"; }
for e in "$@"; do run $e; done
