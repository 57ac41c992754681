#!/bin/bash
# tile sweep A/B on config 4: the one-kernel tile sweep, the three kernels on its layout, the five-kernel layout
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lattice_gpu.py -x -q -m gpu 2>&1 | tail -5
for mode in ${MODES:-fused CARMEL_HIP_TILE_SWEEP_KERNEL=0 CARMEL_HIP_TILE_SWEEP=0}; do
  echo "== $mode"
  unset CARMEL_HIP_TILE_SWEEP_KERNEL CARMEL_HIP_TILE_SWEEP
  [ "$mode" != fused ] && export $mode
  for cfg in ${CFGS:-c4}; do
    bash tools/kstats.sh $cfg 2>&1 | grep -E "^[0-9]|trans_|sweep_|Traceback|Error|error"
  done
done
