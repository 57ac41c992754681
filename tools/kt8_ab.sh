#!/bin/bash
# experiment: the transposition kernels' times with tiles of 8192 positions (a build with -DTRANS_KT=8 beside the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for lib in "" carmel_amd/variants/libcarmel_kt8.so; do
  echo "== lib: ${lib:-default}"
  if [ -n "$lib" ]; then export CARMEL_HIP_LIB=$R/$lib; fi
  bash tools/kstats.sh c4 2>&1 | grep -E "^[0-9]|trans_|sweep_lane" 
done
