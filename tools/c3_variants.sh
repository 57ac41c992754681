#!/bin/bash
# A/B of unrolled-sweep builds on config 3: each variant library (libcarmel_hip_<name>.so, built with other -DU_BATCH /
# -DU_WAVES_PER_EU) is copied over libcarmel_hip.so on the GPU box.  usage: bash tools/c3_variants.sh name:wgs_per_cu ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp carmel_amd/libcarmel_hip.so /tmp/orig.so
run() { echo "== $1 wgs/cu=$2: $(CARMEL_HIP_UNROLLED_WGS_PER_CU=$2 bash tools/c3_run.sh 200000 3 2>&1 | grep 'timing: i=3 estimate')"; }
run default 2
for v in "$@"; do
  lib=${v%%:*}; k=${v##*:}
  cp carmel_amd/libcarmel_hip_$lib.so carmel_amd/libcarmel_hip.so
  run $lib $k
done
cp /tmp/orig.so carmel_amd/libcarmel_hip.so
