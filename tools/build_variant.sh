#!/bin/bash
# builds carmel_amd/libcarmel_hip<suffix>.so with extra -D flags (A/B experiments; bench with CARMEL_HIP_LIB=...)
# usage: bash tools/build_variant.sh _t8 -DTRANS_KT=8 -DTRANS_TILE_WAVES=8
set -e
suf=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
D=$R/carmel_amd/csrc_var$suf
rm -rf $D && mkdir -p $D && cp -r $R/carmel_amd/csrc/. $D/ && rm -f $D/*.o
make -C $D -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result $*" LIB=../libcarmel_hip$suf.so ../libcarmel_hip$suf.so > /tmp/build$suf.log 2>&1 || { tail -20 /tmp/build$suf.log; exit 1; }
rm -rf $D
ls -la $R/carmel_amd/libcarmel_hip$suf.so
