#!/bin/bash
# builds an A/B copy of the library with extra -D flags: mk_ab.sh <outdir> <flags...>
OUT=$1; shift
mkdir -p $OUT
SRC=/root/repo/carmel_amd/csrc
cd $SRC
HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics"
CXX="-O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result"
pids=()
for f in kernels.hip tile_sweep.hip engine.cpp engine_unrolled.cpp compose.hip matrix_fb.hip lattice_gpu.hip comm.cpp exchange.cpp dense.hip unrolled.hip; do
  /opt/rocm/bin/hipcc $CXX $HIPFLAGS "$@" -c $f -o $OUT/${f%.*}.o & pids+=($!)
done
for f in gibbs_exact.hip gibbs.hip forest_exact.hip forest.hip; do
  /opt/rocm/bin/hipcc $CXX $HIPFLAGS -ffp-contract=off "$@" -c $f -o $OUT/${f%.*}.o & pids+=($!)
done
/opt/rocm/bin/hipcc $CXX "$@" -c unrolled.cpp -o $OUT/unrolled_host.o & pids+=($!)
/opt/rocm/bin/hipcc $CXX "$@" -c lattice.cpp -o $OUT/lattice.o & pids+=($!)
/opt/rocm/bin/hipcc $CXX "$@" -c host_api.cpp -o $OUT/host_api.o & pids+=($!)
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc -shared -fPIC $HIPFLAGS -o $OUT/libcarmel_hip.so $OUT/*.o -lpthread -ldl -lrt
ls -la $OUT/libcarmel_hip.so
