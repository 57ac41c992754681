#!/bin/bash
# The host-only sources (lattice builder, blocked-transposition tables, host inspection API) under AddressSanitizer and
# UBSan.  GPU sanitizers are not available on the pool; this is the CPU-side check.  usage: bash tools/asan_host.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/carmel_asan
mkdir -p $OUT
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -I$ROOT/include \
    $ROOT/carmel_amd/csrc/lattice.cpp $ROOT/carmel_amd/csrc/host_api.cpp $ROOT/carmel_amd/csrc/unrolled.cpp \
    -o $OUT/libcarmel_host_asan.so -lpthread
cat > $OUT/run.py <<PY
import ctypes as C, numpy as np, sys
sys.path.insert(0, "$ROOT")
lib = C.CDLL("$OUT/libcarmel_host_asan.so")
from carmel_amd import synth
def ptr(a): return a.ctypes.data_as(C.c_void_p) if a is not None else None
lib.carmel_hip_host_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64] + [C.c_void_p] * 4 + [C.c_uint64] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int]
lib.carmel_hip_host_dims.argtypes = [C.c_void_p, C.c_void_p]
lib.carmel_hip_host_transpose.argtypes = [C.c_void_p] * 12
lib.carmel_hip_host_free.argtypes = [C.c_void_p]
import os
for cfg in [dict(ns=300, deg=6, pairs=3000, lo=3, hi=25), dict(ns=40, deg=8, pairs=500, lo=3, hi=12), dict(ns=2000, deg=10, pairs=20000, lo=5, hi=40),
            dict(ns=12, deg=3, pairs=2000, lo=10, hi=60, win=4), dict(ns=30, deg=4, pairs=1500, lo=20, hi=90, win=4)]:
    os.environ["CARMEL_HIP_LANE_WINDOW_MIN"] = str(cfg.get("win", 40))  # the last two: windowed lane groups
    w = synth.random_wfst(cfg["ns"], cfg["deg"], n_sym=6, p_eps=0.15, seed=3)
    c = synth.random_walk_corpus(w, cfg["pairs"], min_arcs=cfg["lo"], max_arcs=cfg["hi"], seed=3, out_degree=cfg["deg"])
    h = C.c_void_p()
    assert lib.carmel_hip_host_build(C.byref(h), w.n_states, w.final, w.n_arcs, ptr(w.src), ptr(w.dst), ptr(w.isym), ptr(w.osym),
                                     c.n_pairs, ptr(c.in_off), ptr(c.in_sym), ptr(c.out_off), ptr(c.out_sym), ptr(c.weight), 1, 4, 0, 0, -1) == 0
    dims, td = np.zeros(19, np.uint64), np.zeros(6, np.uint64)
    lib.carmel_hip_host_dims(h, ptr(dims))
    lib.carmel_hip_host_transpose(h, ptr(td), *([None] * 10))
    print(cfg, "kept pairs", int(dims[6]), "lattice arcs", int(dims[15]), "items", int(td[0]), "buckets", int(td[1]))
    lib.carmel_hip_host_free(h)
print("asan/ubsan run complete: no reports")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python3 $OUT/run.py
# the front end's host-only paths (reader, writer, composition with and without -a, --normby / --number-from /
# --write-loaded) under the same sanitizers, linked against the (uninstrumented) library; nothing here touches a GPU
B=$OUT/carmel_asan_bin
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include $ROOT/carmel_amd/csrc/host/carmel_main.cpp \
    -o $B -L$ROOT/carmel_amd -lcarmel_hip -Wl,-rpath,$ROOT/carmel_amd -lpthread
G=$ROOT/tests/golden
export ASAN_OPTIONS=detect_leaks=0 CARMEL_TRAINED_DIR=$OUT
: > $OUT/front.err
$B -HJ -q $G/cipher.wfsa $G/cipher.fst > /dev/null 2>> $OUT/front.err
$B -HJm -q --normby=JC --priors=0.5,0.1 --write-loaded=x --number-from=3 $G/cipher.wfsa $G/cipher.fst > /dev/null 2>> $OUT/front.err
$B -a -HJ -q $G/chain.1 $G/chain.2 > /dev/null 2>> $OUT/front.err || true
$B -c $G/tagging.fsa $G/tagging.fst > /dev/null 2>> $OUT/front.err
for f in epron-jpron.fst train.a.w wfst3 tagging.fst; do $B -HJ $G/$f > /dev/null 2>> $OUT/front.err; done
if grep -q "ERROR: AddressSanitizer\|runtime error" $OUT/front.err; then grep "ERROR\|runtime error" $OUT/front.err | head; exit 1; fi
echo "front end under asan/ubsan: no reports"
# forest-em's host-only paths: the forest / normalisation-group / parameter readers, the initial parameters (uniform, -u,
# --random-set, -I with -N) and the option handling up to the first device call, which fails without a GPU (exit != 0 is
# expected here; malformed inputs must be diagnosed, not crash)
FB=$OUT/forest_em_asan_bin
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include $ROOT/carmel_amd/csrc/host/forest_em_main.cpp \
    -o $FB -L$ROOT/carmel_amd -lcarmel_hip -Wl,-rpath,$ROOT/carmel_amd -lpthread
: > $OUT/forest.err
printf '(0.5 0.25 0.125 0.5 0.5 0.5 0.5 0.5 0.5 0.5 0.5 0.5 0.5 0.5 0.5 0.5)\n' > $OUT/fem.init
for args in "" "-u" "--random-set --random-seed=4" "-I $OUT/fem.init -N" "-I $OUT/fem.init -z -N -H" "--crp=5 --crp-parallel" \
            "-v $OUT/v -S $OUT/s -E $OUT/e"; do
  $FB -f $G/fem.forests -n $G/fem.norm -i 2 -o $OUT/fem.out $args > /dev/null 2>> $OUT/forest.err || true
done
$FB -f $G/fem.norm -n $G/fem.norm > /dev/null 2>> $OUT/forest.err || true          # not a forest
$FB -f $G/fem.forests -n $G/fem.forests > /dev/null 2>> $OUT/forest.err || true    # not a group list
head -c 300 $G/fem.forests > $OUT/fem.cut; $FB -f $OUT/fem.cut -n $G/fem.norm > /dev/null 2>> $OUT/forest.err || true  # truncated
printf '(1 2)\n' > $OUT/fem.short; $FB -f $G/fem.forests -n $G/fem.norm -I $OUT/fem.short > /dev/null 2>> $OUT/forest.err || true
$FB -f $G/fem.forests > /dev/null 2>> $OUT/forest.err || true                       # missing normgroups-file
if grep -q "ERROR: AddressSanitizer\|runtime error" $OUT/forest.err; then grep "ERROR: Addr\|runtime error" $OUT/forest.err | head; exit 1; fi
echo "forest-em front end under asan/ubsan: no reports"
