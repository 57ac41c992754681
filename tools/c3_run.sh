#!/bin/bash
# config 3 (cipher cascade) through the command line at a reduced number of lines: explicit lattices
N=${1:-500}
ITERS=${2:-3}
D=/tmp/c3_$N
mkdir -p $D
python3 - <<PY
import sys; sys.path.insert(0, ".")
from carmel_amd import synth
lm, ch, co = synth.cipher_files($N)
open("$D/lm.wfsa", "w").write(lm); open("$D/ch.fst", "w").write(ch); open("$D/corpus", "w").write(co)
PY
export CARMEL_TIMING=1 CARMEL_TRAINED_DIR=$D
time ./carmel_amd/bin/carmel --train-cascade --normby=NC -HJ -M $ITERS $D/corpus $D/lm.wfsa $D/ch.fst 2>&1 | grep -E "^i=|timing|states /|derivations|ERROR|rror" | cut -c1-170
