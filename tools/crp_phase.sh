#!/bin/bash
# per-phase cycles of the lattice sampler's kernels on the tagging cascade: the parallel sweep on the corpus x 100, the exact
# chain on the 1005 sentences (gibbs.hip prints them under CARMEL_HIP_GIBBS_CLK); CARMEL_HIP_GIBBS_REG=0 for the LDS kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - <<PY
import os
g = lambda n: os.path.join("$R", "tests", "golden", n)
open("/tmp/crp_corpus", "w").write(open(g("tagging.data")).read() * 100)
PY
export CARMEL_TRAINED_DIR=/tmp CARMEL_TIMING=1 CARMEL_HIP_GIBBS_CLK=1
for reg in 1 0; do
  echo "== CARMEL_HIP_GIBBS_REG=$reg"
  CARMEL_HIP_GIBBS_REG=$reg $R/carmel_amd/bin/carmel --crp --crp-parallel -M 40 -R 7 /tmp/crp_corpus $R/tests/golden/tagging.fsa $R/tests/golden/tagging.fst 2>&1 | grep -E "cycles per block|timing: gibbs|gibbs wave kernels|rror"
  CARMEL_HIP_GIBBS_REG=$reg $R/carmel_amd/bin/carmel --crp -M 60 -R 7 $R/tests/golden/tagging.data $R/tests/golden/tagging.fsa $R/tests/golden/tagging.fst 2>&1 | grep -E "cycles per block|timing: gibbs|gibbs wave kernels|rror"
done
