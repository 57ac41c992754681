#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_quick
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_unrolled_gpu.py::test_composed_arc_counts_under_the_unrolled_cascade_sweep tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
bash tools/kstats.sh c4a 2>&1 | head -12
bash tools/kstats.sh c4 2>&1 | head -12
