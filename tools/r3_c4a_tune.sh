#!/bin/bash
# c4a tuning sweep + pending tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_tune
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_unrolled_gpu.py::test_composed_arc_counts_under_the_unrolled_cascade_sweep tests/test_gpu_parity.py::test_linear_count_floor_against_the_log_counts tests/test_multirank_gpu.py::test_bench_two_ranks_strong_scaling_is_the_one_rank_run "tests/test_multirank_gpu.py::test_front_end_gpus_switch" -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
run() {
  tag=$1; shift
  env "$@" CARMEL_TIMING=1 timeout 600 python3 bench.py --config c4a --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > $O/c4a_$tag.json 2> $O/c4a_$tag.err
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/c4a_$tag.json") if l.startswith("{")][-1])
    print("$tag", "ms/step %.3f kernel_ms %.3f frac %.4f build %.1fs" % (d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"], d["lattice_build_s"]))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run base X=1
run v3 CARMEL_HIP_LANE_VARIANT=3
run v5 CARMEL_HIP_LANE_VARIANT=5
run v6 CARMEL_HIP_LANE_VARIANT=6
run runs CARMEL_HIP_TRANS_RUNS=1
run lane128 CARMEL_HIP_LANE_STATES=128
grep "timing:   lane piece\|timing: layout\|lattice build" $O/c4a_base.err | head -20
