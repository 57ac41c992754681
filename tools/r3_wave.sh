#!/bin/bash
# wave sweep: parity tests, then bench --config long with kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_wave
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_wave_gpu.py -x -q > $O/pytest_wave.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wave.log
tail -8 $O/pytest_wave.log
if [ -n "$FULL" ]; then
timeout 1500 python3 -m pytest tests/test_bench_workloads_gpu.py -q --durations=10 > $O/pytest_bw.log 2>&1; echo "pytest rc=$?" >> $O/pytest_bw.log
tail -30 $O/pytest_bw.log
fi
bash tools/kstats.sh long 2>&1 | tail -18
cp $R/gpurun_out/long_kernel_stats.csv $O/ 2>/dev/null
CARMEL_TIMING=1 timeout 600 python3 bench.py --config long --no-secondary --steps 10 --cpu-sample-pairs 40 > $O/bench_long.json 2> $O/bench_long.err; echo "bench long rc=$?"; grep timing $O/bench_long.err | head -30
python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench_long.json") if l.startswith("{")][-1])
print("long", d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"], d.get("parity_checked_pairs"), d["lattice_build_s"], d["cpu_baseline"]["value"] if "cpu_baseline" in d else None)
PY
CARMEL_HIP_WAVE_RING=0 bash tools/kstats.sh long 2>&1 | head -4
