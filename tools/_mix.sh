run() { echo "== $*"; env "$@" python bench.py --config mix --no-cpu-baseline --no-secondary --steps 30 --warmup 3 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"; }
run A=1
run CARMEL_HIP_WAVE_GATHER=1
run CARMEL_HIP_TRANS_RUNS=0
run CARMEL_HIP_TRANS_RUNS=0 CARMEL_HIP_WAVE_GATHER=1
run CARMEL_HIP_TRANS_RUNS=0 CARMEL_HIP_WAVE_GATHER=1 CARMEL_HIP_TILE_GATHER=1
CARMEL_HIP_TRANS_RUNS=0 CARMEL_HIP_WAVE_GATHER=1 CARMEL_TIMING=1 python bench.py --config mix --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>&1 | grep "timing: wave\|timing: tile\|timing: trans"
python -m pytest tests/test_bench_workloads_gpu.py -x -q -m gpu 2>&1 | tail -12
