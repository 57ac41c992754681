python -m pytest tests/test_bench_workloads_gpu.py tests/test_forest_gpu.py -x -q -m gpu 2>&1 | tail -4
bash tools/timeline.sh 1 mix
python bench.py --config mix --no-cpu-baseline --no-secondary --steps 30 --warmup 3 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['roofline'].get('frac_iteration'))"
