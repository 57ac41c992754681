for lib in "" "build/g4/libcarmel_hip.so"; do
  echo "== lib=$lib"
  CARMEL_HIP_LIB=$lib python bench.py --config c5 --steps 300 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['workload'][:100])"
done
CARMEL_HIP_LIB=build/g4/libcarmel_hip.so python -m pytest tests/test_forest_gpu.py tests/test_bench_workloads_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --config mix --no-cpu-baseline --no-secondary --steps 50 2>&1 | tail -1 | cut -c1-1200
CARMEL_TIMING=1 python bench.py --config mix --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>&1 | grep "timing:" | head -20
