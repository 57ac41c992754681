#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_lattice_gpu.py tests/test_bench_workloads_gpu.py -x -q -m gpu -k "tile_sweep or scattering or full_size or windowed_corpora or small_corpora or run_length" 2>&1 | tail -8
