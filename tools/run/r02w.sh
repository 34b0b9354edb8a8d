#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02w; mkdir -p $O
FLUIDX_ADVECT_LDS=2 FLUIDX_FUZZ_SEEDS=150 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_lds.txt 2>&1; tail -3 $O/fuzz_lds.txt
FLUIDX_ADVECT_LDS=2 timeout 600 python -m pytest tests/test_gpu_sim.py tests/test_gpu_slabs.py tests/test_gpu_golden.py -m gpu -x -q > $O/sim_lds.txt 2>&1; tail -3 $O/sim_lds.txt
