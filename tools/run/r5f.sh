#!/bin/bash
# fuzz soak of the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5f; mkdir -p $O
FLUIDX_FUZZ_SEEDS=${1:-300} timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
