#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
FLUIDX_FUZZ_SEEDS=150 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
