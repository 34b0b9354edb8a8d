#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
FLUIDX_FUZZ_SEEDS=1000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -3
