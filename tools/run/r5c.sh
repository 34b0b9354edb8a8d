#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_rccl_mock.py -x -q > gpurun_out/r5c_slabs.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r5c_slabs.txt | tail -5
