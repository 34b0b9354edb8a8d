#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed" $O/pytest_gpu.txt | tail -3; grep -n "^FAILED\|^E  " $O/pytest_gpu.txt | head -20
