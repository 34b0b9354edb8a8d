#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
