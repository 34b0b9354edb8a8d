#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02y; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "round2_kernels" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
