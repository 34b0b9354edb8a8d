#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02u; mkdir -p $O
python -m pytest tests/test_gpu_rccl_mock.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
