#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "test_gpu_sim or test_gpu_golden" 2>&1 | tail -2
