#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "temporal_blocking or thick_slabs or round2_kernels or slabs or config4 or x256 or full_size" 2>&1 | grep -E "passed|failed" | tail -1
