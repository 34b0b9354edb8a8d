#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "bench or rccl_mock" 2>&1 | grep -E "passed|failed" | tail -1
