#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "slabs_general_block or slabs_x128 or slabs_equal" 2>&1 | tail -4
