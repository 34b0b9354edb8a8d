#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_sim.py::test_none -k "deferred or advect_lds" 2>&1 | tail -3
