#!/bin/bash
# round 3, final gates: the whole GPU suite and a 1000-seed fuzz soak on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
FLUIDX_FUZZ_SEEDS=1000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu > $O/fuzz_1000.txt 2>&1; tail -1 $O/fuzz_1000.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
