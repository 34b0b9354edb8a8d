#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_freeze.py -x -q > $O/pytest_freeze.txt 2>&1
tail -15 $O/pytest_freeze.txt
timeout 600 python tools/freeze_bench.py --grid 256 --variants > $O/freeze_bench_256.txt 2> $O/freeze_bench.err
cat $O/freeze_bench_256.txt; tail -5 $O/freeze_bench.err
timeout 300 python tools/freeze_bench.py --grid 128 --warm 60 --steps 60 > $O/freeze_bench_128.txt 2>> $O/freeze_bench.err
cat $O/freeze_bench_128.txt
