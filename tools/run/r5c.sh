#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests/test_gpu_freeze.py -x -q 2>&1 | tail -3
python tools/freeze_bench.py --all-active --grid 256
python tools/freeze_bench.py --all-active --grid 128
for g in 128 150 256; do python bench.py --reference-config --grid $g --no-cpu-baseline --no-render | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($g, '%.4g'%d['value'], round(d['ms_per_step'],4), d['stage_ms_per_step'], d['roofline']['avg_launch_us'])"; done
