#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_freeze.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
for rep in 1 2; do for fu in 0 1; do
for g in 128 256; do FLUIDX_FREEZE_FUSE_DIV=$fu timeout 120 python bench.py --reference-config --grid $g --no-cpu-baseline --no-render | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse $fu', $g, '%.4g'%d['value'], round(d['ms_per_step'],4), d['stage_ms_per_step']['divergence'], 'jacobi', round(d['stage_ms_per_step']['jacobi'],4), 'dense', round(d['roofline']['avg_launch_us'],1), 'developed', round(d['developed_plume']['ms_per_step'],4))"; done; done; done
