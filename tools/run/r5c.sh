#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_freeze.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for g in 128 150 256; do python bench.py --reference-config --grid $g --no-cpu-baseline --no-render | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($g, '%.4g'%d['value'], round(d['ms_per_step'],4), 'jacobi', round(d['stage_ms_per_step']['jacobi'],4), d['roofline']['avg_launch_us'])"; done; done
python bench.py --reference-config --grid 512 --steps 6 --warmup 30 --no-cpu-baseline --no-render | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(512, '%.4g'%d['value'], round(d['ms_per_step'],4), d['stage_ms_per_step'], d['roofline']['avg_launch_us'])"
