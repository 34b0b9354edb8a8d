#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "x512 or large_grids or wide_strips or thick or config4 or full_size" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('512^3/80', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], {k:round(v,4) for k,v in d['stage_ms_per_step'].items()}, 'launch us %.2f'%d['roofline']['avg_launch_us'])"
python tools/jacobi_microbench.py --grid 512 --depth 64 --iters 39 --reps 10 --fuse 3 2>/dev/null | tail -1
