#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02x; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "large_grids or temporal_blocking or full_size or thick" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for c in 1 0 1; do FLUIDX_STRIP3_COOP=$c python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-render 2>> $O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('coop $c', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], {k:round(v,4) for k,v in d['stage_ms_per_step'].items()}, 'launch us %.2f'%d['roofline']['avg_launch_us'])"; done
for zc in 16 32 64; do echo -n "R=2 coop zchunk $zc: "; FLUIDX_STRIP3_ZCHUNK=$zc python tools/jacobi_microbench.py --grid 256 --iters 39 --reps 10 --fuse 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/sweep' % d['us_per_sweep'])"; done
