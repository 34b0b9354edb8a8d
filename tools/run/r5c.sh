#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for zc in 16 32 64; do for w in 5 132; do
FLUIDX_ADVECT_ZCHUNK=$zc python bench.py --warmup $w --steps 20 --no-cpu-baseline --no-render --no-developed | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zchunk $zc fixed warm $w', '%.4g'%d['value'], round(d['ms_per_step'],4), 'advect', round(d['stage_ms_per_step']['advect'],4))"
FLUIDX_ADVECT_ZCHUNK=$zc python bench.py --reference-config --warmup $w --steps 20 --no-cpu-baseline --no-render --no-developed | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zchunk $zc reference warm $w', '%.4g'%d['value'], round(d['ms_per_step'],4), 'advect', round(d['stage_ms_per_step']['advect'],4))"
done; done; done
