#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for zc in 16 17 16 17; do FLUIDX_STRIP3_ZCHUNK=$zc python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('zchunk $zc', '%.4g'%d['value'], round(d['ms_per_step'],4), round(d['stage_ms_per_step']['jacobi'],4), round(d['roofline']['avg_launch_us'],2))"; done
