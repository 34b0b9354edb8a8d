#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
run() { env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '%.4g'%d['value'], round(d['ms_per_step'],4), round(d['stage_ms_per_step']['jacobi'],4))"; }
run FLUIDX_STRIP3_ZMEET=0
run FLUIDX_STRIP3_ZMEET=1
run FLUIDX_STRIP3_ZMEET=1 FLUIDX_STRIP3Z_DBG=8
run FLUIDX_STRIP3_ZMEET=1 FLUIDX_STRIP3Z_DBG=72
run FLUIDX_STRIP3_ZMEET=1 FLUIDX_STRIP3Z_DBG=136
