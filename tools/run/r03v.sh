#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env "$@" python bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$ARGS $*', '%.4g'%d['value'], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms_per_step'].items() if k in ('divergence','project')})"; }
ARGS=""; for m in 1,0,0,0 1,0,0,2 1,0,2,2 1,0,0,0 1,0,0,2 1,0,2,2; do run FLUIDX_XCD_REMAP=$m; done
ARGS="--config 5"; for m in 1,0,0,0 1,0,2,2; do run FLUIDX_XCD_REMAP=$m; done
