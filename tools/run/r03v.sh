#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "test_gpu_sim or golden or slabs_x128 or x128" 2>&1 | grep -E "passed|failed" | tail -1
run() { env "$@" python bench.py --grid $G --steps 200 --warmup 10 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid $G $*', '%.4g'%d['value'], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms_per_step'].items() if k!='exchange'})"; }
G=128; run FLUIDX_XCD_REMAP=1,0,1,1; run A=1; run FLUIDX_XCD_REMAP=1,0,1,1; run A=1
