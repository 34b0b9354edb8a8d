#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "x128 or slabs_x128 or round2_kernels" 2>&1 | grep -E "passed|failed" | tail -1
for a in 1 2; do python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('128 run $a', '%.4g'%d['value'], round(d['ms_per_step'],4), round(d['stage_ms_per_step']['jacobi'],4), d['roofline']['kernel'][:18], round(d['roofline']['avg_launch_us'],2))"; done
