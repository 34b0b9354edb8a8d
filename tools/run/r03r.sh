#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "x512 or wide_strips or config4 or full_size or thick_slabs" 2>&1 | grep -E "passed|failed" | tail -1
for a in 1 2; do python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('512/80', '%.4g'%d['value'], round(d['ms_per_step'],3), round(d['stage_ms_per_step']['jacobi'],3), d['roofline']['kernel'][:18], round(d['roofline']['avg_launch_us'],1))"; done
python bench.py --config 4 --loopback 8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 loopback8', '%.4g'%d['value'], round(d['ms_per_step'],3), round(d['stage_ms_per_step']['jacobi'],3), d['roofline']['kernel'][:18], round(d['roofline']['avg_launch_us'],1))"
