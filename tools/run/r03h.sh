#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "jacobi or temporal_blocking or thick_slabs or round2_kernels or config4 or full_size or x512 or slabs_equal" 2>&1 | grep -E "passed|failed" | tail -1
for a in 1 2; do python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('256 run $a', '%.4g'%d['value'], round(d['ms_per_step'],4), round(d['stage_ms_per_step']['jacobi'],4), d['roofline']['kernel'][:18], round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac_compulsory'],3), d['roofline']['other_jacobi_launches'])"; done
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('512/80', '%.4g'%d['value'], round(d['ms_per_step'],3), round(d['stage_ms_per_step']['jacobi'],3), d['roofline']['kernel'][:18], round(d['roofline']['avg_launch_us'],1), d['roofline']['other_jacobi_launches'])"
