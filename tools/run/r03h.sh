#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "x512 or wide_strips or config4 or 512" 2>&1 | tail -1
for a in 0 1 0 1; do FLUIDX_STRIP3H_ASM=$a python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('asm $a', '%.4g'%d['value'], round(d['ms_per_step'],3), round(d['stage_ms_per_step']['jacobi'],3), d['roofline']['kernel'][:18], round(d['roofline']['avg_launch_us'],1))"; done
