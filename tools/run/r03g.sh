#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03g; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "general_block_kernel or jacobi" 2>&1 | tail -1
python bench.py --grid 150 --steps 100 --warmup 10 > $O/bench_150.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_150.json').read().strip().splitlines()[-1])
print('%.4g'%d['value'], round(d['ms_per_step'],4), d['stage_ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])"
