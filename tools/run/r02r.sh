#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02r; mkdir -p $O
python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_fp16.json 2> $O/err.txt; tail -3 $O/err.txt
python -c "
import json
d=json.loads(open('gpurun_out/r02r/bench_fp16.json').read().strip().splitlines()[-1])
print('%.4g'%d['value'], d['ms_per_step'], d['render'])"
