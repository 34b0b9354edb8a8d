#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "general_block_kernel or jacobi or against_oracle or rollout or slabs_equal" 2>&1 | tail -3
for G in 150 100 192; do python bench.py --grid $G --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$G.json 2>/dev/null; python -c "
import json,sys
d=json.loads(open('$O/bench_$G.json').read().strip().splitlines()[-1])
print('grid $G', '%.4g'%d['value'], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms_per_step'].items()}, d['roofline']['kernel'][:24], round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac_compulsory'],3))"; done
