#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i; mkdir -p $O
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $O/bench_20.json 2>> $O/err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03i/bench_20.json').read().strip().splitlines()[-1])
print('%.4g'%d['value'], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms_per_step'].items()}, d.get('timing_marks'), round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac_compulsory'],3))
PY
done
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2>> $O/err.txt
python -m pytest tests -m gpu -x -q -k "bench or rccl_mock" 2>&1 | tail -1
