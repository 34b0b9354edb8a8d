#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03x; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2>> $O/bench.err
python bench.py --config 2 --steps 100 --warmup 16 --no-cpu-baseline > $O/bench_128.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --grid 150 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_150.json 2>> $O/bench.err
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_512_80.json 2>> $O/bench.err
python bench.py --config 4 --loopback 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_loopback8_config4.json 2>> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03x/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('render') or {}
    print(f.split('/')[-1], '%.4g'%d['value'], round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],2), r.get('light_pass_ms'), r.get('view_pass_ms'))
PY
