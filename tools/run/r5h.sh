#!/bin/bash
# round 3: the bench lines of the final tree, taken with its own counter summaries (profiles/r05h_*) already in place
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_warmup5.json 2>> $O/bench.err
python bench.py --config 2 --steps 100 --warmup 16 --no-cpu-baseline > $O/bench_128.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --grid 150 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_150.json 2>> $O/bench.err
python bench.py --reference-config > $O/bench_reference.json 2>> $O/bench.err
python bench.py --reference-config --grid 128 --no-cpu-baseline > $O/bench_reference_128.json 2>> $O/bench.err
python bench.py --reference-config --grid 150 --no-cpu-baseline > $O/bench_reference_150.json 2>> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5h/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], '%.4g'%d['value'], round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'stale', r.get('stale'), 'traffic', r.get('traffic_source'), (d.get('step_fabric_traffic') or {}).get('stale_kernels'), 'developed', round(d['developed_plume']['ms_per_step'],4))
PY
