#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i; mkdir -p $O
python bench.py > $O/bench.json 2> $O/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2>> $O/err.txt
python bench.py --config 2 --steps 100 --warmup 16 --no-cpu-baseline > $O/bench_128.json 2>> $O/err.txt
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_512_80.json 2>> $O/err.txt
python bench.py --config 5 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/err.txt
python bench.py --grid 150 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_150.json 2>> $O/err.txt
for n in 2 4; do python bench.py --loopback $n --steps 25 --warmup 5 --no-cpu-baseline > $O/bench_loopback$n.json 2>> $O/err.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03i/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], '%.4g'%d['value'], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms_per_step'].items()}, d.get('timing_marks'), round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac_compulsory'],3))
PY
python -m pytest tests -m gpu -x -q -k "bench or rccl_mock" 2>&1 | tail -1
