#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err
python bench.py --no-cpu-baseline > $O/bench_256_100.json 2>> $O/bench_256.err
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_512_80.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench_256.json","bench_256_100.json","bench_512_80.json"):
    d=json.loads(open("gpurun_out/r02e/"+f).read().strip().splitlines()[-1])
    print(f, "%.4g"%d["value"], "%.4f ms"%d["ms_per_step"], {k:round(v,4) for k,v in d["stage_ms_per_step"].items()})
PY
