#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "project or golden or rollout or x512 or slabs_equal" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in 1 0; do FLUIDX_PROJECT_V4=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-render > $O/bench_v4_$v.json 2>> $O/bench.err; done
FLUIDX_PROJECT_V4=1 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-render > $O/bench512_v4_1.json 2>> $O/bench.err
FLUIDX_PROJECT_V4=0 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-render > $O/bench512_v4_0.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench_v4_1.json","bench_v4_0.json","bench512_v4_1.json","bench512_v4_0.json"):
    d=json.loads(open("gpurun_out/r02f/"+f).read().strip().splitlines()[-1])
    print(f, "%.4g"%d["value"], "%.4f ms"%d["ms_per_step"], {k:round(v,4) for k,v in d["stage_ms_per_step"].items()})
PY
