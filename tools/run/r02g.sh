#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "x128 or temporal_blocking or full_size" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for sh in 44 42 24 22 43; do FLUIDX_BLOCK_SHAPE=$sh python bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline --no-render > $O/bench128_$sh.json 2>> $O/bench.err; 
FLUIDX_BLOCK_SHAPE=$sh python -m pytest tests -m gpu -x -q -k "x128" 2>&1 | tail -1; done
python - <<'PY'
import json
for sh in (44,42,24,22,43):
    d=json.loads(open("gpurun_out/r02g/bench128_%d.json"%sh).read().strip().splitlines()[-1])
    print(sh, "%.4g"%d["value"], "%.4f ms"%d["ms_per_step"], {k:round(v,4) for k,v in d["stage_ms_per_step"].items()}, d["roofline"]["avg_launch_us"])
PY
