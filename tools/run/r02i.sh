#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "slab or mock or fuzz" > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
for cf in 3 1; do for sch in "2,9" "2,6" "2,3" "0,9" "1,9"; do
  FLUIDX_CHAIN_FUSE=$cf python bench.py --loopback 4 --steps 20 --warmup 4 --no-cpu-baseline --schedule $sch > $O/l4_cf${cf}_${sch/,/_}.json 2>> $O/bench.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02i/l4_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4f ms"%d["ms_per_step"], {k:round(v,3) for k,v in d["stage_ms_per_step"].items()})
PY
