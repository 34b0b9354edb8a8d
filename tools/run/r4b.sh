#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests/test_gpu_freeze.py -q -x 2>&1 | tail -2
python tools/freeze_bench.py --grid 256 --variants > $O/freeze_variants_256.txt 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r4b/freeze_variants_256.txt'):
    d=json.loads(l); print(d['env'], d['ms_per_step'], d['jacobi_ms'])
PY
for nt in 512 1024; do FLUIDX_FREEZE_NT=$nt python tools/freeze_bench.py --grid 128 --warm 60 --steps 60 2>/dev/null | head -1 | cut -c1-300; done
