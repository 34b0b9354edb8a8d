#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02t; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "fp16 or golden or mirror or project or divergence or rollout or slab or uneven or adaptive" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fp16', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], {k:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
