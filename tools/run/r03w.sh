#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "bench_single_gpu_line" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['render'])"
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k:v for k,v in d['render'].items() if k.endswith('_ms')}, d['render'].get('sh_light_probe',{}).get('transform_ms_incl_upload'))"
