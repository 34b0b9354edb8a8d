#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02z; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/err.txt
python -c "
import json
d=json.loads(open('gpurun_out/r02z/bench_driver_shape.json').read().strip().splitlines()[-1])
print('%.4g'%d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['step_fabric_traffic'], d['roofline']['kernel'][:20], d['roofline']['frac'], d['roofline']['frac_compulsory'], d['roofline']['frac_traffic'])"
