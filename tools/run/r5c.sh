#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_slabs.py -x -q -k "bench" 2>&1 | tail -3
python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['ms_per_step'], d['developed_plume'], d['render']['frame'])"
python bench.py --reference-config --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['ms_per_step'], d['developed_plume'], d['render']['frame'])"
