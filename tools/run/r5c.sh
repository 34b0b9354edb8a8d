#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for dbg in 0 1 2 3 4 7; do
for g in 128 256; do FLUIDX_PERSIST_DBG=$dbg timeout 120 python bench.py --reference-config --grid $g --no-cpu-baseline --no-render --no-developed | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $dbg', $g, '%.4g'%d['value'], round(d['ms_per_step'],4), 'jacobi', round(d['stage_ms_per_step']['jacobi'],4))"; done; done
