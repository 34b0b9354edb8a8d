#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02q; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "slab or mock or adaptive or overflow or beyond" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for n in 2 4; do python bench.py --loopback $n --steps 24 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=d['config']['schedule']
print('loopback $n', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], {k:round(v,4) for k,v in d['stage_ms_per_step'].items()}, sc['overlap'], sc['jacobi_round'], round(sc['advect_planes_per_face_and_step'],2), round(sc['sent_MB_per_face_and_step'],1), [(c['overlap'],c['jacobi_round'],round(c['ms_per_step'],2)) for c in sc['candidates']])"; done
