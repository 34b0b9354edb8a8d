#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02o; mkdir -p $O
for coop in 1 0; do for zc in 16 32 64 8; do
  echo -n "coop $coop zchunk $zc: "; FLUIDX_STRIP3_COOP=$coop FLUIDX_STRIP3_ZCHUNK=$zc python tools/jacobi_microbench.py --grid 256 --iters 39 --reps 10 --fuse 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/sweep' % d['us_per_sweep'])"
done; done 2>&1 | tee $O/zchunk.txt
