#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
FLUIDX_ADVECT_DEPTH=0 timeout 600 python -m pytest tests -m gpu -x -q -k "advect" > $O/pytest_advect.txt 2>&1; tail -3 $O/pytest_advect.txt
V="fast;lds,depth=0,zchunk=8;lds,depth=0,zchunk=16;lds,depth=0,zchunk=32;lds,depth=0,zchunk=64;lds,depth=0,zchunk=128"
timeout 300 python tools/advect_microbench.py --grid 512 --steps 12 --iters 10 --reps 5 --variants "$V" > $O/adv_512.txt 2>&1; cat $O/adv_512.txt
timeout 300 python tools/advect_microbench.py --grid 512 --steps 40 --iters 10 --reps 5 --variants "$V" > $O/adv_512b.txt 2>&1; cat $O/adv_512b.txt
timeout 300 python tools/advect_microbench.py --grid 128 --steps 25 --variants "fast;lds,depth=0,zchunk=8;lds,depth=0,zchunk=16;lds,depth=0,zchunk=4" > $O/adv_128.txt 2>&1; cat $O/adv_128.txt
timeout 300 python tools/advect_microbench.py --grid 256 --steps 25 --variants "fast;lds,depth=0,zchunk=8;lds,depth=0,zchunk=16;lds,depth=0,zchunk=12" > $O/adv_256.txt 2>&1; cat $O/adv_256.txt
