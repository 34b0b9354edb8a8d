#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "advect" > $O/pytest_advect.txt 2>&1; tail -15 $O/pytest_advect.txt
timeout 300 python tools/advect_microbench.py --steps 25 > $O/adv_25.txt 2>&1; cat $O/adv_25.txt
timeout 300 python tools/advect_microbench.py --steps 110 --variants "fast;lds;lds,zchunk=32" > $O/adv_110.txt 2>&1; cat $O/adv_110.txt
timeout 300 python tools/advect_microbench.py --grid 512 --steps 12 --iters 10 --reps 5 --variants "fast;lds;lds,zchunk=128;lds,zchunk=64" > $O/adv_512.txt 2>&1; cat $O/adv_512.txt
timeout 300 python tools/advect_microbench.py --grid 128 --steps 25 --variants "fast;lds;lds,zchunk=32;lds,zchunk=8" > $O/adv_128.txt 2>&1; cat $O/adv_128.txt
