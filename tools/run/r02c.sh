#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c; mkdir -p $O
for ty in 8 16; do echo "tile rows $ty"; FLUIDX_ADVECT_TILE_ROWS=$ty python -m pytest tests -m gpu -x -q -k "advect_lds" 2>&1 | tail -1
for st in 25 110; do FLUIDX_ADVECT_TILE_ROWS=$ty timeout 300 python tools/advect_microbench.py --steps $st --variants "fast;lds;lds,zchunk=8;lds,zchunk=32" 2>/dev/null | grep variant; done; done
