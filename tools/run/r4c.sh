#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python - <<'PY'
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import freeze_bench as fb
for grid, warm, steps in ((256, 40, 40), (128, 60, 60)):
    for T in (1, 2, 3, 4):
        for nt in (256, 512):
            for w in (1024, 2048, 4096, 8192):
                if grid == 128 and w != 2048: continue
                r = fb.run(grid, warm, steps, "fp16", {"FLUIDX_FREEZE_T": T, "FLUIDX_FREEZE_NT": nt, "FLUIDX_FREEZE_WGS": w})
                print(grid, "T", T, "NT", nt, "WGS", w, "ms/step", r["ms_per_step"], "jacobi", r["jacobi_ms"], flush=True)
PY
