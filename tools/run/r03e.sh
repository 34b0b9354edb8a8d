#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
for g in 150 128 160 192; do python bench.py --grid $g --steps 100 --warmup 10 --no-cpu-baseline --no-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid $g', '%.4g'%d['value'], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms_per_step'].items()}, d['roofline']['kernel'][:30])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --grid 150 --steps 20 --warmup 2 --no-cpu-baseline --no-render > /dev/null 2>&1
rm -f $O/kt/k_kernel_trace.csv
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r03e/kt/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
