#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 100 --warmup 16 --no-cpu-baseline --no-render > $O/bench_under_rocprof.json 2>/dev/null
rm -f $O/kt/k_kernel_trace.csv; find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import json,csv
d=json.loads(open('gpurun_out/r03q/bench_under_rocprof.json').read().strip().splitlines()[-1])
print('events:', d['roofline']['avg_launch_us'], d['ms_per_step'])
for r in list(csv.DictReader(open('gpurun_out/r03q/kt/k_kernel_stats.csv')))[:6]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
