#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02v; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --loopback 4 --steps 8 --warmup 2 --no-cpu-baseline --schedule 1,9 > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
rows = [r for r in csv.DictReader(open("gpurun_out/r02v/kt/k_kernel_trace.csv")) if "fx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-4 * 4 * 60:]          # roughly the last 4 steps
byk = collections.defaultdict(list)
for r in tail:
    byk[r["Kernel_Name"].split("(")[0][-30:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
print("span %.0f us, sum %.0f us, kernels %d" % (span, sum(sum(v) for v in byk.values()), len(tail)))
for k, v in sorted(byk.items(), key=lambda kv: -sum(kv[1])):
    print("  %-32s n %4d  total %8.1f us  avg %7.1f  min %6.1f max %7.1f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
PY
rm -f $O/kt/k_kernel_trace.csv
