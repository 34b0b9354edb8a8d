#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
for p in 0 1; do PROBE_OVERLAP=1 FLUIDX_COMM_PRIORITY=$p rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$p -o k -- python3 tools/micro/prio_probe.py > /dev/null 2>&1; done
python3 - <<'PY'
import csv, collections
for p in (0, 1):
    rows = list(csv.DictReader(open("gpurun_out/r02j/kt%d/k_kernel_trace.csv" % p)))
    rows = [r for r in rows if "fx::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last 400 kernels ~ the final timed steps
    tail = rows[-1200:]
    byk = collections.defaultdict(list)
    for r in tail:
        byk[(r["Kernel_Name"].split("(")[0][-28:], r["Queue_Id"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
    busy = sum(sum(v) for v in byk.values())
    print("== FLUIDX_COMM_PRIORITY=%d: last %d kernels span %.0f us, sum of kernel durations %.0f us" % (p, len(tail), span, busy))
    for k, v in sorted(byk.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("   %-30s queue %s  n %4d  avg %8.1f us  max %8.1f" % (k[0], k[1], len(v), sum(v) / len(v), max(v)))
PY
rm -f $O/kt*/k_kernel_trace.csv
