#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
hipcc -O2 --offload-arch=gfx950 tools/micro/fetch_calib.cpp -o /tmp/fetch_calib 2>/dev/null || cp tools/micro/fetch_calib /tmp/fetch_calib
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- /tmp/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- /tmp/fetch_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- /tmp/fetch_calib > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
for tag,ctr in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    agg=collections.defaultdict(list)
    for fn in glob.glob("gpurun_out/r03u/%s/*counter_collection.csv"%tag):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"]==ctr: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(ctr, k, "raw KiB %.0f  = %.3f of 1 GiB"%(sum(v)/len(v), sum(v)/len(v)*1024/2**30))
for fn in glob.glob("gpurun_out/r03u/k/*kernel_stats.csv"):
    for r in csv.DictReader(open(fn)): print(r["Name"][:20], r["Calls"], r["AverageNs"], "-> %.2f TB/s"%(2**30/float(r["AverageNs"])/1e3))
PY
