#!/bin/bash
# k_jacobi_strip3z against k_jacobi_strip3c: kernel trace + counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-render"
for zm in 0 1; do
  export FLUIDX_STRIP3_ZMEET=$zm
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$zm -o k -- $B > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$zm -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$zm -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/s$zm -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $O/i$zm -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/t$zm -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/j$zm -o p -- $B > /dev/null 2>&1
  rm -f $O/kt$zm/k_kernel_trace.csv
done
find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections
for zm in (0, 1):
    print("ZMEET", zm)
    for f in glob.glob("gpurun_out/r03c/kt%d/*kernel_stats.csv" % zm):
        for r in csv.DictReader(open(f)):
            if "strip3" in r["Name"]: print("  ", r["Name"][:40], r["Calls"], r["AverageNs"])
    for d in "fwsitj":
        agg = collections.defaultdict(list)
        for f in glob.glob("gpurun_out/r03c/%s%d/*counter_collection.csv" % (d, zm)):
            for r in csv.DictReader(open(f)):
                if "strip3" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print("   %-26s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
