#!/bin/bash
# usage: tools/pmc_probe.sh <outdir> -- collects three PMC passes of a short bench run and prints per-kernel averages
O=${1:-gpurun_out/probe}; mkdir -p $O; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
P2="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1)); rocprofv3 --pmc $P --output-format csv -d $O/p$i -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$O/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split("::")[-1]
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s %14.1f" % (c, sum(v) / len(v)))
PY
