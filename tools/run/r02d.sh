#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
M="python3 tools/advect_microbench.py --steps 25 --reps 3 --variants lds,depth=0;lds,depth=0,dbg=2;fast"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $M > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $M > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/t -o t -- $M > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/s1 -o p -- $M > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/s2 -o p -- $M > /dev/null 2>&1
find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, collections, glob
for fn in sorted(glob.glob("gpurun_out/r02d/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if "advect" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    seen = {}
    for (k, c, d), v in agg.items():
        seen.setdefault((k, c), []).append(sum(v))
    for (k, c), v in seen.items():
        print(fn.split("/")[2], k, c, ["%.4g" % x for x in v[-8:]])
PY
