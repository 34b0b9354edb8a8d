#!/bin/bash
# usage: tools/pmc.sh <tag> "<COUNTER ...>" [filter-regex] -- <python3 args...>
# one rocprofv3 --pmc pass (counters in their own run: no trace domains beside them) of `python3 <args>`; prints the per-kernel
# average of every counter for the kernels matching the filter; csv kept under gpurun_out/<tag>/.
TAG=$1; CNT=$2; shift 2
FILT='.'
if [ "$1" != "--" ]; then FILT=$1; shift; fi
shift
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $O/pmc -o p -- python3 "$@" > $O/stdout.txt 2> $O/stderr.txt
python3 - "$O" "$FILT" <<'PY'
import csv, collections, glob, re, sys
O, filt = sys.argv[1], re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("fx::", ""))
        if filt.search(k):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
