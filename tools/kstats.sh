#!/bin/bash
# usage: tools/kstats.sh <tag> [filter-regex] -- <python3 args...>
# one rocprofv3 --kernel-trace --stats run of `python3 <args>`; prints calls / avg / min / max (us) of the kernels matching the filter
# and keeps the csv under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
TAG=$1; shift
FILT='.'
if [ "$1" != "--" ]; then FILT=$1; shift; fi
shift
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 "$@" > $O/stdout.txt 2> $O/stderr.txt
python3 - "$O" "$FILT" <<'PY'
import csv, glob, re, sys
O, filt = sys.argv[1], re.compile(sys.argv[2])
fs = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv under", O); sys.exit(1)
rows = list(csv.DictReader(open(fs[0])))
print("%-64s %7s %10s %10s %10s %8s" % ("kernel", "calls", "avg us", "min us", "max us", "%"))
for r in rows:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("fx::", "")
    n = re.sub(r"\(.*", "", n)
    if filt.search(n):
        print("%-64s %7s %10.2f %10.2f %10.2f %8s" % (n[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
