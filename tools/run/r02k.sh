#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "slab or mock" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
PROBE_OVERLAP=1 FLUIDX_COMM_PRIORITY=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 tools/micro/prio_probe.py > /dev/null 2>&1
grep -E "face_need|k_copy16" $O/kt/k_kernel_stats.csv | cut -c1-160
rm -f $O/kt/k_kernel_trace.csv
