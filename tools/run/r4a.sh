#!/bin/bash
# round 3, first measurement set: GPU test log, bench lines (default, reference configuration 256^3 / 128^3), kernel stats + PMC traffic + SQ counters
# of the reference configuration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --reference-config > $O/bench_reference.json 2>> $O/bench.err
python bench.py --reference-config --address mirror --no-cpu-baseline > $O/bench_reference_mirror.json 2>> $O/bench.err
python bench.py --reference-config --grid 128 --no-cpu-baseline > $O/bench_reference_128.json 2>> $O/bench.err
FLUIDX_FREEZE_FAST=0 python bench.py --reference-config --steps 20 --warmup 16 --no-cpu-baseline --no-render > $O/bench_reference_one_sweep_per_launch.json 2>> $O/bench.err
B="python3 bench.py --reference-config --steps 4 --warmup 40 --no-cpu-baseline --no-render"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw -o w -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1 -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/sq2 -o p -- $B > /dev/null 2>&1
rm -f $O/kt/k_kernel_trace.csv
find $O -name "*agent_info.csv" -delete
python tools/pmc_summary.py $(find $O/pmcf -name "*counter_collection.csv" | head -1) $(find $O/pmcw -name "*counter_collection.csv" | head -1) --grid 256 --iters 64 --storage fp16 --mode faithful --steps-profiled 44 > $O/pmc_traffic_reference.json
python tools/sq_summary.py $(find $O/sq1 -name "*counter_collection.csv" | head -1) $(find $O/sq2 -name "*counter_collection.csv" | head -1) --grid 256 --iters 64 --storage fp16 --mode faithful > $O/sq_counters_reference.json
find $O -name "*counter_collection.csv" -size +20M -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4a/bench*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    r=d['roofline']; print(f.split('/')[-1], '%.4g'%d['value'], round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'launch us', round(r['avg_launch_us'],2), d.get('stage_ms_per_step'), (r.get('sparse_solver') or {}).get('sweeps_executed_per_solve'))
PY
python __graft_entry__.py smoke 2>&1 | tail -1
