#!/bin/bash
# round 3, measurement set of the pruned tree: bench lines (configs 2-5, 150^3, reference configuration, loop-back 2 / 4 / 8), kernel stats,
# PMC traffic and SQ counters (stamped with kernel source hashes), render cache counters, GPU test log.  Summaries are copied to profiles/r05a_*.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_warmup5.json 2>> $O/bench.err
python bench.py --config 2 --steps 100 --warmup 16 --no-cpu-baseline > $O/bench_128.json 2>> $O/bench.err
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_512_80.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --grid 150 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_150.json 2>> $O/bench.err
python bench.py --reference-config > $O/bench_reference.json 2>> $O/bench.err
python bench.py --reference-config --grid 128 --no-cpu-baseline > $O/bench_reference_128.json 2>> $O/bench.err
for n in 2 4; do python bench.py --loopback $n --steps 25 --warmup 5 --no-cpu-baseline > $O/bench_loopback$n.json 2>> $O/bench.err; done
python bench.py --config 4 --loopback 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_loopback8_config4.json 2>> $O/bench.err
prof() {  # tag, summary args, bench args...
  tag=$1; sargs=$2; shift 2
  B="python3 bench.py --no-cpu-baseline --no-render --no-developed $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o k -- $B > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -o f -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -o w -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1_$tag -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/sq2_$tag -o p -- $B > /dev/null 2>&1
  rm -f $O/kt_$tag/k_kernel_trace.csv
  python tools/pmc_summary.py $(find $O/pmcf_$tag -name "*counter_collection.csv" | head -1) $(find $O/pmcw_$tag -name "*counter_collection.csv" | head -1) $sargs > $O/pmc_traffic_$tag.json
  python tools/sq_summary.py $(find $O/sq1_$tag -name "*counter_collection.csv" | head -1) $(find $O/sq2_$tag -name "*counter_collection.csv" | head -1) $(echo $sargs | sed 's/--steps-profiled [0-9]*//') > $O/sq_counters_$tag.json
  cp $(find $O/kt_$tag -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$tag.csv
  rm -rf $O/kt_$tag $O/pmcf_$tag $O/pmcw_$tag $O/sq1_$tag $O/sq2_$tag
}
prof 256 "--grid 256 --iters 40 --storage fp32" --steps 4 --warmup 1 --config 3
prof 128 "--grid 128 --iters 40 --storage fp32" --steps 4 --warmup 1 --config 2
prof 512_80 "--grid 512 --iters 80 --storage fp32" --steps 4 --warmup 1 --config 4
prof 150 "--grid 150 --iters 40 --storage fp32" --steps 4 --warmup 1 --grid 150
prof reference "--grid 256 --iters 64 --storage fp16 --mode faithful --steps-profiled 44" --steps 4 --warmup 40 --reference-config
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5a/bench*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    r=d.get('roofline') or {}; print(f.split('/')[-1], '%.4g'%d['value'], round(d['ms_per_step'],4), 'frac', round(r.get('frac',0),3), 'launch us', round(r.get('avg_launch_us',0),2), 'stale', r.get('stale'), (d.get('stage_ms_per_step') or {}))
PY
tail -3 $O/bench.err
