#!/bin/bash
# round 3, reference configuration (<= 64 sweeps + early-out, RGBA16F) after the XCD-aware tile hand-out: bench lines (256^3, 128^3, 150^3, 512^3),
# kernel stats, per-launch trace of one step, PMC traffic and SQ counters (stamped with kernel source hashes).  Summaries -> profiles/r05b_*.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
python bench.py --reference-config > $O/bench_reference.json 2> $O/bench.err
python bench.py --reference-config --address mirror --no-cpu-baseline > $O/bench_reference_mirror.json 2>> $O/bench.err
python bench.py --reference-config --grid 128 --no-cpu-baseline > $O/bench_reference_128.json 2>> $O/bench.err
python bench.py --reference-config --grid 150 --no-cpu-baseline > $O/bench_reference_150.json 2>> $O/bench.err
python bench.py --reference-config --grid 512 --steps 6 --warmup 30 --no-cpu-baseline --no-render > $O/bench_reference_512.json 2>> $O/bench.err
prof() {  # tag, summary args, bench args...
  tag=$1; sargs=$2; shift 2
  B="python3 bench.py --no-cpu-baseline --no-render --no-developed $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o k -- $B > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -o f -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -o w -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1_$tag -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/sq2_$tag -o p -- $B > /dev/null 2>&1
  python - $O/kt_$tag $tag <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
last=max(i for i,n in enumerate(names) if 'k_advect' in n and 'k_advect_far' not in n)
out=open('gpurun_out/r5d/last_step_%s.txt'%sys.argv[2],'w')
for r in rows[last:]:
    if 'rocclr' in r['Kernel_Name']: break
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=r['Kernel_Name']; n=n[n.find('k_'):].split('(')[0]
    out.write('%-34s %8.2f us   grid %8s x %4s\n'%(n[:34],(e-s)/1e3,r.get('Grid_Size_X',r.get('Grid_Size')),r.get('Workgroup_Size_X',r.get('Workgroup_Size'))))
PY
  python tools/pmc_summary.py $(find $O/pmcf_$tag -name "*counter_collection.csv" | head -1) $(find $O/pmcw_$tag -name "*counter_collection.csv" | head -1) $sargs > $O/pmc_traffic_$tag.json
  python tools/sq_summary.py $(find $O/sq1_$tag -name "*counter_collection.csv" | head -1) $(find $O/sq2_$tag -name "*counter_collection.csv" | head -1) $(echo $sargs | sed 's/--steps-profiled [0-9]*//') > $O/sq_counters_$tag.json
  cp $(find $O/kt_$tag -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$tag.csv
  rm -rf $O/kt_$tag $O/pmcf_$tag $O/pmcw_$tag $O/sq1_$tag $O/sq2_$tag
}
prof reference "--grid 256 --iters 64 --storage fp16 --mode faithful --steps-profiled 44" --steps 4 --warmup 40 --reference-config
prof reference_128 "--grid 128 --iters 64 --storage fp16 --mode faithful --steps-profiled 44" --steps 4 --warmup 40 --reference-config --grid 128
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5d/bench*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    r=d.get('roofline') or {}; print(f.split('/')[-1], '%.4g'%d['value'], round(d['ms_per_step'],4), 'frac', round(r.get('frac',0),3), 'launch us', round(r.get('avg_launch_us',0),2), 'stale', r.get('stale'), (d.get('stage_ms_per_step') or {}))
PY
tail -3 $O/bench.err
