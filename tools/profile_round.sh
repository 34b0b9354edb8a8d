#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_round.sh <tag>          e.g.  gpurun -- 'tools/profile_round.sh r06'
# The measurement set of one round, into gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>_*):
#   bench lines     default (config 3), the driver's --steps 20 --warmup 5, configs 2 / 4 / 5, 150^3, 384^3, 1024^3, the reference's own configuration
#                   (256^3, 128^3, 150^3), in-process slab groups (shared and peer, N = 2 / 4; config 4 on 8 slabs)
#   per workload    rocprofv3 --kernel-trace --stats, HBM-side traffic (--pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as
#                   MI355X_MICROARCH.md prescribes: tools/pmc_summary.py) and SQ issue / wait counters (tools/sq_summary.py), every
#                   summary stamped with the source hash of the kernel it was taken on
#   render          kernel times + L2 / L1 hit rates of the render kernels at frame 132 (tools/render_pmc_summary.py), configs 3 and 5
# Counter passes never share a run with a trace domain other than the kernel trace's own (this pool refuses such runs).
TAG=${1:-r10}
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; if [ -z "$ONLY_RENDER" ]; then rm -rf $O; rm -f profiles/${TAG}_pmc_traffic_*.json profiles/${TAG}_sq_counters_*.json; fi; mkdir -p $O
rm -f profiles/${TAG}_render_pmc_*.json
prof() {  # tag, summary args, bench args...
  tag=$1; sargs=$2; shift 2
  # (every pass runs exactly --warmup + --steps steps on ONE context: the summaries divide dispatch counts by the steps profiled)
  B="python3 bench.py --no-cpu-baseline --no-render --no-developed --no-warm-leg $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o k -- $B > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -o f -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -o w -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1_$tag -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/sq2_$tag -o p -- $B > /dev/null 2>&1
  python tools/pmc_summary.py $(find $O/pmcf_$tag -name "*counter_collection.csv" | head -1) $(find $O/pmcw_$tag -name "*counter_collection.csv" | head -1) $sargs > $O/pmc_traffic_$tag.json
  python tools/sq_summary.py $(find $O/sq1_$tag -name "*counter_collection.csv" | head -1) $(find $O/sq2_$tag -name "*counter_collection.csv" | head -1) $(echo $sargs | sed 's/--steps-profiled [0-9]*//') > $O/sq_counters_$tag.json
  cp $(find $O/kt_$tag -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$tag.csv
  rm -rf $O/kt_$tag $O/pmcf_$tag $O/pmcw_$tag $O/sq1_$tag $O/sq2_$tag
}
if [ -z "$ONLY_RENDER" ]; then
prof 256 "--grid 256 --iters 40 --storage fp32" --steps 4 --warmup 1 --config 3
prof 128 "--grid 128 --iters 40 --storage fp32" --steps 4 --warmup 1 --config 2
prof 512_80 "--grid 512 --iters 80 --storage fp32" --steps 4 --warmup 1 --config 4
prof 150 "--grid 150 --iters 40 --storage fp32" --steps 4 --warmup 1 --grid 150
prof 384 "--grid 384 --iters 40 --storage fp32" --steps 4 --warmup 1 --grid 384
prof reference "--grid 256 --iters 64 --storage fp16 --mode faithful --steps-profiled 44" --steps 4 --warmup 40 --reference-config
fi
render() {  # tag, summary args, bench args...
  tag=$1; sargs=$2; shift 2
  B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-developed --no-warm-leg $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/rkt_$tag -o k -- $B > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/rl2_$tag -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/rl1_$tag -o p -- $B > /dev/null 2>&1
  python tools/render_pmc_summary.py $(find $O/rkt_$tag -name "*kernel_stats.csv" | head -1) $(find $O/rl2_$tag -name "*counter_collection.csv" | head -1) $(find $O/rl1_$tag -name "*counter_collection.csv" | head -1) $sargs > $O/render_pmc_$tag.json
  cp $(find $O/rkt_$tag -name "*kernel_stats.csv" | head -1) $O/render_${tag}_kernel_stats.csv
  rm -rf $O/rkt_$tag $O/rl2_$tag $O/rl1_$tag
}
render config3 "--grid 256 --storage fp32" --config 3
render config5 "--grid 256 --storage fp16 --has-sh" --config 5
# the counter summaries become the committed ones of this round BEFORE the bench lines run: bench.py quotes `roofline.traffic` and the
# render cache figures from profiles/ (on the box: in the snapshot; copy the same files into profiles/ at home)
mkdir -p profiles
if [ -n "$ONLY_RENDER" ]; then cp $O/render_pmc_*.json profiles/ 2>/dev/null; for f in $O/render_pmc_*.json; do cp $f profiles/${TAG}_$(basename $f); done; exit 0; fi
for f in $O/pmc_traffic_*.json $O/sq_counters_*.json $O/render_pmc_*.json; do cp $f profiles/${TAG}_$(basename $f); done
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_warmup5.json 2>> $O/bench.err
python bench.py --config 2 --steps 100 --warmup 16 --no-cpu-baseline > $O/bench_128.json 2>> $O/bench.err
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_512_80.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --grid 150 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_150.json 2>> $O/bench.err
python bench.py --grid 384 --steps 30 --warmup 10 --no-cpu-baseline --no-render > $O/bench_384.json 2>> $O/bench.err
python bench.py --grid 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-render --no-developed --no-warm-leg > $O/bench_1024.json 2>> $O/bench.err
python bench.py --reference-config > $O/bench_reference.json 2>> $O/bench.err
python bench.py --reference-config --grid 128 --no-cpu-baseline > $O/bench_reference_128.json 2>> $O/bench.err
python bench.py --reference-config --grid 150 --no-cpu-baseline > $O/bench_reference_150.json 2>> $O/bench.err
for g in shared peer; do
  for n in 2 4; do python bench.py --loopback $n --group $g --steps 25 --warmup 5 --no-cpu-baseline --no-render --no-developed > $O/bench_loopback${n}_$g.json 2>> $O/bench.err; done
  python bench.py --config 4 --loopback 8 --group $g --steps 6 --warmup 2 --no-cpu-baseline --no-render --no-developed > $O/bench_loopback8_config4_$g.json 2>> $O/bench.err
done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    r = d.get('roofline') or {}; rn = d.get('render') or {}
    print(f.split('/')[-1], '%.4g' % d['value'], round(d['ms_per_step'], 4), 'frac', round(r.get('frac', 0), 3), 'launch us', round(r.get('avg_launch_us', 0), 2), 'stale', r.get('stale'),
          {k: round(v, 4) for k, v in (d.get('stage_ms_per_step') or {}).items()}, 'render', {k: round(rn[k], 4) for k in ('light_pass_ms', 'view_pass_ms') if k in rn})
PY
tail -3 $O/bench.err
