#!/bin/bash
# round 3: render leg at the pinned frame (132): bench lines config 3 / 5, kernel stats + L2 / L1 hit rates of the render kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
for cfg in 3 5; do
  B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --config $cfg"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$cfg -o k -- $B > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2_$cfg -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/l1_$cfg -o p -- $B > /dev/null 2>&1
  rm -f $O/kt$cfg/k_kernel_trace.csv
  st=fp32; sh=""; if [ $cfg = 5 ]; then st=fp16; sh="--has-sh"; fi
  python tools/render_pmc_summary.py $(find $O/kt$cfg -name "*kernel_stats.csv" | head -1) $(find $O/l2_$cfg -name "*counter_collection.csv" | head -1) $(find $O/l1_$cfg -name "*counter_collection.csv" | head -1) --grid 256 --storage $st $sh > $O/render_pmc_config$cfg.json
done
find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['render']
    print(f.split('/')[-1], '%.4g'%d['value'], 'frame', r['frame'], 'light ms %.4f view ms %.4f'%(r['light_pass_ms'], r['view_pass_ms']), 'samples/s %.3g view %.3g light %.3g'%(r['samples_per_s'], r['view_samples_per_s'], r['light_samples_per_s']), 'per ray %.1f'%r['mean_samples_per_view_ray'], {k:round(v['GBps']) for k,v in r['bound'].items() if isinstance(v,dict) and 'GBps' in v})
for f in sorted(glob.glob('gpurun_out/r4e/render_pmc*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], {k:(round(e.get('avg_us',0),1), round(e.get('l2_hit_rate',-1),3), round(e.get('l1_hit_rate',-1),3)) for k,e in d['kernels'].items()})
PY
tail -3 $O/bench.err
