#!/bin/bash
# round 3: kernel traces of the reference configuration at the reference's small grids (128^3 = FluidX12.cpp:44, 150^3 = Bin/FluidGI.bat:1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
for g in 128 150 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$g -o k -- python3 bench.py --no-cpu-baseline --no-render --reference-config --grid $g --steps 20 --warmup 40 > $O/bench_kt_$g.json 2> $O/err_$g.txt
  cp $(find $O/kt_$g -name "*kernel_stats.csv" | head -1) $O/kernel_stats_reference_$g.csv
  python - $O/kt_$g $g <<'PY'
import sys,glob,csv,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last step's launches in order: name, duration, gap to previous end
names=[r['Kernel_Name'] for r in rows]
last=max(i for i,n in enumerate(names) if 'k_advect' in n and 'k_advect_far' not in n)
prev=None
out=open('gpurun_out/r5b/last_step_%s.txt'%sys.argv[2],'w')
for r in rows[last:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.write('%-40s dur %7.2f us gap %7.2f us grid %s wg %s\n'%(r['Kernel_Name'].split('(')[0][-40:],(e-s)/1e3,((s-prev)/1e3 if prev else 0),r.get('Grid_Size_X',r.get('Grid_Size')),r.get('Workgroup_Size_X',r.get('Workgroup_Size'))))
    prev=e
PY
  rm -rf $O/kt_$g
done
cat $O/last_step_128.txt
