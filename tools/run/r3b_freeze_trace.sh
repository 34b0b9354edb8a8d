#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/fz_run.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fluidx12_amd as fx
f = fx.Fluid(); assert f.Init(0, 0, (256, 256, 256), storage="fp16", jacobi_iters=64, jacobi_mode="faithful")
dt = np.float32(f.default_time_step())
for k in range(int(sys.argv[1])):
    f.UpdateFrame(dt, k % 3); f.Simulate(k % 3)
f.Synchronize()
PY
FLUIDX_FREEZE_WGS=4096 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 /tmp/fz_run.py 44 > $O/trace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
fn = glob.glob('gpurun_out/r3b/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: find last k_freeze_dense
idx = [i for i, r in enumerate(rows) if 'k_freeze_dense' in r['Kernel_Name']]
i0 = idx[-1]
out = []
for r in rows[i0 - 3:i0 + 20]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    out.append('%-40s %8.2f us  grid %s' % (r['Kernel_Name'][:40], d, r.get('Grid_Size')))
print('\n'.join(out))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a = agg[r['Kernel_Name'][:50]]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print('%-52s n=%5d total %.1f us avg %.2f' % (k, v[0], v[1], v[1] / v[0]))
PY
