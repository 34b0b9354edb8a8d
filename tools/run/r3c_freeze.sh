#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_freeze.py -x -q > $O/pytest_freeze.txt 2>&1
tail -8 $O/pytest_freeze.txt
timeout 600 python tools/freeze_bench.py --grid 256 > $O/freeze_bench_256.txt 2> $O/freeze_bench.err
cut -c1-400 $O/freeze_bench_256.txt; tail -5 $O/freeze_bench.err
timeout 300 python tools/freeze_bench.py --grid 128 --warm 60 --steps 60 > $O/freeze_bench_128.txt 2>> $O/freeze_bench.err
cat $O/freeze_bench_128.txt
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
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 /tmp/fz_run.py 44 > $O/trace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
fn = glob.glob('gpurun_out/r3c/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_freeze_dense' in r['Kernel_Name']]
i0 = idx[-1]
print(' '.join('%.1f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows[i0:i0 + 18]))
print('step span us', (int(rows[i0+17]['End_Timestamp']) - int(rows[i0]['Start_Timestamp']))/1e3)
PY
