O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rccl_mock.py -x -q -k "failed_link or byte_count or match_single" > $O/rccl.log 2>&1; tail -15 $O/rccl.log
timeout 600 python -m pytest tests/test_gpu_sim.py -x -q -k "2d_tile" > $O/t2d.log 2>&1; tail -5 $O/t2d.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $O/bench4.json 2> $O/bench4.err; tail -2 $O/bench4.err
FLUIDX_JACOBI_PREFER4=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $O/bench3.json 2> $O/bench3.err
python - <<'PY'
import json
for f in ("bench4","bench3"):
    d=json.loads([l for l in open("gpurun_out/r5e/%s.json"%f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d.get("stage_ms_per_step"), d["roofline"].get("kernel","")[:40], d["roofline"].get("avg_launch_us"), d.get("developed_plume",{}).get("ms_per_step"))
PY
