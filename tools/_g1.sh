O=gpurun_out/r5i; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5i/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d.get("stage_ms_per_step"), d["roofline"].get("avg_launch_us"), d.get("developed_plume",{}).get("ms_per_step"))
PY
