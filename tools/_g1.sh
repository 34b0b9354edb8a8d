O=gpurun_out/r5m; mkdir -p $O
export PYTHONPATH=.
{ echo "tools/long_run_parity.py on tree $(cat .tree 2>/dev/null): default schedule (fours at 256^3) vs the plainest kernels, digests of velocity+colour+pressure every 100 steps";
  echo "== 256^3 fp32 fixed 40 (300 steps)"; timeout 900 python tools/long_run_parity.py 256 300 fp32;
  echo "== 256^3 fp16 faithful (300 steps)"; timeout 900 python tools/long_run_parity.py 256 300 fp16 faithful;
  echo "== 128^3 fp16 faithful (200 steps)"; timeout 600 python tools/long_run_parity.py 128 200 fp16 faithful; } > $O/long_run_parity.txt 2>&1
tail -20 $O/long_run_parity.txt
FLUIDX_FUZZ_SEEDS=1000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/fuzz_soak.txt 2>&1; tail -4 $O/fuzz_soak.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python bench.py --reference-config > $O/bench_reference.json 2> $O/bench.err
python bench.py --reference-config --grid 128 --no-cpu-baseline > $O/bench_reference_128.json 2>> $O/bench.err
python bench.py --reference-config --grid 150 --no-cpu-baseline > $O/bench_reference_150.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench_reference","bench_reference_128","bench_reference_150"):
    d=json.loads(open("gpurun_out/r5m/%s.json"%f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), d.get("developed_plume"))
PY
