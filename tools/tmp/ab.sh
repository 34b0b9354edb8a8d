for a in "--config 2" "--grid 150" "--config 3" "--config 5"; do
python bench.py $a --steps 4 --warmup 1 --no-cpu-baseline --no-developed 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d[\"render\"]; print('$a', round(r[\"light_pass_ms\"],4), round(r[\"view_pass_ms\"],4), r[\"cube_size\"], r[\"ray_samples\"])"; done
python -m pytest tests/test_gpu_render.py -x -q -m gpu 2>&1 | tail -1
