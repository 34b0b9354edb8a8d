cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_sim.py tests/test_gpu_slabs.py -m gpu -q -k "x512 or 512 or config4 or register_strips or thick" --timeout=900 2>&1 | tail -3
python bench.py --config 4 --loopback 8 --group shared --steps 6 --warmup 2 --no-cpu-baseline --no-render --no-developed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loopback8', d['ms_per_step'], d['stage_ms_per_step'], d.get('multi_rank_parity'))"
export FLUIDX_LIB_PATH=tools/_variants/libfluidx_hip_lab.so
FLUIDX_STRIP4T_512=0 python bench.py --config 4 --loopback 8 --group shared --steps 6 --warmup 2 --no-cpu-baseline --no-render --no-developed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loopback8 strip4x only', d['ms_per_step'], d['stage_ms_per_step'])"
