cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python bench.py --config 4 --loopback 8 --group shared --steps 6 --warmup 2 --no-cpu-baseline --no-render --no-developed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loopback8 config4', d['ms_per_step'], d['stage_ms_per_step'], d.get('multi_rank_parity'), d['config']['schedule']['overlap'], d['config']['schedule']['jacobi_round'])"
