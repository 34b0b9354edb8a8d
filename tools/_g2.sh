cd $GRAFT_REPO_ROOT
FLUIDX_FUZZ_SEEDS=4000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "wide" --timeout=900 2>&1 | grep -E "passed|failed|Error|assert|dims" | tail -12
