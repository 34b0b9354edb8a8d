O=gpurun_out/r5k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bc6h.py tests/test_gpu_cxx_dropin.py tests/test_gpu_golden.py tests/test_gpu_freeze.py -x -q > $O/t.log 2>&1; tail -12 $O/t.log
