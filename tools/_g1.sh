O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sim.py -x -q -k "four_sweeps or runs_fours" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_rccl_mock.py -x -q -k "failed_link" 2>&1 | tail -3
MB="tools/jacobi_microbench.py --grid 256 --iters 40 --reps 10 --fuse 4"
python $MB 2>&1 | grep us_per
python $MB 2>&1 | grep us_per
