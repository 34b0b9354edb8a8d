cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sim.py -m gpu -q -k "any_row_length" --timeout=600 2>&1 | tail -4
FLUIDX_LIB_PATH=tools/_variants/libfluidx_hip_lab.so timeout 900 python -m pytest tests/test_gpu_sim.py -m gpu -q -k "any_row_length or x512" --timeout=600 2>&1 | tail -4
for dims in "1024 1024" "1024 128" "1024 64" "2048 128" "384 384" "768 768"; do set -- $dims; echo -n "$1 x $1 x $2: "; timeout 300 python tools/jacobi_microbench.py --grid $1 --depth $2 --iters 40 --reps 4 2>&1 | tail -1 | cut -c60-130; done
