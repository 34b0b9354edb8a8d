cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sim.py -m gpu -q -k "any_row_length or x512" --timeout=600 2>&1 | tail -3
export FLUIDX_LIB_PATH=tools/_variants/libfluidx_hip_lab.so
timeout 900 python -m pytest tests/test_gpu_sim.py -m gpu -q -k "any_row_length or x512" --timeout=600 2>&1 | tail -3
for t in 0 1 0 1; do echo -n "256^3 tiled=$t: "; FLUIDX_STRIP4T_256=$t timeout 300 python tools/jacobi_microbench.py --grid 256 --iters 40 --reps 20 2>&1 | tail -1 | cut -c60-130; done
for dims in "264 264" "320 64" "320 320" "384 32" "384 96" "384 384" "640 64" "640 640" "768 32" "768 768" "1024 1024"; do set -- $dims; echo -n "$1 x $1 x $2: "; timeout 300 python tools/jacobi_microbench.py --grid $1 --depth $2 --iters 40 --reps 8 2>&1 | tail -1 | cut -c60-130; done
