cd $GRAFT_REPO_ROOT
export FLUIDX_LIB_PATH=tools/_variants/libfluidx_hip_lab.so
for dims in "264 8" "264 32" "264 264" "320 16" "320 64" "384 4" "384 8" "384 16" "384 32" "384 96" "1024 4" "1024 8" "2048 4"; do
  set -- $dims
  for f in 2000000000 1; do
    echo -n "$1 x $1 x $2 four-from $f: "; FLUIDX_STRIP4T_FROM=$f timeout 300 python tools/jacobi_microbench.py --grid $1 --depth $2 --iters 40 --reps 5 2>&1 | tail -1 | cut -c60-200
  done
done
