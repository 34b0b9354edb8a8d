O=gpurun_out/r5o; mkdir -p $O
run() { export FLUIDX_BUILD_STRIP3_DEFS="$1"; python -m fluidx12_amd.build > $O/build.log 2>&1 || tail -5 $O/build.log; echo "== defs='$1'";
  for a in "--grid 512 --iters 24 --reps 4" "--grid 512 --depth 64 --iters 24 --reps 10" "--grid 256 --depth 128 --iters 24 --reps 20" "--grid 256 --iters 24 --reps 10"; do echo "$a: $(python tools/jacobi_microbench.py $a --fuse 3 2>&1 | grep -o '"us_per_sweep": [0-9.]*')"; done; }
{
run "-DFX_S3_COND_STORE"
run ""
timeout 900 python -m pytest tests/test_gpu_sim.py tests/test_gpu_slabs.py -x -q -k "three_sweeps or x512 or thick_slabs or default_schedule" 2>&1 | tail -2
run "-DFX_S3_COND_STORE"
run ""
} > $O/variants.txt 2>&1
cat $O/variants.txt
