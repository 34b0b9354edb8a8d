O=gpurun_out/r5j; mkdir -p $O
MB="tools/jacobi_microbench.py --grid 256 --iters 40 --reps 10 --fuse 4"
run() { export FLUIDX_BUILD_STRIP4_DEFS="$1"; python -m fluidx12_amd.build > $O/build.log 2>&1 || tail -5 $O/build.log; echo "== defs='$1'"; python $MB 2>&1 | grep us_per;  python $MB 2>&1 | grep us_per; }
{
run ""
timeout 600 python -m pytest tests/test_gpu_sim.py -x -q -k "four_sweeps or runs_fours" 2>&1 | tail -2
run "-DFX_S4_LATE_PREFETCH"
run ""
} > $O/variants.txt 2>&1
cat $O/variants.txt
