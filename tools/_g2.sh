cd $GRAFT_REPO_ROOT
export FLUIDX_LIB_PATH=tools/_variants/libfluidx_hip_lab.so
for g in 384 264 320; do echo "== long run $g"; timeout 1200 python tools/long_run_parity.py $g 200 fp32 2>&1 | tail -3 | cut -c1-160; done
echo "== long run 256"; timeout 1200 python tools/long_run_parity.py 256 300 fp32 2>&1 | tail -3 | cut -c1-160
echo "== long run 384 fp16"; timeout 1200 python tools/long_run_parity.py 384 100 fp16 2>&1 | tail -2 | cut -c1-160
