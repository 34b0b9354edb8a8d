#!/bin/bash
# round 3: long-run parity of the final tree's default kernels against the plainest ones (256^3: 600 steps fp32, 400 steps fp16; the
# reference's configuration: 128^3 x 300 and 256^3 x 150 steps against k_jacobi_generic)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
python tools/long_run_parity.py 256 600 fp32 > $O/long_run_256_fp32.txt 2>&1; echo "rc $?" >> $O/long_run_256_fp32.txt
python tools/long_run_parity.py 256 400 fp16 > $O/long_run_256_fp16.txt 2>&1; echo "rc $?" >> $O/long_run_256_fp16.txt
python tools/long_run_parity.py 128 300 fp16 faithful > $O/long_run_128_reference.txt 2>&1; echo "rc $?" >> $O/long_run_128_reference.txt
python tools/long_run_parity.py 256 150 fp16 faithful > $O/long_run_256_reference.txt 2>&1; echo "rc $?" >> $O/long_run_256_reference.txt
python tools/long_run_parity.py 150 150 fp16 faithful > $O/long_run_150_reference.txt 2>&1; echo "rc $?" >> $O/long_run_150_reference.txt
for f in $O/*.txt; do echo $f; tail -n 2 $f; done
