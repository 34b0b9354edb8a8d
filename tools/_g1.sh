hipcc --offload-arch=gfx950 -O2 tools/micro/issue_rate.cpp -o /tmp/issue_rate && /tmp/issue_rate > gpurun_out/issue_rate.txt 2>&1; cat gpurun_out/issue_rate.txt
