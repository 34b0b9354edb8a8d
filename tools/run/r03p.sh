#!/bin/bash
# round 2 (second half), final-state measurements: bench lines (configs 2-5, loop-back 2/4/8), kernel traces, PMC traffic, SQ counters, GPU test log
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2>> $O/bench.err
python bench.py --config 2 --steps 100 --warmup 16 --no-cpu-baseline > $O/bench_128.json 2>> $O/bench.err
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_512_80.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --grid 150 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_150.json 2>> $O/bench.err
for n in 2 4; do python bench.py --loopback $n --steps 25 --warmup 5 --no-cpu-baseline > $O/bench_loopback$n.json 2>> $O/bench.err; done
python bench.py --config 4 --loopback 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_loopback8_config4.json 2>> $O/bench.err
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-render"
for cfg in 3 2 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$cfg -o k -- $B --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf$cfg -o f -- $B --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw$cfg -o w -- $B --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1_$cfg -o p -- $B --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/sq2_$cfg -o p -- $B --config $cfg > /dev/null 2>&1
  rm -f $O/kt$cfg/k_kernel_trace.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt150 -o k -- $B --grid 150 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf150 -o f -- $B --grid 150 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw150 -o w -- $B --grid 150 > /dev/null 2>&1
rm -f $O/kt150/k_kernel_trace.csv
find $O -name "*agent_info.csv" -delete
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1
