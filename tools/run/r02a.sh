#!/bin/bash
# round 2, GPU session a: new config-4 tests, then the round-1 state measured again as this round's baseline
# (bench lines for configs 2/3/4, kernel traces, PMC traffic at 512^3/80 and 128^3, SQ counters at 256^3)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "config4 or x512_full_step or full_size_properties or bench_multi_rank" > $O/pytest_new.txt 2>&1
tail -5 $O/pytest_new.txt
python bench.py --steps 20 --warmup 5 > $O/bench_256.json 2> $O/bench_256.err
python bench.py --config 2 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_128.json 2>> $O/bench.err
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_512_80.json 2>> $O/bench.err
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-render"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt256 -o k -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt512 -o k -- $B --config 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt128 -o k -- $B --config 2 > /dev/null 2>&1
for cfg in 4 2 3; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf$cfg -o f -- $B --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw$cfg -o w -- $B --config $cfg > /dev/null 2>&1
done
for cfg in 3 4; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1_$cfg -o p -- $B --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/sq2_$cfg -o p -- $B --config $cfg > /dev/null 2>&1
done
# keep only the small files
find $O -name "*agent_info.csv" -delete
ls -la $O $O/*/ | head -80
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
