export TMPDIR=/tmp
O=gpurun_out/sqr; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-developed --no-preheat --config 3"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1 -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/sq2 -o p -- $B > /dev/null 2>&1
python tools/sq_summary.py $(find $O/sq1 -name "*counter_collection.csv" | head -1) $(find $O/sq2 -name "*counter_collection.csv" | head -1) --grid 256 --iters 40 --storage fp32 > $O/sq.json
rm -rf $O/sq1 $O/sq2
