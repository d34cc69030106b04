#!/bin/bash
# SQ counter passes for the three FFT kernels (bench.py, 3 steps): gpurun_out/pmc_sq/{a,b}
set -e
export TMPDIR=/tmp
out=gpurun_out/pmc_sq
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/a -o a --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES -d $out/b -o b --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
python3 tools/pmc_summary.py $out/a/a_counter_collection.csv > $out/a.txt
python3 tools/pmc_summary.py $out/b/b_counter_collection.csv > $out/b.txt
cat $out/a.txt $out/b.txt
