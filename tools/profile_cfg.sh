#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_cfg.sh <outdir> <name> <case> [reps] [variant library]
# One configuration of tools/cfg_workload.py through rocprofv3: kernel stats, then FETCH_SIZE, WRITE_SIZE and the two SQ
# counter sets, each in a pass of its own (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never with
# trace domains other than the kernel trace).  Leaves <outdir>/<name>_{kernel_stats.csv,pmc_*.txt,traffic.json,run.json}.
set -e
out=$1; name=$2; shift 2
mkdir -p $out
export TMPDIR=/tmp
w=$out/.work_$name
rm -rf $w
python3 tools/cfg_workload.py "$@" > $out/${name}_run.json
rocprofv3 --kernel-trace --stats -d $w/stats -o s --output-format csv -- python3 tools/cfg_workload.py "$@" > /dev/null
cp $(find $w/stats -name "*kernel_stats.csv") $out/${name}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $w/fetch -o f --output-format csv -- python3 tools/cfg_workload.py "$@" > /dev/null
rocprofv3 --pmc WRITE_SIZE -d $w/write -o w --output-format csv -- python3 tools/cfg_workload.py "$@" > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $w/sqa -o a --output-format csv -- python3 tools/cfg_workload.py "$@" > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES -d $w/sqb -o b --output-format csv -- python3 tools/cfg_workload.py "$@" > /dev/null
python3 tools/pmc_summary.py $(find $w/sqa -name "*counter_collection.csv") > $out/${name}_pmc_sq_a.txt
python3 tools/pmc_summary.py $(find $w/sqb -name "*counter_collection.csv") > $out/${name}_pmc_sq_b.txt
python3 tools/make_traffic.py $(find $w/fetch -name "*counter_collection.csv") $(find $w/write -name "*counter_collection.csv") $out/${name}_traffic.json "$name" > $out/${name}_traffic.txt
rm -rf $w
echo "profiled $name: $(cat $out/${name}_run.json)"
