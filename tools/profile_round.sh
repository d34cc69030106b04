#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> [main|cfg|all]
# bench line + rocprofv3 kernel stats + the two PMC passes (separate runs) + SQ counters -> gpurun_out/prof_<tag>/
# (main = the bench command's passes, cfg = the per-configuration sets: two gpurun calls fit the call's time limit comfortably)
set -e
tag=${1:-x}
part=${2:-all}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
if [ "$part" != "cfg" ]; then
python3 bench.py --steps 20 --warmup 5 > $out/bench.json
rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $out/bench_stats_run.json
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/sqa -o a --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES -d $out/sqb -o b --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
python3 tools/pmc_summary.py $(find $out/sqa -name "*counter_collection.csv") > $out/pmc_sq_a.txt
python3 tools/pmc_summary.py $(find $out/sqb -name "*counter_collection.csv") > $out/pmc_sq_b.txt
python3 tools/pmc_summary.py $(find $out/fetch -name "*counter_collection.csv") > $out/pmc_fetch_size.txt
python3 tools/pmc_summary.py $(find $out/write -name "*counter_collection.csv") > $out/pmc_write_size.txt
nslab=$(python3 -c "import json;print(json.load(open('$out/bench.json'))['pipeline']['yz_slabs'])")
python3 tools/make_traffic.py $(find $out/fetch -name "*counter_collection.csv") $(find $out/write -name "*counter_collection.csv") $out/traffic.json "$tag" $nslab > $out/traffic.txt
cp $(find $out/stats -name "*kernel_stats.csv") $out/kernel_stats.csv
rm -rf $out/fetch $out/write $out/sqa $out/sqb $out/stats
fi
if [ "$part" = "main" ]; then ls $out; exit 0; fi
# every other BASELINE configuration / path: kernel stats, FETCH_SIZE / WRITE_SIZE and the SQ sets per configuration (tools/profile_cfg.sh)
for c in ref refone 512 2048 f64 f64ln rank0 rank3 rank3direct; do
  bash tools/profile_cfg.sh $out $c $c 3
done
ls $out
