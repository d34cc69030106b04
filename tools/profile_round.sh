#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# bench line + rocprofv3 kernel stats + the two PMC passes (separate runs) -> gpurun_out/prof_<tag>/
set -e
tag=${1:-x}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 2 > $out/bench.json
rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/bench_stats_run.json
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
find $out -name "*.csv" | head -20
