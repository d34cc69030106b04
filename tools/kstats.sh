#!/bin/bash
# usage (GPU box): bash tools/kstats.sh <outdir> <case> [reps] [variant library]  -- rocprofv3 kernel stats of one tools/cfg_workload.py case
set -e
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
w=$out/.ks_$1
rm -rf $w
rocprofv3 --kernel-trace --stats -d $w -o s --output-format csv -- python3 tools/cfg_workload.py "$@" > $out/$1_run.json
cp $(find $w -name "*kernel_stats.csv") $out/$1_kernel_stats.csv
rm -rf $w
python3 - $out/$1_kernel_stats.csv <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(rf::.*', '', r['Name']).replace('void rf::', '').replace('rf::', '').replace('(anonymous namespace)::', '')
    if float(r['Percentage']) > 0.05:
        print("%-118s calls %5s avg_us %9.1f pct %s" % (n[:118], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
