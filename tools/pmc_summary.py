#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters: tools/pmc_summary.py <counter_collection.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"]
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void rf::", "").replace("rf::", "")[:100]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
