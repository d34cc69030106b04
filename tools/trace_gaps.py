#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints, for the last graph-replayed realisations, the time inside kernels, the gaps
between consecutive kernels and the per-kernel-name sums: how much of a step is launch gap / drain.  usage: trace_gaps.py trace.csv [n_last]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 350
rows = rows[-n_last:]
busy = 0
gaps = []
names = collections.defaultdict(lambda: [0, 0])
for a, b in zip(rows, rows[1:]):
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    k = r["Kernel_Name"].split("<")[0].split("(")[0][-40:] + ("|" + r["Kernel_Name"][60:100] if "<" in r["Kernel_Name"] else "")
    names[k][0] += d
    names[k][1] += 1
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
g = sorted(gaps)
print("kernels %d  span %.3f ms  in kernels %.3f ms  gaps %.3f ms (%.1f %%)  median gap %.2f us  p90 %.2f us  max %.1f us" %
      (len(rows), span / 1e6, busy / 1e6, sum(gaps) / 1e6, 100.0 * sum(gaps) / span, g[len(g) // 2] / 1e3, g[int(len(g) * 0.9)] / 1e3, g[-1] / 1e3))
neg = [x for x in gaps if x < 0]
print("overlapping pairs:", len(neg))
for k, (d, c) in sorted(names.items(), key=lambda kv: -kv[1][0]):
    print("  %9.3f ms  %5d x %7.1f us  %s" % (d / 1e6, c, d / c / 1e3, k))
