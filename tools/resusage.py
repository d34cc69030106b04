#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr dump files)."""
import re
import subprocess
import sys

KEYS = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("spill", r"VGPRs Spill"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]"), ("SGPR", r"SGPRs")]


def main():
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    txt = open(sys.argv[1]).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    for b in blocks:
        name = b.split("\n")[0].strip().split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = dem.replace("rf::", "")
        if pat and not re.search(pat, short):
            continue
        vals = []
        for label, k in KEYS:
            m = re.search(k + r": (\d+)", b)
            vals.append("%s=%s" % (label, m.group(1) if m else "?"))
        print(" ".join(vals), " ", short[:150])


if __name__ == "__main__":
    main()
