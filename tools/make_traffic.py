#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into per-kernel HBM
bytes per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): both counters are in
KiB; FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled.
usage: tools/make_traffic.py <fetch.csv> <write.csv> <out.json> [note] [y/z launches per realisation]"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    try:
        import subprocess
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "unknown"
    out = {"note": note, "source": "profiles/traffic_latest.json (%s) @ commit %s" % (note, commit), "unit": "bytes per launch",
           "correction": "read = 2 * FETCH_SIZE * 1024 (gfx950 wide-read under-count), write = WRITE_SIZE * 1024",
           "grid": [int(v) for v in sys.argv[6].split("x")] if len(sys.argv) > 6 else [1024, 1024, 1024], "yz_slabs": int(sys.argv[5]) if len(sys.argv) > 5 else None,
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "rf::" not in k:
            continue
        rd, wr = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out["kernels"][short] = {"read": rd, "write": wr, "total": rd + wr}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-110s read %.3e write %.3e" % (k[:110], v["read"], v["write"]))


if __name__ == "__main__":
    main()
