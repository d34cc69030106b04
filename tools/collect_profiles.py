#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked): the bench line, kernel stats and PMC
summaries of the bench command as profiles/<tag>_*, the per-configuration sets as profiles/<tag>_cfg/*, and rebuild
profiles/traffic_latest.json (per-launch HBM bytes of the bench command's kernels + a `configs` section).  usage: collect_profiles.py <tag>"""
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
cfg = os.path.join(dst, tag + "_cfg")
os.makedirs(cfg, exist_ok=True)
top = {"bench.json": "bench.json", "kernel_stats.csv": "kernel_stats.csv", "pmc_fetch_size.txt": "pmc_fetch_size.txt",
       "pmc_write_size.txt": "pmc_write_size.txt", "pmc_sq_a.txt": "pmc_sq_a.txt", "pmc_sq_b.txt": "pmc_sq_b.txt",
       "bench_stats_run.json": "bench_under_rocprof.json"}
for a, b in top.items():
    shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))
workloads = {"ref": "1024^3 float32, rng='reference' (MT19937 replay + deviate-reading generation pass), two calls",
             "refone": "1024^3 float32, rng='reference' as ONE device call (rf_realise_batch_reference with one seed: what Generator runs)",
             "512": "512^3 float32, one realisation (config 2)", "2048": "2048^3 float32 on one GPU (config 4's kernels at full length)",
             "f64": "1024^3 float64", "f64ln": "1024^3 float64 + lognormal, fused (config 5)",
             "rank0": "rank 0 of 2048^3 / 8 (virtual ranks): forward + backward halves", "rank3": "rank 3 of 2048^3 / 8 (virtual ranks)",
             "rank3direct": "rank 3 of 2048^3 / 8 through the pipelined batch of the DIRECT exchange (its y pass stores into its own receive buffers)"}
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, cwd=root).strip()
t = json.load(open(os.path.join(src, "traffic.json")))
t["source"] = "profiles/traffic_latest.json (%s: profiles/%s_* and profiles/%s_cfg/) @ commit %s" % (tag, tag, tag, commit)
t["configs"] = {}
for name, what in workloads.items():
    for f in sorted(os.listdir(src)):
        if f.startswith(name + "_"):
            shutil.copy(os.path.join(src, f), os.path.join(cfg, f))
    tj = os.path.join(src, name + "_traffic.json")
    if os.path.exists(tj):
        c = json.load(open(tj))
        t["configs"][name] = {"workload": what, "grid": [2048] * 3 if name in ("2048", "rank0", "rank3", "rank3direct") else ([512] * 3 if name == "512" else [1024] * 3),
                              "run": json.load(open(os.path.join(src, name + "_run.json"))), "kernels": c["kernels"]}
json.dump(t, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print("profiles/%s_*, profiles/%s_cfg/ (%d files), traffic_latest.json: %d kernels + %d configurations"
      % (tag, tag, len(os.listdir(cfg)), len(t["kernels"]), len(t["configs"])))
