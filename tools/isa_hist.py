#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a hipcc -S dump: tools/isa_hist.py file.s <regex on mangled name>"""
import collections
import re
import sys

CATS = [
    ("vmem", r"^(global_|buffer_|flat_|scratch_)"), ("lds", r"^ds_"), ("smem", r"^s_load|^s_buffer"),
    ("trans", r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_"), ("quarter_int", r"^v_(mul_lo_u32|mul_hi_u32|mad_u64_u32|mul_hi_i32|mad_i64_i32)"),
    ("f64", r"^v_.*_f64"), ("valu_f32", r"^v_.*f32"), ("valu_int", r"^v_"), ("s_nop", r"^s_nop"),
    ("waitcnt", r"^s_waitcnt"), ("barrier", r"^s_barrier"), ("branch", r"^s_cbranch|^s_branch"), ("salu", r"^s_"),
]


def main():
    lines = open(sys.argv[1]).read().split("\n")
    pat = re.compile(sys.argv[2])
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ": " in l and pat.search(l.split(":")[0]):
            end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
            ins = [x.strip().split()[0] for x in lines[i + 1:end] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            cat = collections.Counter()
            for op in ins:
                for name, rx in CATS:
                    if re.match(rx, op):
                        cat[name] += 1
                        break
                else:
                    cat["other"] += 1
            print(l.split(":")[0][:110])
            print("  total", len(ins), dict(cat))
            if len(sys.argv) > 3:
                for k, v in collections.Counter(ins).most_common(int(sys.argv[3])):
                    print("   ", v, k)


if __name__ == "__main__":
    main()
