#!/usr/bin/env python3
"""Weight the static instruction mix of one kernel (hipcc -S dump) with the per-opcode VALU throughputs measured
by tools/op_rate.hip on MI355X (cycles per wave64 instruction per SIMD).  usage: isa_cost.py file.s <regex> [top]"""
import collections
import re
import sys

def cost(op):
    if not op.startswith("v_"):
        return 0.0
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", op): return 8.2
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op): return 16.3
    if re.search(r"_f64", op): return 5.0
    if op.startswith("v_pk_"): return 4.5
    if re.match(r"v_(mad_u64_u32|mad_i64_i32|mul_lo_u32|mul_hi_u32|mul_hi_i32|mul_u32_u24|mul_hi_u32_u24|mad_u32_u24|mad_i32_i24|mul_i32_i24)", op): return 4.3
    if re.match(r"v_(lshl_add_u32|lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|alignbit_b32|perm_b32|med3|fract|cvt_|add3_u32|lshl_or_b32|and_or_b32|or3_b32|xad_u32|add_lshl_u32|bfe_|bfi_|mad_|min3|max3|ldexp|rndne|floor|trunc|cndmask_b32_e64|readlane|readfirstlane|cmp_.*_e64)", op): return 4.2
    if re.match(r"v_(fma_f32|fmac_f32|fmamk_f32|fmaak_f32|add_f32|sub_f32|subrev_f32|mul_f32|max_f32|min_f32)", op): return 2.8
    return 2.4

def main():
    lines = open(sys.argv[1]).read().split("\n")
    pat = re.compile(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ": " in l and pat.search(l.split(":")[0]):
            end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
            ins = [x.strip().split()[0] for x in lines[i + 1:end] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            cnt = collections.Counter(ins)
            tot = sum(cost(o) * n for o, n in cnt.items())
            print(l.split(":")[0][:120])
            print("  VALU instructions %d, weighted cycles per wave %.0f" % (sum(n for o, n in cnt.items() if o.startswith("v_")), tot))
            for o, n in sorted(cnt.items(), key=lambda kv: -cost(kv[0]) * kv[1])[:top]:
                print("   %6.0f cyc (%4.1f%%)  %4d x %-22s @%.1f" % (cost(o) * n, 100 * cost(o) * n / tot, n, o, cost(o)))

if __name__ == "__main__":
    main()
