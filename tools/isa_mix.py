#!/usr/bin/env python3
"""Static instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing, priced with the per-class pipe costs measured by
tools/ubench/valu_classes.hip (profiles/r05_valu_classes.txt): cycles a wave64 instruction of the class holds its SIMD's
vector pipe.  usage: tools/isa_mix.py <listing.s> <mangled kernel name substring> [waves per SIMD per launch]"""
import collections, re, sys
src = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(\S*%s\S*):[^\n]*\n(.*?)^\s*s_endpgm" % re.escape(key), src, flags=re.S | re.M)
if not m:
    raise SystemExit("kernel not found")
body = m.group(2)
ops = collections.Counter()
for ln in body.splitlines():
    ln = ln.strip()
    if not ln or ln[0] in ";./" or ln.endswith(":"):
        continue
    ops[ln.split()[0]] += 1

def cls(op):
    if not op.startswith("v_"):
        return None
    if op.startswith(("v_mfma", "v_smfmac")):
        return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane<->sgpr"
    if re.match(r"v_(log|exp|rcp|rsq|sqrt|sin|cos)_f32", op):
        return "transcendental f32"
    if re.search(r"_f64|_b64|_u64|_i64", op):
        return "64-bit"
    if re.match(r"v_(fma|fmac|add|sub|subrev|mul|mac|max|min)_f32(_e32|_e64)?$", op):
        return "f32 arithmetic (fast)"
    if re.match(r"v_pk_", op):
        return "packed"
    if "dpp" in op:
        return "dpp"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_cmp"):
        return "compare"
    return "other 32-bit (int / logic / mov / cvt)"

COST = {"f32 arithmetic (fast)": 2.4, "transcendental f32": 8.0, "64-bit": 4.6, "dpp": 4.2, "cndmask": 4.2, "compare": 4.2,
        "other 32-bit (int / logic / mov / cvt)": 4.2, "packed": 4.8, "lane<->sgpr": 4.2, "mfma": 0.0}
by = collections.Counter()
for op, n in ops.items():
    c = cls(op)
    if c:
        by[c] += n
tot = sum(by.values())
print("kernel", m.group(1))
print("static instructions: %d total, %d VALU, %d SALU, %d LDS, %d VMEM/FLAT" % (
    sum(ops.values()), tot, sum(n for o, n in ops.items() if o.startswith("s_")), sum(n for o, n in ops.items() if o.startswith("ds_")),
    sum(n for o, n in ops.items() if o.startswith(("global_", "buffer_", "flat_", "scratch_")))))
cyc = 0.0
for c, n in by.most_common():
    cyc += n * COST[c]
    print("  %-42s %6d  x %.1f cycles = %8.0f" % (c, n, COST[c], n * COST[c]))
print("  vector-pipe cycles per wave (static, straight-line): %.0f  (%.2f cycles per VALU instruction on average)" % (cyc, cyc / max(tot, 1)))
top = sorted(((n, o) for o, n in ops.items() if o.startswith("v_")), reverse=True)[:25]
print("  most frequent:", ", ".join("%s %d" % (o, n) for n, o in top))
