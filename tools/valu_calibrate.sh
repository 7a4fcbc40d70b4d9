#!/bin/bash
# Run on the GPU box: tools/ubench/valu_classes.bin plain (event timing per instruction class) and under rocprofv3 --pmc
# (two counter passes, kernel-trace only): what SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CYCLES read for a vector pipe
# that is known to be saturated by one instruction class.  -> gpurun_out/valu_cal/{plain.txt,pass1.txt,pass2.txt}
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/valu_cal
mkdir -p "$OUT"
BIN=$REPO/tools/ubench/valu_classes.bin
$BIN > "$OUT/plain.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
summ() {
python3 - "$1" <<'PY'
import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"k<(\d+)>", k)
    if not m: continue
    key = (int(m.group(1)), r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = agg.get(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# keep the LAST dispatch of every mode (the timed one: 64 iterations)
last = {}
for (mode, disp), v in agg.items():
    if mode not in last or int(disp) > last[mode][0]:
        last[mode] = (int(disp), v)
for mode in sorted(last):
    print("mode", mode, {c: round(x, 1) for c, x in sorted(last[mode][1].items())})
PY
}
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT/p1" -o p1 --output-format csv -- $BIN > "$OUT/p1.log" 2>&1
F=$(find "$OUT/p1" -name '*counter_collection.csv' | head -1); [ -n "$F" ] && summ "$F" > "$OUT/pass1.txt"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU -d "$OUT/p2" -o p2 --output-format csv -- $BIN > "$OUT/p2.log" 2>&1
F=$(find "$OUT/p2" -name '*counter_collection.csv' | head -1); [ -n "$F" ] && summ "$F" > "$OUT/pass2.txt"
rm -rf "$OUT/p1" "$OUT/p2"
cat "$OUT/plain.txt" "$OUT/pass1.txt" "$OUT/pass2.txt"; tail -3 "$OUT/p2.log"
