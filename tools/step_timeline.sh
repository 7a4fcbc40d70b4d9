#!/bin/bash
# Where does a stationary step go?  rocprofv3 --kernel-trace over a short bench run, then the launches of the LAST steps
# in order: start offset, duration, and the GAP to the previous kernel's end (dependent launches on one stream).
# usage (GPU box): tools/step_timeline.sh [tag] [bench args]        -> gpurun_out/timeline_<tag>.txt
set -u
TAG=${1:-r04}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_tl_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o "$TAG" --output-format csv -- \
  python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
F=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
python3 - "$F" > "$REPO/gpurun_out/timeline_$TAG.txt" <<'PY'
import csv, sys, re, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sg::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(k):
    m = re.search(r"sg::(?:fast::)?(k_[a-z0-9_]+)", k)
    if m and m.group(1) == "k_gate_onepass" and re.search(r"k_gate_onepass<\d+, \w+, \w+, true[,>]", k):
        return "k_gate_onepass<redo>"    # second launch of the in-kernel floor test (returns at once when no chunk reported)
    return m.group(1) if m else k[:30]
# a step starts at the first noise-statistics kernel (k_stft of the clip)
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_stft"]
steps = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
steps = [s for s in steps if any(short(r["Kernel_Name"]) == "k_gate_onepass" for r in s)][-8:]
print("# last %d steps of `bench.py --steps 10` under rocprofv3 --kernel-trace (GPU timestamps, ns -> us)" % len(steps))
agg = collections.OrderedDict()
for s in steps:
    t0 = int(s[0]["Start_Timestamp"]); prev_end = None
    for r in s:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = short(r["Kernel_Name"])
        d = agg.setdefault(k + "#" + str(sum(1 for x in agg if x.startswith(k + "#"))) if False else k, [0.0, 0.0, 0])
        d[0] += (b - a) / 1e3; d[1] += ((a - prev_end) / 1e3 if prev_end is not None else 0.0); d[2] += 1
        prev_end = b
    span = (int(s[-1]["End_Timestamp"]) - t0) / 1e3
    agg.setdefault("__span", [0.0, 0.0, 0]); agg["__span"][0] += span; agg["__span"][2] += 1
print("%-22s %10s %14s" % ("kernel", "run us", "gap before us"))
tot_run = tot_gap = 0.0
for k, (run, gap, n) in agg.items():
    if k == "__span": continue
    per_step = n / len(steps)
    print("%-22s %10.2f %14.2f   (x%.0f per step)" % (k, run / len(steps), gap / len(steps), per_step))
    tot_run += run / len(steps); tot_gap += gap / len(steps)
print("%-22s %10.2f %14.2f" % ("sum", tot_run, tot_gap))
print("first kernel start -> last kernel end: %.2f us per step" % (agg["__span"][0] / agg["__span"][2]))
s = steps[-1]; t0 = int(s[0]["Start_Timestamp"])
print("\n# the last step, launch by launch: start offset, duration, gap before (us)")
prev = None
for r in s:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-22s start %8.2f  run %8.2f  gap %6.2f  grid %s wg %s" % (short(r["Kernel_Name"]), (a - t0) / 1e3, (b - a) / 1e3,
          ((a - prev) / 1e3 if prev else 0.0), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
    prev = b
PY
cat "$REPO/gpurun_out/timeline_$TAG.txt"
