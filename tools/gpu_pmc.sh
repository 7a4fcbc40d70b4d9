#!/bin/bash
# Run on the GPU box: rocprofv3 PMC pass(es) over a short bench run (kernel-trace + counters only).
# usage: tools/gpu_pmc.sh <tag> "<COUNTER1 COUNTER2 ...>" [bench args...]
set -u
TAG=${1:-pmc}; shift
CTRS=${1:-SQ_WAVES}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"}
rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT" -o "$TAG" --output-format csv -- $CMD > "$OUT/bench.json" 2> "$OUT/bench.err"
F=$(find "$OUT" -name '*counter_collection.csv' | head -1)
if [ -n "$F" ]; then
python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "sg::" not in r["Kernel_Name"]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in agg:
    print(k, "dispatches", cnt[k], {c: round(v / cnt[k], 1) for c, v in agg[k].items()})
PY
else
  echo "no counter csv"; tail -5 "$OUT/bench.err"
fi
