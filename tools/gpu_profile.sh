#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of a short bench run.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o "$TAG" --output-format csv -- \
  python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name '*kernel_stats.csv' | head -3
F=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && head -30 "$F"
# keep the trace small: drop the per-dispatch csv, keep stats
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
