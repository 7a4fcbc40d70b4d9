#!/bin/bash
# Per-kernel rocprofv3 averages of reduce_noise over n_fft (2 min of audio).  usage: tools/prof_nfft.sh <tag> "256 512 2048" [stat|nonstat]
set -u
TAG=${1:-nfft}; NFFTS=${2:-"256 512 2048"}; KIND=${3:-stat}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in $NFFTS; do
  D=/tmp/prof_nfft_$n
  rocprofv3 --kernel-trace --stats -d $D -o p --output-format csv -- python $REPO/tools/prof_nfft_loop.py $n $KIND > /dev/null 2>&1
  F=$(find $D -name '*kernel_stats.csv' | head -1)
  [ -n "$F" ] && (head -1 "$F"; grep "sg::" "$F") > "$OUT/nfft${n}_${KIND}_kernel_stats.csv"; [ -n "$F" ] && grep -v "sg::" "$F" | head -8 > "$OUT/nfft${n}_${KIND}_other_kernels.csv"
  echo "== n_fft $n $KIND"; cut -d, -f1-4 "$OUT/nfft${n}_${KIND}_kernel_stats.csv" | cut -c1-160
done
