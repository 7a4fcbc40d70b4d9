#!/bin/bash
# Round-6 measurements in one gpurun call: GPU test suite, bench lines (configs[1], [2], [3]-share), rocprofv3 kernel stats of the
# bench command, PMC traffic, throughput over n_fft (incl. the mixed-radix sizes) and sample dtypes, step timeline.
# usage (on the GPU box): tools/round_final_r06.sh <tag>      -> gpurun_out/<tag>/
set -u
TAG=${1:-r06_v1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
(python -m pytest tests -m gpu -q 2>&1 | tail -5) > "$OUT/pytest_gpu.txt"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --workload config3 --no-cpu-baseline > "$OUT/bench_config3.json" 2>> "$OUT/bench.err"
python bench.py --workload config4 --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/bench_config4.json" 2>> "$OUT/bench.err"
for S in 2 3; do python bench.py --streams $S --no-extras --no-cpu-baseline > "$OUT/bench_streams$S.json" 2>> "$OUT/bench.err"; done   # serving mode: S calls in flight (not the headline)
tools/gpu_profile.sh ${TAG}k --no-extras > "$OUT/prof.log" 2>&1
tools/gpu_profile.sh ${TAG}k3 --no-extras --workload config3 > "$OUT/prof3.log" 2>&1
for t in ${TAG}k ${TAG}k3; do F=$(find gpurun_out/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && (head -1 "$F"; grep "sg::" "$F") > "$OUT/${t}_kernel_stats.csv"; done
tools/gpu_traffic.sh > "$OUT/traffic.log" 2>&1
cp gpurun_out/traffic/traffic.json gpurun_out/traffic/traffic_detail.json "$OUT/" 2>/dev/null
NFFT=256,400,512,1000,1024,1536,2048,3000,4096,8192 WARM=20 REPS=20 python tools/time_nfft.py > "$OUT/time_nfft.txt" 2>&1
tools/prof_nfft.sh ${TAG}_nfft "256 512 2048 400 1000" stat > /dev/null 2>&1
tools/prof_nfft.sh ${TAG}_nfft "256 512 2048" nonstat > /dev/null 2>&1
cp gpurun_out/${TAG}_nfft/*kernel_stats.csv "$OUT/" 2>/dev/null
tools/step_timeline.sh $TAG > "$OUT/step_timeline.log" 2>&1; cp gpurun_out/timeline_$TAG.txt "$OUT/" 2>/dev/null
python tools/rowgate_scale.py > "$OUT/rowgate_scale.txt" 2>&1; cp gpurun_out/rowgate_scale.json "$OUT/" 2>/dev/null
cat "$OUT/pytest_gpu.txt"; head -c 400 "$OUT/bench.json"; echo; tail -3 "$OUT/traffic.log" | cut -c1-300
