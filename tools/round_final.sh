#!/bin/bash
# Final measurements of a round, one gpurun call: GPU test suite, bench lines (configs[1], [2], [3]-share), rocprofv3
# kernel stats of the bench command, PMC HBM traffic (configs[1], [2], [4]), throughput over n_fft and sample dtypes.
# usage (on the GPU box): tools/round_final.sh <tag>      -> gpurun_out/<tag>/
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
(python -m pytest tests -m gpu -q 2>&1 | tail -5) > "$OUT/pytest_gpu.txt"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --workload config3 --no-cpu-baseline > "$OUT/bench_config3.json" 2>> "$OUT/bench.err"
python bench.py --workload config4 --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/bench_config4.json" 2>> "$OUT/bench.err"
tools/gpu_profile.sh ${TAG}k --no-extras > "$OUT/prof.log" 2>&1
tools/gpu_profile.sh ${TAG}k3 --no-extras --workload config3 > "$OUT/prof3.log" 2>&1
tools/gpu_traffic.sh > "$OUT/traffic.log" 2>&1
cp gpurun_out/traffic/traffic.json gpurun_out/traffic/traffic_detail.json "$OUT/" 2>/dev/null
for t in ${TAG}k ${TAG}k3; do F=$(find gpurun_out/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && (head -1 "$F"; grep "sg::" "$F") > "$OUT/${t}_kernel_stats.csv"; done
python tools/time_nfft.py > "$OUT/time_nfft.json" 2>&1
python tools/time_dtypes.py > "$OUT/time_dtypes.txt" 2>&1
python tools/prof_torchgate.py > "$OUT/prof_torchgate.txt" 2>&1
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_fb -o fb --output-format csv -- python $REPO/tools/prof_fwdbwd.py > /dev/null 2>&1; F=$(find $REPO/gpurun_out/prof_${TAG}_fb -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && (head -1 "$F"; grep "sg::" "$F") > "$OUT/fwdbwd_kernel_stats.csv")
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_i16 -o i16 --output-format csv -- python $REPO/tools/prof_int16.py > /dev/null 2>&1; F=$(find $REPO/gpurun_out/prof_${TAG}_i16 -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && (head -1 "$F"; grep "sg::" "$F") > "$OUT/int16_kernel_stats.csv")
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_x -o x --output-format csv -- python $REPO/tools/prof_exact.py > /dev/null 2>&1; F=$(find $REPO/gpurun_out/prof_${TAG}_x -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && (head -1 "$F"; grep "sg::" "$F") > "$OUT/float64_pipeline_kernel_stats.csv")
python tools/time_host_path.py > "$OUT/time_host_path.txt" 2>&1
cat "$OUT/pytest_gpu.txt"; head -c 300 "$OUT/bench.json"; echo; tail -3 "$OUT/traffic.log"
python tools/rowgate_scale.py > "$OUT/rowgate_scale.txt" 2>&1; cp gpurun_out/rowgate_scale.json "$OUT/" 2>/dev/null
python tools/rowgate_probe.py > "$OUT/rowgate_probe.txt" 2>&1; cp gpurun_out/rowgate_probe.json "$OUT/" 2>/dev/null
tools/step_timeline.sh $TAG > "$OUT/step_timeline.log" 2>&1; cp gpurun_out/timeline_$TAG.txt "$OUT/" 2>/dev/null
