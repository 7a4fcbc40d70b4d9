#!/bin/bash
# What would "the noise statistics inside the gate launch"  buy?  Upper bound by emulation: libraries built with
#   -DSG_EXP_SKIP_STATS            sg_noise_stats returns at once after the first call (thresholds kept): the step without its three
#                                  statistics launches
#   -DOP_EXP_STALL_NS=<ns>         every first-round tile of k_gate_onepass waits at its decision stage until <ns> after its
#                                  workgroup started -- the moment an in-launch statistics chain would publish the thresholds
# (slots the statistics workgroups would occupy are NOT emulated: optimistic by a few us).  Build here, run on the GPU box:
#   tools/experiments/stats_in_launch.sh build ; gpurun -- 'tools/experiments/stats_in_launch.sh run'
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  tools/ab_build.sh exp_nostats -DSG_EXP_SKIP_STATS &
  for ns in 10000 15000 20000 25000 30000; do tools/ab_build.sh exp_stall$ns -DSG_EXP_SKIP_STATS -DOP_EXP_STALL_NS=$ns & done
  wait
else
  OUT=gpurun_out/exp_stats; mkdir -p $OUT
  for f in default noisereduce_amd/_ab/lib_exp_nostats.so noisereduce_amd/_ab/lib_exp_stall*.so; do
    if [ $f = default ]; then unset SG_LIB_PATH; else export SG_LIB_PATH=$PWD/$f; fi
    echo -n "$(basename $f) "; python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('ms_per_step_median'))"
  done | tee $OUT/exp.txt
fi
