#!/bin/bash
# (diagnosis) the persistent gate next to a matmul loop on a second stream (tools/experiments/persist_tilecount.py), once per variant:
# resident workgroups per CU 3 / 2 / 1 with the in-tree library, then the A/B libraries named on the command line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/persist_bisect; mkdir -p $OUT
run() { echo "== $1"; timeout 240 python tools/experiments/persist_tilecount.py 2>&1 | grep -v amdgpu.ids | tail -n 8; }
for n in 3 2 1; do SG_ONEPASS_WG_PER_CU=$n run "default wg_per_cu=$n"; done 2>&1 | tee $OUT/default.txt
for t in "$@"; do SG_LIB_PATH=$PWD/noisereduce_amd/_ab/lib_$t.so run "lib_$t"; done 2>&1 | tee $OUT/libs.txt
