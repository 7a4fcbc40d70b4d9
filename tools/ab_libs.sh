#!/bin/bash
# run tools/ab_persist.py once per library in noisereduce_amd/_ab/ (and the in-tree one): one JSON line each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-ab_libs}; mkdir -p $OUT
for f in default noisereduce_amd/_ab/lib_*.so; do
  case $f in *trace*) continue;; esac
  if [ $f = default ]; then unset SG_LIB_PATH; else export SG_LIB_PATH=$PWD/$f; fi
  echo -n "$(basename $f) " ; MODES=${MODES:-0,2} ROUNDS=${ROUNDS:-4} timeout 120 python tools/ab_persist.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['wall_ms_median'], {m: v.get('k_gate_onepass (fft+decide+smooth+mask+ifft+ola)') for m,v in d['event_ms'].items()})"
done | tee $OUT/ab.txt
