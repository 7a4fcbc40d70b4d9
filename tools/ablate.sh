#!/bin/bash
# Build k_apply_fast ablation variants (SG_ABLATE bit mask, fastpath.hpp) into noisereduce_amd/_ab/
# usage: tools/ablate.sh "0 1 2 4 8 16"   (here, cross-compiling);  then on the GPU box:
#        for f in noisereduce_amd/_ab/*.so; do SG_LIB_PATH=$PWD/$f python bench.py --steps 30 --no-cpu-baseline; done
set -e
cd "$(dirname "$0")/.."
mkdir -p noisereduce_amd/_ab
for m in ${1:-0 1 2 4 8 16}; do
  ( cd noisereduce_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared api.hip nonstat_mask.hip \
      -o ../_ab/lib_ab$m.so -Xclang -target-feature -Xclang -packed-fp32-ops -DSG_ABLATE=$m $SG_EXTRA 2>/dev/null ) &
done
wait
ls -la noisereduce_amd/_ab
