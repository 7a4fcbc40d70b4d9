#!/bin/bash
# Build the current tree into noisereduce_amd/_ab/lib_<tag>.so (cross-compiling here); run variants on the GPU box with
#   for f in noisereduce_amd/_ab/*.so; do SG_LIB_PATH=$PWD/$f python tools/time_onepass.py; done
# usage: tools/ab_build.sh <tag> [extra hipcc flags]      (flags of __graft_entry__.build(): no SLP vectorisation)
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p noisereduce_amd/_ab
cd noisereduce_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
  -Wl,--version-script=exports.map api.hip nonstat_mask.hip mixed.hip -o ../_ab/lib_$TAG.so -fno-slp-vectorize "$@" 2>/dev/null
ls -la ../_ab/lib_$TAG.so
