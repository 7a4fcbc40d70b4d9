#!/bin/bash
# Host-side AddressSanitizer run of the C ABI: the library's HOST code (table builders, workspace
# management, argument handling -- the device code is untouched) and the plain-C caller tests/c_abi/example.c are built
# with -fsanitize=address and the caller is run on the GPU box.
#   here:        tools/asan_check.sh build        -> noisereduce_amd/_ab/lib_asan.so, noisereduce_amd/_ab/c_abi_asan
#   GPU box:     tools/asan_check.sh run [n]      -> prints the caller's output; ASan aborts on any finding
set -e
cd "$(dirname "$0")/.."
CLANG=/opt/rocm/lib/llvm/bin/clang
RT=$(dirname "$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)")
if [ "${1:-build}" = build ]; then
  mkdir -p noisereduce_amd/_ab
  (cd noisereduce_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fvisibility=hidden \
     -Wl,--version-script=exports.map api.hip nonstat_mask.hip -o ../_ab/libmi355gate_asan.so -ldl \
     -Xclang -target-feature -Xclang -packed-fp32-ops -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
     -shared-libasan 2>/dev/null)
  $CLANG -std=c99 -g -fsanitize=address -shared-libasan tests/c_abi/example.c -Iinclude -Lnoisereduce_amd/_ab -lmi355gate_asan \
     -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$RT" -o noisereduce_amd/_ab/c_abi_asan
  ls -la noisereduce_amd/_ab/libmi355gate_asan.so noisereduce_amd/_ab/c_abi_asan
else
  # the HIP runtime maps device memory into the shadow gap and keeps allocations alive at exit
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:abort_on_error=1 LD_LIBRARY_PATH="$RT:$LD_LIBRARY_PATH" \
    noisereduce_amd/_ab/c_abi_asan ${2:-700000} && echo "asan_check: clean"
fi
