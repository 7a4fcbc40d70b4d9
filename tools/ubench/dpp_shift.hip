// Full-wave lane shifts on gfx950: DPP wave_shr:1 / wave_shl:1 (VALU modifier) against ds_bpermute_b32 (LDS crossbar).
// Checks the semantics (lane l <- lane l -/+ 1 across the 16-lane row and 32-lane boundaries, zero at the wave's end with
// bound_ctrl) and times a boxcar of width 6 built both ways.   hipcc --offload-arch=gfx950 -O3 dpp_shift.hip -o dpp_shift
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__global__ void k_sem(float* o) {
  const float v = (float)(threadIdx.x + 1);
  o[threadIdx.x] = shr1(v);
  o[64 + threadIdx.x] = shl1(v);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* o, int iters) {
  float x[8];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (float)(lane * (j + 1) % 7);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float b = x[j];
      if (MODE == 0) {
#pragma unroll
        for (int d = 0; d < 5; ++d) b = shr1(b) + x[j];
      } else {
        float acc = x[j];
#pragma unroll
        for (int d = 1; d <= 5; ++d) acc += __shfl(x[j], lane - d);
        b = acc;
      }
      x[j] = b * 0.125f;
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j];
  o[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d;
  hipMalloc(&d, 4 << 20);
  k_sem<<<1, 64>>>(d);
  std::vector<float> h(128);
  hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    if (h[l] != (l ? (float)l : 0.f)) ++bad;
    if (h[64 + l] != (l < 63 ? (float)(l + 2) : 0.f)) ++bad;
  }
  printf("semantics: %d mismatches (shr lanes 0,1,16,32: %g %g %g %g; shl lanes 15,31,62,63: %g %g %g %g)\n", bad, h[0], h[1], h[16],
         h[32], h[64 + 15], h[64 + 31], h[64 + 62], h[64 + 63]);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, blocks = 256 * 8;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) k_rate<0><<<blocks, 256>>>(d, iters); else k_rate<1><<<blocks, 256>>>(d, iters);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double shifts = (double)blocks * 4 * iters * 8 * 5;   // wave-level shift+add pairs
    printf("%s: %.3f ms, %.2f cycles per (shift+add) per CU at 1.95 GHz\n", mode ? "ds_bpermute + add" : "dpp wave_shr:1 add", ms,
           ms * 1e-3 * 1.95e9 * 256 / shifts);
  }
  return 0;
}
