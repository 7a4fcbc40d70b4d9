// Micro-benchmark: issue rate of independent non-packed v_fma_f32 / v_add_f32 vs v_pk_fma_f32 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      } else {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 v = {r[i], r[i + 1]};
          f2 av = {a, a}, bv = {b, b};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
          r[i] = v.x; r[i + 1] = v.y;
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int flops_per_instr, int instr_per_rep) {
  const int blocks = 256 * 8, iters = 2000;
  float* out; CHK(hipMalloc(&out, blocks * 256 * 4));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  double instr = (double)blocks * 4 /*waves*/ * iters * 8 * instr_per_rep;
  double flops = instr * 64 * flops_per_instr;
  // cycles per wave-instruction per SIMD: 1024 SIMDs at 2.4 GHz
  double cyc = ms * 1e-3 * 2.4e9 * 1024 / instr;
  printf("%-14s %.3f ms  %.1f TFLOP/s  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, flops / ms / 1e9, cyc);
  return 0;
}

int main() {
  run<0>("v_fma_f32", 2, 16);
  run<1>("v_add_f32", 1, 16);
  run<2>("v_pk_fma_f32", 4, 8);
  return 0;
}
