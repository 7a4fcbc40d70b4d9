// Micro-benchmark: do the matrix pipe and the vector pipe of a CDNA4 SIMD run side by side?
//
// One workgroup of 12 wavefronts per CU (3 per SIMD, the occupancy of k_gate_onepass).  On every SIMD, wave role by mode:
//   mode 0  V V V   three waves of float32 FMA chains (the gate kernel today: everything on the vector pipe)
//   mode 1  V V -   two waves of FMA chains, the third exits at once
//   mode 2  V V M   two waves of FMA chains + one wave issuing v_mfma_f32_16x16x4_f32 back to back
//   mode 3  - - M   the MFMA wave alone
//   mode 4  V V m   as mode 2, but the MFMA wave interleaves its products with the VALU work a DFT16-as-GEMM stage needs
//                   around them (operand shuffles: 2 v_mov per product)
// Per wave: shader-clock cycles for a fixed number of instructions; reported: VALU wave-instructions per cycle per SIMD
// and MFMA products per cycle per SIMD for each mode.  If mode 2's VALU rate equals mode 1's, work moved from the
// vector pipe to the matrix pipe is free for the neighbouring waves (MI355X_MICROARCH.md: separate pipes).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o gpurun_out/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int WAVES = 12;
constexpr int V_PER_IT = 32 * 16;     // VALU wave-instructions per loop iteration
constexpr int M_PER_IT = 32 * 4;      // MFMA wave-instructions per loop iteration

__global__ __launch_bounds__(WAVES * 64) void k(float* out, long long* cyc, int mode, int iters, float a, float b) {
  const int wave = threadIdx.x >> 6;
  const int slot = wave >> 2;           // waves w, w + 4, w + 8 share SIMD w % 4 (round-robin placement); slot 0..2
  // roles: slot 0, 1 = VALU (modes 0, 1, 2, 4), slot 2 = VALU (mode 0) / MFMA (modes 2, 3, 4) / idle (mode 1)
  int role;                              // 0 idle, 1 VALU, 2 MFMA, 3 MFMA + shuffles
  if (slot < 2) role = (mode == 3) ? 0 : 1;
  else role = mode == 0 ? 1 : (mode == 1 ? 0 : (mode == 4 ? 3 : 2));
  long long dt = 0;
  float res = 0.f;
  if (role == 1) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
      asm volatile(".rept 32\n"
                   "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                   "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                   "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                   "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                   ".endr\n"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                     "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                   : "v"(a), "v"(b));
    }
    dt = clock64() - t0;
#pragma unroll
    for (int i = 0; i < 16; ++i) res += r[i];
  } else if (role == 2 || role == 3) {
    v4f acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float x = threadIdx.x * 0.01f, y = 1.0f + threadIdx.x * 0.001f, s0 = a, s1 = b;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
      if (role == 2) {
        asm volatile(".rept 32\n"
                     "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                     "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                     ".endr\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(x), "v"(y));
      } else {
        asm volatile(".rept 32\n"
                     "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mov_b32 %6, %4\n v_mov_b32 %7, %5\n"
                     "v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n v_mov_b32 %6, %5\n v_mov_b32 %7, %4\n"
                     "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mov_b32 %6, %4\n v_mov_b32 %7, %5\n"
                     "v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n v_mov_b32 %6, %5\n v_mov_b32 %7, %4\n"
                     ".endr\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(x), "v"(y), "v"(s0), "v"(s1));
      }
    }
    dt = clock64() - t0;
#pragma unroll
    for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    res += s0 + s1;
  }
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * WAVES + wave] = dt;
  if (res == 12345.678f) out[0] = res;
}

int main() {
  int dev = 0;
  CHK(hipSetDevice(dev));
  hipDeviceProp_t pr;
  CHK(hipGetDeviceProperties(&pr, dev));
  const int cus = pr.multiProcessorCount;
  float* out;
  long long* cyc;
  CHK(hipMalloc(&out, 64));
  CHK(hipMalloc(&cyc, sizeof(long long) * cus * WAVES));
  const int iters = 2000;
  printf("# %s, %d CUs; one workgroup of %d waves per CU (3 per SIMD); %d iterations of %d VALU / %d MFMA wave-instructions\n",
         pr.gcnArchName, cus, WAVES, iters, V_PER_IT, M_PER_IT);
  printf("# mode | VALU waves: cycles per VALU wave-instruction (per wave) -> VALU wave-instr per cycle per SIMD | MFMA wave: cycles "
         "per v_mfma_f32_16x16x4_f32 -> products per cycle per SIMD | kernel ms\n");
  const char* names[5] = {"V V V", "V V -", "V V M", "- - M", "V V M+2mov"};
  for (int mode = 0; mode < 5; ++mode) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {     // first launch warms up
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(cus), dim3(WAVES * 64), 0, 0, out, cyc, mode, iters, 1.0001f, 0.5f);
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
    }
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h((size_t)cus * WAVES);
    CHK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double vsum = 0, msum = 0;
    int vn = 0, mn = 0;
    for (int b = 0; b < cus; ++b)
      for (int w = 0; w < WAVES; ++w) {
        const long long c = h[(size_t)b * WAVES + w];
        if (c == 0) continue;
        const int slot = w >> 2;
        const bool is_m = slot == 2 && mode >= 2;
        if (is_m) { msum += (double)c; ++mn; } else { vsum += (double)c; ++vn; }
      }
    const double vc = vn ? vsum / vn / ((double)iters * V_PER_IT) : 0.0;         // cycles per VALU instr, per wave
    const double mc = mn ? msum / mn / ((double)iters * M_PER_IT) : 0.0;
    const int vw = mode == 0 ? 3 : (mode == 3 ? 0 : 2);                          // VALU waves per SIMD
    printf("%-11s | %6.3f -> %6.3f | %7.3f -> %6.4f | %.3f\n", names[mode], vc, vc > 0 ? vw / vc : 0.0, mc, mc > 0 ? 1.0 / mc : 0.0, ms);
  }
  printf("# float32 FMA peak per SIMD: 0.5 wave-instructions per cycle (32 lanes x 2 flops: 157.3 TFLOP/s at 2.4 GHz);\n"
         "# v_mfma_f32_16x16x4_f32 = 2048 flops per product: 256 flops per cycle per SIMD at one product per 8 cycles\n");
  return 0;
}
