// Micro-benchmark 2: issue rate of f32 VALU forms on gfx950, measured with the shader clock (s_memtime) per wave
// and with HIP events for the whole grid, at 1 / 2 / 4 / 8 waves per SIMD.  Every timed loop body is ONE asm
// statement (.rept) on compiler-allocated registers: no moves, no re-packing by the compiler between instructions.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate2.hip -o gpurun_out/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum { M_FMA = 0, M_FMAC, M_MUL, M_ADD, M_PKFMA, M_PKFMA_SEL, M_PKMUL, M_PKADD, M_FMA64, M_PKFMA16, M_FMA_DEP1, M_PKFMA_DEP1,
       M_MIX, N_MODES };
static const char* names[N_MODES] = {"v_fma_f32 x16 chains", "v_fmac_f32 x16", "v_mul_f32 x16", "v_add_f32 x16",
                                     "v_pk_fma_f32 x8 pairs", "v_pk_fma_f32 op_sel/neg x8", "v_pk_mul_f32 x8", "v_pk_add_f32 x8",
                                     "v_fma_f64 x8", "v_pk_fma_f32 x16 pairs", "v_fma_f32 1 chain (latency)",
                                     "v_pk_fma_f32 1 chain (latency)", "v_pk_fma_f32 + v_fma_f32 alternating"};
// wave-instructions per .rept body, flops per lane per instruction
static const int instr_per_body[N_MODES] = {16, 16, 16, 16, 8, 8, 8, 8, 8, 16, 16, 16, 16};
static const int flops_per_instr[N_MODES] = {2, 2, 1, 1, 4, 4, 2, 2, 2, 4, 2, 4, 3};

constexpr int REPT = 32;   // bodies per loop iteration

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float a, float b) {
  float r[16];
  f2 p[16];
  double d[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x * 0.001f + i; p[i] = {r[i], r[i] + 0.5f}; }
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = r[i];
  f2 av = {a, a * 0.999f}, bv = {b, b * 1.01f};
  double ad = a, bd = b;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == M_FMA) {
      asm volatile(".rept 32\n"
                   "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                   "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                   "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                   "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                   ".endr"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                     "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                   : "v"(a), "v"(b));
    } else if constexpr (MODE == M_FMAC) {
      asm volatile(".rept 32\n"
                   "v_fmac_f32 %0, %16, %17\n v_fmac_f32 %1, %16, %17\n v_fmac_f32 %2, %16, %17\n v_fmac_f32 %3, %16, %17\n"
                   "v_fmac_f32 %4, %16, %17\n v_fmac_f32 %5, %16, %17\n v_fmac_f32 %6, %16, %17\n v_fmac_f32 %7, %16, %17\n"
                   "v_fmac_f32 %8, %16, %17\n v_fmac_f32 %9, %16, %17\n v_fmac_f32 %10, %16, %17\n v_fmac_f32 %11, %16, %17\n"
                   "v_fmac_f32 %12, %16, %17\n v_fmac_f32 %13, %16, %17\n v_fmac_f32 %14, %16, %17\n v_fmac_f32 %15, %16, %17\n"
                   ".endr"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                     "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                   : "v"(a), "v"(b));
    } else if constexpr (MODE == M_MUL || MODE == M_ADD) {
#define BODY2(OP)                                                                                                              \
      asm volatile(".rept 32\n" OP " %0, %0, %16\n " OP " %1, %1, %16\n " OP " %2, %2, %16\n " OP " %3, %3, %16\n " OP         \
                   " %4, %4, %16\n " OP " %5, %5, %16\n " OP " %6, %6, %16\n " OP " %7, %7, %16\n " OP " %8, %8, %16\n " OP    \
                   " %9, %9, %16\n " OP " %10, %10, %16\n " OP " %11, %11, %16\n " OP " %12, %12, %16\n " OP                   \
                   " %13, %13, %16\n " OP " %14, %14, %16\n " OP " %15, %15, %16\n .endr"                                      \
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),           \
                     "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])      \
                   : "v"(a))
      if constexpr (MODE == M_MUL) BODY2("v_mul_f32"); else BODY2("v_add_f32");
    } else if constexpr (MODE == M_PKFMA) {
      asm volatile(".rept 32\n"
                   "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   ".endr"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                   : "v"(av), "v"(bv));
    } else if constexpr (MODE == M_PKFMA_SEL) {
      // the complex-multiply forms: broadcast one half of a source, swap halves, negate one half
      asm volatile(".rept 32\n"
                   "v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n"
                   "v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n"
                   "v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n"
                   "v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n"
                   ".endr"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                   : "v"(av), "v"(bv));
    } else if constexpr (MODE == M_PKMUL || MODE == M_PKADD) {
#define BODY3(OP)                                                                                                              \
      asm volatile(".rept 32\n" OP " %0, %0, %8\n " OP " %1, %1, %8\n " OP " %2, %2, %8\n " OP " %3, %3, %8\n " OP             \
                   " %4, %4, %8\n " OP " %5, %5, %8\n " OP " %6, %6, %8\n " OP " %7, %7, %8\n .endr"                           \
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])            \
                   : "v"(av))
      if constexpr (MODE == M_PKMUL) BODY3("v_pk_mul_f32"); else BODY3("v_pk_add_f32");
    } else if constexpr (MODE == M_FMA64) {
      asm volatile(".rept 32\n"
                   "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                   "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                   ".endr"
                   : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                   : "v"(ad), "v"(bd));
    } else if constexpr (MODE == M_PKFMA16) {
      asm volatile(".rept 32\n"
                   "v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
                   "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
                   "v_pk_fma_f32 %8, %8, %16, %17\n v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n"
                   "v_pk_fma_f32 %12, %12, %16, %17\n v_pk_fma_f32 %13, %13, %16, %17\n v_pk_fma_f32 %14, %14, %16, %17\n v_pk_fma_f32 %15, %15, %16, %17\n"
                   ".endr"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]),
                     "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15])
                   : "v"(av), "v"(bv));
    } else if constexpr (MODE == M_FMA_DEP1) {
      asm volatile(".rept 512\n v_fma_f32 %0, %0, %1, %2\n .endr" : "+v"(r[0]) : "v"(a), "v"(b));
    } else if constexpr (MODE == M_PKFMA_DEP1) {
      asm volatile(".rept 512\n v_pk_fma_f32 %0, %0, %1, %2\n .endr" : "+v"(p[0]) : "v"(av), "v"(bv));
    } else if constexpr (MODE == M_MIX) {
      asm volatile(".rept 32\n"
                   "v_pk_fma_f32 %0, %0, %16, %17\n v_fma_f32 %8, %8, %18, %19\n v_pk_fma_f32 %1, %1, %16, %17\n v_fma_f32 %9, %9, %18, %19\n"
                   "v_pk_fma_f32 %2, %2, %16, %17\n v_fma_f32 %10, %10, %18, %19\n v_pk_fma_f32 %3, %3, %16, %17\n v_fma_f32 %11, %11, %18, %19\n"
                   "v_pk_fma_f32 %4, %4, %16, %17\n v_fma_f32 %12, %12, %18, %19\n v_pk_fma_f32 %5, %5, %16, %17\n v_fma_f32 %13, %13, %18, %19\n"
                   "v_pk_fma_f32 %6, %6, %16, %17\n v_fma_f32 %14, %14, %18, %19\n v_pk_fma_f32 %7, %7, %16, %17\n v_fma_f32 %15, %15, %18, %19\n"
                   ".endr"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(r[8]),
                     "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                   : "v"(av), "v"(bv), "v"(a), "v"(b));
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i] + p[i].x + p[i].y;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
int run(int wg_per_cu) {
  const int blocks = 256 * wg_per_cu, iters = 200;
  float* out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  long long* cyc; CHK(hipMalloc(&cyc, (size_t)blocks * 4 * 8));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 5, 1.0001f, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
  CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  std::vector<long long> hc((size_t)blocks * 4);
  CHK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
  std::sort(hc.begin(), hc.end());
  const double per_wave = (double)iters * 32 * instr_per_body[MODE] * ((MODE == M_FMA_DEP1 || MODE == M_PKFMA_DEP1) ? 1.0 : 1.0);
  const double med = (double)hc[hc.size() / 2];
  const double instr = (double)blocks * 4 * per_wave;
  const double flops = instr * 64 * flops_per_instr[MODE];
  // s_memtime ticks at a constant 100 MHz on gfx9 (not the shader clock): report wave time in ns and derive the
  // issue interval from the event-timed wall clock instead
  printf("%-40s wg/CU %d  %8.3f ms  %7.1f TFLOP/s  %6.2f ns per wave-instr per SIMD (wall)  median wave ticks %.0f\n",
         names[MODE], wg_per_cu, ms, flops / ms / 1e9, ms * 1e6 * 1024 / instr, med);
  CHK(hipFree(out)); CHK(hipFree(cyc));
  return 0;
}

template <int MODE>
int sweep() {
  for (int w : {1, 2, 4, 8})
    if (run<MODE>(w)) return 1;
  return 0;
}

int main() {
  hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
  printf("device %s  CUs %d  clock %d kHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
  sweep<M_FMA>(); sweep<M_FMAC>(); sweep<M_MUL>(); sweep<M_ADD>(); sweep<M_PKFMA>(); sweep<M_PKFMA16>(); sweep<M_PKFMA_SEL>();
  sweep<M_PKMUL>(); sweep<M_PKADD>(); sweep<M_MIX>(); sweep<M_FMA64>();
  run<M_FMA_DEP1>(1); run<M_PKFMA_DEP1>(1);
  return 0;
}
