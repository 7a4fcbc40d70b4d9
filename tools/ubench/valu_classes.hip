// Micro-benchmark 3 (round 5): what one wave64 VALU instruction of each CLASS the one-pass gate uses costs the vector pipe,
// and what rocprofv3's SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES read for it -- the calibration VERDICT r4
// item 3(a) asks for ("0.37 of the issue slots, or 0.74 if every instruction held the pipe a quad-cycle").
// Every timed body is ONE asm statement (.rept) on 16 independent registers; each mode is its own kernel (= its own
// dispatch in the counter CSV).  Grid = CUs x 4 workgroups of 256 threads (4 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_classes.hip -o tools/ubench/valu_classes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define R16(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                      OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n" \
                      OP " %8, %8" TAIL "\n" OP " %9, %9" TAIL "\n" OP " %10, %10" TAIL "\n" OP " %11, %11" TAIL "\n" \
                      OP " %12, %12" TAIL "\n" OP " %13, %13" TAIL "\n" OP " %14, %14" TAIL "\n" OP " %15, %15" TAIL "\n"
#define REGS32 "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), \
               "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
#define REGS64 "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), \
               "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15])

enum { M_FMA32, M_ADD32, M_CNDMASK, M_MOV_DPP, M_ADD_DPP, M_LOG, M_EXP, M_RCP, M_SQRT, M_FMA64, M_ADD64, M_MUL64, M_CVT_F64_F32,
       M_LSHL64, M_AND_OR, M_PERM, M_MAD_U32_U24, M_ADD3, M_CNDMASK_SGPR, M_CNDMASK_VCC_DST, M_MOV, N_MODES };
static const char* names[N_MODES] = {"v_fma_f32", "v_add_f32", "v_cndmask_b32 (vcc)", "v_mov_b32_dpp row_shr:1", "v_add_f32_dpp row_shr:1",
                                     "v_log_f32", "v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_fma_f64", "v_add_f64", "v_mul_f64",
                                     "v_cvt_f64_f32", "v_lshlrev_b64", "v_and_or_b32", "v_perm_b32", "v_mad_u32_u24", "v_add3_u32", "v_cndmask_b32_e64 (sgpr pair)", "v_cndmask_b32 (vcc, other sources)", "v_mov_b32"};
constexpr int REPT = 32, PER_BODY = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float r[16];
  double d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { r[i] = 1.0f + threadIdx.x * 0.001f + i; d[i] = r[i]; }
  const double ad = a;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == M_FMA32) asm volatile(".rept 32\n" R16("v_fma_f32", ", %16, %17") ".endr" : REGS32 : "v"(a), "v"(b));
    else if constexpr (MODE == M_ADD32) asm volatile(".rept 32\n" R16("v_add_f32", ", %16") ".endr" : REGS32 : "v"(a));
    else if constexpr (MODE == M_CNDMASK) asm volatile("v_cmp_gt_f32 vcc, %16, %17\n.rept 32\n" R16("v_cndmask_b32", ", %16, vcc") ".endr" : REGS32 : "v"(a), "v"(b) : "vcc");
    else if constexpr (MODE == M_MOV_DPP) asm volatile(".rept 32\n" R16("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf") ".endr" : REGS32);
    else if constexpr (MODE == M_ADD_DPP) asm volatile(".rept 32\n" R16("v_add_f32_dpp", ", %16 row_shr:1 row_mask:0xf bank_mask:0xf") ".endr" : REGS32 : "v"(a));
    else if constexpr (MODE == M_LOG) asm volatile(".rept 32\n" R16("v_log_f32", "") ".endr" : REGS32);
    else if constexpr (MODE == M_EXP) asm volatile(".rept 32\n" R16("v_exp_f32", "") ".endr" : REGS32);
    else if constexpr (MODE == M_RCP) asm volatile(".rept 32\n" R16("v_rcp_f32", "") ".endr" : REGS32);
    else if constexpr (MODE == M_SQRT) asm volatile(".rept 32\n" R16("v_sqrt_f32", "") ".endr" : REGS32);
    else if constexpr (MODE == M_FMA64) asm volatile(".rept 32\n" R16("v_fma_f64", ", %16, %16") ".endr" : REGS64 : "v"(ad));
    else if constexpr (MODE == M_ADD64) asm volatile(".rept 32\n" R16("v_add_f64", ", %16") ".endr" : REGS64 : "v"(ad));
    else if constexpr (MODE == M_MUL64) asm volatile(".rept 32\n" R16("v_mul_f64", ", %16") ".endr" : REGS64 : "v"(ad));
    else if constexpr (MODE == M_CVT_F64_F32) {
      asm volatile(".rept 32\n"
                   "v_cvt_f64_f32 %0, %16\n v_cvt_f64_f32 %1, %17\n v_cvt_f64_f32 %2, %18\n v_cvt_f64_f32 %3, %19\n"
                   "v_cvt_f64_f32 %4, %16\n v_cvt_f64_f32 %5, %17\n v_cvt_f64_f32 %6, %18\n v_cvt_f64_f32 %7, %19\n"
                   "v_cvt_f64_f32 %8, %16\n v_cvt_f64_f32 %9, %17\n v_cvt_f64_f32 %10, %18\n v_cvt_f64_f32 %11, %19\n"
                   "v_cvt_f64_f32 %12, %16\n v_cvt_f64_f32 %13, %17\n v_cvt_f64_f32 %14, %18\n v_cvt_f64_f32 %15, %19\n"
                   ".endr" : REGS64 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
    } else if constexpr (MODE == M_LSHL64) {
      asm volatile(".rept 32\n"
                   "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                   "v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %6, 1, %6\n v_lshlrev_b64 %7, 1, %7\n"
                   "v_lshlrev_b64 %8, 1, %8\n v_lshlrev_b64 %9, 1, %9\n v_lshlrev_b64 %10, 1, %10\n v_lshlrev_b64 %11, 1, %11\n"
                   "v_lshlrev_b64 %12, 1, %12\n v_lshlrev_b64 %13, 1, %13\n v_lshlrev_b64 %14, 1, %14\n v_lshlrev_b64 %15, 1, %15\n"
                   ".endr" : REGS64);
    } else if constexpr (MODE == M_AND_OR) asm volatile(".rept 32\n" R16("v_and_or_b32", ", %16, %17") ".endr" : REGS32 : "v"(a), "v"(b));
    else if constexpr (MODE == M_PERM) asm volatile(".rept 32\n" R16("v_perm_b32", ", %16, %17") ".endr" : REGS32 : "v"(a), "v"(b));
    else if constexpr (MODE == M_MAD_U32_U24) asm volatile(".rept 32\n" R16("v_mad_u32_u24", ", %16, %17") ".endr" : REGS32 : "v"(a), "v"(b));
    else if constexpr (MODE == M_CNDMASK_SGPR) {
      unsigned long long msk;
      asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(msk) : "v"(a), "v"(b));
      asm volatile(".rept 32\n" R16("v_cndmask_b32_e64", ", %16, %17") ".endr" : REGS32 : "v"(a), "s"(msk));
    } else if constexpr (MODE == M_CNDMASK_VCC_DST) {
      asm volatile("v_cmp_gt_f32 vcc, %16, %17\n.rept 32\n"
                   "v_cndmask_b32 %0, %16, %17, vcc\n v_cndmask_b32 %1, %16, %17, vcc\n v_cndmask_b32 %2, %16, %17, vcc\n v_cndmask_b32 %3, %16, %17, vcc\n"
                   "v_cndmask_b32 %4, %16, %17, vcc\n v_cndmask_b32 %5, %16, %17, vcc\n v_cndmask_b32 %6, %16, %17, vcc\n v_cndmask_b32 %7, %16, %17, vcc\n"
                   "v_cndmask_b32 %8, %16, %17, vcc\n v_cndmask_b32 %9, %16, %17, vcc\n v_cndmask_b32 %10, %16, %17, vcc\n v_cndmask_b32 %11, %16, %17, vcc\n"
                   "v_cndmask_b32 %12, %16, %17, vcc\n v_cndmask_b32 %13, %16, %17, vcc\n v_cndmask_b32 %14, %16, %17, vcc\n v_cndmask_b32 %15, %16, %17, vcc\n"
                   ".endr" : REGS32 : "v"(a), "v"(b) : "vcc");
    } else if constexpr (MODE == M_MOV) {
      asm volatile(".rept 32\n"
                   "v_mov_b32 %0, %16\n v_mov_b32 %1, %16\n v_mov_b32 %2, %16\n v_mov_b32 %3, %16\n v_mov_b32 %4, %16\n v_mov_b32 %5, %16\n v_mov_b32 %6, %16\n v_mov_b32 %7, %16\n"
                   "v_mov_b32 %8, %16\n v_mov_b32 %9, %16\n v_mov_b32 %10, %16\n v_mov_b32 %11, %16\n v_mov_b32 %12, %16\n v_mov_b32 %13, %16\n v_mov_b32 %14, %16\n v_mov_b32 %15, %16\n"
                   ".endr" : REGS32 : "v"(a));
    }
    else if constexpr (MODE == M_ADD3) asm volatile(".rept 32\n" R16("v_add3_u32", ", %16, %17") ".endr" : REGS32 : "v"(a), "v"(b));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i] + (float)d[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
int run(float* out, int cus, int clock_khz) {
  const int iters = 64;
  const int wgs = cus * 4;     // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 4, 1.0001f, 0.5f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_simd = (double)iters * REPT * PER_BODY * 4;   // 4 waves per SIMD
  const double ns = ms * 1e6 / instr_per_simd;
  printf("%-28s %8.3f ms  %6.3f ns per wave-instruction per SIMD = %5.2f cycles at %.2f GHz   (%.0f instructions per SIMD)\n", names[MODE], ms, ns,
         ns * clock_khz * 1e-6, clock_khz * 1e-6, instr_per_simd);
  if constexpr (MODE + 1 < N_MODES) return run<MODE + 1>(out, cus, clock_khz);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  float* out;
  CHK(hipMalloc(&out, 4096));
  // warm the clocks
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<M_FMA32>, dim3(p.multiProcessorCount * 4), dim3(256), 0, 0, out, 64, 1.0001f, 0.5f);
  CHK(hipDeviceSynchronize());
  return run<0>(out, p.multiProcessorCount, p.clockRate);
}
