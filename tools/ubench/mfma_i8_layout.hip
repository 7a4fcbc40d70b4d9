// Checks the operand layout of v_mfma_i32_16x16x32_i8 assumed by onepass.hpp:
//   A[i][k]: lane i + 16*(k/8), byte k%8 ;  B[k][j]: lane j + 16*(k/8), byte k%8 ;  D[i][j]: lane j + 16*(i/4), reg i%4
// build: hipcc --offload-arch=gfx950 -O2 mfma_i8_layout.hip -o mfma_i8_layout ; prints "layout OK" or the mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const signed char* A, const signed char* B, int* D) {
  const int l = threadIdx.x;
  long a = 0, b = 0;
  for (int e = 0; e < 8; ++e) {
    a |= (long)(unsigned char)A[(l % 16) * 32 + 8 * (l / 16) + e] << (8 * e);
    b |= (long)(unsigned char)B[(8 * (l / 16) + e) * 16 + (l % 16)] << (8 * e);
  }
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = c[r];
}
int main() {
  signed char hA[16 * 32], hB[32 * 16];
  int hD[256], ref[256];
  srand(1);
  for (auto& x : hA) x = (signed char)(rand() % 37);
  for (auto& x : hB) x = (signed char)(rand() % 19 - 3);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      int s = 0;
      for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[kk * 16 + j];
      ref[i * 16 + j] = s;
    }
  signed char *dA, *dB; int* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) { if (bad < 5) printf("mismatch at %d: %d vs %d\n", i, hD[i], ref[i]); ++bad; }
  printf(bad ? "layout WRONG (%d mismatches)\n" : "layout OK\n", bad);
  return bad != 0;
}
