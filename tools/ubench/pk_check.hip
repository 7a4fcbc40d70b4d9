#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));
// p = e + w*o, w = (c, -s) forward:  p.x = e.x + o.x c + o.y s ; p.y = e.y + o.y c - o.x s
__device__ __forceinline__ f2 tw_add(f2 e, f2 o, f2 cs /* sgpr pair (c, s) */) {
  f2 t, p;
  // t = (o.y * s + e.x, o.x * (-s) + e.y):  src0 = o swapped, src1 = s broadcast (hi of cs) with neg_hi, src2 = e
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(t) : "v"(o), "s"(cs), "v"(e));
  // p = (o.x * c + t.x, o.y * c + t.y): src1 = c broadcast (lo of cs)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(p) : "v"(o), "s"(cs), "v"(t));
  return p;
}
__device__ __forceinline__ f2 two_minus(f2 e, f2 p, f2 two) {   // 2e - p
  f2 q;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(e), "s"(two), "v"(p));
  return q;
}
__global__ void k(const float* in, float* out, float c, float s) {
  const int i = threadIdx.x;
  f2 e = {in[4 * i], in[4 * i + 1]}, o = {in[4 * i + 2], in[4 * i + 3]};
  f2 cs = {c, s};
  f2 two = {2.0f, 2.0f};
  // make cs / two uniform (SGPR): readfirstlane
  f2 p = tw_add(e, o, cs);
  f2 q = two_minus(e, p, two);
  // scalar reference
  float px = fmaf(o.x, c, fmaf(o.y, s, e.x));
  float py = fmaf(o.y, c, fmaf(-o.x, s, e.y));
  float qx = fmaf(2.0f, e.x, -px), qy = fmaf(2.0f, e.y, -py);
  out[8 * i + 0] = p.x; out[8 * i + 1] = p.y; out[8 * i + 2] = q.x; out[8 * i + 3] = q.y;
  out[8 * i + 4] = px; out[8 * i + 5] = py; out[8 * i + 6] = qx; out[8 * i + 7] = qy;
}
int main() {
  float h[256 * 4], *d, *o, ho[256 * 8];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 10007) / 997.0f - 5.0f;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, 0.92387953f, 0.38268343f);
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i)
    for (int j = 0; j < 4; ++j)
      if (memcmp(&ho[8 * i + j], &ho[8 * i + 4 + j], 4)) { if (bad < 5) printf("mismatch lane %d comp %d: %g vs %g\n", i, j, ho[8*i+j], ho[8*i+4+j]); ++bad; }
  printf("pk_check: %d mismatches of 1024\n", bad);
  return 0;
}
