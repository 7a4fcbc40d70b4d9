// Micro-benchmark: what does the work ticket of k_gate_onepass cost?  Every workgroup of that kernel takes its tile with one
// returning atomicAdd on ONE device word (dispatch-order independence of the tile hand-offs).  A launch of 7152 workgroups (configs[1])
// whose only work is that atomic took 83 us (round 4, the "second launch" of the in-kernel floor test) -- about 11 ns per
// same-address atomic, serialised.  This program times a grid of G workgroups of 256 threads, 50 KB of dynamic LDS each
// (three per CU, as the gate), that
//   mode 0  exit at once
//   mode 1  one returning atomicAdd on one word (thread 0), result broadcast through LDS, exit
//   mode 2  as 1, on one word per XCD (blockIdx % 8)
//   mode 3  as 1, without a return value (fire and forget)
//   mode 4  one plain load of that word, exit
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/ticket_atomic.hip -o gpurun_out/ticket_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(unsigned* ctr, unsigned* out, int mode) {
  extern __shared__ unsigned sm[];
  if (mode == 0) return;
  if (threadIdx.x == 0) {
    if (mode == 1) sm[0] = atomicAdd(ctr, 1u);
    else if (mode == 2) sm[0] = atomicAdd(ctr + 64 * (blockIdx.x & 7), 1u);
    else if (mode == 3) { __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sm[0] = 1u; }
    else sm[0] = *(volatile unsigned*)ctr;
  }
  __syncthreads();
  if (sm[0] == 0xffffffffu) out[blockIdx.x] = 1u;   // keep the value alive
}

int main() {
  unsigned *ctr, *out;
  CHK(hipMalloc(&ctr, 4096));
  CHK(hipMalloc(&out, 1 << 20));
  CHK(hipMemset(ctr, 0, 4096));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 51200));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const char* names[5] = {"exit", "atomicAdd (returning), one word", "atomicAdd (returning), word per XCD", "atomic add, no return", "plain load"};
  printf("# grid | mode | us per launch (median of 20, HIP events around each launch)\n");
  for (int G : {256, 768, 2064, 8256}) {
    for (int mode = 0; mode < 5; ++mode) {
      float ts[20];
      for (int r = 0; r < 23; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(G), dim3(256), 51200, 0, ctr, out, mode);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) ts[r - 3] = ms * 1000.f;
      }
      for (int i = 0; i < 20; ++i) for (int j = i + 1; j < 20; ++j) if (ts[j] < ts[i]) { float t = ts[i]; ts[i] = ts[j]; ts[j] = t; }
      printf("%5d | %-36s | %8.2f\n", G, names[mode], ts[10]);
    }
  }
  return 0;
}
