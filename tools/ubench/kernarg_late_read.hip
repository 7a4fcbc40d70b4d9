// Does a kernel's kernel-argument segment stay intact for the kernel's whole life while ANOTHER host thread launches and
// synchronises kernels on a second stream?  A long-running kernel re-reads its arguments through an opaque pointer to the
// kernel-argument segment (scalar loads, like late_args() in fastpath.hpp) and compares them with the by-value copies it got
// at its start.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/kl tools/ubench/kernarg_late_read.hip -lpthread && /tmp/kl
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <thread>
struct Args { unsigned long long a[24]; unsigned* bad; unsigned loops; };
__global__ void k_long(Args A) {
  for (unsigned it = 0; it < A.loops; ++it) {
    const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const Args* L = reinterpret_cast<const Args*>((const char*)kp);
    bool same = true;
    for (int i = 0; i < 24; ++i) same = same && L->a[i] == A.a[i];
    same = same && L->bad == A.bad && L->loops == A.loops;
    if (!same && threadIdx.x == 0) atomicAdd(A.bad, 1u);
    __builtin_amdgcn_s_sleep(100);
  }
}
__global__ void k_small(float* p, float v) { p[threadIdx.x] += v; }
int main() {
  unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  float* buf; hipMalloc(&buf, 1024);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  std::atomic<bool> stop{false};
  std::thread other([&] { float v = 0; while (!stop) { for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s2, buf, v += 1.f); hipStreamSynchronize(s2); } });
  for (int call = 0; call < 4000; ++call) {
    Args A; for (int i = 0; i < 24; ++i) A.a[i] = 0x1234567800000000ull + (unsigned long long)call * 100 + i;
    A.bad = bad; A.loops = 400;   // ~400 x (100 x 64 clocks) = ~1 ms per workgroup
    hipLaunchKernelGGL(k_long, dim3(512), dim3(64), 0, s1, A);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s1, buf + 128, 1.f);
    if (call % 8 == 7) hipStreamSynchronize(s1);
  }
  hipStreamSynchronize(s1); stop = true; other.join();
  unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("late kernel-argument reads that differed from the by-value copy: %u (4000 launches x 512 workgroups x 400 re-reads)\n", h);
  return 0;
}
