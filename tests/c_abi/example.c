/* Plain-C use of libmi355gate.so: no torch, no C++ -- just the HIP runtime for device memory.
 * Stationary reduce_noise of a mono float32 recording that already lives in HBM.
 *
 *   hipcc -x c tests/c_abi/example.c -Iinclude -Lnoisereduce_amd -lmi355gate -o example   (needs a GPU to run)
 *
 * tests/test_host_cpu.py compiles this file (syntax + link against the built library) on the
 * CPU-only build host; tests/test_gpu_parity.py runs it on the GPU and checks it against the
 * Python path. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mi355gate.h"

/* the three HIP runtime calls used here, declared by hand so that plain gcc can compile the file */
typedef int hipError_t;
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, int kind); /* 1 = H2D, 2 = D2H */
hipError_t hipDeviceSynchronize(void);

static int fail(sg_handle* h, const char* what, int rc) {
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, sg_last_error(h));
  return 1;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 480000;
  float* y = (float*)malloc((size_t)n * sizeof(float));
  uint32_t s = 12345u;
  for (int64_t i = 0; i < n; ++i) { /* deterministic pseudo-noise + a tone-like ramp */
    s = s * 1664525u + 1013904223u;
    y[i] = 0.2f * ((float)(s >> 8) / 8388608.0f - 1.0f) + 0.3f * (float)((i % 48) - 24) / 24.0f;
  }
  sg_params p = {0};
  p.variant = SG_VARIANT_S; p.stationary = 1;
  p.n_fft = 1024; p.win_length = 1024; p.hop_length = 256;
  p.n_grad_freq = 5; p.n_grad_time = 9; p.smooth_mask = 1;   /* 500 Hz / 50 ms at 48 kHz */
  p.chunk_size = 600000; p.padding = 30000;
  p.prop_decrease = 1.0; p.n_std_thresh = 1.5; p.top_db = 80.0; p.ddof = 0;
  sg_handle* h = NULL;
  int rc = sg_create(&p, NULL, &h);
  if (rc) return fail(NULL, "sg_create", rc);
  void *d_in = NULL, *d_out = NULL;
  if (hipMalloc(&d_in, (size_t)n * 4) || hipMalloc(&d_out, (size_t)n * 4)) return fail(h, "hipMalloc", -1);
  hipMemcpy(d_in, y, (size_t)n * 4, 1);
  const int64_t n_clip = n < p.chunk_size ? n : p.chunk_size;  /* stationary.py:61-64 */
  if ((rc = sg_noise_stats(h, d_in, SG_F32, 1, n_clip, n, NULL))) return fail(h, "sg_noise_stats", rc);
  if ((rc = sg_process_chunks(h, d_in, SG_F32, d_out, SG_F32, 1, n, n, n, 0, n, n > p.chunk_size, 0, 0, NULL)))
    return fail(h, "sg_process_chunks", rc);
  hipDeviceSynchronize();
  float* out = (float*)malloc((size_t)n * sizeof(float));
  hipMemcpy(out, d_out, (size_t)n * 4, 2);
  double e_in = 0.0, e_out = 0.0;
  for (int64_t i = 0; i < n; ++i) { e_in += (double)y[i] * y[i]; e_out += (double)out[i] * out[i]; }
  printf("samples %lld  energy in %.6e  out %.6e  first %.9g %.9g %.9g\n", (long long)n, e_in, e_out,
         out[1000], out[n / 2], out[n - 1000]);
  hipFree(d_in); hipFree(d_out);
  sg_destroy(h);
  free(y); free(out);
  return 0;
}
