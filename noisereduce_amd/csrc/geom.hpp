// The two structs every kernel of the path takes by value (split out of kernels.hpp so that translation units which
// only hold kernel templates -- nonstat_mask.hip -- need not pull in the non-template kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg {

// How a unit's samples map onto the caller's planar (rows, N) buffer.  Sample s of unit
// u = (row r, chunk i) is x[r*stride + i*cs - pad + s] when that index is inside [0, N),
// else 0 (SpectralGate._read_chunk, base.py:130-142).
struct View {
  const void* x;
  int dtype;        // SG_F32 ...
  int64_t stride;   // elements between rows
  int64_t N;        // samples per row
  int64_t lo, hi;   // readable index range [lo, hi) of a row (0, N unless the caller holds halos)
  int64_t cs;       // chunk step (0 when n_chunks == 1)
  int64_t pad;      // zero/neighbour padding before the chunk start
  int64_t Lp;       // samples per unit window
  int32_t n_chunks; // units per row
  int64_t c0;       // index of the row's first chunk (sub-range filtering: chunks c0 .. c0 + n_chunks - 1)
  int64_t unit0;    // global index of this batch's first unit (unit = row*n_chunks + chunk)
};

struct Geom {
  int32_t n, W, H, F, FS, padL;  // padL: zero extension before sample 0 (W/2 scipy, n/2 torch)
  int64_t T;                     // frames per unit
  int64_t Lout;                  // valid ISTFT samples per unit
};


// ---- per-sample access to a unit window, shared by every kernel file ----
__device__ __forceinline__ double load_sample(const void* p, int dtype, int64_t idx) {
  switch (dtype) {
    case 0: return (double)((const float*)p)[idx];
    case 1: return ((const double*)p)[idx];
    case 2: return (double)((const int16_t*)p)[idx];
    default: return (double)((const int32_t*)p)[idx];
  }
}

__device__ __forceinline__ double view_sample(const View& v, int64_t row, int64_t chunk, int64_t s) {
  if (s < 0 || s >= v.Lp) return 0.0;
  int64_t g = chunk * v.cs - v.pad + s;
  if (g < v.lo || g >= v.hi) return 0.0;
  return load_sample(v.x, v.dtype, row * v.stride + g);
}

// Maximum that KEEPS a NaN (numpy / torch maxima do; fmax drops it): a band that holds a NaN has a NaN maximum, and
// the reference's `max(dB, rowmax - top_db) > thresh` is then False for the whole band.  The canonical positive NaN also
// wins the bit-pattern atomicMax of the per-band maxima.
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? (double)NAN : fmax(a, b); }

// Start of the `len` samples [s0, s0 + len) of a unit window when they are all readable float32
// samples (no zero padding, no conversion), else nullptr: frames take the direct-load path in the
// interior and the checked per-sample path (view_sample) at the edges / for other dtypes.
__device__ __forceinline__ const float* frame_ptr_f32(const View& v, int64_t row, int64_t chunk, int64_t s0,
                                                      int64_t len) {
  if (v.dtype != 0 || s0 < 0 || s0 + len > v.Lp) return nullptr;
  const int64_t g = chunk * v.cs - v.pad + s0;
  if (g < v.lo || g + len > v.hi) return nullptr;
  return (const float*)v.x + row * v.stride + g;
}

__device__ __forceinline__ void store_sample(void* p, int dtype, int64_t idx, float val) {
  switch (dtype) {
    case 0: ((float*)p)[idx] = val; break;
    case 1: ((double*)p)[idx] = (double)val; break;
    case 2: ((int16_t*)p)[idx] = (int16_t)val; break;  // truncation, like ndarray.astype
    default: ((int32_t*)p)[idx] = (int32_t)val; break;
  }
}

// dB of one cell exactly as the reference writes it: 20*log10(|Z| + eps)
// (spectralgate/utils.py:15; torchgate/utils.py:22), |Z| = sqrt(P) * mag_scale.
__device__ __forceinline__ double cell_db(double P, double mag_scale) {
  return 20.0 * log10(sqrt(P) * mag_scale + 2.220446049250313e-16);
}

}  // namespace sg
