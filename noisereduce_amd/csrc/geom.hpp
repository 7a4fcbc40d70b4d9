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

}  // namespace sg
