// Compare constants of the stationary gate's decision kernels (split out of fastpath.hpp: shared with mixed.hip, which must
// not pull in the non-template kernels of kernels.hpp / fused.hpp).
#pragma once
#include "geom.hpp"

namespace sg {

struct ThreshConsts {
  // all device pointers
  const double* T2;        // [F]  compare constant on the RAW power |X|^2 (see k_prep_thresh)
  const double* thresh;    // [F]  dB threshold (for the floor test)
  const double* pmax;      // [units][FS] per-(unit, band) max raw power, 0 where not computed
  const int* need_floor;   // [units] 1: this unit's -top_db floor may be live -> pmax is valid;
                           //         2: the unit holds a non-finite sample -> no cell of it passes (T2_NEVER)
  // need_tag != 0 (one-pass gate, in-kernel floor test): the words are TAGGED, (tag << 2) | flags, raised with atomicMax
  // by the gate's tiles (2 beats 1) -- a word with another tag is a leftover of an earlier call and reads as 0, so
  // nobody has to clear the array between calls
  unsigned need_tag = 0;
};
// The one-pass gate's floor test reads its bound on max|x| as OP_ALIM_BLOCKS bit patterns (one per 64-band block of the
// noise statistics' final kernel, which derives them without a cross-block reduction) and takes their minimum:
// alim[2 .. 2 + OP_ALIM_BLOCKS); alim[1] is the tag of the last call in which a unit reported.
constexpr int OP_ALIM_BLOCKS = 9;
__device__ __forceinline__ int need_of(const ThreshConsts& tc, int64_t u) {
  const unsigned w = (unsigned)tc.need_floor[u];
  if (tc.need_tag == 0u) return (int)w;
  return (w >> 2) == tc.need_tag ? (int)(w & 3u) : 0;
}

// Compare constant meaning "no cell passes".  A band with a NaN threshold (NaN in the noise clip: stationary.py:75-81
// give mean(NaN) = NaN, and `dB > NaN` is False) and every band of a unit with a non-finite sample (np.max over a
// band that holds a NaN is NaN: _amp_to_db makes the whole band NaN, stationary.py:96-106) gate everything.
constexpr double T2_NEVER = 1e300;
// LDS layout of the float32 compare constants of the decision stages: lane c's 32 entries start at c * 36 floats (a
// pitch of 32 puts the lanes of equal parity on the same banks: every read was an 8-way conflict -- 45 % of the LDS
// cycles of k_gate_onepass); 36 keeps 16-byte alignment and spreads the 16 lanes over all 64 banks.
constexpr int T2_PITCH = 36, T2_POS512 = 16 * T2_PITCH, T2_FLOATS = 592;
__host__ __device__ constexpr int t2_pos(int i) { return i >= 512 ? T2_POS512 : (i >> 5) * T2_PITCH + (i & 31); }
// float32 copy of a (4x) compare constant for the float32 decision kernels: -1 ("all pass") and T2_NEVER map to
// huge finite values of either sign -- P - T overflows when squared, so the ambiguity test fails by itself
__device__ __forceinline__ float t2_to_f32(double v, double scale) {
  return v < 0.0 ? -3.0e38f : (v > 1e37 ? 3.0e38f : (float)(scale * v));
}

// In-kernel floor test of a decision kernel (the protocol of the one-pass gate, onepass.hpp "floor test"), for the kernels
// whose tiles stage EVERY sample of a unit's window (k_decide_fast512 / 256 / 2048: their frames cover the window, padding
// included): instead of reading the recording once more before the gate (k_unit_absmax + k_prep_thresh: 18 us of a 150 us
// call on two minutes of audio) every tile compares the largest sample it staged with the bound a_lim under which no band's
// -top_db floor can be live (alim[2 .. 2 + nb): one bit pattern per 64-band block of the statistics' last kernel, minimum
// taken here).  A tile whose test fires reports its unit -- need_floor[u] = (tag << 2) | 1 (2: a non-finite sample) by
// atomicMax, the unit's band maxima cleared for the float64 pre-pass -- and stamps alim[1] / the host-mapped word; the
// call's follow-up launches (pre-pass, the decision kernel's REDO instantiation) return at once unless a unit reported.
struct FloorLazy {
  unsigned* alim;      // null: the flags in ThreshConsts::need_floor were computed a priori
  int nb;              // bounds in alim[2 .. 2 + nb)
  unsigned* live;      // host-mapped: "a unit of launch `epoch` reported"
  unsigned epoch;
};
// the bound, loaded early (one vector load per lane, reduced over the wavefront); 0 when the test is off
__device__ __forceinline__ unsigned floor_lazy_bound(const FloorLazy& L, int lane) {
  if (L.alim == nullptr) return 0xffffffffu;
  unsigned a = L.alim[2 + (lane % L.nb)];
  for (int off = 32; off > 0; off >>= 1) a = min(a, (unsigned)__shfl_xor((int)a, off));
  return a;
}
// mi: largest |sample| this thread staged, as a bit pattern (sign cleared: NaN / Inf order above every finite value)
__device__ __forceinline__ void floor_lazy_report(const ThreshConsts& tc, const FloorLazy& L, unsigned bound, unsigned mi,
                                                  int64_t u, int FS, int lane) {
  if (L.alim == nullptr) return;
  if (__any(mi >= bound)) {   // wave-uniform, rare
    const bool nonfinite = __any(mi >= 0x7f800000u);
    double* pm = const_cast<double*>(tc.pmax) + u * FS;
    for (int f = lane; f < FS; f += 64) pm[f] = 0.0;
    if (lane == 0) {
      atomicMax(reinterpret_cast<unsigned*>(const_cast<int*>(tc.need_floor)) + u, (tc.need_tag << 2) | (nonfinite ? 2u : 1u));
      L.alim[1] = tc.need_tag;
      *L.live = L.epoch;
    }
  }
}

// prop_decrease < 1 (stationary.py:108-114 applies it BEFORE the zero-padded smoothing): mask = (p K + (1 - p) E) / ktot with
// E[t][f] = tri_valid(nt, t, T) tri_valid(nf, f, F), the triangle's weight over the taps that fall inside the field
// (closed form of its tails; half-width w, index i of [0, n)).
__device__ __forceinline__ float tri_valid(int w, int64_t i, int64_t n) {
  const int64_t l = i < w ? w - i : 0, r = (n - 1 - i) < w ? w - (n - 1 - i) : 0;
  return (float)((int64_t)(w + 1) * (w + 1) - l * (l + 1) / 2 - r * (r + 1) / 2);
}

}  // namespace sg
