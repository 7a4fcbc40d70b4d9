// Non-stationary mask of variant S (nonstationary.py:59-87) in two passes over the magnitude field:
//
//   S = filtfilt(one-pole)(|X|) along time   (get_time_smoothed_representation, nonstationary.py:106-115;
//                                             scipy filtfilt(padtype=None): forward seeded with A[0], backward
//                                             seeded with the forward pass's last value)
//   raw = sigmoid(((|X| - S) / S - thresh) * slope),   M = p * smooth(raw) + (1 - p)
//
// k_iir_sigmoid_seg + k_smooth_tiled moved the field 8 times (three sweeps of the recurrence over |X| and the
// raw mask, then the smoothing pass).  The recurrence is linear, so a time tile's contribution is fixed by two
// numbers per band -- its zero-state forward end value and the backward zero-state sum of its forward values --
// and the state entering any tile follows from a short chain over tiles:
//
//   k_iir_part   (reads |X| once)      per (tile, band): e_f = sum_t b c^(end-1-t) A[t],
//                                      E0 = sum_t b c^(t-start) s0[t]   (s0 = zero-state forward response)
//   k_iir_chain  (tiny)                forward state before / backward state after every tile
//   k_iir_mask   (reads |X| once more, writes M) per (time tile + nt halo rows, bin block incl. nf halo columns):
//                                      a column of the tile in registers; forward sweep; backward sweep that REGENERATES
//                                      the forward values in reverse (s_f[t-1] = (s_f[t] - b A[t]) / c: error growth
//                                      c^-rows, 1.25 at the default 2 s time constant) and writes the sigmoid in place;
//                                      separable triangle smoothing; p * . + (1 - p); store.
//
// The chain only knows the states at TILE boundaries; the states at the outer edges of a tile's halo rows follow
// from them inside k_iir_mask by the same inversion over the nt halo rows it has loaded anyway (forward state:
// s_f[t-1] = (s_f[t] - b A[t]) / c down the leading halo; backward state: S[t+1] = (S[t] - b s_f[t]) / c up the
// trailing halo, beside the forward sweep).  Rounds 2-3 split every tile at ts + nt and te - nt instead: three
// (e_f, E0) pairs per tile and band, 49 MB of partials and as many carries per 10-minute call, and a chain three
// times as long.  All recurrences in float64 (the reference's precision).
#pragma once
#include "kernels.hpp"

namespace sg {

constexpr int NS_TT = 64;      // frames per time tile
constexpr int NS_MAX_NF = 24;  // k_iir_mask: a wave holds 64 - 2 nf output bins (the DPP boxcars cost 2 nf adds per value)

struct NsTiling {
  int64_t T;
  int nt;
  __host__ __device__ int64_t n_tiles() const { return (T + NS_TT - 1) / NS_TT; }
};

// partials [unit][tile][2][FS]
__global__ __launch_bounds__(256) void k_iir_part(const float* __restrict__ A, Geom g, NsTiling tl, double b,
                                                  double* __restrict__ part) {
  const int l = threadIdx.x & 63;
  const int f = blockIdx.x * 64 + l;
  const int64_t k = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  const int64_t u = blockIdx.z;
  if (f >= g.F || k >= tl.n_tiles()) return;
  const double c = 1.0 - b;
  const float* a = A + u * g.T * g.FS + f;
  const int64_t ts = k * NS_TT, te = ts + NS_TT < g.T ? ts + NS_TT : g.T;
  double e = 0.0, E0 = 0.0, pw = b;
  for (int64_t t = ts; t < te; t += 16) {   // 16 rows in flight (the recurrence is serial, its operands are not)
    float av[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) av[q] = a[(t + q < te ? t + q : te - 1) * g.FS];
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (t + q < te) {
        e = b * (double)av[q] + c * e;
        E0 += pw * e;
        pw *= c;
      }
  }
  double* o = part + ((u * tl.n_tiles() + k) * 2) * (int64_t)g.FS + f;
  o[0] = e;
  o[g.FS] = E0;
}

// carries [unit][tile][2][FS]: [0] forward state before the tile's first frame (s_f[ts - 1]; s_f[-1] := A[0], the
// lfilter_zi steady state), [1] backward state at its end (S[te]; S[T] := s_f[T - 1], the seed of the backward pass)
__global__ __launch_bounds__(64) void k_iir_chain(const float* __restrict__ A, const double* __restrict__ part,
                                                  Geom g, NsTiling tl, double b, double* __restrict__ carry,
                                                  int64_t n_units) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* pw1 = reinterpret_cast<double*>(smem);  // [nk] c^len
  const int64_t nk = tl.n_tiles();
  const int nj = (int)nk;
  double* pw2 = pw1 + nj;                          // [nk] 1 - c^(2 len)
  const double c = 1.0 - b;
  for (int j = threadIdx.x; j < nj; j += 64) {
    const int64_t t0 = (int64_t)j * NS_TT, t1 = t0 + NS_TT < g.T ? t0 + NS_TT : g.T;
    const double len = (double)(t1 - t0);
    pw1[j] = pow(c, len);
    pw2[j] = 1.0 - pow(c, 2.0 * len);
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_units * g.FS) return;
  const int64_t u = i / g.FS;
  const int f = (int)(i % g.FS);
  if (f >= g.F) return;
  const double* pb = part + (u * nk * 2) * (int64_t)g.FS + f;
  double* cb = carry + (u * nk * 2) * (int64_t)g.FS + f;
  double s = (double)A[u * g.T * g.FS + f];  // s[-1] = A[0]  (lfilter_zi steady state)
  // the chain is serial, its operands are not: the partials of the NEXT 16 tiles are in flight while this batch's
  // recurrence runs (double-buffered by hand; a batch is one dependent round trip to memory otherwise)
  const int64_t st2 = 2 * (int64_t)g.FS;
  constexpr int CB = 16;
  {
    double e[2][CB];
    auto fetch = [&](int j0, double* d) {
#pragma unroll
      for (int q = 0; q < CB; ++q) d[q] = j0 + q < nj ? pb[(j0 + q) * st2] : 0.0;
    };
    auto run = [&](int j0, const double* d) {
#pragma unroll
      for (int q = 0; q < CB; ++q)
        if (j0 + q < nj) {
          cb[(j0 + q) * st2] = s;
          s = d[q] + pw1[j0 + q] * s;
        }
    };
    fetch(0, e[0]);
    for (int j0 = 0; j0 < nj; j0 += 2 * CB) {
      fetch(j0 + CB, e[1]);
      run(j0, e[0]);
      fetch(j0 + 2 * CB, e[0]);
      run(j0 + CB, e[1]);
    }
  }
  // backward: S[end_j] given; S[start_j] = E_b + c^len S[end_j],
  // E_b = sum_t b c^(t-start) s_f[t] = E0 + s_in * b c (1 - c^(2 len)) / (1 - c^2)   (s_f = s0 + c^(t-start+1) s_in)
  double S = s;  // seed: the forward pass's last value
  const double gq = b * c / (1.0 - c * c);
  {
    double e0[2][CB], sin[2][CB];
    auto fetch = [&](int j1, double* d0, double* d1) {
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const int j = j1 - q;
        d0[q] = j >= 0 ? pb[j * st2 + g.FS] : 0.0;
        d1[q] = j >= 0 ? cb[j * st2] : 0.0;
      }
    };
    auto run = [&](int j1, const double* d0, const double* d1) {
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const int j = j1 - q;
        if (j >= 0) {
          cb[j * st2 + g.FS] = S;
          S = (d0[q] + d1[q] * gq * pw2[j]) + pw1[j] * S;
        }
      }
    };
    fetch(nj - 1, e0[0], sin[0]);
    for (int j1 = nj - 1; j1 >= 0; j1 -= 2 * CB) {
      fetch(j1 - CB, e0[1], sin[1]);
      run(j1, e0[0], sin[0]);
      fetch(j1 - 2 * CB, e0[0], sin[0]);
      run(j1 - CB, e0[1], sin[1]);
    }
  }
}

// sigmoid_ratio (kernels.hpp) with v_rcp_f32 in place of the two IEEE divisions (10 instructions each; the mask is a
// float32 field, 1 ulp of the reciprocal is 1e-7 of it).  v_rcp_f32 flushes denormals: a smoothed magnitude below
// 1e-30 (or NaN) takes the IEEE form -- digital silence included: 0 / 0 = NaN as in nonstationary.py:75.
__device__ __forceinline__ float sigmoid_ratio_rcp(double av, double s, float nthresh, float slope) {
  const float num = (float)(av - s), den = (float)s;
  float ratio = num * __builtin_amdgcn_rcpf(den);
  if (__builtin_expect(!(den >= 1e-30f), 0)) ratio = num / den;
  return __builtin_amdgcn_rcpf(1.0f + __expf(-(ratio - nthresh) * slope));
}

// lane l <- lane l -/+ 1 of the whole wavefront, 0 at the end (DPP wave_shr:1 / wave_shl:1, bound_ctrl): folded into the
// consuming VALU instruction by the compiler (v_add_f32_dpp)
__device__ __forceinline__ float lane_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// One thread = one bin column of a time tile (NS_TT frames + NT halo rows each side), everything in REGISTERS:
// no LDS, no barriers, occupancy bounded by registers only.
//   loads (all rows in flight) -> forward sweep -> backward sweep regenerating the forward values in reverse
//   (s_f[t-1] = (s_f[t] - b A[t]) / c) with the sigmoid written in place -> triangle smoothing along t as two
//   running boxcar sums (float64 accumulators) -> smoothing along f through DPP lane shifts (a wave covers
//   64 - 2 nf output bins plus nf halo columns per side) -> p * . + (1 - p) -> store.
// NT is a template parameter: the row arrays must be indexed statically.  EDGE: tiles that touch frame 0 / T
// (rows outside [0, T) are the smoothing's zero padding and are skipped by the recurrence).
template <int NT, bool EDGE>
__device__ __forceinline__ void ns_mask_tile(const float* __restrict__ A, const double* __restrict__ carry,
                                             const Geom& g, const NsTiling& tl, double b, double nthresh, double slope,
                                             int nf, float p, float* __restrict__ M, int64_t k) {
  constexpr int ROWS = NS_TT + 2 * NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int BW = 64 - 2 * nf;
  const int f = (blockIdx.x * 4 + wave) * BW - nf + lane;
  const int64_t nk = tl.n_tiles();
  const int64_t u = blockIdx.z;
  if ((blockIdx.x * 4 + wave) * BW >= g.F) return;   // wave-uniform
  const int64_t ts = k * NS_TT, te = ts + NS_TT < g.T ? ts + NS_TT : g.T;
  const bool col_on = f >= 0 && f < g.F;
  const int fc = f < 0 ? 0 : (f >= g.F ? g.F - 1 : f);
  const int64_t ta = ts - NT > 0 ? ts - NT : 0, tb = te + NT < g.T ? te + NT : g.T;
  const int ra = (int)(ta - (ts - NT)), rb = (int)(tb - (ts - NT));   // valid rows [ra, rb): block-uniform
  const int n_out = (int)(te - ts);
  float x[ROWS];
  {
    const float* colp = A + (u * g.T) * g.FS + fc;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int64_t t = ts - NT + r;
      if (EDGE) t = t < 0 ? 0 : (t >= g.T ? g.T - 1 : t);   // clamped address, value masked below
      x[r] = colp[t * g.FS];
    }
  }
  {
    const double c = 1.0 - b, rc = 1.0 / c;
    const double* cb = carry + ((u * nk + k) * 2) * (int64_t)g.FS + fc;
    double s = cb[0];          // s_f[ts - 1]
    double S = cb[g.FS];       // S[te]
    // forward state at the outer edge of the leading halo: s_f[ta - 1], down the halo rows
#pragma unroll
    for (int r = NT - 1; r >= 0; --r) {
      const double sp = (s - b * (double)x[r]) * rc;
      s = (!EDGE || r >= ra) ? sp : s;
    }
    // forward sweep; beside it, up the trailing halo rows: S[t + 1] = (S[t] - b s_f[t]) / c  ->  S[tb]
    // (sched_barriers: fully unrolled, the scheduler would otherwise convert every row to float64 up front --
    // two registers per row -- and spill)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if ((r & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const double sn = b * (double)x[r] + c * s;
      const bool in = !EDGE || (r >= ra && r < rb);
      s = in ? sn : s;
      if (r >= NT + NS_TT || (EDGE && r >= NT)) {        // rows past the tile's last frame (interior tiles: 64 frames)
        const double Sn = (S - b * sn) * rc;
        S = (in && (!EDGE || r >= NT + n_out)) ? Sn : S;
      }
    }
    if (EDGE && tb == g.T) S = s;                          // the backward pass's seed, exactly
    __builtin_amdgcn_sched_barrier(0);
    const float nth = (float)nthresh, slp = (float)slope;
#pragma unroll
    for (int r = ROWS - 1; r >= 0; --r) {
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      const bool in = !EDGE || (r >= ra && r < rb);
      float xv = x[r];
      asm volatile("" : "+v"(xv));           // not a CSE of the forward sweep's conversion (82 doubles kept = spills)
      const double av = (double)xv;
      const double Sn = b * s + c * S;       // s = s_f[t]
      float m = sigmoid_ratio_rcp(av, Sn, nth, slp);
      asm volatile("" : "+v"(m));            // evaluated HERE (else it is sunk to its use and (av, S) stay live per row)
      const double sp = (s - b * av) * rc;   // s_f[t - 1]
      S = in ? Sn : S;
      s = in ? sp : s;
      x[r] = (in && col_on) ? m : 0.f;       // zero padding outside the recording / the spectrum
    }
  }
  // ---- smoothing along t: triangle = boxcar(NT+1) * boxcar(NT+1); y[i] = sum_{e<=NT} B[i+e], B[r] = sum_{d<=NT} x[r+d]
  {
    constexpr int W = NT + 1, NB = NS_TT + NT;
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < W; ++d) acc += (double)x[d];
#pragma unroll
    for (int r = 0; r < NB; ++r) {           // in place: B[r] overwrites x[r]
      if ((r & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const double xo = (double)x[r], xn = r + W < ROWS ? (double)x[r + W] : 0.0;
      x[r] = (float)acc;
      acc += xn - xo;
    }
    acc = 0.0;
#pragma unroll
    for (int e = 0; e < W; ++e) acc += (double)x[e];
    const float inv = 1.0f / (float)(W * W);
#pragma unroll
    for (int i = 0; i < NS_TT; ++i) {        // in place: y[i] overwrites B[i]
      if ((i & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const double bo = (double)x[i], bn = i + W < NB ? (double)x[i + W] : 0.0;
      x[i] = (float)acc * inv;
      acc += bn - bo;
    }
  }
  // ---- smoothing along f through the wave + prop_decrease (applied AFTER smoothing, nonstationary.py:78-84).
  // The normalised triangle of half-width nf is boxcar(nf+1) * boxcar(nf+1) / (nf+1)^2 (utils.py:45-60: linspace
  // ramps k / (nf+1), divided by their sum nf+1), and a full-wave lane shift by one is a DPP modifier of the VALU add
  // (wave_shr:1 / wave_shl:1, zero shifted in at the wave's ends -- halo lanes): 2 nf adds per value at the VALU rate,
  // where 2 nf + 1 ds_bpermute through the LDS crossbar took 4.3 x as long each (tools/ubench/dpp_shift.hip).
  //   B[l] = sum_{d<=nf} x[l-d]  (right shifts),   y[l] = sum_{e<=nf} B[l+e]  (left shifts)  = sum_a tri[a] x[l+a-nf]
  const bool out_on = lane >= nf && lane < 64 - nf && f < g.F;
  const float q = 1.0f - p;
  const float ps = p / (float)((nf + 1) * (nf + 1));
  float* mp = M + (u * g.T + ts) * g.FS + fc;
#pragma unroll
  for (int i0 = 0; i0 < NS_TT; i0 += 16) {    // fully unrolled: x[] must stay in registers; 16 independent chains
    float acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = x[i0 + m];
    for (int a = 0; a < nf; ++a) {
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m] = lane_shr1(acc[m]) + x[i0 + m];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) x[i0 + m] = acc[m];
    for (int a = 0; a < nf; ++a) {
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m] = lane_shl1(acc[m]) + x[i0 + m];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
      if (out_on && (!EDGE || i0 + m < n_out)) mp[(i0 + m) * (int64_t)g.FS] = ps * acc[m] + q;   // interior tiles: 64 rows
  }
}

// grid (bin blocks, time tiles, units).  Interior tiles (every row of the tile and its halos inside [0, T)) take
// the predicate-free instantiation, the first tile and the last one or two the EDGE one (block-uniform branch).
template <int NT>
__global__ __launch_bounds__(256, (NT <= 9 ? 3 : 2)) void k_iir_mask(const float* __restrict__ A, const double* __restrict__ carry,
                                                     Geom g, NsTiling tl, double b, double nthresh, double slope,
                                                     int nf, float p, float* __restrict__ M) {
  const int64_t k = blockIdx.y;
  const bool edge = k * NS_TT - NT < 0 || (k + 1) * NS_TT + NT > g.T;
  if (edge) ns_mask_tile<NT, true>(A, carry, g, tl, b, nthresh, slope, nf, p, M, k);
  else ns_mask_tile<NT, false>(A, carry, g, tl, b, nthresh, slope, nf, p, M, k);
}

}  // namespace sg
