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
//   k_iir_part   (reads |X| once)      per (tile, sub-tile, band): e_f = sum_t b c^(end-1-t) A[t],
//                                      E0 = sum_t b c^(t-start) s0[t]   (s0 = zero-state forward response)
//   k_iir_chain  (tiny)                forward states before / backward states after every sub-tile
//   k_iir_mask   (reads |X| once more, writes M) per (time tile + nt halo rows, 128-bin block incl. nf halo):
//                                      |X| tile -> LDS; forward sweep; backward sweep that REGENERATES the forward
//                                      values in reverse (s_f[t-1] = (s_f[t] - b A[t]) / c: error growth c^-rows,
//                                      1.25 at the default 2 s time constant) and writes the sigmoid in place;
//                                      separable triangle smoothing of the tile in LDS; p * . + (1 - p); store.
//
// Every tile is split at ts + nt and te - nt so that the chain also yields the states at the edges of the
// neighbours' halo rows.  All recurrences in float64 (the reference's precision); the previous kernels rounded
// the forward pass to float32 between sweeps.
#pragma once
#include "kernels.hpp"

namespace sg {

constexpr int NS_TT = 64;      // frames per time tile
constexpr int NS_MAX_NF = 8;   // k_iir_mask: a wave holds 64 - 2 nf output bins

struct NsTiling {
  int64_t T;
  int nt;
  __host__ __device__ int64_t n_tiles() const { return (T + NS_TT - 1) / NS_TT; }
  __host__ __device__ void bounds(int64_t k, int sub, int64_t& a, int64_t& b) const {
    const int64_t ts = k * NS_TT, te = ts + NS_TT < T ? ts + NS_TT : T;
    const int64_t sa = ts + nt < te ? ts + nt : te;
    const int64_t sb = te - nt > sa ? te - nt : sa;
    a = sub == 0 ? ts : (sub == 1 ? sa : sb);
    b = sub == 0 ? sa : (sub == 1 ? sb : te);
  }
};

// partials [unit][tile][sub][2][FS]
__global__ __launch_bounds__(256) void k_iir_part(const float* __restrict__ A, Geom g, NsTiling tl, double b,
                                                  double* __restrict__ part) {
  const int l = threadIdx.x & 63;
  const int f = blockIdx.x * 64 + l;
  const int64_t k = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  const int64_t u = blockIdx.z;
  if (f >= g.F || k >= tl.n_tiles()) return;
  const double c = 1.0 - b;
  const float* a = A + u * g.T * g.FS + f;
  for (int sub = 0; sub < 3; ++sub) {
    int64_t t0, t1;
    tl.bounds(k, sub, t0, t1);
    double e = 0.0, E0 = 0.0, pw = b;
#pragma unroll 4
    for (int64_t t = t0; t < t1; ++t) {
      e = b * (double)a[t * g.FS] + c * e;
      E0 += pw * e;
      pw *= c;
    }
    double* o = part + (((u * tl.n_tiles() + k) * 3 + sub) * 2) * (int64_t)g.FS + f;
    o[0] = e;
    o[g.FS] = E0;
  }
}

// carries [unit][tile][sub][2][FS]: [0] forward state before the sub-tile's first frame (s_f[start - 1]),
// [1] backward state at its end (S[end]; S[T] := s_f[T - 1], the seed of the backward pass)
__global__ __launch_bounds__(64) void k_iir_chain(const float* __restrict__ A, const double* __restrict__ part,
                                                  Geom g, NsTiling tl, double b, double* __restrict__ carry,
                                                  int64_t n_units) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* pw1 = reinterpret_cast<double*>(smem);  // [3 nk] c^len
  const int64_t nk = tl.n_tiles();
  const int nj = (int)(nk * 3);
  double* pw2 = pw1 + nj;                          // [3 nk] 1 - c^(2 len)
  const double c = 1.0 - b;
  for (int j = threadIdx.x; j < nj; j += 64) {
    int64_t t0, t1;
    tl.bounds(j / 3, j % 3, t0, t1);
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
  const double* pb = part + (u * nk * 3 * 2) * (int64_t)g.FS + f;
  double* cb = carry + (u * nk * 3 * 2) * (int64_t)g.FS + f;
  double s = (double)A[u * g.T * g.FS + f];  // s[-1] = A[0]  (lfilter_zi steady state)
  // the chain is serial, its operands are not: 32 sub-tiles' partials are fetched together
  const int64_t st2 = 2 * (int64_t)g.FS;
  constexpr int CB = 32;
  for (int j0 = 0; j0 < nj; j0 += CB) {
    double e[CB];
#pragma unroll
    for (int q = 0; q < CB; ++q) e[q] = j0 + q < nj ? pb[(j0 + q) * st2] : 0.0;
#pragma unroll
    for (int q = 0; q < CB; ++q)
      if (j0 + q < nj) {
        cb[(j0 + q) * st2] = s;
        s = e[q] + pw1[j0 + q] * s;
      }
  }
  // backward: S[end_j] given; S[start_j] = E_b + c^len S[end_j],
  // E_b = sum_t b c^(t-start) s_f[t] = E0 + s_in * b c (1 - c^(2 len)) / (1 - c^2)   (s_f = s0 + c^(t-start+1) s_in)
  double S = s;  // seed: the forward pass's last value
  const double gq = b * c / (1.0 - c * c);
  constexpr int CB2 = 16;
  for (int j1 = nj - 1; j1 >= 0; j1 -= CB2) {
    double e0[CB2], sin[CB2];
#pragma unroll
    for (int q = 0; q < CB2; ++q) {
      const int j = j1 - q;
      e0[q] = j >= 0 ? pb[j * st2 + g.FS] : 0.0;
      sin[q] = j >= 0 ? cb[j * st2] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < CB2; ++q) {
      const int j = j1 - q;
      if (j >= 0) {
        cb[j * st2 + g.FS] = S;
        S = (e0[q] + sin[q] * gq * pw2[j]) + pw1[j] * S;
      }
    }
  }
}

// One thread = one bin column of a time tile (NS_TT frames + NT halo rows each side), everything in REGISTERS:
// no LDS, no barriers, occupancy bounded by registers only.
//   loads (all rows in flight) -> forward sweep -> backward sweep regenerating the forward values in reverse
//   (s_f[t-1] = (s_f[t] - b A[t]) / c) with the sigmoid written in place -> triangle smoothing along t as two
//   running boxcar sums (float64 accumulators) -> smoothing along f through lane shuffles (a wave covers
//   64 - 2 nf output bins plus nf halo columns per side) -> p * . + (1 - p) -> store.
// NT is a template parameter: the row arrays must be indexed statically.  EDGE: tiles that touch frame 0 / T
// (rows outside [0, T) are the smoothing's zero padding and are skipped by the recurrence).
template <int NT, bool EDGE>
__device__ __forceinline__ void ns_mask_tile(const float* __restrict__ A, const double* __restrict__ carry,
                                             const Geom& g, const NsTiling& tl, double b, double nthresh, double slope,
                                             const float* __restrict__ kf, int nf, float p, float* __restrict__ M,
                                             int64_t k) {
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
    const double* cb = carry + (u * nk * 3 * 2) * (int64_t)g.FS + fc;
    const int64_t jf = k == 0 ? 0 : (k - 1) * 3 + 2;          // forward state before frame ta
    const int64_t jb = te == g.T ? k * 3 + 2 : (k + 1) * 3;   // backward state at frame tb
    double s = cb[(jf * 2) * g.FS];
    double S = cb[(jb * 2 + 1) * g.FS];
    // (sched_barriers: fully unrolled, the scheduler would otherwise convert every row to float64 up front --
    // two registers per row -- and spill)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if ((r & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const double sn = b * (double)x[r] + c * s;
      s = (!EDGE || (r >= ra && r < rb)) ? sn : s;
    }
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
      float m = sigmoid_ratio(av, Sn, nth, slp);
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
  // ---- smoothing along f through the wave + prop_decrease (applied AFTER smoothing, nonstationary.py:78-84)
  const bool out_on = lane >= nf && lane < 64 - nf && f < g.F;
  const float q = 1.0f - p;
  float* mp = M + (u * g.T + ts) * g.FS + fc;
  // 16 rows per tap: the tap weight is fetched once (scalar) and the 16 shuffles / FMAs are independent
#pragma unroll
  for (int i0 = 0; i0 < NS_TT; i0 += 16) {    // fully unrolled: x[] must stay in registers
    float acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = 0.f;
    for (int a = 0; a <= 2 * nf; ++a) {
      const float wgt = kf[a];
      const int src = lane + a - nf;
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m] += wgt * __shfl(x[i0 + m], src);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
      if (out_on && i0 + m < n_out) mp[(i0 + m) * (int64_t)g.FS] = p * acc[m] + q;
  }
}

// grid (bin blocks, time tiles, units).  Interior tiles (every row of the tile and its halos inside [0, T)) take
// the predicate-free instantiation, the first tile and the last one or two the EDGE one (block-uniform branch).
template <int NT>
__global__ __launch_bounds__(256, (NT <= 9 ? 3 : 2)) void k_iir_mask(const float* __restrict__ A, const double* __restrict__ carry,
                                                     Geom g, NsTiling tl, double b, double nthresh, double slope,
                                                     const float* __restrict__ kf, int nf, float p,
                                                     float* __restrict__ M) {
  const int64_t k = blockIdx.y;
  const bool edge = k * NS_TT - NT < 0 || (k + 1) * NS_TT + NT > g.T;
  if (edge) ns_mask_tile<NT, true>(A, carry, g, tl, b, nthresh, slope, kf, nf, p, M, k);
  else ns_mask_tile<NT, false>(A, carry, g, tl, b, nthresh, slope, kf, nf, p, M, k);
}

}  // namespace sg
