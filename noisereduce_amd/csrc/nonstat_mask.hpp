// Register-tile mask kernels of the non-stationary gates (variant S: k_iir_mask, variant T: k_box_mask) -- kernel
// TEMPLATES only (34 instantiations x 2 bodies: their own translation unit, nonstat_mask.hip, compiled beside api.hip).
// The description of the two-pass scheme is in nonstat.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geom.hpp"

namespace sg {

constexpr int NS_TT = 64;      // frames per time tile
constexpr int NS_MAX_NF = 24;  // k_iir_mask: a wave holds 64 - 2 nf output bins (the DPP boxcars cost 2 nf adds per value)

struct NsTiling {
  int64_t T;
  int nt;
  int64_t k0 = 0;   // first tile of this launch (grid.y <= 65535: very long windows take several launches)
  int tt = NS_TT;   // frames per tile (the float64 pipeline of exact.hpp chains tiles of 32: two register rows per frame)
  __host__ __device__ int64_t n_tiles() const { return (T + tt - 1) / tt; }
};

// sigmoid_ratio (kernels.hpp) with v_rcp_f32 in place of the two IEEE divisions (10 instructions each; the mask is a
// float32 field, 1 ulp of the reciprocal is 1e-7 of it).  v_rcp_f32 flushes denormals: a smoothed magnitude below
// 1e-30 (or NaN) takes the IEEE form -- digital silence included: 0 / 0 = NaN as in nonstationary.py:75.
__device__ __forceinline__ float sigmoid_ratio_rcp(double av, double s, float nthresh, float slope) {
  const float num = (float)(av - s), den = (float)s;
  float ratio = num * __builtin_amdgcn_rcpf(den);
  if (__builtin_expect(!(den >= 1e-30f), 0)) ratio = num / den;
  return __builtin_amdgcn_rcpf(1.0f + __expf(-(ratio - nthresh) * slope));
}

// XCD-aware tile order (see k_iir_mask): a bijection of the grid's linear index, x fastest
constexpr unsigned SG_XCDS = 8;
__device__ __forceinline__ void xcd_remap(unsigned& bx, unsigned& by, unsigned& bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y;
  const unsigned total = gx * gy * gridDim.z;
  const unsigned L = bx + gx * (by + gy * bz);
  const unsigned xcd = L % SG_XCDS, slot = L / SG_XCDS;
  const unsigned q = total / SG_XCDS, r = total % SG_XCDS;
  const unsigned P = xcd * q + (xcd < r ? xcd : r) + slot;   // XCD j owns q + (j < r) positions
  bx = P % gx;
  by = (P / gx) % gy;
  bz = P / (gx * gy);
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
                                             int nf, float p, float* __restrict__ M, int64_t k, int bx, int64_t u) {
  constexpr int ROWS = NS_TT + 2 * NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int BW = 64 - 2 * nf;
  const int f = (bx * 4 + wave) * BW - nf + lane;
  const int64_t nk = tl.n_tiles();
  if ((bx * 4 + wave) * BW >= g.F) return;   // wave-uniform
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
#ifndef NS_ABLATE_TSMOOTH   // (development: -DNS_ABLATE_TSMOOTH removes this stage to measure what it costs; results are wrong)
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
#endif
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
  // Which tile: workgroups are dispatched round-robin over the 8 XCDs, each with its own L2, and neighbouring tiles
  // share their halos (nt rows of the next time tile, nf columns of the next bin block: 40 % of what a tile loads).  Deal
  // every XCD a CONTIGUOUS run of the launch's tiles (bin block fastest, then time, then unit) so that those re-reads hit
  // its L2: dispatch index L -> XCD L % 8, its (L / 8)-th workgroup -> position off(L % 8) + L / 8.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  xcd_remap(bx, by, bz);
  const int64_t k = tl.k0 + by;
  const bool edge = k * NS_TT - NT < 0 || (k + 1) * NS_TT + NT > g.T;
  if (edge) ns_mask_tile<NT, true>(A, carry, g, tl, b, nthresh, slope, nf, p, M, k, (int)bx, (int64_t)bz);
  else ns_mask_tile<NT, false>(A, carry, g, tl, b, nthresh, slope, nf, p, M, k, (int)bx, (int64_t)bz);
}

// ------------------------------------------------------------------------------------------------------------
// Variant T (TorchGate), non-stationary: the same register tile for
//   S = conv1d(|X|, ones(KB), "same") / KB along time (zero padded, left pad (KB-1)//2: torchgate.py:179-190),
//   raw = sigmoid(((|X| - S) / S - thresh) * slope)   (torchgate.py:193-196),
//   M = p * smooth(raw) + (1 - p) * smooth(1)         (prop_decrease BEFORE the zero-padded smoothing: torchgate.py:241-249)
// instead of k_boxcar_sigmoid + k_smooth_tiled (the raw field written and re-read, the smoothing through LDS).
// One thread = one bin column: NS_TT + 2 NT smoothing rows + KB - 1 more for the moving mean's windows, all in
// registers; the window sum slides in float64 like k_boxcar_sigmoid's; the sigmoid of row r overwrites x[r] (rows
// r+1.. only need x[r+1..]); smoothing as in ns_mask_tile (running boxcars along t, DPP lane shifts along f).
template <int NT, int KB, bool EDGE>
__device__ __forceinline__ void box_mask_tile(const float* __restrict__ A, const Geom& g, double nthresh, double slope,
                                              int nf, float p, float* __restrict__ M, int64_t k) {
  constexpr int LEFT = (KB - 1) / 2;
  constexpr int SR = NS_TT + 2 * NT;          // smoothing rows
  constexpr int ROWS = SR + KB - 1;           // loaded rows: frames ts - NT - LEFT ..
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int BW = 64 - 2 * nf;
  const int f = (blockIdx.x * 4 + wave) * BW - nf + lane;
  const int64_t u = blockIdx.z;
  if ((blockIdx.x * 4 + wave) * BW >= g.F) return;   // wave-uniform
  const int64_t ts = k * NS_TT, te = ts + NS_TT < g.T ? ts + NS_TT : g.T;
  const bool col_on = f >= 0 && f < g.F;
  const int fc = f < 0 ? 0 : (f >= g.F ? g.F - 1 : f);
  const int64_t tfirst = ts - NT - LEFT;      // frame of x[0]
  const int n_out = (int)(te - ts);
  float x[ROWS];
  {
    const float* colp = A + (u * g.T) * g.FS + fc;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int64_t t = tfirst + r;
      if (EDGE) {
        const bool in = t >= 0 && t < g.T;
        t = t < 0 ? 0 : (t >= g.T ? g.T - 1 : t);
        const float v = colp[t * g.FS];
        x[r] = in ? v : 0.f;                  // the moving mean's zero padding
      } else {
        x[r] = colp[t * g.FS];
      }
    }
  }
  {
    const float nth = (float)nthresh, slp = (float)slope;
    const double rk = 1.0 / (double)KB;
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < KB; ++j) sum += (double)x[j];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
      if ((r & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const double S = sum * rk;
      float xa = x[r + LEFT];
      asm volatile("" : "+v"(xa));
      float m = sigmoid_ratio_rcp((double)xa, S, nth, slp);
      asm volatile("" : "+v"(m));
      const double xo = (double)x[r], xn = r + KB < ROWS ? (double)x[r + KB] : 0.0;
      sum += xn - xo;
      bool on = col_on;
      if (EDGE) {
        const int64_t t = ts - NT + r;
        on = on && t >= 0 && t < g.T;         // the smoothing's zero padding outside the row / the spectrum
      }
      x[r] = on ? m : 0.f;
    }
  }
  // ---- smoothing along t (triangle = boxcar(NT+1) * boxcar(NT+1)), in place on x[0 .. SR)
  {
    constexpr int W = NT + 1, NB = NS_TT + NT;
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < W; ++d) acc += (double)x[d];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      if ((r & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const double xo = (double)x[r], xn = r + W < SR ? (double)x[r + W] : 0.0;
      x[r] = (float)acc;
      acc += xn - xo;
    }
    acc = 0.0;
#pragma unroll
    for (int e = 0; e < W; ++e) acc += (double)x[e];
#pragma unroll
    for (int i = 0; i < NS_TT; ++i) {
      if ((i & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const double bo = (double)x[i], bn = i + W < NB ? (double)x[i + W] : 0.0;
      x[i] = (float)acc;
      acc += bn - bo;
    }
  }
  // ---- smoothing along f (DPP boxcars) + prop_decrease with the zero-padded filter's weight inside the field
  const bool out_on = lane >= nf && lane < 64 - nf && f < g.F;
  const float inv_all = 1.0f / (float)((NT + 1) * (NT + 1) * (nf + 1) * (nf + 1));
  const float ps = p * inv_all;
  // valid taps along f for this bin (closed form of the triangle's tails), x valid taps along t per row below
  float ef;
  {
    const int fl = fc < nf ? nf - fc : 0, fr = (g.F - 1 - fc) < nf ? nf - (g.F - 1 - fc) : 0;
    ef = (float)((nf + 1) * (nf + 1) - fl * (fl + 1) / 2 - fr * (fr + 1) / 2);
  }
  const float qe = (1.0f - p) * ef * inv_all;
  float* mp = M + (u * g.T + ts) * g.FS + fc;
#pragma unroll
  for (int i0 = 0; i0 < NS_TT; i0 += 16) {
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
    for (int m = 0; m < 16; ++m) {
      float et = (float)((NT + 1) * (NT + 1));
      if (EDGE) {
        const int64_t t = ts + i0 + m;
        const int64_t tl = t < NT ? NT - t : 0, tr = (g.T - 1 - t) < NT ? NT - (g.T - 1 - t) : 0;
        et = (float)((int64_t)(NT + 1) * (NT + 1) - tl * (tl + 1) / 2 - tr * (tr + 1) / 2);
      }
      if (out_on && (!EDGE || i0 + m < n_out)) mp[(i0 + m) * (int64_t)g.FS] = ps * acc[m] + qe * et;
    }
  }
}

// grid (bin blocks, time tiles, units); interior tiles take the predicate-free instantiation
template <int NT, int KB>
__global__ __launch_bounds__(256, 2) void k_box_mask(const float* __restrict__ A, Geom g, double nthresh, double slope,
                                                     int nf, float p, float* __restrict__ M, int64_t k0) {
  constexpr int LEFT = (KB - 1) / 2;
  const int64_t k = k0 + blockIdx.y;
  const bool edge = k * NS_TT - NT - LEFT < 0 || (k + 1) * NS_TT + NT + (KB - 1 - LEFT) > g.T;
  if (edge) box_mask_tile<NT, KB, true>(A, g, nthresh, slope, nf, p, M, k);
  else box_mask_tile<NT, KB, false>(A, g, nthresh, slope, nf, p, M, k);
}

// launchers (nonstat_mask.hip): the instantiation for a time half-width nt, hipErrorInvalidValue if there is none
hipError_t launch_iir_mask(int nt, dim3 grid, hipStream_t st, const float* mag, const double* carry, Geom g, NsTiling tl,
                           double b, double nthresh, double slope, int nf, float p, float* M);
hipError_t launch_box_mask(int nt, int kbox, dim3 grid, hipStream_t st, const float* mag, Geom g, double nthresh,
                           double slope, int nf, float p, float* M, int64_t k0);
constexpr int NS_IIR_MAX_NT = 20;   // k_iir_mask<0 .. 20>
// (round 5) ... plus the half-widths of the default 50 ms at a hop of 64 samples (n_fft = 256) and 32 / 44.1 / 48 kHz -- the
// same at a hop of 128 and twice the rate: 138 register rows per bin column, two waves per SIMD.  Anything else beyond 20
// takes k_iir_mask<0> + k_smooth_tiled (the raw field through HBM, the smoothing through LDS: 63 + 254 us where this takes ~150).
__host__ __device__ constexpr bool ns_iir_nt_ok(int nt) { return (nt >= 1 && nt <= NS_IIR_MAX_NT) || nt == 25 || nt == 34 || nt == 37; }
constexpr int NS_BOX_MAX_NT = 12;   // k_box_mask<0 .. 12, 20>
constexpr int NS_BOX_KB = 20;       // the moving-mean length k_box_mask is built for (TorchGate's default)

}  // namespace sg
