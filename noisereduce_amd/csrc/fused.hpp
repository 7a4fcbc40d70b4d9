// Fused stationary path (variant S): no time-frequency field wider than one BIT per cell
// is written by the decision stage.
//
//   k_unit_absmax   max|x| per unit (upper bound of every |X[k]| -> is the -top_db floor live?)
//   k_prep_thresh   per-band compare constants from the dB threshold (monotone transform of
//                   20*log10(|Z|+eps) > thresh  <=>  |X|^2 > T2[f]) + per-unit floor flags
//   k_stft_bits     float64 STFT of every frame; MODE_MAX: per-(unit,band) max power (only for
//                   units whose floor may be live), MODE_DECIDE: mask bits via wave ballot
//   k_smooth_bits   separable triangular smoothing on the bit field in exact integer
//                   arithmetic -> uint16 weight sums K (mask = K / ktot)
// fast::k_apply_fast reads K directly (lane order); k_k16_to_mask expands it to a float mask for
// the general apply kernels.
#pragma once
#include "kernels.hpp"
#include "fastpath.hpp"

namespace sg {


// ---------------------------------------------------------------------------------------
__global__ void k_unit_absmax(View view, int64_t n_units, unsigned* __restrict__ umax_bits) {
  // grid: (splits, units).  Non-negative floats order like their bit patterns.
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  // max |x| on the BIT PATTERNS (sign cleared): non-negative floats order like unsigned integers, and NaN / Inf
  // patterns are larger than every finite one -- a non-finite sample survives the reduction (fmaxf drops NaN)
  unsigned mi = 0u;
  auto ab = [](float x) -> unsigned { return __float_as_uint(x) & 0x7fffffffu; };
  // the unit's window clipped to the readable part of the row (everything else is zero)
  const int64_t g0 = chunk * view.cs - view.pad;
  const int64_t s_lo = max<int64_t>(0, view.lo - g0), s_hi = min<int64_t>(view.Lp, view.hi - g0);
  if (view.dtype == 0) {
    const float* src = (const float*)view.x + row * view.stride + g0;
    // 16-byte loads over the aligned middle, scalar loads over the two ragged ends
    const int64_t a_lo = s_lo + ((4 - ((reinterpret_cast<uintptr_t>(src + s_lo) >> 2) & 3)) & 3);
    const int64_t n4 = a_lo < s_hi ? (s_hi - a_lo) / 4 : 0;
    const float4* s4 = reinterpret_cast<const float4*>(src + a_lo);
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      float4 v4 = s4[i];
      mi = max(max(mi, max(ab(v4.x), ab(v4.y))), max(ab(v4.z), ab(v4.w)));
    }
    if (blockIdx.x == 0) {
      for (int64_t s = s_lo + threadIdx.x; s < min(a_lo, s_hi); s += blockDim.x) mi = max(mi, ab(src[s]));
      for (int64_t s = a_lo + 4 * n4 + threadIdx.x; s < s_hi; s += blockDim.x) mi = max(mi, ab(src[s]));
    }
  } else if (view.dtype == 2) {
    // int16 recordings (the float64 pipeline's floor test): eight samples per 16-byte load, the maximum taken on the
    // integers ((float)|x| is exact and monotone: the same bound as the per-sample path below, which took 39 us of a
    // 0.67 ms ten-minute call)
    const int16_t* src = (const int16_t*)view.x + row * view.stride + g0;
    const int64_t a_lo = s_lo + ((8 - ((reinterpret_cast<uintptr_t>(src + s_lo) >> 1) & 7)) & 7);
    const int64_t n8 = a_lo < s_hi ? (s_hi - a_lo) / 8 : 0;
    const uint4* s8 = reinterpret_cast<const uint4*>(src + a_lo);
    int m = 0;
    auto two = [&](unsigned w) {
      const int lo = (int)(short)(w & 0xffffu), hi = (int)w >> 16;
      m = max(m, max(lo < 0 ? -lo : lo, hi < 0 ? -hi : hi));
    };
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
      const uint4 q = s8[i];
      two(q.x); two(q.y); two(q.z); two(q.w);
    }
    if (blockIdx.x == 0) {
      auto one = [&](int64_t s) { const int x = src[s]; m = max(m, x < 0 ? -x : x); };
      for (int64_t s = s_lo + threadIdx.x; s < min(a_lo, s_hi); s += blockDim.x) one(s);
      for (int64_t s = a_lo + 8 * n8 + threadIdx.x; s < s_hi; s += blockDim.x) one(s);
    }
    mi = ab((float)m);
  } else {
    for (int64_t s = s_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < s_hi;
         s += (int64_t)gridDim.x * blockDim.x)
      mi = max(mi, ab((float)view_sample(view, row, chunk, s)));
  }
  // float(double) rounds to nearest: inflate by one ulp so the bound stays an upper bound
  if (mi < 0x7f800000u) mi = __float_as_uint(__uint_as_float(mi) * 1.0000002f);
  for (int off = 32; off > 0; off >>= 1) mi = max(mi, (unsigned)__shfl_xor((int)mi, off));
  // one atomic per block: same-address L2 atomics serialise
  __shared__ unsigned s_m[16];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mi;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mi = max(mi, s_m[w]);
    atomicMax(&umax_bits[u], mi);
  }
}

// T2[f]: 20*log10(|X|*mag_scale + eps) > thresh[f]   <=>   |X|^2 > T2[f]   with
//   Tm = (10^(thresh/20) - eps) / mag_scale ;  T2 = Tm^2  (Tm > 0)
// and the two degenerate cases pinned to what the reference's formula gives for |X| = 0:
//   zero cell passes (20*log10(eps) > thresh)  -> T2 = -1  (everything passes)
//   zero cell fails                            -> T2 = max(T2, 0)  (0 > 0 is false)
// need_floor[u]: the floor max(dB, rowmax - top_db) can only lift a cell above thresh[f] if
// rowmax_dB - top_db > thresh[f]; |X[k]| <= sum|x w| <= max|x| * sum|w| bounds rowmax.
// Also keeps two buffers clean so that no memset launch sits on the critical path: umax_bits (zeroed after it
// was read: k_unit_absmax of the NEXT call accumulates into it with atomicMax) and the pmax rows of the units
// whose floor can be live (the floor pre-pass accumulates into them; nobody reads the other rows).
// live_host (host-mapped, may be null): set to `stamp` when some unit's floor may be live -- the one-pass gate's host
// side reads it (without synchronising) to choose between this a-priori test and the in-kernel one, see
// k_prep_thresh_lazy.
__global__ void k_prep_thresh(const double* __restrict__ thresh, int F, double mag_scale, double sum_abs_w,
                              double top_db, unsigned* __restrict__ umax_bits, int64_t n_units,
                              double* __restrict__ T2, int* __restrict__ need_floor, double* __restrict__ pmax,
                              int FS, unsigned* __restrict__ live_host, unsigned stamp) {
  const double eps = 2.220446049250313e-16;
  __shared__ double s_min[256];
  double mn = 1e300;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    double th = thresh[f];
    mn = fmin(mn, th);
    if (blockIdx.x == 0) {
      double zero_db = 20.0 * log10(eps);
      double t2;
      if (th != th) {
        t2 = T2_NEVER;   // NaN threshold: `dB > NaN` is False for every cell of the band
      } else if (zero_db > th) {
        t2 = -1.0;
      } else {
        double tm = (exp10(th / 20.0) - eps) / mag_scale;
        t2 = tm > 0.0 ? tm * tm : 0.0;
      }
      T2[f] = t2;
    }
  }
  s_min[threadIdx.x] = mn;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + o]);
    __syncthreads();
  }
  const double min_thresh = s_min[0];
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units;
       u += (int64_t)gridDim.x * blockDim.x) {
    const unsigned mbits = umax_bits[u];
    double ub = (double)__uint_as_float(mbits) * sum_abs_w * mag_scale;
    umax_bits[u] = 0u;
    double ub_db = 20.0 * log10(ub + eps) + 1e-6;  // margin covers log10/rounding slack
    // 2: a NaN / Inf sample in the unit -- every band's maximum is NaN, no cell passes (T2_NEVER)
    const int need = mbits >= 0x7f800000u ? 2 : ((ub_db - top_db > min_thresh) ? 1 : 0);
    need_floor[u] = need;
    if (need) {
      for (int f = 0; f < FS; ++f) pmax[u * (int64_t)FS + f] = 0.0;
      if (live_host) *live_host = stamp;
    }
  }
}

// The one-pass gate's variant (onepass.hpp): k_unit_absmax reads the whole recording once more (26 us of a 345 us
// call at 10 minutes of 48 kHz) only to learn that no unit's floor can be live.  Here the gate kernel makes that test
// itself on the samples it stages anyway: this kernel turns the dB test of k_prep_thresh into ONE compare constant on
// max|x|,
//   20 log10(max|x| sum|w| mag_scale + eps) + 1e-6 - top_db > min_f thresh[f]   <=>   max|x| > a_lim,
// published as the bit pattern of a float a little BELOW a_lim (non-negative floats order like their bit patterns; a
// test that fires too often only costs time: a flagged unit takes the exact floor path).  Only needed when the threshold
// did not come from sg_noise_stats, whose last kernel (k_colstats1, Colstats1Fin) derives the same constants itself.
__global__ void k_prep_thresh_lazy(const double* __restrict__ thresh, int F, double mag_scale, double sum_abs_w,
                                   double top_db, double* __restrict__ T2, unsigned* __restrict__ alim_bits,
                                   int nb = OP_ALIM_BLOCKS /* bounds to write: one per 64-band block, all equal here */) {
  const double eps = 2.220446049250313e-16;
  __shared__ double s_min[256];
  double mn = 1e300;
  const double zero_db = 20.0 * log10(eps);
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const double th = thresh[f];
    mn = fmin(mn, th);
    double t2;
    if (th != th) {
      t2 = T2_NEVER;
    } else if (zero_db > th) {
      t2 = -1.0;
    } else {
      const double tm = (exp10(th / 20.0) - eps) / mag_scale;
      t2 = tm > 0.0 ? tm * tm : 0.0;
    }
    T2[f] = t2;
  }
  s_min[threadIdx.x] = mn;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // k_prep_thresh: ub = float(max|x| (1 + 2^-22)) sum|w| mag_scale;  need <=> 20 log10(ub + eps) + 1e-6 - top_db > min
    const double lim = (exp10((s_min[0] + top_db - 1e-6) / 20.0) - eps) / (sum_abs_w * mag_scale);
    unsigned bits;
    if (!(lim > 0.0)) {
      bits = 0u;                               // (or NaN) every tile reports: the floor path is exact for every unit
    } else {
      // 1e-6 relative slack covers the float rounding of max|x| (k_unit_absmax inflates by an ulp) and exp10's error
      const float lf = __double2float_rd(lim * (1.0 - 1e-6));
      bits = __float_as_uint(lf);              // +Inf (limit beyond float): only non-finite samples report
    }
    for (int b = 0; b < nb; ++b) alim_bits[2 + b] = bits;   // (the gate takes the minimum of the block bounds)
  }
}

// ---------------------------------------------------------------------------------------
// float64 STFT + decision.  One wavefront per frame (LDS Stockham core of fft_wave.hpp).
// MODE 0 (max): only for units with need_floor: atomic max of the raw power per band.
// MODE 1 (decide): bits[u][t][w] (64 bins per word) = |X|^2 > T2[f]  ||  floor lifts the band.
// ---------------------------------------------------------------------------------------
// NT: threads per frame (64: one wavefront; 256: the whole workgroup shares a frame -- from N = 2048 on, where a single
// wave's 32 float64 points per pass cost 256 VGPRs + scratch).  WAVES = frames in flight per workgroup.
template <int N, int WAVES, int FPW, int MODE, int NT = 64>
__global__ __launch_bounds__(WAVES * NT) void k_stft_bits(View view, Geom g, const cx<double>* __restrict__ tw_g,
                                                          const double* __restrict__ wfull, ThreshConsts tc,
                                                          double mag_scale, double top_db,
                                                          unsigned long long* __restrict__ pmax_bits,
                                                          unsigned long long* __restrict__ bits, int wpr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cx<double>* tw = reinterpret_cast<cx<double>*>(smem);
  cx<double>* bufs = tw + N;
  double* sT2 = reinterpret_cast<double*>(bufs + WAVES * lpn<double>(N));  // [N+1] compare constants
  const int lane = threadIdx.x % NT;   // thread within the frame's team
  const int wave = threadIdx.x / NT;
  cx<double>* buf = bufs + wave * lpn<double>(N);
  const int64_t u = blockIdx.y;
  const int need = need_of(tc, u);
  const bool floor_live = need == 1;
  if (MODE == 0 && !floor_live) return;  // whole block: uniform
  stage_twiddles<WAVES * NT, N>(tw, tw_g, (int)threadIdx.x);
  if (MODE == 1) {
    for (int i = threadIdx.x; i <= N; i += WAVES * NT) {
      double t2 = tc.T2[i];
      if (floor_live) {
        // band lifted by the floor: rowmax_dB - top_db > thresh  =>  every cell passes
        double fl = cell_db(tc.pmax[u * g.FS + i], mag_scale) - top_db;
        if (fl > tc.thresh[i]) t2 = -1.0;
      }
      if (need == 2) t2 = T2_NEVER;
      sT2[i] = t2;
    }
  }
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  double vmax[N / NT + 1];
#pragma unroll
  for (int m = 0; m <= N / NT; ++m) vmax[m] = 0.0;
  for (int fi = 0; fi < FPW; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * FPW + fi) * WAVES + wave;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    for (int j = lane; j < N; j += NT) {
      cx<double> z = {0.0, 0.0};
      if (valid) {
        z.x = view_sample(view, row, chunk, s0 + 2 * j) * wfull[2 * j];
        z.y = view_sample(view, row, chunk, s0 + 2 * j + 1) * wfull[2 * j + 1];
      }
      buf[lp<double>(j)] = z;
    }
    SG_PASS_SYNC();
    wave_fft<double, N, false, NT>(buf, tw, lane);
    unsigned long long* brow = bits + ((u * g.T + t) * (int64_t)wpr);
#pragma unroll
    for (int m = 0; m <= N / NT; ++m) {
      const int k = lane + NT * m;
      bool pred = false;
      if (k <= N) {
        cx<double> a = buf[lp<double>(k == N ? 0 : k)];
        cx<double> b = buf[lp<double>((k == 0 || k == N) ? 0 : N - k)];
        cx<double> w = tw[k == N ? 0 : k];
        cx<double> X = rfft_bin(a, b, w, k, N);
        double P = X.x * X.x + X.y * X.y;
        if (MODE == 0) vmax[m] = fmax(vmax[m], valid ? P : 0.0);
        else pred = P > sT2[k];
      }
      if (MODE == 1) {
        // a hardware wave covers 64 consecutive bins: its ballot is word k / 64 of the frame's row
        unsigned long long word = __ballot(pred);
        if (valid && (lane & 63) == 0 && k <= N) brow[k >> 6] = word;
      }
    }
    SG_PASS_SYNC();
  }
  if (MODE == 0) {
#pragma unroll
    for (int m = 0; m <= N / NT; ++m) {
      const int k = lane + NT * m;
      if (k <= N) atomicMax(&pmax_bits[u * g.FS + k], (unsigned long long)__double_as_longlong(vmax[m]));
    }
  }
}

// ---------------------------------------------------------------------------------------
// float32 STFT + decision with exact refinement, any power-of-two n_fft (the default geometry has
// its own register-resident version, fast::k_decide_fast).  |X|^2 comes from the float32 LDS
// Stockham core; a cell is AMBIGUOUS when the float32 value cannot be told from the compare constant:
//     | |X| - sqrt(T2) | <= delta,   delta = 2^-16 * || window * frame ||_2
// (>= 20x the error bound of a float32 FFT of up to 4096 points), tested without square roots as
// (P - T2)^2 <= 2 delta^2 (P + T2).  Ambiguous cells (~1e-5 of all) are re-evaluated exactly: the
// whole wavefront sums the n-term float64 DFT of that bin.  The bits are therefore identical to those
// of the float64 kernel k_stft_bits<MODE 1> (checked by tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------
// (round 5) NT = lanes that share one frame (a TEAM; WAVES = teams per workgroup).  A 64-lane wavefront on a 128-point
// transform leaves 48 lanes idle in two of the three Stockham passes; 16-lane teams run four frames per wavefront with the
// same butterflies in the same order (bit-identical spectra).  Ballots stay wavefront-wide: a team takes its NT bits of
// each ballot, and the exact re-evaluation of an ambiguous cell is done by the whole wavefront for whichever team owns it.
template <int N, int WAVES, int FPW, int NT = 64>
__global__ __launch_bounds__(WAVES * NT) void k_decide_lds(View view, Geom g, const cx<float>* __restrict__ tw_g,
                                                           const float* __restrict__ win32,
                                                           const cx<double>* __restrict__ tw64,
                                                           const double* __restrict__ win64, ThreshConsts tc,
                                                           double mag_scale, double top_db,
                                                           unsigned long long* __restrict__ bits, int wpr) {
  static_assert(NT == 64 || NT == 32 || NT == 16, "team = a power-of-two part of a wavefront");
  constexpr int SY = 1;                     // teams are (parts of) wavefronts, buffers team-private: wave-level pass syncs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cx<float>* tw = reinterpret_cast<cx<float>*>(smem);
  cx<float>* bufs = tw + N;
  float* sT2 = reinterpret_cast<float*>(bufs + WAVES * N);  // [N+1] compare constants (float32)
  constexpr int TPW = 64 / NT;                // teams per wavefront
  const int lane64 = threadIdx.x & 63;
  const int lane = threadIdx.x % NT;          // lane within the team
  const int team = threadIdx.x / NT;          // team within the workgroup
  const int tq = lane64 / NT;                 // team within the wavefront
  cx<float>* buf = bufs + team * N;
  const int64_t u = blockIdx.y;
  const int need = tc.need_floor[u];
  const bool floor_live = need == 1;
  auto t2eff = [&](int k) -> double {  // exact compare constant of band k (-1: every cell passes)
    double t2 = tc.T2[k];
    if (floor_live) {
      const double fl = cell_db(tc.pmax[u * g.FS + k], mag_scale) - top_db;
      if (fl > tc.thresh[k]) t2 = -1.0;
    }
    if (need == 2) t2 = T2_NEVER;
    return t2;
  };
  stage_twiddles<WAVES * NT, N>(tw, tw_g, (int)threadIdx.x);
  // ("every cell passes" as a huge negative constant: P - T > 0 and the ambiguity test fails by itself)
  stage_t2_plain<WAVES * NT, N + 1>(sT2, tc.T2, need, 1.0, (int)threadIdx.x, t2eff);
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  for (int fi = 0; fi < FPW; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * FPW + fi) * WAVES + team;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    float nrm2 = 0.f;
    const float* fp = valid ? frame_ptr_f32(view, row, chunk, s0, 2 * N) : nullptr;  // team-uniform
    if (fp) {
      for (int j = lane; j < N; j += NT) {
        const cx<float> z = {fp[2 * j] * win32[2 * j], fp[2 * j + 1] * win32[2 * j + 1]};
        nrm2 += z.x * z.x + z.y * z.y;
        buf[j] = z;
      }
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<float> z = {0.f, 0.f};
        if (valid) {
          z.x = (float)view_sample(view, row, chunk, s0 + 2 * j) * win32[2 * j];
          z.y = (float)view_sample(view, row, chunk, s0 + 2 * j + 1) * win32[2 * j + 1];
        }
        nrm2 += z.x * z.x + z.y * z.y;
        buf[j] = z;
      }
    }
    for (int off = NT / 2; off > 0; off >>= 1) nrm2 += __shfl_xor(nrm2, off);
    // 2 delta^2 = 2 * 2^-32 * nrm2; a silent frame (nrm2 == 0) has no ambiguous cells
    const float d2 = nrm2 > 0.f ? 2.0f * 2.3283064e-10f * nrm2 : -1.0f;
    team_sync<SY>();
    wave_fft<float, N, false, NT, SY>(buf, tw, lane);
    unsigned long long* brow = bits + ((u * g.T + t) * (int64_t)wpr);
    constexpr int NW = N / 64 + 1;            // words per frame (F = N + 1 bins)
    unsigned long long acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] = 0ull;
#pragma unroll
    for (int m = 0; m <= N / NT; ++m) {
      const int k = lane + NT * m;
      bool pred = false, amb = false;
      if (k <= N) {
        const cx<float> a = buf[k == N ? 0 : k];
        const cx<float> b = buf[(k == 0 || k == N) ? 0 : N - k];
        const cx<float> w = tw[k == N ? 0 : k];
        const cx<float> X = rfft_bin(a, b, w, k, N);
        const float P = X.x * X.x + X.y * X.y;
        const float T = sT2[k];
        const float diff = P - T;
        pred = diff > 0.f;
        amb = valid && diff * diff <= d2 * (P + T);
      }
      // exact re-evaluation, one cell at a time, whole wave cooperating (wave-uniform loop)
      unsigned long long pending = __ballot(amb);
      while (pending) {
        const int src = __ffsll((long long)pending) - 1;
        pending &= pending - 1;
        const int ks = (src % NT) + NT * m;
        // the frame of the team that owns the cell (the wavefront's teams are consecutive frames)
        const int64_t s0s = (t - tq + src / NT) * g.H - g.padL;
        double re = 0.0, im = 0.0;
        for (int i = lane64; i < 2 * N; i += 64) {
          const double xv = view_sample(view, row, chunk, s0s + i) * win64[i];
          const int j = (int)(((int64_t)ks * i) & (2 * N - 1));
          cx<double> w = tw64[j & (N - 1)];
          if (j >= N) { w.x = -w.x; w.y = -w.y; }
          re += xv * w.x;
          im += xv * w.y;
        }
        for (int off = 32; off > 0; off >>= 1) {
          re += __shfl_xor(re, off);
          im += __shfl_xor(im, off);
        }
        const bool pass = re * re + im * im > t2eff(ks);
        if (lane64 == src) pred = pass;
      }
      const unsigned long long word = __ballot(pred);
      if constexpr (NT == 64) {
        acc[m < NW ? m : 0] = word;           // (m < NW always: N / 64 + 1 ballots)
      } else {
        const unsigned long long seg = (word >> (NT * tq)) & ((1ull << NT) - 1ull);
        acc[(NT * m) >> 6] |= seg << ((NT * m) & 63);
      }
    }
    if (valid && lane == 0) {
#pragma unroll
      for (int w = 0; w < NW; ++w) brow[w] = acc[w];
    }
    team_sync<SY>();
  }
}

// ---------------------------------------------------------------------------------------
// Integer mask smoothing.  K[t][f] = sum_{a,b} vf[a] vt[b] bit[t+b][f+a], vf/vt the integer
// triangles [1..m+1..1] (base.py:7-29 times (m+1)), zero outside the unit's (F, T) field
// (fftconvolve mode="same", stationary.py:114).  Exact: K <= (nf+1)^2 (nt+1)^2 <= 65535.
// Block = one tile of TT frames of one unit.  Phase 1: along f, from the bit words, with the
// two-boxcar recurrence  c[f+1]-c[f] = sum(x[f+1..f+m+1]) - sum(x[f-m..f]) ; phase 2: the same
// recurrence along t on the phase-1 counts held in LDS.
// ---------------------------------------------------------------------------------------
constexpr int SM_TT = 64;   // output frames per block
constexpr int SM_SEG = 64;  // recurrence segment length (both phases)

__device__ __forceinline__ int bit_at(const unsigned long long* __restrict__ rowbits, int f, int F) {
  if (f < 0 || f >= F) return 0;
  return (int)((rowbits[f >> 6] >> (f & 63)) & 1ull);
}

// CT: uint8_t when (nf+1)^2 <= 255, else uint16_t (phase-1 counts held in LDS).
__host__ __device__ inline size_t smooth_cf_bytes(int rows, int F, int ct_size) {
  size_t b = (size_t)rows * ((F + 3) & ~3) * ct_size;
  return (b + 15) & ~(size_t)15;
}

template <typename CT>
__global__ __launch_bounds__(256) void k_smooth_bits(const unsigned long long* __restrict__ bits, Geom g, int wpr,
                                                     int nf, int nt, unsigned short* __restrict__ K,
                                                     int perm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int rows = SM_TT + 2 * nt;     // halo rows on both sides
  const int FP = (g.F + 3) & ~3;       // LDS row pitch
  CT* cf = reinterpret_cast<CT*>(smem);                                               // [rows][FP]
  unsigned long long* wb = reinterpret_cast<unsigned long long*>(smem + smooth_cf_bytes(rows, g.F, sizeof(CT)));
  const int64_t u = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * SM_TT;  // first output frame of the tile
  // stage the bit rows (zero rows outside [0, T))
  for (int i = threadIdx.x; i < rows * wpr; i += blockDim.x) {
    int r = i / wpr, w = i % wpr;
    int64_t t = t0 - nt + r;
    wb[i] = (t >= 0 && t < g.T) ? bits[(u * g.T + t) * (int64_t)wpr + w] : 0ull;
  }
  __syncthreads();
  // phase 1: along f.  task = (row r, segment s)
  const int nseg = (g.F + SM_SEG - 1) / SM_SEG;
  for (int task = threadIdx.x; task < rows * nseg; task += blockDim.x) {
    const int r = task / nseg, f0 = (task % nseg) * SM_SEG;
    const unsigned long long* rb = wb + (size_t)r * wpr;
    int c = 0, R = 0, L = 0;
    for (int a = -nf; a <= nf; ++a) c += (nf + 1 - (a < 0 ? -a : a)) * bit_at(rb, f0 + a, g.F);
    for (int j = 1; j <= nf + 1; ++j) R += bit_at(rb, f0 + j, g.F);
    for (int j = 0; j <= nf; ++j) L += bit_at(rb, f0 - j, g.F);
    const int f1 = min(f0 + SM_SEG, g.F);
    for (int f = f0; f < f1; ++f) {
      cf[(size_t)r * FP + f] = (CT)c;
      c += R - L;
      R += bit_at(rb, f + nf + 2, g.F) - bit_at(rb, f + 1, g.F);
      L += bit_at(rb, f + 1, g.F) - bit_at(rb, f - nf, g.F);
    }
  }
  __syncthreads();
  // phase 2: along t on cf.  task = (bin f, time segment)
  const int tsegs = 4, tlen = SM_TT / tsegs;
  // tasks are ordered by OUTPUT position so that the uint16 stores of a wavefront are contiguous
  // even in the permuted layout of the fast apply kernel (fast::perm_pos).
  for (int task = threadIdx.x; task < g.F * tsegs; task += blockDim.x) {
    const int pos = task % g.F, ts = task / g.F;
    const int f = perm ? fast::perm_inv(pos) : pos;
    const int r0 = nt + ts * tlen;  // LDS row of the first output frame of this segment
    auto at = [&](int r) -> int { return (r >= 0 && r < rows) ? (int)cf[(size_t)r * FP + f] : 0; };
    int c = 0, R = 0, L = 0;
    for (int b = -nt; b <= nt; ++b) c += (nt + 1 - (b < 0 ? -b : b)) * at(r0 + b);
    for (int j = 1; j <= nt + 1; ++j) R += at(r0 + j);
    for (int j = 0; j <= nt; ++j) L += at(r0 - j);
    for (int i = 0; i < tlen; ++i) {
      const int r = r0 + i;
      const int64_t t = t0 + ts * tlen + i;
      if (t < g.T) K[(u * g.T + t) * (int64_t)g.FS + pos] = (unsigned short)c;
      c += R - L;
      R += at(r + nt + 2) - at(r + 1);
      L += at(r + 1) - at(r - nt);
    }
  }
}

// Second-generation smoothing kernel (nf <= 30): phase 1 keeps a 128-bit sliding window of the
// bit row in registers (no LDS reads in the recurrence) and stores 4 counts per LDS write;
// phase 2 walks one output column per thread over the whole tile with batched LDS reads.
// Frames outside [t_begin, t_end) are neither read as outputs nor written.  The tile height tt (<= SM2_TT frames) is a
// launch parameter: long frames (n_fft = 4096: 2049 bins per row) take lower tiles so that the tile fits the LDS.
constexpr int SM2_TT = 64;
constexpr int SM2_THREADS = 576;
// (round 5) tallest tile tried for a row of F bins: short rows take tall tiles (fewer halo rows per output -- the time
// half-width of n_fft = 256 at 48 kHz is 37 frames: 64-frame tiles ran phase 1 over 2.2 x the rows) and split them into
// segments for phase 2
__host__ __device__ inline int smooth2_tt_max(int F) { return F <= 160 ? 256 : (F <= 320 ? 128 : SM2_TT); }

__device__ __forceinline__ unsigned long long funnel_r(unsigned long long lo, unsigned long long hi, int sh) {
  // bits [sh, sh+64) of the 128-bit value hi:lo, 0 <= sh < 64
  return sh == 0 ? lo : ((lo >> sh) | (hi << (64 - sh)));
}

__host__ __device__ inline int smooth2_pitch(int F) { return ((F + 3) & ~3) + 4 * ((F + 31) / 32) + 4; }
__host__ __device__ inline size_t smooth2_cf_bytes(int rows, int F, int ct_size) {
  size_t b = (size_t)rows * smooth2_pitch(F) * ct_size;
  return (b + 15) & ~(size_t)15;
}

template <typename CT>
__global__ __launch_bounds__(SM2_THREADS) void k_smooth_bits2(const unsigned long long* __restrict__ bits, Geom g,
                                                               int wpr, int nf, int nt,
                                                               unsigned short* __restrict__ K, int perm,
                                                               int64_t t_begin, int64_t t_end,
                                                               const unsigned long long* __restrict__ ftab, int tt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int rows = tt + 2 * nt;
  // phase-1 counts are stored at column f + 4*(f/32): the phase-2 column walk visits bins 32 apart
  // with consecutive lanes (lane order of the apply kernel), which would be an 8-way bank conflict
  // on a dense row
  const int FP = smooth2_pitch(g.F);
  const int WP = wpr + 2;  // one zero word on each side of every bit row
  CT* cf = reinterpret_cast<CT*>(smem);  // [rows + 2][FP]: two zero rows behind the tile (phase 2 reads them)
  unsigned long long* wb =
      reinterpret_cast<unsigned long long*>(smem + smooth2_cf_bytes(rows + 2, g.F, sizeof(CT)));
  for (int i = threadIdx.x; i < 2 * FP; i += SM2_THREADS) cf[(size_t)rows * FP + i] = (CT)0;
  const int64_t u = blockIdx.y;
  const int64_t t0 = t_begin + (int64_t)blockIdx.x * tt;
  for (int i = threadIdx.x; i < rows * WP; i += SM2_THREADS) {
    const int r = i / WP, w = i - r * WP - 1;
    const int64_t t = t0 - nt + r;
    unsigned long long word = 0ull;
    if (w >= 0 && w < wpr && t >= 0 && t < g.T) {
      word = bits[(u * g.T + t) * (int64_t)wpr + w];
      const int nvalid = g.F - 64 * w;  // clear bits beyond bin F-1 (they are unspecified)
      if (nvalid < 64) word &= (nvalid <= 0 ? 0ull : ((1ull << nvalid) - 1ull));
    }
    wb[i] = word;
  }
  __syncthreads();
  // ---- phase 1: along f ----------------------------------------------------------------
  if (ftab != nullptr && sizeof(CT) == 1) {
    // table form (8 + 2 nf <= 18 window bits): the counts of 8 adjacent bins are linear in the 18
    // window bits, so they are the sum of two 512-entry tables of 8 packed byte counts.
    unsigned long long* tab = wb + (size_t)rows * WP;  // [2][512] staged in LDS
    for (int i = threadIdx.x; i < 1024; i += SM2_THREADS) tab[i] = ftab[i];
    __syncthreads();
    const int ngrp = (g.F + 7) / 8;
    for (int task = threadIdx.x; task < rows * ngrp; task += SM2_THREADS) {
      const int r = task / ngrp, grp = task - r * ngrp;
      const int f = 8 * grp;
      const unsigned long long* rb = wb + (size_t)r * WP + 1;  // word w at rb[w], rb[-1] = 0
      const int start = f - nf + 64;                           // bit index in the stream that begins at rb[-1]
      const int wi = (start >> 6) - 1, sh = start & 63;
      const unsigned win = (unsigned)(funnel_r(rb[wi], rb[wi + 1], sh)) & 0x3ffffu;
      const unsigned long long cnt = tab[win & 511u] + tab[512 + (win >> 9)];
      CT* out = cf + (size_t)r * FP + f + ((f >> 5) << 2);
      *reinterpret_cast<unsigned*>(out) = (unsigned)cnt;
      *reinterpret_cast<unsigned*>(out + 4) = (unsigned)(cnt >> 32);
    }
    __syncthreads();
  } else {
  const unsigned long long m1 = (1ull << (nf + 1)) - 1ull;
  for (int task = threadIdx.x; task < rows * wpr; task += SM2_THREADS) {
    const int r = task / wpr, w = task - r * wpr;
    const unsigned long long* rb = wb + (size_t)r * WP + 1 + w;
    // 128-bit window starting at bin f0 - nf, f0 = 64 w
    unsigned long long lo = funnel_r(rb[-1], rb[0], 64 - nf);
    unsigned long long hi = funnel_r(rb[0], rb[1], 64 - nf);
    int c = 0;
    for (int i = 0; i <= 2 * nf; ++i) c += (nf + 1 - (i < nf ? nf - i : i - nf)) * (int)((lo >> i) & 1ull);
    int R = __popcll((lo >> (nf + 1)) & m1);
    int L = __popcll(lo & m1);
    CT* out = cf + (size_t)r * FP + 64 * w + 8 * w;  // column(f) = f + 4*(f/32)
    const int nb = min(64, g.F - 64 * w);
    for (int f4 = 0; f4 < nb; f4 += 4) {
      unsigned packed = 0;
      int vals[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vals[e] = c;
        packed |= ((unsigned)c & 0xffu) << (8 * e);
        c += R - L;
        const int bA = (int)((lo >> (nf + 1)) & 1ull);      // bit(f + 1)
        R += (int)((lo >> (2 * nf + 2)) & 1ull) - bA;       // + bit(f + nf + 2) - bit(f + 1)
        L += bA - (int)(lo & 1ull);                         // + bit(f + 1) - bit(f - nf)
        lo = (lo >> 1) | (hi << 63);
        hi >>= 1;
      }
      const int colo = f4 + ((f4 >> 5) << 2);
      if (sizeof(CT) == 1) {
        *reinterpret_cast<unsigned*>(out + colo) = packed;  // FP is a multiple of 4: rows stay aligned
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[colo + e] = (CT)vals[e];
      }
    }
  }
  __syncthreads();
  }
  // ---- phase 2: along t, one output position per thread ------------------------------------
  // The walk reads rows r - nt .. r + nt + 2 for r = nt .. nt + tt - 1: never below row 0, at most two
  // rows past the tile -- those two rows exist and are zero (cleared above), so no bounds checks
  // (they were scalar compares/selects per read: the CU's one scalar unit was the bottleneck).
  // (round 5) Short rows (F = 129 / 257: n_fft = 256 / 512) used to leave 447 / 319 of the 576 threads idle here: the tile's
  // frames are split into nseg = 576 / F segments, one (segment, position) per thread, each segment with its own start-up sum
  // (2 nt + 2 reads) -- integer arithmetic, so the counts do not depend on where a segment starts.
  const int nseg = g.F < SM2_THREADS ? max(1, SM2_THREADS / g.F) : 1;
  const int slen = (tt + nseg - 1) / nseg;
  for (int item = threadIdx.x; item < nseg * g.F; item += SM2_THREADS) {
    const int seg = nseg == 1 ? 0 : item / g.F;
    const int pos = item - seg * g.F;
    const int i0 = seg * slen;                           // first output frame of the segment (tile-local)
    const int f = perm ? fast::perm_inv(pos) : pos;
    const CT* col = cf + f + ((f >> 5) << 2) + (size_t)i0 * FP;
    int c = 0, R = 0, L = 0;
    for (int b = -nt; b <= nt + 1; ++b) {
      const int x = (int)col[(size_t)(nt + b) * FP];
      if (b <= nt) c += (nt + 1 - (b < 0 ? -b : b)) * x;
      if (b >= 1) R += x;
      if (b <= 0) L += x;
    }
    unsigned short* kout = K + (u * g.T + t0 + i0) * (int64_t)g.FS + pos;
    const int n_tile = (int)min<int64_t>(tt, min<int64_t>(t_end, g.T) - t0);
    const int n_out = min(slen, n_tile - i0);
    const CT* pa = col + (size_t)(2 * nt + 2) * FP;  // row r + nt + 2
    const CT* pb = col + (size_t)(nt + 1) * FP;      // row r + 1
    const CT* pc = col;                              // row r - nt
#pragma unroll 4
    for (int i = 0; i < n_out; ++i) {
      const int xa = (int)*pa, xb = (int)*pb, xc = (int)*pc;
      pa += FP; pb += FP; pc += FP;
      *kout = (unsigned short)c;
      kout += g.FS;
      c += R - L;
      R += xa - xb;
      L += xb - xc;
    }
  }
}

// no smoothing: K = bit (ktot = 1)
__global__ void k_bits_to_k16(const unsigned long long* __restrict__ bits, Geom g, int wpr,
                              unsigned short* __restrict__ K, int64_t n_units, int perm) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i % g.FS);
    if (pos >= g.F) continue;
    const int f = perm ? fast::perm_inv(pos) : pos;
    const int64_t ut = i / g.FS;
    K[i] = (unsigned short)((bits[ut * wpr + (f >> 6)] >> (f & 63)) & 1ull);
  }
}

// M[t][f] from K:  p * K/ktot + (1-p) * edge(f, t)  (edge = valid-tap weight fraction when the
// reference applies prop_decrease before smoothing, else 1).  Written as float for the v1
// apply kernel (general geometries); the fast apply kernel evaluates this on the fly.
// (round 5) eight cells per thread -- FS is a multiple of 16, so a group of eight never straddles a frame: one 16-byte load,
// two 16-byte stores and ONE division per group (the first version divided twice per cell in 64 bits and took 194 us for two
// minutes of audio at n_fft = 256: 15 M cells, the longest kernel of that call).
__global__ void k_k16_to_mask(const unsigned short* __restrict__ K, Geom g, int nf, int nt, float inv_ktot,
                              float p, int prop_before, int smooth, float* __restrict__ M, int64_t n_units) {
  const int gpr = g.FS >> 3;                              // groups per frame
  const int64_t groups = n_units * g.T * gpr;
  const bool edge_live = prop_before && smooth;
  const float q = 1.0f - p;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t rowi;
    int gi;
    if (groups <= 0x7fffffffll) {                         // (uniform) 32-bit division
      const unsigned r = (unsigned)i / (unsigned)gpr;
      rowi = r;
      gi = (int)((unsigned)i - r * (unsigned)gpr);
    } else {
      rowi = i / gpr;
      gi = (int)(i - rowi * gpr);
    }
    const int f0 = gi * 8;
    if (f0 >= g.F) continue;                              // padding group of the frame
    const uint4 kk = *reinterpret_cast<const uint4*>(K + i * 8);
    const unsigned kw[4] = {kk.x, kk.y, kk.z, kk.w};
    float out[8];
    float et = 1.0f;
    if (edge_live) {
      // integer triangle sums over the valid taps
      const int64_t t = rowi % g.T;
      const int64_t tlo = max<int64_t>(-nt, -t), thi = min<int64_t>(nt, g.T - 1 - t);
      int s = 0;
      for (int a = (int)tlo; a <= (int)thi; ++a) s += nt + 1 - (a < 0 ? -a : a);
      et = (float)s;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = f0 + e;
      const float kv = (float)((kw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
      float edge = 1.0f;
      if (edge_live) {
        const int flo = max(-nf, -f), fhi = min(nf, g.F - 1 - f);
        int s = 0;
        for (int a = flo; a <= fhi; ++a) s += nf + 1 - (a < 0 ? -a : a);
        edge = (float)s * et * inv_ktot;
      }
      out[e] = p * (kv * inv_ktot) + q * edge;
    }
    float* dst = M + i * 8;
    if (f0 + 8 <= g.F) {
      *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(out[4], out[5], out[6], out[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (f0 + e < g.F) dst[e] = out[e];
    }
  }
}


// K stored in the lane order of the fused apply kernel -> float mask in natural bin order
// (full reduction: mask = K / ktot); used when a training step needs the mask for the adjoint.
__global__ void k_k16_to_mask_perm(const unsigned short* __restrict__ K, Geom g, float inv_ktot,
                                   float* __restrict__ M, int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i % g.FS);
    if (pos >= g.F) continue;
    M[i - pos + fast::perm_inv(pos)] = (float)K[i] * inv_ktot;
  }
}

}  // namespace sg
