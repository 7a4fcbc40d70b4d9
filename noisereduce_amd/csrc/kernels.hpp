// Device kernels of the spectral-gating path (v1: every field materialised in HBM).
//
// Layout of all time-frequency fields: [unit][frame t][FS] with FS = round_up(F, 16),
// bin index contiguous -- one STFT frame is one contiguous row, so the FFT kernels
// store/load rows with lane-contiguous (coalesced) accesses and the time recurrences
// walk rows with lanes = bins.
//
// A "unit" is one independently filtered signal window: a (channel, chunk) pair of the
// reference's chunk grid (base.py:144-156) or one batch row of TorchGate.forward.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_wave.hpp"
#include "geom.hpp"

namespace sg {

// (load_sample, view_sample, nanmax, frame_ptr_f32, store_sample, cell_db: geom.hpp -- shared with the translation units that
// hold kernel templates only)

// The readable part [lo, hi) of every row as float32 -- what every float32 transform kernel makes of a sample
// anyway ((float)sample): non-float32 recordings are converted ONCE instead of per frame and kernel on the
// checked per-sample path.  out[r * (hi - lo) + i] = (float)x[r * stride + lo + i].
__global__ void k_to_f32(const void* __restrict__ x, int dtype, int64_t stride, int64_t lo, int64_t len, int64_t rows,
                         float* __restrict__ out) {
  const int64_t n = rows * len;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / len, j = i - r * len;
    out[i] = (float)load_sample(x, dtype, r * stride + lo + j);
  }
}

// ---------------------------------------------------------------------------------------
// Forward STFT.  One wavefront per frame, FPW frames per wave, WAVES waves per block.
//   TC = double: stores the power |X|^2 (float64) -- the decision-critical quantity
//                (the stationary mask is a hard compare).
//   TC = float : stores the magnitude |X| (float32) for the non-stationary masks.
// X is the UNSCALED transform of window * frame (scipy's 1/sum(w) is applied by consumers).
// Optionally dumps X itself (float64 pairs, [u][t][F]) for the stage tap sg_stft.
// ---------------------------------------------------------------------------------------
template <typename TC, int N, int WAVES, int FPW, int NT = 64>
__global__ __launch_bounds__(WAVES * NT) void k_stft(View view, Geom g, const cx<TC>* __restrict__ tw_g,
                                                     const TC* __restrict__ wfull,
                                                     double* __restrict__ P_out, float* __restrict__ mag_out,
                                                     double* __restrict__ z_out, double z_scale,
                                                     unsigned long long* __restrict__ pmax_bits) {
  constexpr int SY = NT <= 64 ? 1 : NT;   // a team of <= 64 lanes is (part of) one wavefront and its buffer is its own
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cx<TC>* tw = reinterpret_cast<cx<TC>*>(smem);
  cx<TC>* bufs = tw + N;
  const int lane = threadIdx.x % NT;  // thread within the frame's team (NT = 64: one wavefront)
  const int wave = threadIdx.x / NT;
  cx<TC>* buf = bufs + wave * lpn<TC>(N);
  stage_twiddles<WAVES * NT, N>(tw, tw_g, (int)threadIdx.x);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  double vmax[N / NT + 1];  // running max power of this lane's bins (pmax_bits != nullptr)
#pragma unroll
  for (int m = 0; m <= N / NT; ++m) vmax[m] = 0.0;
  for (int fi = 0; fi < FPW; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * FPW + fi) * WAVES + wave;
    const bool valid = t < g.T;
    // gather window * frame as complex pairs (x[2j], x[2j+1])
    const int64_t s0 = t * g.H - g.padL;
    const float* fp = valid ? frame_ptr_f32(view, row, chunk, s0, 2 * N) : nullptr;  // team-uniform
    if (fp) {
      for (int j = lane; j < N; j += NT)
        buf[lp<TC>(j)] = {(TC)fp[2 * j] * wfull[2 * j], (TC)fp[2 * j + 1] * wfull[2 * j + 1]};
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<TC> z = {(TC)0, (TC)0};
        if (valid) {
          z.x = (TC)view_sample(view, row, chunk, s0 + 2 * j) * wfull[2 * j];
          z.y = (TC)view_sample(view, row, chunk, s0 + 2 * j + 1) * wfull[2 * j + 1];
        }
        buf[lp<TC>(j)] = z;
      }
    }
    team_sync<SY>();
    wave_fft<TC, N, false, NT, SY>(buf, tw, lane);
    if (valid) {
      const int64_t rowoff = (u * g.T + t) * g.FS;
#pragma unroll
      for (int m = 0; m <= N / NT; ++m) {
        const int k = lane + NT * m;
        if (k > N) continue;
        cx<TC> a = buf[lp<TC>(k == N ? 0 : k)];
        cx<TC> b = buf[lp<TC>((k == 0 || k == N) ? 0 : N - k)];
        cx<TC> w = tw[k == N ? 0 : k];
        cx<TC> X = rfft_bin(a, b, w, k, N);
        const double Pk = (double)X.x * (double)X.x + (double)X.y * (double)X.y;
        vmax[m] = nanmax(vmax[m], Pk);
        if (P_out) P_out[rowoff + k] = Pk;
        if (mag_out) mag_out[rowoff + k] = sqrtf((float)(X.x * X.x + X.y * X.y));
        if (z_out) {
          int64_t zo = ((u * g.T + t) * g.F + k) * 2;
          z_out[zo] = (double)X.x * z_scale;
          z_out[zo + 1] = (double)X.y * z_scale;
        }
      }
    }
    team_sync<SY>();
  }
  // per-(unit, band) max power, order-independent (non-negative doubles order like their bits)
  if (pmax_bits) {
#pragma unroll
    for (int m = 0; m <= N / NT; ++m) {
      const int k = lane + NT * m;
      if (k <= N) atomicMax(&pmax_bits[u * g.FS + k], (unsigned long long)__double_as_longlong(vmax[m]));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Apply + inverse: frame -> FFT (fp32) -> X * M[t][k] -> inverse FFT -> * window / N
// -> windowed segment seg[u][t][0..n) ready for overlap-add.  `adjoint` swaps nothing here:
// the backward pass of TorchGate uses the same kernel on grad_out frames (the operator
// frame -> window -> rfft -> mask -> irfft -> window is symmetric up to the Hermitian weights,
// handled by the caller through the window/normalisation tables).
// ---------------------------------------------------------------------------------------
template <int N, int WAVES, int FPW, int NT = 64>
__global__ __launch_bounds__(WAVES * NT) void k_apply_istft(View view, Geom g, const cx<float>* __restrict__ tw_g,
                                                            const float* __restrict__ win_a,  // analysis window (n)
                                                            const float* __restrict__ win_s,  // synthesis window (n), incl. 1/N
                                                            const float* __restrict__ M, float* __restrict__ seg,
                                                            // (round 5) K16 != nullptr: the mask is K16 * kscale, the integer weight
                                                            // sums of the smoothed bit mask read as they are (no float mask field)
                                                            const unsigned short* __restrict__ K16 = nullptr,
                                                            float kscale = 0.f) {
  constexpr int SY = NT <= 64 ? 1 : NT;   // (as in k_stft)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cx<float>* tw = reinterpret_cast<cx<float>*>(smem);
  cx<float>* bufs = tw + N;
  const int lane = threadIdx.x % NT;  // thread within the frame's team (NT = 64: one wavefront)
  const int wave = threadIdx.x / NT;
  cx<float>* buf = bufs + wave * lpn<float>(N);
  stage_twiddles<WAVES * NT, N>(tw, tw_g, (int)threadIdx.x);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  for (int fi = 0; fi < FPW; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * FPW + fi) * WAVES + wave;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    const float* fp = valid ? frame_ptr_f32(view, row, chunk, s0, 2 * N) : nullptr;  // team-uniform
    if (fp) {
      for (int j = lane; j < N; j += NT)
        buf[lp<float>(j)] = {fp[2 * j] * win_a[2 * j], fp[2 * j + 1] * win_a[2 * j + 1]};
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<float> z = {0.f, 0.f};
        if (valid) {
          z.x = (float)view_sample(view, row, chunk, s0 + 2 * j) * win_a[2 * j];
          z.y = (float)view_sample(view, row, chunk, s0 + 2 * j + 1) * win_a[2 * j + 1];
        }
        buf[lp<float>(j)] = z;
      }
    }
    team_sync<SY>();
    wave_fft<float, N, false, NT, SY>(buf, tw, lane);
    // split -> mask -> merge, pairwise in place: task k handles bins k and N-k.
    if (valid) {
      const float* Mrow = M + (u * g.T + t) * g.FS;
      const unsigned short* Krow = K16 + (u * g.T + t) * g.FS;
      auto mask_at = [&](int k) -> float { return K16 ? (float)Krow[k] * kscale : Mrow[k]; };
      for (int k = lane; k <= N / 2; k += NT) {
        if (k == 0) {
          cx<float> a = buf[lp<float>(0)];
          float y0 = (a.x + a.y) * mask_at(0);
          float yN = (a.x - a.y) * mask_at(N);
          buf[lp<float>(0)] = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
        } else {
          cx<float> a = buf[lp<float>(k)], b = buf[lp<float>(N - k)];
          cx<float> w = tw[k];
          // X[k] = E + w O ; X[N-k] = conj(E) - conj(w) conj(O)
          cx<float> E = {(a.x + b.x) * 0.5f, (a.y - b.y) * 0.5f};
          cx<float> O = {(a.y + b.y) * 0.5f, (b.x - a.x) * 0.5f};
          cx<float> wO = cmul(w, O);
          float mk = mask_at(k), mn = mask_at(N - k);
          cx<float> Yk = {(E.x + wO.x) * mk, (E.y + wO.y) * mk};
          cx<float> Yn = {(E.x - wO.x) * mn, (-E.y + wO.y) * mn};  // X[N-k] * mn
          // merge: E' = (Yk + conj Yn)/2 ; O' = (Yk - conj Yn)/2 * conj(w) ; Zc'[k] = E' + i O'
          cx<float> Ep = {(Yk.x + Yn.x) * 0.5f, (Yk.y - Yn.y) * 0.5f};
          cx<float> D = {(Yk.x - Yn.x) * 0.5f, (Yk.y + Yn.y) * 0.5f};
          cx<float> wc = {w.x, -w.y};
          cx<float> Op = cmul(D, wc);
          buf[lp<float>(k)] = {Ep.x - Op.y, Ep.y + Op.x};
          if (k != N - k) {
            // Zc'[N-k] = conj(E') + i conj(O')  (E', O' are spectra of real sequences)
            buf[lp<float>(N - k)] = {Ep.x + Op.y, -Ep.y + Op.x};
          }
        }
      }
    }
    team_sync<SY>();
    wave_fft<float, N, true, NT, SY>(buf, tw, lane);
    if (valid) {
      float2* srow = reinterpret_cast<float2*>(seg + (u * g.T + t) * (int64_t)g.n);
      for (int j = lane; j < N; j += NT) {
        cx<float> z = buf[lp<float>(j)];
        srow[j] = make_float2(z.x * win_s[2 * j], z.y * win_s[2 * j + 1]);
      }
    }
    team_sync<SY>();
  }
}

// ---------------------------------------------------------------------------------------
// Overlap-add gather (scipy/_spectral_py.py:1708-1725; torch.istft): one thread per kept
// output sample.  out[p] = sum_t seg[t][e - tH] / sum_t w^2[e - tH],  e = p + padL.
// ---------------------------------------------------------------------------------------
struct OutMap {
  void* out;
  int dtype;
  int64_t stride;   // elements between rows
  int64_t p0, p1;   // kept range of unit-local sample positions [p0, p1)
  int64_t g_step;   // destination index of position p of chunk i: i*g_step + (p - p0) - g0
  int64_t g0;       // destination offset subtracted (start_frame)
  int64_t g_lo, g_hi;  // keep only destination indices (before -g0) in [g_lo, g_hi)
};

__global__ void k_ola(View view, Geom g, OutMap om, const float* __restrict__ seg,
                      const float* __restrict__ wsq /* analysis*synthesis window product (n) */,
                      int normalize) {
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  const int64_t p = om.p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= om.p1) return;
  const int64_t gi = chunk * om.g_step + (p - om.p0);
  if (gi < om.g_lo || gi >= om.g_hi) return;
  float val = 0.f;
  if (p < g.Lout) {
    const int64_t e = p + g.padL;
    int64_t t_hi = e / g.H;
    if (t_hi > g.T - 1) t_hi = g.T - 1;
    int64_t t_lo = (e - g.n + g.H) / g.H;  // ceil((e - n + 1) / H)
    if (e - g.n + 1 <= 0) t_lo = 0;
    float acc = 0.f, norm = 0.f;
    for (int64_t t = t_lo; t <= t_hi; ++t) {
      int m = (int)(e - t * g.H);
      acc += seg[(u * g.T + t) * (int64_t)g.n + m];
      norm += wsq[m];
    }
    val = normalize ? acc / (norm > 1e-10f ? norm : 1.0f) : acc;
  }
  store_sample(om.out, om.dtype, row * om.stride + gi - om.g0, val);
}

// gq[b][p] = g[b][p] / env(p), p < Lout (0 beyond): first step of the adjoint of the ISTFT
// normalisation (env = sum of window^2 over the frames covering p, guarded like the forward).
__global__ void k_env_scale(const void* __restrict__ gsrc, int dtype, int64_t stride, Geom g,
                            const float* __restrict__ wsq, float* __restrict__ gq, int64_t Lq) {
  const int64_t b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Lq) return;
  float val = 0.f;
  if (p < g.Lout) {
    const int64_t e = p + g.padL;
    int64_t t_hi = e / g.H;
    if (t_hi > g.T - 1) t_hi = g.T - 1;
    int64_t t_lo = (e - g.n + g.H) / g.H;
    if (e - g.n + 1 <= 0) t_lo = 0;
    float norm = 0.f;
    for (int64_t t = t_lo; t <= t_hi; ++t) norm += wsq[(int)(e - t * g.H)];
    val = (float)load_sample(gsrc, dtype, b * stride + p) / (norm > 1e-10f ? norm : 1.0f);
  }
  gq[b * Lq + p] = val;
}

// ---------------------------------------------------------------------------------------
// Per-band statistics over time (lanes = bins).  block = 64 bins x TG time groups.
// ---------------------------------------------------------------------------------------
constexpr int STAT_TG = 4;

// cell_db without the library logarithm and square root, for kernels whose time IS those two (k_row_decide: 8 M cells
// x ~110 float64 operations).  With y = sqrt(P) s:  20 log10(y + eps) = 10 log10(2) log2(P) + 20 log10(s) +
// (20 / ln 10) log1p(eps / y), and
//   log2(P)  = exponent + log2(m), m in [1, 2): table of 128 centres c_i (top 7 mantissa bits): r = m / c_i - 1 as ONE
//              fma with the rounded reciprocal t_i = rd(1 / c_i), |r| <= 2^-8; log2(m) = -log2(t_i) [tabulated for the
//              ROUNDED t_i, so the split is exact] + log2(1 + r) [degree-6 series, next term < 2e-18];
//   log1p(x) = x for x = eps / y < 2^-27 (error x^2 / 2 < 3e-17), y from the hardware reciprocal square root (its
//              2^-26 relative accuracy is ample for a term this small).
// Cells outside that range (y <= eps 2^27: silence; non-finite) take the plain formula.  Agreement with cell_db:
// a few 1e-14 dB, the size of cell_db's own rounding.
struct DbFast {
  const double* tab;   // [128][2] = {t_i, -log2(t_i)} (device; staged in LDS by the kernel)
  double db0;          // 20 log10(mag_scale)
  double pmin;         // (eps 2^27 / mag_scale)^2
  double eps_s;        // eps / mag_scale
};
__device__ __forceinline__ double db_fast(double P, const double* __restrict__ s_tab, const DbFast& k, double mag_scale) {
  if (!(P > k.pmin && P < 1e300)) return cell_db(P, mag_scale);
  const long long b = __double_as_longlong(P);
  const int e = (int)(b >> 52) - 1023;
  const int i = (int)(b >> 45) & 127;
  const double m = __longlong_as_double((b & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
  const double2 t = reinterpret_cast<const double2*>(s_tab)[i];
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, -1.0 / 6.0, 0.2);
  p = fma(r, p, -0.25);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  p = fma(r, p, 1.0);
  const double l2 = ((double)e + t.y) + (r * p) * 1.4426950408889634074;   // 1 / ln 2
  const double rho = k.eps_s * __builtin_amdgcn_rsq(P);
  return fma(3.0102999566398119521, l2, k.db0) + 8.6858896380650365530 * rho;   // 10 log10(2); 20 / ln 10
}

// Time is split into gridDim.z slices (deterministic two-stage reduction: partials, then a
// fixed-order final sum) so that a single unit (the noise clip) still fills the chip.
__global__ __launch_bounds__(64 * STAT_TG) void k_colmax(const double* __restrict__ P, Geom g,
                                                         double* __restrict__ pmax_part) {
  __shared__ double red[STAT_TG][64];
  const int f = blockIdx.x * 64 + (threadIdx.x & 63);
  const int tg = threadIdx.x >> 6;
  const int64_t u = blockIdx.y;
  const int ts = blockIdx.z, nts = gridDim.z;
  const int64_t tb = g.T * ts / nts, te = g.T * (ts + 1) / nts;
  double m = 0.0;
  if (f < g.F)
    for (int64_t t = tb + tg; t < te; t += STAT_TG) m = nanmax(m, P[(u * g.T + t) * g.FS + f]);
  red[tg][threadIdx.x & 63] = m;
  __syncthreads();
  if (tg == 0 && f < g.F) {
    for (int i = 1; i < STAT_TG; ++i) m = nanmax(m, red[i][threadIdx.x & 63]);
    pmax_part[(u * nts + ts) * g.FS + f] = m;
  }
}

// pmax[u][f] = max over the slices (also the input of k_decide)
__global__ void k_colmax_final(const double* __restrict__ pmax_part, Geom g, int nts, double* __restrict__ pmax,
                               int64_t n_units) {
  const int64_t n = n_units * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = i / g.FS;
    const int f = (int)(i % g.FS);
    double m = 0.0;
    if (f < g.F) {
#pragma unroll 8
      for (int ts = 0; ts < nts; ++ts) m = nanmax(m, pmax_part[(u * nts + ts) * g.FS + f]);
    }
    pmax[i] = m;
  }
}

// partial sums of dBfl - rowmax_dB and its square, dBfl = max(dB, rowmax_dB - top_db)
__global__ __launch_bounds__(64 * STAT_TG) void k_colstats(const double* __restrict__ P, Geom g,
                                                           const double* __restrict__ pmax, double mag_scale,
                                                           double top_db, double* __restrict__ s_part) {
  __shared__ double r1[STAT_TG][64], r2[STAT_TG][64];
  const int l = threadIdx.x & 63;
  const int f = blockIdx.x * 64 + l;
  const int tg = threadIdx.x >> 6;
  const int64_t u = blockIdx.y;
  const int ts = blockIdx.z, nts = gridDim.z;
  const int64_t tb = g.T * ts / nts, te = g.T * (ts + 1) / nts;
  double s1 = 0.0, s2 = 0.0;
  if (f < g.F) {
    const double mdb = cell_db(pmax[u * g.FS + f], mag_scale);
    for (int64_t t = tb + tg; t < te; t += STAT_TG) {
      double d = cell_db(P[(u * g.T + t) * g.FS + f], mag_scale) - mdb;  // <= 0
      d = (d != d) ? d : fmax(d, -top_db);   // (fmax would drop a NaN; numpy / torch keep it)
      s1 += d;
      s2 += d * d;
    }
  }
  r1[tg][l] = s1;
  r2[tg][l] = s2;
  __syncthreads();
  if (tg == 0 && f < g.F) {
    for (int i = 1; i < STAT_TG; ++i) {
      s1 += r1[i][l];
      s2 += r2[i][l];
    }
    s_part[((u * nts + ts) * 2 + 0) * g.FS + f] = s1;
    s_part[((u * nts + ts) * 2 + 1) * g.FS + f] = s2;
  }
}

// ---------------------------------------------------------------------------------------
// Single-pass column statistics (few units: the noise clip).  The -top_db floor max(dB, rowmax - top_db)
// needs the band maximum BEFORE the moments -- two passes over the power field and four launches
// (k_colmax, k_colmax_final, k_colstats, k_colstats_final).  But only a band's very smallest cells can lie
// more than top_db below its maximum (in practice a handful of cells of the real-valued DC / Nyquist bins):
// one pass gathers max, the TWO smallest powers of every time slice and the moments of the UNfloored dB about
// a pivot (the band's first frame); the final kernel replaces the contribution of the cells that turn out to
// be floored.  If both tracked minima of some slice are floored there may be a third: that band (rare) is
// recomputed exactly by its whole wave.
// ---------------------------------------------------------------------------------------
constexpr int STAT1_NP = 5;  // partials per (slice, band): max, min1, min2, s1, s2
#ifndef STAT1_BATCH
#define STAT1_BATCH 16
#endif

__device__ __forceinline__ void min2_push(double& m1, double& m2, double x) {  // keep the two smallest
  const double lo = fmin(m1, x), hi = fmax(m1, x);
  m1 = lo;
  m2 = fmin(m2, hi);
}

__global__ __launch_bounds__(64 * STAT_TG) void k_colstats1(const double* __restrict__ P, Geom g, double mag_scale,
                                                            double* __restrict__ part /* [u][nts][NP][FS] */,
                                                            DbFast dbk) {
  __shared__ double r[STAT1_NP][STAT_TG][64];
  __shared__ __attribute__((aligned(16))) double s_tab[256];
  s_tab[threadIdx.x] = dbk.tab[threadIdx.x];   // 64 * STAT_TG = 256 threads
  __syncthreads();
  const int l = threadIdx.x & 63;
  const int f = blockIdx.x * 64 + l;
  const int tg = threadIdx.x >> 6;
  const int64_t u = blockIdx.y;
  const int ts = blockIdx.z, nts = gridDim.z;
  const int64_t tb = g.T * ts / nts, te = g.T * (ts + 1) / nts;
  double mx = 0.0, m1 = 1e300, m2 = 1e300, s1 = 0.0, s2 = 0.0;
  if (f < g.F) {
    // the thread's cells in batches of 8 independent loads (the pivot cell rides in the first batch): the pass is a
    // chain of memory round trips, not a stream
    const double* col = P + (u * g.T) * g.FS + f;
    const double p0 = col[0];
    double pivot = 0.0;
    constexpr int NB = STAT1_BATCH;   // cells in flight per thread
    for (int64_t t0 = tb + tg; t0 < te; t0 += NB * STAT_TG) {
      double pv[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int64_t t = t0 + q * STAT_TG;
        pv[q] = t < te ? col[t * g.FS] : 0.0;
      }
      if (t0 == tb + tg) pivot = cell_db(p0, mag_scale);   // (plain formula: k_colstats1_final recomputes it)
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        if (t0 + q * STAT_TG < te) {
          mx = fmax(mx, pv[q]);
          min2_push(m1, m2, pv[q]);
          const double d = db_fast(pv[q], s_tab, dbk, mag_scale) - pivot;
          s1 += d;
          s2 += d * d;
        }
      }
    }
  }
  r[0][tg][l] = mx; r[1][tg][l] = m1; r[2][tg][l] = m2; r[3][tg][l] = s1; r[4][tg][l] = s2;
  __syncthreads();
  if (tg == 0 && f < g.F) {
    for (int i = 1; i < STAT_TG; ++i) {
      mx = fmax(mx, r[0][i][l]);
      min2_push(m1, m2, r[1][i][l]);
      min2_push(m1, m2, r[2][i][l]);
      s1 += r[3][i][l];
      s2 += r[4][i][l];
    }
    double* o = part + ((u * nts + ts) * STAT1_NP) * (int64_t)g.FS + f;
    o[0] = mx; o[g.FS] = m1; o[2 * g.FS] = m2; o[3 * g.FS] = s1; o[4 * g.FS] = s2;
  }
}

// one workgroup = 64 bands of one unit; the STAT_TG thread groups split the slices
constexpr int STAT1_MAXS = 16;  // slices per thread group (nts <= STAT_TG * STAT1_MAXS).  (round 6: 32 -- 128 slices -- measured: k_colstats1
                                // 13.7 -> 10.1 us at n_fft = 256, 8.4 -> 7.5 at 1024, but k_colstats1_final 11 - 13 -> 16.6 us everywhere: reverted)
// gc (variant S noise statistics, unit 0 only): the stationary gate's compare constants ride along instead of taking
// a launch of their own (k_prep_thresh_lazy, fused.hpp: same arithmetic) -- T2[f] per band, and per BAND BLOCK the floor
// test's bound on max|x| from that block's minimum threshold: alim_b[block].  The gate takes the minimum of the blocks'
// bounds itself (the bound is monotone in the threshold), so no block waits for another here.
struct GateConsts {
  double* T2;           // [F] or nullptr
  unsigned* alim_b;     // [band blocks] bit patterns of the bounds
  double sum_abs_w;
};
__global__ __launch_bounds__(64 * STAT_TG) void k_colstats1_final(const double* __restrict__ part,
                                                                  const double* __restrict__ P, Geom g, int nts,
                                                                  double mag_scale, double top_db, double n_std,
                                                                  int ddof, double* __restrict__ pmax,
                                                                  double* __restrict__ thresh, GateConsts gc) {
  __shared__ double r[3][STAT_TG][64];
  const int l = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + l;
  const int64_t u = blockIdx.y;
  const int64_t i = u * g.FS + f;
  const bool live = f < g.F;
  const double Tn = (double)g.T;
  double mx = 0.0, s1 = 0.0, s2 = 0.0;
  const double p0 = live ? P[(u * g.T) * g.FS + f] : 1.0;   // pivot cell: loaded with the partials, one round trip
  double m1[STAT1_MAXS];
  {
    // (round 6) every slice's four partials loaded UNCONDITIONALLY from a clamped address, selected afterwards: behind
    // `if (live && ts < nts)` the loads of one slice waited for the previous slice's -- sixteen dependent round trips
    const int fc = live ? f : g.F - 1;
    double a0[STAT1_MAXS], a3[STAT1_MAXS], a4[STAT1_MAXS];
#pragma unroll
    for (int k = 0; k < STAT1_MAXS; ++k) {
      const int ts = tg + STAT_TG * k;
      const double* o = part + ((u * nts + (ts < nts ? ts : nts - 1)) * STAT1_NP) * (int64_t)g.FS + fc;
      a0[k] = o[0];
      m1[k] = o[g.FS];
      a3[k] = o[3 * g.FS];
      a4[k] = o[4 * g.FS];
    }
#pragma unroll
    for (int k = 0; k < STAT1_MAXS; ++k) {
      const bool on = live && tg + STAT_TG * k < nts;
      if (on) {
        mx = fmax(mx, a0[k]);
        s1 += a3[k];
        s2 += a4[k];
      } else {
        m1[k] = 1e300;
      }
    }
  }
  r[0][tg][l] = mx;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < STAT_TG; ++k) mx = fmax(mx, r[0][k][l]);
  const double mdb = cell_db(mx, mag_scale);
  const double pivot = live ? cell_db(p0, mag_scale) : 0.0;
  {
    // cells below the floor: replace d by the floored value (both about the pivot).
    // dB < mdb - top_db  <=>  P < Pfl (cell_db is monotone; at the boundary both forms of the cell give the
    // same floored value): the logarithm is only evaluated for cells that ARE floored
    const double dfl = (mdb - top_db) - pivot;
    const double am = (exp10((mdb - top_db) / 20.0) - 2.220446049250313e-16) / mag_scale;
    const double Pfl = am > 0.0 ? am * am : 0.0;
    // (hot path: compares only.  The corrections sit in a ROLLED loop that re-reads the two minima -- unrolled with
    // its two inlined logarithms per slice the kernel was 40 KB of straight-line code, fetched cold on every call)
    // (round 5) Which of the thread's slices have a floored minimum: a bit per slice, OR-ed over the wavefront; the loop
    // below only visits those (typically one or two slices of the DC / Nyquist bands; it used to walk all sixteen with a
    // dependent load each whenever any lane of the wavefront had one -- the two workgroups that hold those bands set the
    // kernel's duration)
    unsigned km = 0u;
#pragma unroll
    for (int k = 0; k < STAT1_MAXS; ++k) km |= (m1[k] < Pfl ? 1u : 0u) << k;
    unsigned wor = km;
    for (int off = 32; off > 0; off >>= 1) wor |= (unsigned)__shfl_xor((int)wor, off);
    {
      while (wor) {                            // wave-uniform
        const int k = __ffs((int)wor) - 1;
        wor &= wor - 1u;
        const int ts = tg + STAT_TG * k;     // (< nts: slices beyond it never set a bit)
        bool both = false;
        if ((km >> k) & 1u) {
          const double* o = part + ((u * nts + ts) * STAT1_NP) * (int64_t)g.FS + f;
          const double a1 = o[g.FS], a2 = o[2 * g.FS];
          both = a1 < Pfl && a2 < Pfl;     // a third floored cell of this slice would go unseen: rescan the slice
          if (a1 < Pfl && !both) {
            const double d = cell_db(a1, mag_scale) - pivot;
            if (d < dfl) {
              s1 += dfl - d;
              s2 += dfl * dfl - d * d;
            }
          }
        }
        // (round 5) rare: a slice whose two tracked minima are both under the floor.  The wavefront scans that slice of
        // that band for EVERY cell under the floor (compares; a logarithm per floored cell) and the band's lane takes the
        // correction -- until round 5 such a band was recomputed whole, one logarithm per cell of all T frames by one
        // wavefront: 40 us of a 12 us kernel whenever a recording's DC / Nyquist band had such a slice (the benchmark
        // recording at n_fft = 256: 0.234 instead of 0.157 ms per call).
        unsigned long long todo = __ballot(both);
        while (todo) {
          const int src = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int fb = blockIdx.x * 64 + src;
          const double Pfl_s = __shfl(Pfl, src), piv_s = __shfl(pivot, src), dfl_s = __shfl(dfl, src);
          const int64_t tb = g.T * ts / nts, te = g.T * (ts + 1) / nts;
          double c1 = 0.0, c2 = 0.0;
          for (int64_t t0 = tb + l; t0 < te; t0 += 64 * 4) {
            double pv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int64_t t = t0 + 64 * q;
              pv[q] = t < te ? P[(u * g.T + t) * g.FS + fb] : 1e300;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (pv[q] < Pfl_s) {
                const double d = cell_db(pv[q], mag_scale) - piv_s;
                if (d < dfl_s) {
                  c1 += dfl_s - d;
                  c2 += dfl_s * dfl_s - d * d;
                }
              }
          }
          for (int off = 32; off > 0; off >>= 1) {
            c1 += __shfl_xor(c1, off);
            c2 += __shfl_xor(c2, off);
          }
          if (l == src) {
            s1 += c1;
            s2 += c2;
          }
        }
      }
    }
  }
  __syncthreads();
  r[1][tg][l] = s1; r[2][tg][l] = s2;
  __syncthreads();
  if (tg != 0) return;
  double thr_out = (double)NAN;
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int k = 0; k < STAT_TG; ++k) {  // fixed order: deterministic
    s1 += r[1][k][l];
    s2 += r[2][k][l];
  }
  if (live) {
    pmax[i] = mx;
    double var = (s2 - s1 * s1 / Tn) / (Tn - (double)ddof);
    if (var < 0.0) var = 0.0;
    thr_out = (pivot + s1 / Tn) + sqrt(var) * n_std;
    thresh[i] = thr_out;
  } else if (f < g.FS) {
    pmax[i] = 0.0;
  }
  if (gc.T2 == nullptr || u != 0) return;
  // ---- the gate's compare constants for these 64 bands (wave 0 holds their thresholds) ----
  const double eps = 2.220446049250313e-16;
  double mn = 1e300;
  if (live) {
    const double zero_db = 20.0 * log10(eps);
    double t2;
    if (thr_out != thr_out) {
      t2 = 1e300;   // T2_NEVER (fastpath.hpp): NaN threshold, no cell of the band passes
    } else if (zero_db > thr_out) {
      t2 = -1.0;    // a zero cell already passes: every cell does
    } else {
      const double tm = (exp10(thr_out / 20.0) - eps) / mag_scale;
      t2 = tm > 0.0 ? tm * tm : 0.0;
    }
    gc.T2[f] = t2;
    mn = fmin(mn, thr_out);
  }
  for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off));
  if (l == 0) {
    // need  <=>  20 log10(max|x| sum|w| mag_scale + eps) + 1e-6 - top_db > min thresh   <=>   max|x| > lim
    const double lim = (exp10((mn + top_db - 1e-6) / 20.0) - eps) / (gc.sum_abs_w * mag_scale);
    unsigned bits = 0u;                        // lim <= 0 or NaN: every tile reports (the floor path is exact for every unit)
    if (lim > 0.0) bits = __float_as_uint(__double2float_rd(lim * (1.0 - 1e-6)));   // +Inf: only non-finite samples report
    gc.alim_b[blockIdx.x] = bits;
  }
}

// thresh[u][f] = mean_t(dBfl) + n_std * std_t(dBfl)   (stationary.py:75-81; torchgate.py:158-160)
__global__ void k_colstats_final(const double* __restrict__ s_part, Geom g, int nts,
                                 const double* __restrict__ pmax, double mag_scale, double n_std, int ddof,
                                 double* __restrict__ thresh, int64_t n_units) {
  const int64_t n = n_units * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = i / g.FS;
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
    for (int ts = 0; ts < nts; ++ts) {
      s1 += s_part[((u * nts + ts) * 2 + 0) * g.FS + f];
      s2 += s_part[((u * nts + ts) * 2 + 1) * g.FS + f];
    }
    const double Tn = (double)g.T;
    const double mean_d = s1 / Tn;
    double var = (s2 - s1 * s1 / Tn) / (Tn - (double)ddof);
    if (var < 0.0) var = 0.0;
    thresh[i] = (cell_db(pmax[i], mag_scale) + mean_d) + sqrt(var) * n_std;
  }
}

// raw[u][t][f] = (max(dB, rowmax_dB - top_db) > thresh[f])   (stationary.py:96-106;
// torchgate.py:161-164 with amp_to_db's floor, torchgate/utils.py:23).
__global__ void k_decide(const double* __restrict__ P, Geom g, const double* __restrict__ pmax,
                         const double* __restrict__ thresh, int64_t thresh_ustride, double mag_scale,
                         double top_db, float* __restrict__ raw, int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    const int64_t u = i / (g.T * g.FS);
    double db = cell_db(P[i], mag_scale);
    double fl = cell_db(pmax[u * g.FS + f], mag_scale) - top_db;
    db = fmax(db, fl);
    // (np.maximum(x, NaN) is NaN: a band whose maximum is NaN passes nowhere)
    raw[i] = (fl == fl && db > thresh[u * thresh_ustride + f]) ? 1.0f : 0.0f;
  }
}

// Per-(unit, band) compare constant in the POWER domain (as k_prep_thresh does per band for variant S):
//   max(dB, rowmax_dB - top_db) > thresh   <=>   floor lifts the band  ||  |X|^2 > T2
// T2 = -1: every cell passes (floor above the threshold, or a zero cell already passes).
__global__ void k_t2_rows(const double* __restrict__ thresh, int64_t thresh_ustride, const double* __restrict__ pmax,
                          Geom g, double mag_scale, double top_db, double* __restrict__ T2, int64_t n_units) {
  const double eps = 2.220446049250313e-16;
  const int64_t n = n_units * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = i / g.FS;
    const int f = (int)(i % g.FS);
    double t2 = 0.0;
    if (f < g.F) {
      const double th = thresh[u * thresh_ustride + f];
      const double fl = cell_db(pmax[i], mag_scale) - top_db;
      if (th != th || fl != fl) {
        t2 = 1e300;   // NaN threshold or NaN in the band (its maximum is NaN): no cell passes (T2_NEVER)
      } else if (fl > th || 20.0 * log10(eps) > th) {
        t2 = -1.0;
      } else {
        const double tm = (exp10(th / 20.0) - eps) / mag_scale;
        t2 = tm > 0.0 ? tm * tm : 0.0;
      }
    }
    T2[i] = t2;
  }
}

// Short rows (TorchGate: 63 frames per 1 s clip): statistics, compare constants and decisions of one
// (row, 64 bands) tile in ONE kernel -- the tile of powers (T x 64 float64 <= 64 KB) is read once into
// LDS instead of four passes over the field (column max, moments, constants, compare).
//   thresh_in == nullptr: threshold from the row's own statistics (torchgate.py:158-160), same
//   summation order as k_colstats/k_colstats_final with one time slice; else thresh_in[u*ustride + f].
// Outputs: bits (64 bands per word), pmax and thresh_out (stage taps).
__global__ __launch_bounds__(64 * STAT_TG) void k_row_decide(const double* __restrict__ P, Geom g,
                                                             const double* __restrict__ thresh_in,
                                                             int64_t thresh_ustride, double mag_scale,
                                                             double top_db, double n_std, int ddof,
                                                             double* __restrict__ pmax, double* __restrict__ thresh_out,
                                                             unsigned long long* __restrict__ bits, int wpr,
                                                             DbFast dbk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* tile = reinterpret_cast<double*>(smem);  // [T][64]
  __shared__ double r1[STAT_TG][64], r2[STAT_TG][64];
  __shared__ __attribute__((aligned(16))) double s_tab[256];
  s_tab[threadIdx.x] = dbk.tab[threadIdx.x];   // 64 * STAT_TG = 256 threads
  const double eps = 2.220446049250313e-16;
  const int l = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int w = blockIdx.x;
  const int f = 64 * w + l;
  const int64_t u = blockIdx.y;
  const bool on = f < g.F;
  double m = 0.0;
  for (int64_t t0 = tg; t0 < g.T; t0 += 16 * STAT_TG) {   // 16 independent loads in flight per thread
    double pv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t t = t0 + q * STAT_TG;
      pv[q] = (on && t < g.T) ? P[(u * g.T + t) * g.FS + f] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t t = t0 + q * STAT_TG;
      if (t < g.T) tile[t * 64 + l] = pv[q];
      m = (m != m || pv[q] != pv[q]) ? NAN : fmax(m, pv[q]);   // NaN-sticky like torch.max: the band is NaN then
    }
  }
  r1[tg][l] = m;
  __syncthreads();
  for (int i = 0; i < STAT_TG; ++i) {
    const double mm = r1[i][l];
    m = (m != m || mm != mm) ? NAN : fmax(m, mm);
  }
  // band-level quantities (two logarithms, a square root, an exp10: ~300 float64 instructions) are evaluated by ONE of
  // the four thread groups and handed to the others through LDS -- they only depend on the band
  __shared__ double s_band[2][64];   // [0] maximum in dB, [1] compare constant
  if (tg == 0) s_band[0][l] = cell_db(m, mag_scale);
  __syncthreads();  // (also: r1 is reused below)
  const double mdb = s_band[0][l];
  double th = 0.0;
  if (thresh_in == nullptr) {
    double s1 = 0.0, s2 = 0.0;
    for (int64_t t = tg; t < g.T; t += STAT_TG) {
      double d = db_fast(tile[t * 64 + l], s_tab, dbk, mag_scale) - mdb;  // <= 0
      d = (d != d) ? d : fmax(d, -top_db);   // (fmax would drop a NaN; numpy / torch keep it)
      s1 += d;
      s2 += d * d;
    }
    r1[tg][l] = s1;
    r2[tg][l] = s2;
    __syncthreads();
    if (tg == 0) {
      s1 = r1[0][l];
      s2 = r2[0][l];
      for (int i = 1; i < STAT_TG; ++i) {
        s1 += r1[i][l];
        s2 += r2[i][l];
      }
      const double Tn = (double)g.T;
      const double mean_d = s1 / Tn;
      double var = (s2 - s1 * s1 / Tn) / (Tn - (double)ddof);
      if (var < 0.0) var = 0.0;
      th = (mdb + mean_d) + sqrt(var) * n_std;
    }
  } else if (tg == 0) {
    th = on ? thresh_in[u * thresh_ustride + f] : 0.0;
  }
  // compare constant in the power domain (see k_t2_rows)
  double t2 = 0.0;
  if (tg == 0) {
    if (on) {
      const double fl = mdb - top_db;
      if (th != th || fl != fl) {
        t2 = 1e300;   // NaN threshold / NaN in the band: no cell passes (T2_NEVER)
      } else if (fl > th || 20.0 * log10(eps) > th) {
        t2 = -1.0;
      } else {
        const double tm = (exp10(th / 20.0) - eps) / mag_scale;
        t2 = tm > 0.0 ? tm * tm : 0.0;
      }
    }
    s_band[1][l] = t2;
  }
  __syncthreads();
  t2 = s_band[1][l];
  if (tg == 0 && f < g.FS) {
    pmax[u * g.FS + f] = on ? m : 0.0;
    if (thresh_out && on) thresh_out[u * g.FS + f] = th;
  }
  for (int64_t t = tg; t < g.T; t += STAT_TG) {
    const bool pred = on && tile[t * 64 + l] > t2;
    const unsigned long long word = __ballot(pred);
    if (l == 0) bits[(u * g.T + t) * (int64_t)wpr + w] = word;
  }
}

// bits[u][t][w] = P > T2[u][f]: one wavefront per 64-bin word, a pure compare (no log10 per cell)
__global__ void k_decide_bits_t2(const double* __restrict__ P, Geom g, const double* __restrict__ T2,
                                 unsigned long long* __restrict__ bits, int wpr, int64_t n_units) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = n_units * g.T * wpr;
  for (int64_t wd = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); wd < nwords;
       wd += (int64_t)gridDim.x * (blockDim.x >> 6)) {
    const int w = (int)(wd % wpr);
    const int64_t ut = wd / wpr;
    const int64_t u = ut / g.T;
    const int f = 64 * w + lane;
    const bool pred = f < g.F && P[ut * g.FS + f] > T2[u * g.FS + f];
    const unsigned long long word = __ballot(pred);
    if (lane == 0) bits[wd] = word;
  }
}

// ---------------------------------------------------------------------------------------
// Non-stationary raw masks.
// ---------------------------------------------------------------------------------------
// S: forward-backward one-pole smoother == scipy filtfilt([b],[1,b-1],padtype=None)
// (nonstationary.py:106-115), then sigmoid((A-S)/S - thresh) * slope) (nonstationary.py:70-76).
// One thread per (unit, bin); lanes = bins (coalesced row walk).  raw is used as scratch for
// the forward pass.
// sigmoid(((A - S) / S - thresh) * slope): the difference in float64 (cancellation), the smooth
// remainder (division, exp) in float32 -- the mask is a float32 field and no decision hangs on it.
__device__ __forceinline__ float sigmoid_ratio(double av, double s, float nthresh, float slope) {
  const float ratio = (float)(av - s) / (float)s;
  return 1.0f / (1.0f + __expf(-(ratio - nthresh) * slope));
}

__global__ void k_iir_sigmoid(const float* __restrict__ A, Geom g, double b, double nthresh, double slope,
                              float* __restrict__ raw) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t u = blockIdx.y;
  if (f >= g.F) return;
  const float* a = A + u * g.T * g.FS + f;
  float* r = raw + u * g.T * g.FS + f;
  const double c = 1.0 - b;
  double s = (double)a[0];
  for (int64_t t = 0; t < g.T; ++t) {
    s = b * (double)a[t * g.FS] + c * s;
    r[t * g.FS] = (float)s;
  }
  // backward pass on the forward output, seeded with its last value
  double fprev = s;
  for (int64_t t = g.T - 1; t >= 0; --t) {
    double fw = (double)r[t * g.FS];
    if (t == g.T - 1) fw = fprev;  // exact (unrounded) last forward value
    s = b * fw + c * s;
    double av = (double)a[t * g.FS];
    r[t * g.FS] = sigmoid_ratio(av, s, (float)nthresh, (float)slope);
  }
}

// Segmented form of k_iir_sigmoid: block = 64 bins x NSEG time segments of one unit.  Every
// segment first runs the recurrence from a zero state (its response to its own input), the
// carries s_in(seg) = s_end(seg-1) are chained through LDS (s_end = local_end + c^len * s_in), then
// the segment re-runs with the right initial state.  Same recurrence, 16x the parallelism.
constexpr int IIR_NSEG = 16;

__global__ __launch_bounds__(64 * IIR_NSEG) void k_iir_sigmoid_seg(const float* __restrict__ A, Geom g, double b,
                                                                   double nthresh, double slope,
                                                                   float* __restrict__ raw) {
  __shared__ double s_end[IIR_NSEG][64];
  __shared__ double s_pow[IIR_NSEG];
  __shared__ double s_seed[64];
  const int l = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + l;
  const int64_t u = blockIdx.y;
  const bool on = f < g.F;
  const int64_t ts = g.T * seg / IIR_NSEG, te = g.T * (seg + 1) / IIR_NSEG;
  const float* a = A + u * g.T * g.FS + (on ? f : 0);
  float* r = raw + u * g.T * g.FS + (on ? f : 0);
  const double c = 1.0 - b;
  if (l == 0) s_pow[seg] = pow(c, (double)(te - ts));
  // ---- forward ----
  double e = 0.0;
  if (on) {
#pragma unroll 8
    for (int64_t t = ts; t < te; ++t) e = b * (double)a[t * g.FS] + c * e;
  }
  s_end[seg][l] = e;
  __syncthreads();
  double carry = on ? (double)a[0] : 0.0;  // s[-1] = A[0]  (lfilter_zi steady state)
  for (int k = 0; k < seg; ++k) carry = s_end[k][l] + s_pow[k] * carry;
  double s = carry;
  // the backward recurrence's zero-state response of this segment, s_back[ts] = sum_t b c^(t-ts) fw[t],
  // is a weighted sum of the forward values: accumulated here in forward order (saves one pass over r)
  e = 0.0;
  if (on) {
    double pw = b;
#pragma unroll 8
    for (int64_t t = ts; t < te; ++t) {
      s = b * (double)a[t * g.FS] + c * s;
      const float sf = (float)s;
      r[t * g.FS] = sf;
      // the backward pass reads the ROUNDED forward value, except the exact last one (seed)
      e += pw * ((t == g.T - 1) ? s : (double)sf);
      pw *= c;
    }
  }
  if (seg == IIR_NSEG - 1) s_seed[l] = s;  // exact forward value at T-1
  __syncthreads();  // s_end (forward carries) fully consumed, s_seed visible
  // ---- backward on the forward output ----
  const double seed = s_seed[l];
  s_end[seg][l] = e;
  __syncthreads();
  carry = seed;  // backward pass is seeded with the forward pass's last value
  for (int k = IIR_NSEG - 1; k > seg; --k) carry = s_end[k][l] + s_pow[k] * carry;
  s = carry;
  if (on) {
#pragma unroll 4
    for (int64_t t = te - 1; t >= ts; --t) {
      double fw = (t == g.T - 1) ? seed : (double)r[t * g.FS];
      s = b * fw + c * s;
      double av = (double)a[t * g.FS];
      r[t * g.FS] = sigmoid_ratio(av, s, (float)nthresh, (float)slope);
    }
  }
}

// T: boxcar moving mean conv1d(ones(k), padding="same")/k, left pad (k-1)//2
// (torchgate.py:179-190), then sigmoid((ratio - x0)/temp) (torchgate.py:193-196).
// Block = 64 bins x 4 time segments of BOX_TSEG frames: every thread seeds its window sum directly
// (kbox loads) and slides it over its own frames -- T/BOX_TSEG-fold the parallelism of one serial
// walk per band (TorchGate rows are short: 63 frames at 16 kHz / 1 s).
constexpr int BOX_TSEG = 8;
__global__ __launch_bounds__(256) void k_boxcar_sigmoid(const float* __restrict__ A, Geom g, int kbox,
                                                        double nthresh, double slope, float* __restrict__ raw) {
  const int f = blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t u = blockIdx.z;
  const int64_t t0 = ((int64_t)blockIdx.y * 4 + (threadIdx.x >> 6)) * BOX_TSEG;
  if (f >= g.F || t0 >= g.T) return;
  const float* a = A + u * g.T * g.FS + f;
  float* r = raw + u * g.T * g.FS + f;
  const int left = (kbox - 1) / 2;
  // window of frame t: [t - left, t - left + kbox), zero outside [0, T)
  double sum = 0.0;
  for (int64_t j = max<int64_t>(t0 - left, 0); j < min<int64_t>(t0 - left + kbox, g.T); ++j) sum += (double)a[j * g.FS];
  const int64_t t1 = min<int64_t>(t0 + BOX_TSEG, g.T);
  for (int64_t t = t0; t < t1; ++t) {
    const double S = sum / (double)kbox;
    r[t * g.FS] = sigmoid_ratio((double)a[t * g.FS], S, (float)nthresh, (float)slope);
    const int64_t drop = t - left, add = t - left + kbox;
    if (drop >= 0) sum -= (double)a[drop * g.FS];
    if (add < g.T) sum += (double)a[add * g.FS];
  }
}

// ---------------------------------------------------------------------------------------
// Mask smoothing: separable triangular FIR, zero padded ("same"), two passes.
// final = p * conv(raw) + (1-p) * edge   where edge = conv(1) (zero-padded) when the reference
// applies prop_decrease BEFORE smoothing (stationary.py:108-114; torchgate.py:241-249) and
// 1 when it applies it AFTER (nonstationary.py:78-84).
// ---------------------------------------------------------------------------------------
__global__ void k_smooth_f(const float* __restrict__ raw, Geom g, const float* __restrict__ kf, int nf,
                           float* __restrict__ tmp, int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    float acc = 0.f;
    for (int a = -nf; a <= nf; ++a) {
      int ff = f + a;
      if (ff >= 0 && ff < g.F) acc += kf[a + nf] * raw[i + a];
    }
    tmp[i] = acc;
  }
}

__global__ void k_smooth_t(const float* __restrict__ tmp, Geom g, const float* __restrict__ kt, int nt,
                           const float* __restrict__ kf, int nf, float p, int prop_before,
                           float* __restrict__ M, int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    const int64_t t = (i / g.FS) % g.T;
    float acc = 0.f, et = 0.f;
    for (int b = -nt; b <= nt; ++b) {
      int64_t tt = t + b;
      if (tt >= 0 && tt < g.T) {
        acc += kt[b + nt] * tmp[i + (int64_t)b * g.FS];
        et += kt[b + nt];
      }
    }
    float edge = 1.0f;
    if (prop_before) {  // conv(1) with zero padding = (valid freq taps) * (valid time taps)
      float ef = 0.f;
      for (int a = -nf; a <= nf; ++a)
        if (f + a >= 0 && f + a < g.F) ef += kf[a + nf];
      edge = ef * et;
    }
    M[i] = p * acc + (1.0f - p) * edge;
  }
}

// LDS-tiled version of k_smooth_f + k_smooth_t: one block = TT frames x FB bins of one unit.
// raw tile (+halo) -> LDS, f-pass LDS -> LDS, t-pass LDS -> global.  Each thread produces FOUR
// adjacent outputs along the filter axis from a sliding register window (one LDS read per tap per
// four outputs); taps and per-tile edge factors sit in LDS.  LDS pitches are odd: the f-pass walks
// rows with consecutive lanes, the t-pass walks bins with consecutive lanes -- both conflict-free.
// The kernel is latency-bound (one global round trip per block, then LDS work): the tile is sized
// for occupancy -- 64 x 64 outputs = 46 KB of LDS at the 48 kHz default (3 blocks/CU); 32 x 128
// (54 KB, 2 blocks/CU) was 1.6x slower, smaller tiles pay too much halo.
constexpr int SMF_TT = 64, SMF_FB = 64, SMF_KT = 192;

__global__ __launch_bounds__(256) void k_smooth_tiled(const float* __restrict__ raw, Geom g,
                                                      const float* __restrict__ kf, int nf,
                                                      const float* __restrict__ kt, int nt, float p,
                                                      int prop_before, float* __restrict__ M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int rows = SMF_TT + 2 * nt;
  const int cols = SMF_FB + 2 * nf;
  const int tp = cols | 1;             // odd pitch of the raw tile
  constexpr int BP = SMF_FB + 1;       // odd pitch of the f-pass result
  // taps first (16-byte aligned, padded to multiples of 4): broadcast LDS reads instead of one
  // scalar-cache round trip per tap inside the filter loops
  float* skf = reinterpret_cast<float*>(smem);       // [64]
  float* skt = skf + 64;                             // [SMF_KT]: time half-widths up to 95 (n_fft = 256 at 48 kHz: nt = 37)
  float* sef = skt + SMF_KT;                         // [SMF_FB] conv(1) along f under zero padding
  float* set_ = sef + SMF_FB;                        // [SMF_TT] conv(1) along t
  float* tile = set_ + SMF_TT;                       // [rows][tp]  raw, zero outside the field
  float* buf = tile + (size_t)rows * tp;             // [rows][BP]  after the f-pass
  if (threadIdx.x < 64) skf[threadIdx.x] = (int)threadIdx.x <= 2 * nf ? kf[threadIdx.x] : 0.f;
  if (threadIdx.x < SMF_KT) skt[threadIdx.x] = (int)threadIdx.x <= 2 * nt ? kt[threadIdx.x] : 0.f;
  const int64_t u = blockIdx.z;
  const int64_t t0 = (int64_t)blockIdx.y * SMF_TT;
  const int f0 = blockIdx.x * SMF_FB;
  if (threadIdx.x >= 64 && threadIdx.x < 64 + SMF_FB + SMF_TT) {
    // edge factors of this tile, straight from the global taps (read again below through LDS)
    const int j = threadIdx.x - 64;
    float e = 0.f;
    if (j < SMF_FB) {
      const int f = f0 + j;
      for (int a = -nf; a <= nf; ++a)
        if (f + a >= 0 && f + a < g.F) e += kf[a + nf];
      sef[j] = e;
    } else {
      const int64_t t = t0 + (j - SMF_FB);
      for (int b = -nt; b <= nt; ++b)
        if (t + b >= 0 && t + b < g.T) e += kt[b + nt];
      set_[j - SMF_FB] = e;
    }
  }
  // tile load: ALL of a thread's global loads are issued before the first LDS store, so the block
  // pays one memory round trip instead of one per batch (the kernel is latency-bound otherwise)
  constexpr int LB = 32;
  for (int i0 = threadIdx.x; i0 < rows * cols; i0 += 256 * LB) {
    float vals[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int i = i0 + 256 * j;
      const int r = i / cols, cidx = i - r * cols;
      const int64_t t = t0 - nt + r;
      const int f = f0 - nf + cidx;
      vals[j] = (i < rows * cols && t >= 0 && t < g.T && f >= 0 && f < g.F) ? raw[(u * g.T + t) * g.FS + f] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int i = i0 + 256 * j;
      if (i < rows * cols) tile[(i / cols) * tp + (i % cols)] = vals[j];
    }
  }
  __syncthreads();
  // f-pass: item = (row r, group of 4 bins); consecutive lanes take consecutive rows
  for (int it = threadIdx.x; it < rows * (SMF_FB / 4); it += 256) {
    const int r = it % rows, c0 = (it / rows) * 4;
    const float* src = tile + (size_t)r * tp + c0;  // src[a] = raw at bin (f0 + c0 - nf + a)
    float w0 = src[0], w1 = src[1], w2 = src[2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int a = 0; a <= 2 * nf; ++a) {
      const float w3 = src[a + 3];
      const float k = skf[a];
      a0 += k * w0; a1 += k * w1; a2 += k * w2; a3 += k * w3;
      w0 = w1; w1 = w2; w2 = w3;
    }
    float* dst = buf + (size_t)r * BP + c0;
    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
  }
  __syncthreads();
  // t-pass: item = (group of 4 frames, bin); consecutive lanes take consecutive bins
  for (int it = threadIdx.x; it < (SMF_TT / 4) * SMF_FB; it += 256) {
    const int cidx = it % SMF_FB, r0 = (it / SMF_FB) * 4;
    const int f = f0 + cidx;
    const float* src = buf + (size_t)r0 * BP + cidx;  // src[b*BP] = frame (t0 + r0 - nt + b)
    float w0 = src[0], w1 = src[BP], w2 = src[2 * BP];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int b = 0; b <= 2 * nt; ++b) {
      const float w3 = src[(size_t)(b + 3) * BP];
      const float k = skt[b];
      a0 += k * w0; a1 += k * w1; a2 += k * w2; a3 += k * w3;
      w0 = w1; w1 = w2; w2 = w3;
    }
    if (f >= g.F) continue;
    const float accs[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t t = t0 + r0 + e;
      if (t >= g.T) break;
      float edge = 1.0f;
      // conv(1) under zero padding differs from 1 only within nf bins / nt frames of the border
      if (prop_before && (f < nf || f >= g.F - nf || t < nt || t >= g.T - nt)) edge = sef[cidx] * set_[r0 + e];
      M[(u * g.T + t) * g.FS + f] = p * accs[e] + (1.0f - p) * edge;
    }
  }
}

// no smoothing: M = p*raw + (1-p)
__global__ void k_prop_only(const float* __restrict__ raw, Geom g, float p, float* __restrict__ M,
                            int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    M[i] = p * raw[i] + (1.0f - p);
  }
}

// yn[s] = mean over channels (np.mean(axis=0), stationary.py:61): sequential fp64 sum / C.
__global__ void k_channel_mean(const void* __restrict__ x, int dtype, int64_t C, int64_t n, int64_t stride,
                               double* __restrict__ out) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n;
       s += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int64_t c = 0; c < C; ++c) acc += load_sample(x, dtype, c * stride + s);
    out[s] = acc / (double)C;
  }
}


}  // namespace sg
