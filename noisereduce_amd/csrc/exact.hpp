// EXACT path (variant S): the whole gate evaluated in float64, field by field, like the reference does it
// (spectralgate/base.py:140 allocates float64 chunks; stationary.py:83-127, nonstationary.py:47-97).
//
// Why it exists: the reference casts its float64 result to the input dtype (base.py:217-226) -- for int16 / int32
// recordings (what scipy.io.wavfile.read returns) a TRUNCATION.  A float32 pipeline is accurate to ~2e-7 of peak,
// which moves about 1 % of the samples of an int16 recording across an integer boundary (+-1 LSB).  Integer outputs
// therefore take this path by default: float64 power field -> float64 decisions / floor / sigmoid -> float64
// separable smoothing -> float64 masked inverse transform -> float64 overlap-add -> truncation: the integers of the
// reference.  It is the materialised pipeline (every field through HBM, 8 bytes per cell): an order of magnitude
// slower than the fused float32 kernels -- SG_OPT_FAST_INTEGER selects those (<= 1 LSB off on ~1 % of the samples).
#pragma once
#include "kernels.hpp"
#include "czt.hpp"
#include "nonstat_mask.hpp"   // NsTiling

namespace sg {
namespace exact {

// (m + 1 - |a|) / (m + 1)^2: one axis of the reference's smoothing filter after its normalisation
// (base.py:7-29: outer([1..m+1..1] / (m+1)) / sum)
__device__ __forceinline__ double tap(int m, int a) {
  const int aa = a < 0 ? -a : a;
  return (double)(m + 1 - aa) / ((double)(m + 1) * (double)(m + 1));
}

// Non-stationary raw mask (nonstationary.py:59-76): A = |X| (any common scale cancels), S = filtfilt(one pole)(A)
// along time, raw = 1 / (1 + exp(-((A - S) / S - thresh) * slope)).  One thread per (unit, band), float64 throughout;
// the forward pass is parked in `raw`.
__global__ void kx_iir_sigmoid(const double* __restrict__ P, Geom g, double b, double nthresh, double slope,
                               double* __restrict__ raw) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t u = blockIdx.y;
  if (f >= g.F) return;
  const double* p = P + u * g.T * g.FS + f;
  double* r = raw + u * g.T * g.FS + f;
  const double c = 1.0 - b;
  double s = sqrt(p[0]);
  for (int64_t t = 0; t < g.T; ++t) {
    s = b * sqrt(p[t * g.FS]) + c * s;
    r[t * g.FS] = s;
  }
  // backward pass over the forward output, seeded with its last value (scipy filtfilt, padtype=None)
  for (int64_t t = g.T - 1; t >= 0; --t) {
    s = b * r[t * g.FS] + c * s;
    const double a = sqrt(p[t * g.FS]);
    r[t * g.FS] = 1.0 / (1.0 + exp(-((a - s) / s - nthresh) * slope));
  }
}

// The same mask tile-parallel (round 5; nonstat.hpp's scheme in double): kx_iir_sigmoid walks a band's 2579 frames
// serially with one thread per (unit, band) -- 24 k threads on 256 CUs, 2.85 ms of a 6.7 ms call.  The recurrence is
// linear: a tile of XIIR_TT frames contributes two numbers per band (kx_iir_part), k_iir_chain<double, true> turns them
// into the states entering every tile, and kx_iir_apply runs both sweeps of a tile from those states with the tile's
// magnitudes and forward values in registers -- the operations of kx_iir_sigmoid on every frame, in the same order within
// a tile; what differs is how the state ENTERING a tile was formed (closed-form combination instead of the running
// value: a few 1e-16 relative).
constexpr int XIIR_TT = 32;
__global__ __launch_bounds__(256) void kx_iir_part(const double* __restrict__ P, Geom g, NsTiling tl, double b,
                                                   double* __restrict__ part) {
  const unsigned FSu = (unsigned)g.FS;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const int64_t nk = tl.n_tiles();
  if (idx >= (unsigned)nk * FSu) return;
  const int64_t k = idx / FSu;
  const int f = (int)(idx % FSu);
  if (f >= g.F) return;
  const int64_t u = blockIdx.y;
  const double c = 1.0 - b;
  const double* a = P + u * g.T * g.FS + f;
  const int64_t ts = k * XIIR_TT, te = ts + XIIR_TT < g.T ? ts + XIIR_TT : g.T;
  double x[XIIR_TT];
#pragma unroll
  for (int q = 0; q < XIIR_TT; ++q) x[q] = a[(ts + q < te ? ts + q : te - 1) * g.FS];   // all loads in flight
  double e = 0.0, E0 = 0.0, pw = b;
#pragma unroll
  for (int q = 0; q < XIIR_TT; ++q)
    if (ts + q < te) {
      e = b * sqrt(x[q]) + c * e;
      E0 += pw * e;
      pw *= c;
    }
  double* o = part + ((u * nk + k) * 2) * (int64_t)g.FS + f;
  o[0] = e;
  o[g.FS] = E0;
}

// carries [unit][tile][2][FS] of k_iir_chain: [0] forward state before the tile (s_f[ts - 1]), [1] backward state at its
// end (S[te]; S[T] = s_f[T - 1])
__global__ __launch_bounds__(256) void kx_iir_apply(const double* __restrict__ P, const double* __restrict__ carry, Geom g,
                                                    NsTiling tl, double b, double nthresh, double slope,
                                                    double* __restrict__ raw) {
  const unsigned FSu = (unsigned)g.FS;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const int64_t nk = tl.n_tiles();
  if (idx >= (unsigned)nk * FSu) return;
  const int64_t k = idx / FSu;
  const int f = (int)(idx % FSu);
  if (f >= g.F) return;
  const int64_t u = blockIdx.y;
  const double c = 1.0 - b;
  const int64_t ts = k * XIIR_TT, te = ts + XIIR_TT < g.T ? ts + XIIR_TT : g.T;
  const double* a = P + (u * g.T + ts) * g.FS + f;
  double* r = raw + (u * g.T + ts) * g.FS + f;
  const int n = (int)(te - ts);
  const double* cb = carry + ((u * nk + k) * 2) * (int64_t)g.FS + f;
  double s = cb[0];
  double S = cb[g.FS];
  double x[XIIR_TT], sf[XIIR_TT];
#pragma unroll
  for (int q = 0; q < XIIR_TT; ++q) x[q] = a[(int64_t)(q < n ? q : n - 1) * g.FS];
#pragma unroll
  for (int q = 0; q < XIIR_TT; ++q) {
    x[q] = sqrt(x[q]);
    if (q < n) s = b * x[q] + c * s;
    sf[q] = s;
  }
  if (te == g.T) S = s;   // the backward pass's seed: the forward pass's last value, exactly
#pragma unroll
  for (int q = XIIR_TT - 1; q >= 0; --q)
    if (q < n) {
      S = b * sf[q] + c * S;
      r[(int64_t)q * g.FS] = 1.0 / (1.0 + exp(-((x[q] - S) / S - nthresh) * slope));
    }
}

// separable triangle smoothing, zero padded ("same"), float64.  TIN: float (0/1 decisions) or double (sigmoid).
template <typename TIN>
__global__ void kx_smooth_f(const TIN* __restrict__ raw, Geom g, int nf, double* __restrict__ tmp, int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    double acc = 0.0;
    for (int a = -nf; a <= nf; ++a) {
      const int ff = f + a;
      if (ff >= 0 && ff < g.F) acc += tap(nf, a) * (double)raw[i + a];
    }
    tmp[i] = acc;
  }
}

// final = p * conv(raw) + (1 - p) * edge: edge = conv(1) (zero padded) when prop_decrease is applied BEFORE the
// smoothing (stationary.py:108-114), 1 when it is applied after (nonstationary.py:78-84)
__global__ void kx_smooth_t(const double* __restrict__ tmp, Geom g, int nt, int nf, double p, int prop_before,
                            double* __restrict__ M, int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    const int64_t t = (i / g.FS) % g.T;
    double acc = 0.0, et = 0.0;
    for (int b = -nt; b <= nt; ++b) {
      const int64_t tt = t + b;
      if (tt >= 0 && tt < g.T) {
        acc += tap(nt, b) * tmp[i + (int64_t)b * g.FS];
        et += tap(nt, b);
      }
    }
    double edge = 1.0;
    if (prop_before) {
      double ef = 0.0;
      for (int a = -nf; a <= nf; ++a)
        if (f + a >= 0 && f + a < g.F) ef += tap(nf, a);
      edge = ef * et;
    }
    M[i] = p * acc + (1.0 - p) * edge;
  }
}

// LDS-tiled kx_smooth_f + kx_smooth_t (k_smooth_tiled of kernels.hpp in double; round 5): one block = XT frames x XB bins of
// one unit; raw tile (+ halo) -> LDS, f-pass LDS -> LDS, t-pass LDS -> global.  The taps are computed once per block
// (kx_smooth_f / kx_smooth_t divide per tap and cell: 2.0 ms of a 6.7 ms non-stationary call on ten minutes of int16 audio).
// Same sums in the same order as the two direct kernels (a = -nf .. nf, then b = -nt .. nt; taps outside the field skipped =
// zero in the tile): identical results.
// The kernel is latency-bound (three barrier-separated phases, 64 KB of LDS: two blocks per CU), so it runs 1024 threads per
// block (1.01 ms with 256 threads and one dependent LDS read per tap -> 0.73 with four taps per step at 512 threads -> 0.55 at
// 1024 -> 0.50 with one t-pass item per thread and the row loads of the tile issued together; ten minutes of 48 kHz).
constexpr int XSM_TT = 32, XSM_FB = 64, XSM_KMAX = 192;
__host__ __device__ inline size_t xsm_lds_bytes(int nf, int nt) {
  const int rows = XSM_TT + 2 * nt + 3, cols = XSM_FB + 2 * nf + 3;   // + 3: the sliding windows read 3 entries past the last tap
  return ((size_t)rows * (cols | 1) + (size_t)rows * (XSM_FB + 1) + 2 * XSM_KMAX + XSM_FB + XSM_TT) * sizeof(double);
}
// One sliding step of FOUR adjacent outputs over FOUR consecutive taps (k[0..3]; inputs w0..w2 carried in, w3..w6 fresh):
// every output still sums its taps in ascending order, one fused multiply-add per tap -- the sums of the one-tap loop, without
// its register rotation (three 64-bit moves per tap) and with four LDS reads in flight instead of one dependent read per tap.
#define XSM_STEP4(k, w0, w1, w2, w3, w4, w5, w6) \
  a0 += k[0] * w0; a1 += k[0] * w1; a2 += k[0] * w2; a3 += k[0] * w3; \
  a0 += k[1] * w1; a1 += k[1] * w2; a2 += k[1] * w3; a3 += k[1] * w4; \
  a0 += k[2] * w2; a1 += k[2] * w3; a2 += k[2] * w4; a3 += k[2] * w5; \
  a0 += k[3] * w3; a1 += k[3] * w4; a2 += k[3] * w5; a3 += k[3] * w6;
constexpr int XSM_THREADS = 1024;
template <typename TIN>
__global__ __launch_bounds__(XSM_THREADS) void kx_smooth_tiled(const TIN* __restrict__ raw, Geom g, int nf, int nt, double p, int prop_before,
                                                               double* __restrict__ M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHR = XSM_THREADS;
  const int rows = XSM_TT + 2 * nt, cols = XSM_FB + 2 * nf;
  const int tp = (cols + 3) | 1;
  constexpr int BP = XSM_FB + 1;
  double* skf = reinterpret_cast<double*>(smem);   // [XSM_KMAX]
  double* skt = skf + XSM_KMAX;                    // [XSM_KMAX]
  double* sef = skt + XSM_KMAX;                    // [XSM_FB] conv(1) along f under zero padding
  double* set_ = sef + XSM_FB;                     // [XSM_TT] conv(1) along t
  double* tile = set_ + XSM_TT;                    // [rows + 3][tp]
  double* buf = tile + (size_t)(rows + 3) * tp;    // [rows + 3][BP]
  const int64_t u = blockIdx.z;
  const int64_t t0 = (int64_t)blockIdx.y * XSM_TT;
  const int f0 = blockIdx.x * XSM_FB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // raw tile: a wave per row, lanes along the bins (coalesced 512-byte reads; no division per element)
  for (int r = wave; r < rows + 3; r += NTHR / 64) {
    const int64_t t = t0 - nt + r;
    const bool rv = r < rows && t >= 0 && t < g.T;
    const TIN* src = raw + (u * g.T + (rv ? t : 0)) * g.FS;
    // (clamped loads issued together, selected afterwards: the row's two column passes are in flight at once)
    const int fa = f0 - nf + lane, fb = fa + 64;
    const TIN va = src[min(max(fa, 0), g.F - 1)], vb = src[min(max(fb, 0), g.F - 1)];
    tile[r * tp + lane] = (rv && lane < cols && fa >= 0 && fa < g.F) ? (double)va : 0.0;
    if (lane + 64 < tp) tile[r * tp + lane + 64] = (rv && lane + 64 < cols && fb >= 0 && fb < g.F) ? (double)vb : 0.0;
    for (int cidx = lane + 128; cidx < tp; cidx += 64) {
      const int f = f0 - nf + cidx;
      tile[r * tp + cidx] = (rv && cidx < cols && f >= 0 && f < g.F) ? (double)src[f] : 0.0;
    }
  }
  for (int i = threadIdx.x; i < XSM_KMAX; i += NTHR) {
    skf[i] = i <= 2 * nf ? tap(nf, i - nf) : 0.0;
    skt[i] = i <= 2 * nt ? tap(nt, i - nt) : 0.0;
  }
  if (prop_before && threadIdx.x < XSM_FB + XSM_TT) {
    const int j = threadIdx.x;
    double e = 0.0;
    if (j < XSM_FB) {
      const int f = f0 + j;
      for (int a = -nf; a <= nf; ++a)
        if (f + a >= 0 && f + a < g.F) e += tap(nf, a);
      sef[j] = e;
    } else {
      const int64_t t = t0 + (j - XSM_FB);
      for (int b = -nt; b <= nt; ++b)
        if (t + b >= 0 && t + b < g.T) e += tap(nt, b);
      set_[j - XSM_FB] = e;
    }
  }
  __syncthreads();
  // f-pass: item = (row, group of 4 bins), consecutive lanes take consecutive rows; FOUR adjacent outputs from a sliding
  // register window; every output sums its taps in the order a = -nf .. nf
  const int kf_n = 2 * nf + 1, kt_n = 2 * nt + 1;
  for (int it = threadIdx.x; it < (rows + 3) * (XSM_FB / 4); it += NTHR) {
    const int q = it / (rows + 3), r = it - q * (rows + 3), c0 = q * 4;
    const double* src = tile + (size_t)r * tp + c0;   // src[a] = raw at bin f0 + c0 - nf + a
    double w0 = src[0], w1 = src[1], w2 = src[2];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int a = 0;
    for (; a + 4 <= kf_n; a += 4) {
      const double w3 = src[a + 3], w4 = src[a + 4], w5 = src[a + 5], w6 = src[a + 6];
      const double k[4] = {skf[a], skf[a + 1], skf[a + 2], skf[a + 3]};
      XSM_STEP4(k, w0, w1, w2, w3, w4, w5, w6)
      w0 = w4; w1 = w5; w2 = w6;
    }
    for (; a < kf_n; ++a) {
      const double w3 = src[a + 3];
      const double k = skf[a];
      a0 += k * w0; a1 += k * w1; a2 += k * w2; a3 += k * w3;
      w0 = w1; w1 = w2; w2 = w3;
    }
    double* dst = buf + (size_t)r * BP + c0;
    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
  }
  __syncthreads();
  // t-pass: item = (pair of frames, bin), consecutive lanes take consecutive bins (a block's 1024 items: one per thread)
  for (int it = threadIdx.x; it < (XSM_TT / 2) * XSM_FB; it += NTHR) {
    const int cidx = it % XSM_FB, r0 = (it / XSM_FB) * 2;
    const int f = f0 + cidx;
    const double* src = buf + (size_t)r0 * BP + cidx;   // src[b * BP] = frame t0 + r0 - nt + b
    double w0 = src[0];
    double a0 = 0.0, a1 = 0.0;
    int b = 0;
    for (; b + 4 <= kt_n; b += 4) {
      const double* s1 = src + (size_t)(b + 1) * BP;
      const double w1 = s1[0], w2 = s1[BP], w3 = s1[2 * BP], w4 = s1[3 * BP];
      const double k0 = skt[b], k1 = skt[b + 1], k2 = skt[b + 2], k3 = skt[b + 3];
      a0 += k0 * w0; a1 += k0 * w1;
      a0 += k1 * w1; a1 += k1 * w2;
      a0 += k2 * w2; a1 += k2 * w3;
      a0 += k3 * w3; a1 += k3 * w4;
      w0 = w4;
    }
    for (; b < kt_n; ++b) {
      const double w1 = src[(size_t)(b + 1) * BP];
      const double k = skt[b];
      a0 += k * w0; a1 += k * w1;
      w0 = w1;
    }
    if (f >= g.F) continue;
    const double accs[2] = {a0, a1};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t t = t0 + r0 + e;
      if (t >= g.T) break;
      double edge = 1.0;
      if (prop_before) edge = sef[cidx] * set_[r0 + e];
      M[(u * g.T + t) * g.FS + f] = p * accs[e] + (1.0 - p) * edge;
    }
  }
}
#undef XSM_STEP4

template <typename TIN>
__global__ void kx_prop_only(const TIN* __restrict__ raw, Geom g, double p, double* __restrict__ M, int64_t n_units) {
  const int64_t cells = n_units * g.T * g.FS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % g.FS);
    if (f >= g.F) continue;
    M[i] = p * (double)raw[i] + (1.0 - p);
  }
}

// Apply + inverse in float64 (k_apply_istft of kernels.hpp with every type widened): frame -> FFT -> X * M[t][k] ->
// inverse FFT -> synthesis window (win / N: the half-size complex core leaves a factor N = n / 2) -> seg[u][t][0..n).
template <int N, int WAVES, int FPW, int NT = 64>
__global__ __launch_bounds__(WAVES * NT) void kx_apply_istft(View view, Geom g, const cx<double>* __restrict__ tw_g,
                                                             const double* __restrict__ win,
                                                             const double* __restrict__ M, double* __restrict__ seg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef cx<double> cd;
  cd* tw = reinterpret_cast<cd*>(smem);
  cd* bufs = tw + N;
  const int lane = threadIdx.x % NT;
  const int wave = threadIdx.x / NT;
  cd* buf = bufs + wave * lpn<double>(N);
  stage_twiddles<WAVES * NT, N>(tw, tw_g, (int)threadIdx.x);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  const double inv_n = 1.0 / (double)N;
  __syncthreads();
  for (int fi = 0; fi < FPW; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * FPW + fi) * WAVES + wave;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    for (int j = lane; j < N; j += NT) {
      cd z = {0.0, 0.0};
      if (valid) {
        z.x = view_sample(view, row, chunk, s0 + 2 * j) * win[2 * j];
        z.y = view_sample(view, row, chunk, s0 + 2 * j + 1) * win[2 * j + 1];
      }
      buf[lp<double>(j)] = z;
    }
    SG_PASS_SYNC();
    wave_fft<double, N, false, NT>(buf, tw, lane);
    if (valid) {
      const double* Mrow = M + (u * g.T + t) * g.FS;
      for (int k = lane; k <= N / 2; k += NT) {
        if (k == 0) {
          const cd a = buf[lp<double>(0)];
          const double y0 = (a.x + a.y) * Mrow[0];
          const double yN = (a.x - a.y) * Mrow[N];
          buf[lp<double>(0)] = {0.5 * (y0 + yN), 0.5 * (y0 - yN)};
        } else {
          const cd a = buf[lp<double>(k)], b = buf[lp<double>(N - k)];
          const cd w = tw[k];
          const cd E = {(a.x + b.x) * 0.5, (a.y - b.y) * 0.5};
          const cd O = {(a.y + b.y) * 0.5, (b.x - a.x) * 0.5};
          const cd wO = cmul(w, O);
          const double mk = Mrow[k], mn = Mrow[N - k];
          const cd Yk = {(E.x + wO.x) * mk, (E.y + wO.y) * mk};
          const cd Yn = {(E.x - wO.x) * mn, (-E.y + wO.y) * mn};
          const cd Ep = {(Yk.x + Yn.x) * 0.5, (Yk.y - Yn.y) * 0.5};
          const cd D = {(Yk.x - Yn.x) * 0.5, (Yk.y + Yn.y) * 0.5};
          const cd wc = {w.x, -w.y};
          const cd Op = cmul(D, wc);
          buf[lp<double>(k)] = {Ep.x - Op.y, Ep.y + Op.x};
          if (k != N - k) buf[lp<double>(N - k)] = {Ep.x + Op.y, -Ep.y + Op.x};
        }
      }
    }
    SG_PASS_SYNC();
    wave_fft<double, N, true, NT>(buf, tw, lane);
    if (valid) {
      double* srow = seg + (u * g.T + t) * (int64_t)g.n;
      for (int j = lane; j < N; j += NT) {
        const cd z = buf[lp<double>(j)];
        srow[2 * j] = z.x * win[2 * j] * inv_n;
        srow[2 * j + 1] = z.y * win[2 * j + 1] * inv_n;
      }
    }
    SG_PASS_SYNC();
  }
}

// chirp-z form for frame lengths that are not a power of two (k_apply_istft_czt of czt.hpp in float64)
template <int M, int NT, int FR>
__global__ __launch_bounds__(NT* FR) void kx_apply_istft_czt(View view, Geom g, CztTabs<double> tb,
                                                             const double* __restrict__ win,
                                                             const double* __restrict__ Mk, double* __restrict__ seg,
                                                             int fpb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef cx<double> cd;
  const int tl = threadIdx.x % NT, fr = threadIdx.x / NT;
  cd* buf = reinterpret_cast<cd*>(smem) + (size_t)fr * lpn<double>(M);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  const double inv_n = 1.0 / (double)g.n;
  for (int fi = 0; fi < fpb; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpb + fi) * FR + fr;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    for (int j = tl; j < M; j += NT) {
      cd z = {0.0, 0.0};
      if (valid && j < g.n) {
        const double xw = view_sample(view, row, chunk, s0 + j) * win[j];
        const cd c = tb.chirp[j];
        z = {xw * c.x, xw * c.y};
      }
      buf[lp<double>(j)] = z;
    }
    czt_core<double, M, NT>(buf, tb, tl);
    const double* Mrow = Mk + (u * g.T + (valid ? t : 0)) * g.FS;
    for (int j = tl; j < M; j += NT) {
      cd z = {0.0, 0.0};
      if (valid && j < g.n) {
        const cd c = tb.chirp[j];
        cd X = cmul(buf[lp<double>(j)], c);
        const int kk = j < g.F ? j : g.n - j;
        const double m = Mrow[kk];
        if (j == 0 || 2 * j == g.n) X.y = 0.0;
        const cd Yc = {X.x * m, -X.y * m};
        z = cmul(Yc, c);
      }
      buf[lp<double>(j)] = z;
    }
    czt_core<double, M, NT>(buf, tb, tl);
    if (valid) {
      double* srow = seg + (u * g.T + t) * (int64_t)g.n;
      for (int j = tl; j < g.n; j += NT) {
        const cd D = cmul(buf[lp<double>(j)], tb.chirp[j]);
        srow[j] = D.x * win[j] * inv_n;
      }
    }
    SG_PASS_SYNC();
  }
}

// truncating store of a float64 value (ndarray.astype semantics, base.py:217-226)
__device__ __forceinline__ void store_sample_f64(void* p, int dtype, int64_t idx, double val) {
  switch (dtype) {
    case 0: ((float*)p)[idx] = (float)val; break;
    case 1: ((double*)p)[idx] = val; break;
    case 2: ((int16_t*)p)[idx] = (int16_t)val; break;
    default: ((int32_t*)p)[idx] = (int32_t)val; break;
  }
}

// overlap-add gather in float64 (k_ola): out[p] = sum_t seg[t][e - tH] / sum_t w^2[e - tH], e = p + padL
// (scipy/_spectral_py.py:1708-1725: norm > 1e-10 guard)
__global__ void kx_ola(View view, Geom g, OutMap om, const double* __restrict__ seg, const double* __restrict__ win) {
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  const int64_t p = om.p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= om.p1) return;
  const int64_t gi = chunk * om.g_step + (p - om.p0);
  if (gi < om.g_lo || gi >= om.g_hi) return;
  double val = 0.0;
  if (p < g.Lout) {
    const int64_t e = p + g.padL;
    int64_t t_hi = e / g.H;
    if (t_hi > g.T - 1) t_hi = g.T - 1;
    int64_t t_lo = (e - g.n + g.H) / g.H;
    if (e - g.n + 1 <= 0) t_lo = 0;
    double acc = 0.0, norm = 0.0;
    for (int64_t t = t_lo; t <= t_hi; ++t) {
      const int m = (int)(e - t * g.H);
      acc += seg[(u * g.T + t) * (int64_t)g.n + m];
      norm += win[m] * win[m];
    }
    val = acc / (norm > 1e-10 ? norm : 1.0);
  }
  store_sample_f64(om.out, om.dtype, row * om.stride + gi - om.g0, val);
}

}  // namespace exact
}  // namespace sg
