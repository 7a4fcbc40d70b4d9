// STFT / masked ISTFT for frame lengths that are NOT a power of two.
//
// The reference accepts any n_fft (base.py:77-86; scipy.signal.stft -> rfft(n), istft -> irfft(n),
// scipy/signal/_spectral_py.py:2202,1689).  A length-n DFT is evaluated here as a chirp-z (Bluestein)
// convolution on the power-of-two Stockham core of fft_wave.hpp:
//     X[k] = conj(b[k]) * sum_j (x[j] conj(b[j])) b[k - j],      b[m] = exp(i pi m^2 / n)
// i.e. FFT_M -> pointwise product with B = FFT_M(b, wrapped) / M -> inverse FFT_M, M >= 2n - 1.
// The chirp, B and the twiddles are built on the host in extended precision (api.hip: build_czt)
// and stay in HBM (read-only, L2-resident); the M-point work buffer of each frame lives in LDS.
// This is the generality path (4 transforms of size M per frame instead of 2 of size n/2): correct
// for every n in [16, 4096], not tuned -- the tuned kernels are the power-of-two ones.
#pragma once
#include "kernels.hpp"

namespace sg {

template <typename TC>
struct CztTabs {
  const cx<TC>* tw;     // M entries  w_{2M}^k           (master twiddle table of the M-point core)
  const cx<TC>* chirp;  // n entries  conj(b[j]) = exp(-i pi j^2 / n)
  const cx<TC>* bhat;   // M entries  FFT_M(b wrapped) / M
};

// chirp-z DFT of the n values already staged in buf[0..n) (premultiplied by the chirp, zero above n):
// on return buf[k] * chirp[k] is bin k, for every k < n.
template <typename TC, int M, int NT>
__device__ __forceinline__ void czt_core(cx<TC>* buf, const CztTabs<TC>& tb, int tl) {
  SG_PASS_SYNC();
  wave_fft<TC, M, false, NT>(buf, tb.tw, tl);
  for (int j = tl; j < M; j += NT) buf[lp<TC>(j)] = cmul(buf[lp<TC>(j)], tb.bhat[j]);
  SG_PASS_SYNC();
  wave_fft<TC, M, true, NT>(buf, tb.tw, tl);
}

// Forward STFT, outputs as k_stft (kernels.hpp): P (float64 power) / magnitude (float32) / X itself.
// One team of NT threads per frame, FR frames in flight per block, fpb frames per team.
template <typename TC, int M, int NT, int FR>
__global__ __launch_bounds__(NT* FR) void k_stft_czt(View view, Geom g, CztTabs<TC> tb, const TC* __restrict__ wfull,
                                                     double* __restrict__ P_out, float* __restrict__ mag_out,
                                                     double* __restrict__ z_out, double z_scale,
                                                     unsigned long long* __restrict__ pmax_bits, int fpb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tl = threadIdx.x % NT, fr = threadIdx.x / NT;
  cx<TC>* buf = reinterpret_cast<cx<TC>*>(smem) + (size_t)fr * lpn<TC>(M);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  constexpr int VM = M / 4 / NT + 2;  // F <= M/4 + 1 bins, NT per sweep
  double vmax[VM];
#pragma unroll
  for (int m = 0; m < VM; ++m) vmax[m] = 0.0;
  for (int fi = 0; fi < fpb; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpb + fi) * FR + fr;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    for (int j = tl; j < M; j += NT) {
      cx<TC> z = {(TC)0, (TC)0};
      if (valid && j < g.n) {
        const TC xw = (TC)view_sample(view, row, chunk, s0 + j) * wfull[j];
        const cx<TC> c = tb.chirp[j];
        z = {xw * c.x, xw * c.y};
      }
      buf[lp<TC>(j)] = z;
    }
    czt_core<TC, M, NT>(buf, tb, tl);
    if (valid) {
      const int64_t rowoff = (u * g.T + t) * g.FS;
#pragma unroll
      for (int m = 0; m < VM; ++m) {
        const int k = tl + NT * m;
        if (k >= g.F) continue;
        cx<TC> X = cmul(buf[lp<TC>(k)], tb.chirp[k]);
        // rfft of a real frame: bins 0 and n/2 are real (pocketfft returns exactly 0 there)
        if (k == 0 || 2 * k == g.n) X.y = (TC)0;
        const double Pk = (double)X.x * (double)X.x + (double)X.y * (double)X.y;
        vmax[m] = nanmax(vmax[m], Pk);
        if (P_out) P_out[rowoff + k] = Pk;
        if (mag_out) mag_out[rowoff + k] = sqrtf((float)(X.x * X.x + X.y * X.y));
        if (z_out) {
          const int64_t zo = ((u * g.T + t) * g.F + k) * 2;
          z_out[zo] = (double)X.x * z_scale;
          z_out[zo + 1] = (double)X.y * z_scale;
        }
      }
    }
    SG_PASS_SYNC();
  }
  if (pmax_bits) {
#pragma unroll
    for (int m = 0; m < VM; ++m) {
      const int k = tl + NT * m;
      if (k < g.F) atomicMax(&pmax_bits[u * g.FS + k], (unsigned long long)__double_as_longlong(vmax[m]));
    }
  }
}

// Apply + inverse (float32), as k_apply_istft: frame -> DFT_n -> X * M[t][k] (Hermitian: bin k and
// n - k share mask entry min(k, n-k)) -> inverse DFT_n -> real part * synthesis window (incl. 1/n)
// -> seg[u][t][0..n).  The inverse is the forward chirp-z of conj(Y), conjugated.
template <int M, int NT, int FR>
__global__ __launch_bounds__(NT* FR) void k_apply_istft_czt(View view, Geom g, CztTabs<float> tb,
                                                            const float* __restrict__ win_a,
                                                            const float* __restrict__ win_s,
                                                            const float* __restrict__ Mk, float* __restrict__ seg,
                                                            int fpb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tl = threadIdx.x % NT, fr = threadIdx.x / NT;
  cx<float>* buf = reinterpret_cast<cx<float>*>(smem) + (size_t)fr * lpn<float>(M);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  for (int fi = 0; fi < fpb; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpb + fi) * FR + fr;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    for (int j = tl; j < M; j += NT) {
      cx<float> z = {0.f, 0.f};
      if (valid && j < g.n) {
        const float xw = (float)view_sample(view, row, chunk, s0 + j) * win_a[j];
        const cx<float> c = tb.chirp[j];
        z = {xw * c.x, xw * c.y};
      }
      buf[lp<float>(j)] = z;
    }
    czt_core<float, M, NT>(buf, tb, tl);
    // every thread rewrites its own entries: Y[k] = X[k] * m, restaged as conj(Y[k]) * chirp[k]
    const float* Mrow = Mk + (u * g.T + (valid ? t : 0)) * g.FS;
    for (int j = tl; j < M; j += NT) {
      cx<float> z = {0.f, 0.f};
      if (valid && j < g.n) {
        const cx<float> c = tb.chirp[j];
        cx<float> X = cmul(buf[lp<float>(j)], c);
        const int kk = j < g.F ? j : g.n - j;
        const float m = Mrow[kk];
        if (j == 0 || 2 * j == g.n) X.y = 0.f;  // irfft ignores the imaginary part of DC / Nyquist
        const cx<float> Yc = {X.x * m, -X.y * m};
        z = cmul(Yc, c);
      }
      buf[lp<float>(j)] = z;
    }
    czt_core<float, M, NT>(buf, tb, tl);
    if (valid) {
      float* srow = seg + (u * g.T + t) * (int64_t)g.n;
      for (int j = tl; j < g.n; j += NT) {
        const cx<float> D = cmul(buf[lp<float>(j)], tb.chirp[j]);
        srow[j] = D.x * win_s[j];
      }
    }
    SG_PASS_SYNC();
  }
}

}  // namespace sg
