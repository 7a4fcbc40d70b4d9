// STFT / masked ISTFT for LONG frames: every n_fft the per-workgroup kernels do not
// cover -- powers of two from 16384 to 65536 and any other length from 4097 to 32768.  The reference accepts any
// n_fft (base.py:77-86; scipy.signal.stft -> rfft(n), istft -> irfft(n), scipy/signal/_spectral_py.py:2202,1689).
//
// A frame no longer fits a workgroup's LDS, so the length-M complex transform (M = n for a power of two, else the
// chirp-z convolution size M = pow2 >= 2n - 1 of czt.hpp) runs as a FOUR-STEP transform through HBM on a
// [frames][16][M2] work buffer, M = 16 * M2, element j = r * M2 + c:
//     columns:  A[k1][c]  = sum_r x[r][c] w_16^(r k1)          one thread per column, DFT16 in registers;
//     twiddle:  A[k1][c] *= w_M^(c k1)                          adjacent threads = adjacent columns: every row of the
//                                                               work buffer is read and written fully coalesced
//     rows:     X[k1 + 16 k2] = sum_c A[k1][c] w_M2^(c k2)      one workgroup per row, the Stockham core of fft_wave.hpp
// in place: bin k = k1 + 16 k2 ends up at position pos(k) = (k & 15) * M2 + (k >> 4).  The inverse runs the inverse
// steps in reverse order (rows, conjugate twiddle, columns) and so takes that permuted layout back to natural order;
// the chirp-z pair FFT_M -> x B -> IFFT_M therefore needs no reordering at all (B is stored permuted).
// Everything here computes in float64 (a 65536-point float32 chirp-z convolution would eat into the 1e-4 parity
// bar); fields handed to the rest of the pipeline keep their types (float64 power, float32 magnitude / mask / frames).
// Generality path: ~11 passes over a 16 M-byte-per-frame buffer per transform -- HBM-bound, not tuned.
#pragma once
#include "kernels.hpp"

namespace sg {
namespace big {

typedef cx<double> cd;

struct BigTabs {
  const cd* twM;    // M entries   w_M^j
  const cd* tw2;    // M2 entries  w_{2 M2}^k: master twiddle table of the M2-point row transform
  const cd* chirp;  // n entries   exp(-i pi j^2 / n)      (chirp-z only)
  const cd* bhat;   // M entries   FFT_M(b wrapped) / M, PERMUTED: bhat[pos(k)]   (chirp-z only)
  int M, M2, czt;
};

__device__ __forceinline__ int64_t pos_of(int k, int M2) { return (int64_t)(k & 15) * M2 + (k >> 4); }

template <bool INV>
__device__ __forceinline__ void dft16(cd* v) {
  cd e[8], o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
  dft8<INV>(e);
  dft8<INV>(o);
  // w_16^k, k = 0..7
  const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
  const double wc[8] = {1.0, c1, h, s1, 0.0, -s1, -h, -c1};
  const double ws[8] = {0.0, s1, h, c1, 1.0, c1, h, s1};   // sin(2 pi k / 16)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const cd w = {wc[k], INV ? ws[k] : -ws[k]};
    const cd t = cmul(o[k], w);
    v[k] = cadd(e[k], t);
    v[k + 8] = csub(e[k], t);
  }
}

// Stage frames [f0, f0 + nf) of the flattened (unit, frame) index: W[f][j] = x w (x chirp), zero above n.
__global__ __launch_bounds__(256) void k_big_frames(View view, Geom g, BigTabs tb, const double* __restrict__ wfull,
                                                    cd* __restrict__ W, int64_t f0, int64_t nf) {
  const int64_t f = blockIdx.y;
  if (f >= nf) return;
  const int64_t fl = f0 + f, u = fl / g.T, t = fl - u * g.T;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  const int64_t s0 = t * g.H - g.padL;
  cd* Wf = W + f * (int64_t)tb.M;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < tb.M; j += gridDim.x * 256) {
    cd z = {0.0, 0.0};
    if (j < g.n) {
      const double xw = view_sample(view, row, chunk, s0 + j) * wfull[j];
      if (tb.czt) {
        const cd c = tb.chirp[j];
        z = {xw * c.x, xw * c.y};
      } else {
        z.x = xw;
      }
    }
    Wf[j] = z;
  }
}

// columns (+ twiddle): forward = DFT16 then w_M^(c k1); inverse = conj twiddle then inverse DFT16
template <bool INV>
__global__ __launch_bounds__(256) void k_big_cols(cd* __restrict__ W, BigTabs tb, int64_t nf) {
  const int64_t f = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (f >= nf || c >= tb.M2) return;
  cd* Wf = W + f * (int64_t)tb.M + c;
  cd v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = Wf[(int64_t)r * tb.M2];
  if (INV) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      cd w = tb.twM[(int64_t)c * k];
      w.y = -w.y;
      v[k] = cmul(v[k], w);
    }
  }
  dft16<INV>(v);
  if (!INV) {
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tb.twM[(int64_t)c * k]);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) Wf[(int64_t)r * tb.M2] = v[r];
}

// rows: one workgroup per (frame, k1) row of M2 contiguous elements, in LDS.  FUSE_B: the multiplication by the (permuted)
// chirp-z kernel sits between a forward and an inverse row transform, and the inverse row
// transform of a row needs only that row: forward rows -> x B -> inverse rows in one launch.
template <int M2, bool INV, bool FUSE_B>
__global__ __launch_bounds__(256) void k_big_rows(cd* __restrict__ W, BigTabs tb, int64_t nf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* buf = reinterpret_cast<cd*>(smem);
  const int64_t rowi = blockIdx.x;          // f * 16 + k1
  if (rowi >= nf * 16) return;
  cd* Wr = W + rowi * (int64_t)M2;
  const int tl = threadIdx.x;
  for (int j = tl; j < M2; j += 256) buf[lp<double>(j)] = Wr[j];
  SG_PASS_SYNC();
  wave_fft<double, M2, INV, 256>(buf, tb.tw2, tl);
  if constexpr (FUSE_B) {
    // forward rows done: this row holds bins k1 + 16 k2 at k2; x B (permuted the same way), then straight back
    const cd* br = tb.bhat + (rowi & 15) * (int64_t)M2;
    for (int j = tl; j < M2; j += 256) buf[lp<double>(j)] = cmul(buf[lp<double>(j)], br[j]);
    SG_PASS_SYNC();
    wave_fft<double, M2, true, 256>(buf, tb.tw2, tl);
  }
  for (int j = tl; j < M2; j += 256) Wr[j] = buf[lp<double>(j)];
}

// bin k of frame f after the forward transform (power-of-two frames: permuted layout; chirp-z: natural order,
// still to be multiplied by the chirp)
__device__ __forceinline__ cd big_bin(const cd* Wf, const BigTabs& tb, const Geom& g, int k) {
  cd X;
  if (tb.czt) X = cmul(Wf[k], tb.chirp[k]);
  else X = Wf[pos_of(k, tb.M2)];
  if (k == 0 || 2 * k == g.n) X.y = 0.0;   // rfft of a real frame: pocketfft returns exactly 0 there
  return X;
}

// outputs as k_stft (kernels.hpp): power / magnitude / X itself, plus the per-(unit, band) maximum
__global__ __launch_bounds__(256) void k_big_out(const cd* __restrict__ W, Geom g, BigTabs tb, int64_t f0, int64_t nf,
                                                 double* __restrict__ P_out, float* __restrict__ mag_out,
                                                 double* __restrict__ z_out, double z_scale,
                                                 unsigned long long* __restrict__ pmax_bits) {
  const int64_t f = blockIdx.y;
  if (f >= nf) return;
  const int64_t fl = f0 + f, u = fl / g.T;
  const cd* Wf = W + f * (int64_t)tb.M;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < g.F; k += gridDim.x * 256) {
    const cd X = big_bin(Wf, tb, g, k);
    double Pk = X.x * X.x + X.y * X.y;
    if (Pk != Pk) Pk = (double)NAN;   // canonical NaN: wins the bit-pattern maximum
    if (P_out) P_out[fl * g.FS + k] = Pk;
    if (mag_out) mag_out[fl * g.FS + k] = (float)sqrt(Pk);
    if (z_out) {
      z_out[(fl * g.F + k) * 2] = X.x * z_scale;
      z_out[(fl * g.F + k) * 2 + 1] = X.y * z_scale;
    }
    if (pmax_bits) atomicMax(&pmax_bits[u * g.FS + k], (unsigned long long)__double_as_longlong(Pk));
  }
}

// X -> Y = X * mask (Hermitian: bins k and n - k share mask entry min(k, n - k)), staged as the input of the
// inverse transform.  Power-of-two frames: Y over all n bins in the permuted layout (bins above n/2 are the
// conjugates of their mirror bins); chirp-z: conj(Y[k]) * chirp[k] in natural order, zero above n (the inverse DFT
// is the forward chirp-z of conj(Y), conjugated).  In place except for the mirrored bins, hence the second buffer.
template <typename TM>
__global__ __launch_bounds__(256) void k_big_mask(const cd* __restrict__ W, cd* __restrict__ W2, Geom g, BigTabs tb,
                                                  int64_t f0, int64_t nf, const TM* __restrict__ Mk) {
  const int64_t f = blockIdx.y;
  if (f >= nf) return;
  const int64_t fl = f0 + f;
  const cd* Wf = W + f * (int64_t)tb.M;
  cd* Of = W2 + f * (int64_t)tb.M;
  const TM* Mrow = Mk + fl * g.FS;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < tb.M; j += gridDim.x * 256) {
    cd z = {0.0, 0.0};
    if (j < g.n) {
      const int kk = j < g.F ? j : g.n - j;
      cd X = big_bin(Wf, tb, g, kk);
      if (j >= g.F) X.y = -X.y;            // X[n - k] = conj X[k]
      const double m = (double)Mrow[kk];
      if (tb.czt) {
        const cd Yc = {X.x * m, -X.y * m};
        z = cmul(Yc, tb.chirp[j]);
      } else {
        z = {X.x * m, X.y * m};
      }
    }
    if (tb.czt) Of[j] = z;
    else if (j < g.n) Of[pos_of(j, tb.M2)] = z;
  }
}

// time-domain frame * synthesis window (incl. 1 / n) -> seg[u][t][0..n) for k_ola
template <typename TS>
__global__ __launch_bounds__(256) void k_big_seg(const cd* __restrict__ W, Geom g, BigTabs tb, int64_t f0, int64_t nf,
                                                 const double* __restrict__ wfull, TS* __restrict__ seg) {
  const int64_t f = blockIdx.y;
  if (f >= nf) return;
  const int64_t fl = f0 + f;
  const cd* Wf = W + f * (int64_t)tb.M;
  const double inv_n = 1.0 / (double)g.n;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < g.n; j += gridDim.x * 256) {
    double y;
    if (tb.czt) y = cmul(Wf[j], tb.chirp[j]).x;   // Re conj(D) = Re D
    else y = Wf[j].x;
    seg[fl * (int64_t)g.n + j] = (TS)(y * wfull[j] * inv_n);
  }
}

}  // namespace big
}  // namespace sg
