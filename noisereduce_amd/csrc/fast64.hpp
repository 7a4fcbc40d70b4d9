// float64 power spectrogram at the default geometry (n_fft = 1024, hop 256) on the register FFT core.
//
// The float64 statistics (noise clip of the stationary gate, every row of TorchGate: stationary.py:61-81,
// torchgate.py:140-160) need |X|^2 in double precision.  k_stft<double> (kernels.hpp) runs one frame per wavefront
// through a nine-pass LDS Stockham transform; this kernel is the float64 sibling of fast::k_mag_fast: one wavefront
// = 4 frames, lane (g, c) holds 32 complex points of frame g in registers (512 = 32 x 16), ONE exchange through
// LDS (two half-size phases, 16-byte elements, XOR-swizzled so that both the column writes and the row reads touch
// every bank once), real-FFT split in registers, powers stored in natural bin order.  Same result as k_stft<double>
// up to rounding (a different, shorter chain of float64 operations).
#pragma once
#include "fastpath.hpp"

namespace sg {
namespace fast {

typedef cx<double> cd;

// cos(2 pi j / 32), j = 0..8
__device__ constexpr double C32D[9] = {1.0,
                                       0.98078528040323044913,
                                       0.92387953251128675613,
                                       0.83146961230254523708,
                                       0.70710678118654752440,
                                       0.55557023301960222474,
                                       0.38268343236508977173,
                                       0.19509032201612826785,
                                       0.0};
template <int R>
__device__ __forceinline__ constexpr double twcd(int k) {
  int j = k * (32 / R);
  return j <= 8 ? C32D[j] : -C32D[16 - j];
}
template <int R>
__device__ __forceinline__ constexpr double twsd(int k) {
  int j = k * (32 / R);
  return j <= 8 ? C32D[8 - j] : C32D[j - 8];
}

// In-register forward DFT of R points, IN PLACE (iterative decimation in time): input in bit-reversed order
// (v[brev<R>(r)] = x[r] -- free: the indices are compile-time constants), output in natural order.  The recursive
// even/odd form of fastpath.hpp's dft_reg keeps copies of both halves alive; in float64 (32 points = 128 VGPRs)
// that does not fit a 256-register wave.
template <int R>
__host__ __device__ constexpr int brev(int i) {
  int r = 0;
  for (int b = 1; b < R; b <<= 1) {
    r = (r << 1) | (i & 1);
    i >>= 1;
  }
  return r;
}
template <int R, int LEN = 2>
__device__ __forceinline__ void dft_inplace_d(cd* v) {
  if constexpr (LEN <= R) {
    constexpr int H = LEN / 2;
#pragma unroll
    for (int base = 0; base < R; base += LEN) {
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const cd a = v[base + j], b = v[base + j + H];
        if (j == 0) {
          v[base + j] = cadd(a, b);
          v[base + j + H] = csub(a, b);
        } else if (2 * j == H) {
          const cd t = rot90<false>(b);
          v[base + j] = cadd(a, t);
          v[base + j + H] = csub(a, t);
        } else {
          // p = a + w b (4 FMAs), q = a - w b = 2 a - p (2 FMAs); w = w_LEN^j = c - i s
          const double c = twcd<32>(j * (32 / LEN)), s = twsd<32>(j * (32 / LEN));
          cd p;
          p.x = fma(b.x, c, fma(b.y, s, a.x));
          p.y = fma(b.y, c, fma(-b.x, s, a.y));
          v[base + j] = p;
          v[base + j + H] = {fma(2.0, a.x, -p.x), fma(2.0, a.y, -p.y)};
        }
      }
    }
    dft_inplace_d<R, LEN * 2>(v);
  }
}

constexpr int FSLOTS_D = 256;  // complex slots of one frame's half-size exchange slice (16 rows x 16 columns)

// v[brev<32>(r)] = z[c + 16 r]  ->  v[k2] = Zc[row1 + 32 k2], v[16 + k2] = Zc[row2 + 32 k2]
// (fft512_fwd_half in double)
__device__ __forceinline__ void fft512_fwd_half_d(cd* v, cd* fb, const cd* tw512, int c) {
  __builtin_amdgcn_sched_barrier(0);
  dft_inplace_d<32>(v);
  __builtin_amdgcn_sched_barrier(0);
  // twiddle and store FOUR rows at a time (sched_barriers): left alone, the scheduler issues all 31 twiddle loads up
  // front -- 124 more registers beside the 128 of the spectra -- and the kernels spill (k_apply_fast64: 64 VGPRs, a GB of
  // scratch traffic per ten minutes of audio, round 5).  Same arithmetic in the same order: bit-identical results.
  // element (row k1, column c) lives in slot k1 * 16 + (c ^ (k1 & 15)): a row is one 256-byte bank line
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    if (k1 != 0) v[k1] = cmul(v[k1], tw512[k1 * 16 + c]);
    fb[k1 * 16 + (c ^ k1)] = v[k1];
    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_sync();
  {
    const int row = row1(c);
#pragma unroll
    for (int h = 0; h < 16; ++h) v[brev<16>(h)] = fb[row * 16 + (h ^ row)];
  }
  wave_lds_sync();
#pragma unroll
  for (int k1 = 16; k1 < 32; ++k1) {
    v[k1] = cmul(v[k1], tw512[k1 * 16 + c]);
    fb[(k1 - 16) * 16 + (c ^ (k1 - 16))] = v[k1];
    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_sync();
  {
    const int row = row2(c) - 16;
#pragma unroll
    for (int h = 0; h < 16; ++h) v[16 + brev<16>(h)] = fb[row * 16 + (h ^ row)];
  }
  wave_lds_sync();
  dft_inplace_d<16>(v);
  __builtin_amdgcn_sched_barrier(0);
  dft_inplace_d<16>(v + 16);
  __builtin_amdgcn_sched_barrier(0);
}

struct Pow64Args {
  View view;
  Geom g;
  const double* win;                  // analysis window padded to n_fft (double[1024])
  const cd* tw1024;                   // w_1024^k, k = 0..511
  double* P;                          // [units][T][FS] raw power |X|^2
  unsigned long long* pmax_bits;      // optional: per-(unit, band) maximum (atomic max on the bit pattern)
};

// TS: type of the staged samples -- float (float32 / int16 recordings: exact) or double (int32 / float64 recordings: the
// float64 pipeline of a recording whose samples have no exact float32 copy; round 5)
template <int WAVES, bool PMAX, typename TS = float>
__global__ __launch_bounds__(WAVES * 64, 2) void k_power_fast64(Pow64Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* tw512 = reinterpret_cast<cd*>(smem);   // [32][16]: w_512^(k1 c)
  cd* regions = tw512 + FN;
  cd* tw_lo = regions + WAVES * 4 * FSLOTS_D;   // w_1024^0..16 (the split's per-lane twiddles)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  {
    // (every load of the prologue issued before the first store: see stage_tables in fastpath.hpp)
    constexpr int K = (FN + WAVES * 64 - 1) / (WAVES * 64);
    cd t[K];
    const cd tl = A.tw1024[min(tid, 16)];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = min(tid + k * WAVES * 64, FN - 1);
      const int idx = 2 * (i >> 4) * (i & 15);   // w_512^j = w_1024^(2 j), 2 j < 1024
      t[k] = A.tw1024[idx & 511];
      if (idx >= 512) t[k] = {-t[k].x, -t[k].y};
    }
    if (tid < 17) tw_lo[tid] = tl;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = tid + k * WAVES * 64;
      if (i < FN) tw512[i] = t[k];
    }
  }
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  // Stage the workgroup's sample span (16 frames = 19 hops) and the float64 window in the exchange region: ONE
  // global round trip for everything the gather needs; samples outside the readable range are zero
  // (view_sample), so ragged frames take the same path.  float32 samples only (the host checks): staging is float32.
  constexpr bool DS = sizeof(TS) == 8;
  constexpr int NFB = 4 * WAVES, SPAN = (NFB - 1) * 256 + 1024, XPITCH = DS ? 264 : 288;
  constexpr int WOFF = DS ? 40960 : 24576;   // the float64 window behind the staged span
  static_assert((SPAN / 256) * XPITCH * (int)sizeof(TS) <= WOFF && WOFF + 8192 <= WAVES * 4 * FSLOTS_D * 16, "staging fits");
  TS* xs = reinterpret_cast<TS*>(regions);
  double* wl = reinterpret_cast<double*>(reinterpret_cast<char*>(regions) + WOFF);
  const int64_t tqb = (int64_t)blockIdx.x * NFB;
  {
    const int64_t s0b = tqb * 256 - G.padL;
    const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
    const bool blk_in = A.view.dtype == 0 && s0b >= 0 && s0b + SPAN <= A.view.Lp && gb >= A.view.lo &&
                        gb + SPAN <= A.view.hi;
    const float* sp = (const float*)A.view.x + row * A.view.stride + gb;
    {
      constexpr int KW = (512 + WAVES * 64 - 1) / (WAVES * 64);
      double2 w2[KW];
#pragma unroll
      for (int k = 0; k < KW; ++k) w2[k] = reinterpret_cast<const double2*>(A.win)[min(tid + k * WAVES * 64, 511)];
      if constexpr (!DS) {
        if (blk_in && (reinterpret_cast<uintptr_t>(sp) & 15) == 0) {
          stage_span_vec<WAVES * 64, SPAN, XPITCH>(xs, sp, tid);
        } else {
          for (int i = tid; i < SPAN; i += WAVES * 64)
            xs[(i >> 8) * XPITCH + (i & 255)] = (float)view_sample(A.view, row, chunk, s0b + i);
        }
      } else {
        for (int i = tid; i < SPAN; i += WAVES * 64)
          xs[(i >> 8) * XPITCH + (i & 255)] = view_sample(A.view, row, chunk, s0b + i);
      }
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const int i = tid + k * WAVES * 64;
        if (i < 512) reinterpret_cast<double2*>(wl)[i] = w2[k];
      }
    }
  }
  const int64_t tq = tqb + wave * 4;
  __syncthreads();
  const int64_t t = tq + g;
  const bool fvalid = t < G.T;
  cd* fb = regions + (wave * 4 + g) * FSLOTS_D;

  // gather: v[brev(r)] = (x[2c + 32r], x[2c + 32r + 1]) * window
  cd v[32];
  {
    const TS* xl = xs + (4 * wave + g) * XPITCH + 2 * c;
    const double2* wsrc = reinterpret_cast<const double2*>(wl + 2 * c);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const TS* xp = xl + (r >> 3) * XPITCH + 32 * (r & 7);
      const double2 w2 = wsrc[16 * r];
      v[brev<32>(r)] = {(double)xp[0] * w2.x, (double)xp[1] * w2.y};
    }
  }
  __syncthreads();   // every lane has its samples: the staging area becomes the exchange slices
  if (tq >= G.T) return;
  fft512_fwd_half_d(v, fb, tw512, c);

  // real-FFT split (see k_mag_fast): conjugate pair (a, b) = (Zc[k], Zc[512 - k]) -> 2 X[k] = E + w O,
  // 2 X[512 - k]^* = E - w O; raw power = |2X|^2 / 4
  const bool l0 = c == 0;
  const cd wlo = tw_lo[c];
  cd whi = wlo;
  {
    const cd w16 = tw_lo[16];
    if (l0) whi = {-w16.y, w16.x};
  }
  auto sel = [&](cd a0, cd a1) -> cd { return {l0 ? a0.x : a1.x, l0 ? a0.y : a1.y}; };
  double* prow = A.P + (u * G.T + (fvalid ? t : 0)) * (int64_t)G.FS;
  unsigned long long* mrow = PMAX ? A.pmax_bits + u * (int64_t)G.FS : nullptr;
  auto put = [&](int bin, double P4) {
    const double Pv = 0.25 * P4;
    if (fvalid) prow[bin] = Pv;
    if constexpr (PMAX) {
      // maximum over the wave's four frames first (lanes c, c + 16, c + 32, c + 48), one atomic per bin
      double m = fvalid ? Pv : 0.0;
      m = nanmax(m, __shfl_xor(m, 16));
      m = nanmax(m, __shfl_xor(m, 32));
      if (g == 0) atomicMax(&mrow[bin], (unsigned long long)__double_as_longlong(m));
    }
  };
  auto pair_power = [&](cd a, cd b, cd w, double& Pk, double& Pn) {
    const cd E = {a.x + b.x, a.y - b.y};
    const cd O = {a.y + b.y, b.x - a.x};
    const cd wO = cmul(w, O);
    const double px = E.x + wO.x, py = E.y + wO.y, qx = E.x - wO.x, qy = E.y - wO.y;
    Pk = px * px + py * py;
    Pn = qx * qx + qy * qy;
  };
  {
    double Pk, Pn;
    pair_power(v[0], v[31], wlo, Pk, Pn);
    const cd a = v[0];
    const double x0 = 2.0 * (a.x + a.y), xN = 2.0 * (a.x - a.y);
    const double P256 = 4.0 * (v[8].x * v[8].x + v[8].y * v[8].y);
    put(bin_of_entry(c, 0), l0 ? x0 * x0 : Pk);
    put(bin_of_entry(c, 31), l0 ? P256 : Pn);
    // bin 512 belongs to lane 0 only; the other lanes take part in the shuffles of put() with a harmless
    // duplicate of their own entry 0
    put(l0 ? 512 : bin_of_entry(c, 0), l0 ? xN * xN : Pk);
  }
#pragma unroll
  for (int sl = 1; sl < 16; ++sl) {
    const cd a = sl < 8 ? v[sl] : sel(v[8 + sl], v[sl]);
    const cd b = sl < 8 ? sel(v[16 - sl], v[31 - sl]) : sel(v[39 - sl], v[31 - sl]);
    const cd ws = sl < 8 ? wlo : whi;
    const double cc = twcd<32>(sl), ss = twsd<32>(sl);
    const cd w = {ws.x * cc + ws.y * ss, ws.y * cc - ws.x * ss};   // ws * w_32^sl... (c - i s)
    double Pk, Pn;
    pair_power(a, b, w, Pk, Pn);
    put(bin_of_entry(c, sl), Pk);
    put(bin_of_entry(c, 31 - sl), Pn);
    __builtin_amdgcn_sched_barrier(0);   // one slot at a time: hoisted store addresses would spill
  }
}

}  // namespace fast
}  // namespace sg
