// Mixed-radix frames (round 6; SURVEY.md section 8 row f3, VERDICT r5 "missing" item 4).
//
// The reference hands any n_fft to scipy's pocketfft (stationary.py:87-93), which transforms 400, 1000, 1536 or 3000
// points as fast as 512 or 1024.  Up to round 5 every frame length that was not a power of two went through the
// chirp-z kernels of czt.hpp -- two transforms of M >= 2 n complex points where a real frame of n = 2 N samples needs one
// of N: 15 - 70 x slower than the neighbouring powers of two.  Here: a Stockham autosort transform whose radix schedule
// is a RUN-TIME argument (8 / 4 / 2 / 5 / 3, plus a direct small-prime butterfly for 7, 11, 13), on the same real-packed
// layout, master twiddle table (w_2N^k, k < N: it serves the passes -- w_N^t = w_2N^(2t) -- and the real-FFT split) and
// surrounding kernels as the power-of-two LDS path (kernels.hpp k_stft / k_apply_istft, fused.hpp k_stft_bits /
// k_decide_lds), whose outputs these kernels reproduce field for field.  Chirp-z remains for odd n, for n with a prime
// factor above 13 and for N = n / 2 > 2048.
//
// One pass of radix R at stride S (the product of the radices before it), NB = N / R butterflies:
//     butterfly i = p S + q (q < S):   v[j] = x[i + j NB];   v = DFT_R(v);   v[k] *= w_N^(p S k);   y[p S R + q + S k] = v[k]
// -- the formula of FftPass (fft_wave.hpp) with `&` replaced by `%`.  The passes PING-PONG between two buffers: the
// in-place form needs every read of a pass in registers before its first write, i.e. compile-time trip counts.
#pragma once
#include "fused.hpp"

namespace sg {

constexpr int MR_MAXP = 8;        // passes (2^11 = 2048 = 8 8 8 4; 2 3 5 7 11 13 > 2048)
constexpr int MR_MAXR = 13;       // largest radix
struct MrPlan {
  int N;                          // complex length (n_fft / 2)
  int np;                         // passes
  unsigned char R[MR_MAXP];       // radices, in pass order; product = N
};

template <bool INV, typename T>
__device__ __forceinline__ void dft3(cx<T>* v) {
  const T s = (T)0.86602540378443864676;
  const cx<T> a = v[0], t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
  const cx<T> m = {a.x - (T)0.5 * t.x, a.y - (T)0.5 * t.y};
  const cx<T> u = INV ? cx<T>{-s * d.y, s * d.x} : cx<T>{s * d.y, -s * d.x};   // -+ i s d
  v[0] = cadd(a, t);
  v[1] = cadd(m, u);
  v[2] = csub(m, u);
}

template <bool INV, typename T>
__device__ __forceinline__ void dft5(cx<T>* v) {
  const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;
  const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
  const cx<T> x0 = v[0];
  const cx<T> a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
  const cx<T> m1 = {x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y};
  const cx<T> m2 = {x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y};
  const cx<T> n1 = {s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y};
  const cx<T> n2 = {s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y};
  // forward: y1 = m1 - i n1, y4 = m1 + i n1, y2 = m2 - i n2, y3 = m2 + i n2;  -i n = (n.y, -n.x)
  const cx<T> r1 = INV ? cx<T>{-n1.y, n1.x} : cx<T>{n1.y, -n1.x};
  const cx<T> r2 = INV ? cx<T>{-n2.y, n2.x} : cx<T>{n2.y, -n2.x};
  v[0] = {x0.x + a1.x + a2.x, x0.y + a1.y + a2.y};
  v[1] = cadd(m1, r1);
  v[4] = csub(m1, r1);
  v[2] = cadd(m2, r2);
  v[3] = csub(m2, r2);
}

// w_N^t, t < N, from the master table tw[k] = w_2N^k, k < N (conjugated for the inverse transform); N at run time
template <bool INV, typename T>
__device__ __forceinline__ cx<T> mr_tw(const cx<T>* tw, int N, int t) {
  const int k = 2 * t;
  cx<T> w = tw[k < N ? k : k - N];
  if (k >= N) { w.x = -w.x; w.y = -w.y; }
  if (INV) w.y = -w.y;
  return w;
}

template <int R, bool INV, typename T, int NT>
__device__ __forceinline__ void mr_pass(const cx<T>* __restrict__ x, cx<T>* __restrict__ y, const cx<T>* tw, int N, int S,
                                        bool last, int lane) {
  const int NB = N / R;
  for (int i = lane; i < NB; i += NT) {
    const int q = i % S, base = i - q;
    cx<T> v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = x[lp<T>(i + j * NB)];
    const int o = base * R + q;
    if constexpr (R == 7 || R == 11 || R == 13) {
      // direct DFT of a small prime length: y[k] = sum_j x[j] w_R^(j k), w_R^m = w_N^(m N / R), one output at a time
      // (rare sizes: R^2 table lookups instead of R more live values per lane)
      const int step = N / R;
#pragma unroll 1
      for (int k = 0; k < R; ++k) {
        cx<T> acc = v[0];
        int jk = 0;
#pragma unroll
        for (int j = 1; j < R; ++j) {
          jk += k;
          if (jk >= R) jk -= R;
          acc = cadd(acc, cmul(v[j], mr_tw<INV>(tw, N, jk * step)));
        }
        if (!last && k > 0) acc = cmul(acc, mr_tw<INV>(tw, N, base * k));
        y[lp<T>(o + S * k)] = acc;
      }
    } else {
      if constexpr (R == 8 || R == 4 || R == 2) dftR<R, INV>(v);
      else if constexpr (R == 3) dft3<INV>(v);
      else dft5<INV>(v);
      if (!last) {   // the last pass has p == 0: all twiddles are 1
#pragma unroll
        for (int k = 1; k < R; ++k) v[k] = cmul(v[k], mr_tw<INV>(tw, N, base * k));
      }
#pragma unroll
      for (int k = 0; k < R; ++k) y[lp<T>(o + S * k)] = v[k];
    }
  }
}

// Complex transform of a[0 .. N) (unnormalised; INV: exp(+i ...)).  Returns the buffer that holds the result (a or b).
// SY as in fft_wave.hpp: 1 = the team is (part of) one wavefront and the buffers are its own, else a workgroup barrier.
template <typename T, bool INV, int NT, int SY>
__device__ __forceinline__ cx<T>* mr_fft(cx<T>* a, cx<T>* b, const cx<T>* tw, const MrPlan& pl, int lane) {
  int S = 1;
  for (int p = 0; p < pl.np; ++p) {
    const int R = pl.R[p];
    const bool last = p + 1 == pl.np;
    switch (R) {
      case 8: mr_pass<8, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
      case 4: mr_pass<4, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
      case 2: mr_pass<2, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
      case 5: mr_pass<5, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
      case 3: mr_pass<3, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
      case 7: mr_pass<7, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
      case 11: mr_pass<11, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
      default: mr_pass<13, INV, T, NT>(a, b, tw, pl.N, S, last, lane); break;
    }
    team_sync<SY>();
    S *= R;
    cx<T>* t = a; a = b; b = t;
  }
  return a;
}

template <typename T>
__device__ __forceinline__ void mr_stage_twiddles(cx<T>* tw, const cx<T>* __restrict__ tw_g, int N, int tid, int nthr) {
  for (int i0 = 0; i0 < N; i0 += 8 * nthr) {   // eight loads of a thread in flight before its first store
    cx<T> t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = tw_g[min(i0 + tid + k * nthr, N - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + tid + k * nthr;
      if (i < N) tw[i] = t[k];
    }
  }
}

constexpr int MR_MAXM = 9;    // bins per thread: NT = 64 for N <= 512 (9 sweeps cover N + 1 bins), NT = 256 up to N = 2048 (9)

// ---------------------------------------------------------------------------------------
// Forward STFT: k_stft (kernels.hpp) with the frame length at run time.  NT threads per frame (64: one wavefront; 256:
// the workgroup), blockDim.x / NT frames in flight per workgroup, FPW frames per team.
// ---------------------------------------------------------------------------------------
template <typename TC, int NT>
__global__ __launch_bounds__(256, sizeof(TC) == 8 ? 2 : 3) void k_stft_mr(View view, Geom g, MrPlan pl, const cx<TC>* __restrict__ tw_g,
                                                 const TC* __restrict__ wfull, double* __restrict__ P_out,
                                                 float* __restrict__ mag_out, double* __restrict__ z_out, double z_scale,
                                                 unsigned long long* __restrict__ pmax_bits, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<TC>* tw = reinterpret_cast<cx<TC>*>(smem);
  const int lane = threadIdx.x % NT, wave = threadIdx.x / NT;
  cx<TC>* buf0 = tw + N + (size_t)(2 * wave) * lpn<TC>(N);   // (lpn is a constexpr function of its argument: fine at run time)
  cx<TC>* buf1 = buf0 + lpn<TC>(N);
  mr_stage_twiddles(tw, tw_g, N, (int)threadIdx.x, (int)blockDim.x);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  double vmax[MR_MAXM];
#pragma unroll
  for (int m = 0; m < MR_MAXM; ++m) vmax[m] = 0.0;
  for (int fi = 0; fi < fpw; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpw + fi) * teams + wave;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    const float* fp = valid ? frame_ptr_f32(view, row, chunk, s0, 2 * N) : nullptr;  // team-uniform
    if (fp) {
      for (int j = lane; j < N; j += NT)
        buf0[lp<TC>(j)] = {(TC)fp[2 * j] * wfull[2 * j], (TC)fp[2 * j + 1] * wfull[2 * j + 1]};
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<TC> z = {(TC)0, (TC)0};
        if (valid) {
          z.x = (TC)view_sample(view, row, chunk, s0 + 2 * j) * wfull[2 * j];
          z.y = (TC)view_sample(view, row, chunk, s0 + 2 * j + 1) * wfull[2 * j + 1];
        }
        buf0[lp<TC>(j)] = z;
      }
    }
    team_sync<SY>();
    const cx<TC>* Z = mr_fft<TC, false, NT, SY>(buf0, buf1, tw, pl, lane);
    if (valid) {
      const int64_t rowoff = (u * g.T + t) * g.FS;
#pragma unroll
      for (int m = 0; m < MR_MAXM; ++m) {
        const int k = lane + NT * m;
        if (k > N) continue;
        const cx<TC> a = Z[lp<TC>(k == N ? 0 : k)];
        const cx<TC> b = Z[lp<TC>((k == 0 || k == N) ? 0 : N - k)];
        const cx<TC> w = tw[k == N ? 0 : k];
        const cx<TC> X = rfft_bin(a, b, w, k, N);
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
    team_sync<SY>();
  }
  if (pmax_bits) {
#pragma unroll
    for (int m = 0; m < MR_MAXM; ++m) {
      const int k = lane + NT * m;
      if (k <= N) atomicMax(&pmax_bits[u * g.FS + k], (unsigned long long)__double_as_longlong(vmax[m]));
    }
  }
}

// ---------------------------------------------------------------------------------------
// float64 STFT + decision: k_stft_bits (fused.hpp).  MODE 0: band maxima of the units whose floor may be live; MODE 1: bits.
// ---------------------------------------------------------------------------------------
template <int MODE, int NT>
__global__ __launch_bounds__(256, 2) void k_stft_bits_mr(View view, Geom g, MrPlan pl, const cx<double>* __restrict__ tw_g,
                                                      const double* __restrict__ wfull, ThreshConsts tc, double mag_scale,
                                                      double top_db, unsigned long long* __restrict__ pmax_bits,
                                                      unsigned long long* __restrict__ bits, int wpr, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<double>* tw = reinterpret_cast<cx<double>*>(smem);
  double* sT2 = reinterpret_cast<double*>(tw + N + (size_t)(2 * teams) * lpn<double>(N));  // [N + 1] compare constants
  const int lane = threadIdx.x % NT, wave = threadIdx.x / NT;
  cx<double>* buf0 = tw + N + (size_t)(2 * wave) * lpn<double>(N);
  cx<double>* buf1 = buf0 + lpn<double>(N);
  const int64_t u = blockIdx.y;
  const int need = need_of(tc, u);
  const bool floor_live = need == 1;
  if (MODE == 0 && !floor_live) return;  // whole block: uniform
  mr_stage_twiddles(tw, tw_g, N, (int)threadIdx.x, (int)blockDim.x);
  if (MODE == 1) {
    for (int i = threadIdx.x; i <= N; i += blockDim.x) {
      double t2 = tc.T2[i];
      if (floor_live) {
        const double fl = cell_db(tc.pmax[u * g.FS + i], mag_scale) - top_db;
        if (fl > tc.thresh[i]) t2 = -1.0;
      }
      if (need == 2) t2 = T2_NEVER;
      sT2[i] = t2;
    }
  }
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  double vmax[MR_MAXM];
#pragma unroll
  for (int m = 0; m < MR_MAXM; ++m) vmax[m] = 0.0;
  for (int fi = 0; fi < fpw; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpw + fi) * teams + wave;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    for (int j = lane; j < N; j += NT) {
      cx<double> z = {0.0, 0.0};
      if (valid) {
        z.x = view_sample(view, row, chunk, s0 + 2 * j) * wfull[2 * j];
        z.y = view_sample(view, row, chunk, s0 + 2 * j + 1) * wfull[2 * j + 1];
      }
      buf0[lp<double>(j)] = z;
    }
    team_sync<SY>();
    const cx<double>* Z = mr_fft<double, false, NT, SY>(buf0, buf1, tw, pl, lane);
    unsigned long long* brow = bits + ((u * g.T + t) * (int64_t)wpr);
#pragma unroll
    for (int m = 0; m < MR_MAXM; ++m) {
      const int k = lane + NT * m;
      if (NT * m > N) break;   // (team-uniform: the ballot below is the whole wavefront's)
      bool pred = false;
      if (k <= N) {
        const cx<double> a = Z[lp<double>(k == N ? 0 : k)];
        const cx<double> b = Z[lp<double>((k == 0 || k == N) ? 0 : N - k)];
        const cx<double> w = tw[k == N ? 0 : k];
        const cx<double> X = rfft_bin(a, b, w, k, N);
        const double P = X.x * X.x + X.y * X.y;
        if (MODE == 0) vmax[m] = fmax(vmax[m], valid ? P : 0.0);
        else pred = P > sT2[k];
      }
      if (MODE == 1) {
        // a hardware wave covers 64 consecutive bins: its ballot is word k / 64 of the frame's row
        const unsigned long long word = __ballot(pred);
        if (valid && (lane & 63) == 0 && k <= N) brow[k >> 6] = word;
      }
    }
    team_sync<SY>();
  }
  if (MODE == 0) {
#pragma unroll
    for (int m = 0; m < MR_MAXM; ++m) {
      const int k = lane + NT * m;
      if (k <= N) atomicMax(&pmax_bits[u * g.FS + k], (unsigned long long)__double_as_longlong(vmax[m]));
    }
  }
}

// ---------------------------------------------------------------------------------------
// float32 STFT + decision with exact float64 refinement: k_decide_lds (fused.hpp).  One wavefront per frame (NT = 64) or
// the workgroup per frame (NT = 256, N > 1024); bits identical to k_stft_bits_mr<1>.
// ---------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 3) void k_decide_mr(View view, Geom g, MrPlan pl, const cx<float>* __restrict__ tw_g,
                                                   const float* __restrict__ win32, const cx<double>* __restrict__ tw64,
                                                   const double* __restrict__ win64, ThreshConsts tc, double mag_scale,
                                                   double top_db, unsigned long long* __restrict__ bits, int wpr, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<float>* tw = reinterpret_cast<cx<float>*>(smem);
  float* sT2 = reinterpret_cast<float*>(tw + N + (size_t)(2 * teams) * N);  // [N + 1] compare constants (float32)
  float* s_red = sT2 + N + 1;                                               // [4] per-wave partial norms (NT = 256)
  const int lane64 = threadIdx.x & 63, lane = threadIdx.x % NT, team = threadIdx.x / NT;
  cx<float>* buf0 = tw + N + (size_t)(2 * team) * N;
  cx<float>* buf1 = buf0 + N;
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
  mr_stage_twiddles(tw, tw_g, N, (int)threadIdx.x, (int)blockDim.x);
  for (int i = threadIdx.x; i <= N; i += blockDim.x) sT2[i] = t2_to_f32(t2eff(i), 1.0);
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  for (int fi = 0; fi < fpw; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpw + fi) * teams + team;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    float nrm2 = 0.f;
    const float* fp = valid ? frame_ptr_f32(view, row, chunk, s0, 2 * N) : nullptr;  // team-uniform
    if (fp) {
      for (int j = lane; j < N; j += NT) {
        const cx<float> z = {fp[2 * j] * win32[2 * j], fp[2 * j + 1] * win32[2 * j + 1]};
        nrm2 += z.x * z.x + z.y * z.y;
        buf0[j] = z;
      }
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<float> z = {0.f, 0.f};
        if (valid) {
          z.x = (float)view_sample(view, row, chunk, s0 + 2 * j) * win32[2 * j];
          z.y = (float)view_sample(view, row, chunk, s0 + 2 * j + 1) * win32[2 * j + 1];
        }
        nrm2 += z.x * z.x + z.y * z.y;
        buf0[j] = z;
      }
    }
    for (int off = 32; off > 0; off >>= 1) nrm2 += __shfl_xor(nrm2, off);
    if constexpr (NT == 256) {   // the frame's four wavefronts: partial norms through LDS (the pass sync below orders them)
      if (lane64 == 0) s_red[threadIdx.x >> 6] = nrm2;
      __syncthreads();
      nrm2 = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
    // 2 delta^2 = 2 * 2^-32 * nrm2; a silent frame (nrm2 == 0) has no ambiguous cells
    const float d2 = nrm2 > 0.f ? 2.0f * 2.3283064e-10f * nrm2 : -1.0f;
    team_sync<SY>();
    const cx<float>* Z = mr_fft<float, false, NT, SY>(buf0, buf1, tw, pl, lane);
    unsigned long long* brow = bits + ((u * g.T + t) * (int64_t)wpr);
#pragma unroll 1
    for (int m = 0; NT * m <= N; ++m) {
      const int k = lane + NT * m;
      bool pred = false, amb = false;
      if (k <= N) {
        const cx<float> a = Z[k == N ? 0 : k];
        const cx<float> b = Z[(k == 0 || k == N) ? 0 : N - k];
        const cx<float> w = tw[k == N ? 0 : k];
        const cx<float> X = rfft_bin(a, b, w, k, N);
        const float P = X.x * X.x + X.y * X.y;
        const float T = sT2[k];
        const float diff = P - T;
        pred = diff > 0.f;
        amb = valid && diff * diff <= d2 * (P + T);
      }
      // exact re-evaluation, one cell at a time, the cell's wavefront cooperating (wave-uniform loop)
      unsigned long long pending = __ballot(amb);
      while (pending) {
        const int src = __ffsll((long long)pending) - 1;
        pending &= pending - 1;
        const int ks = k - lane64 + src;    // (lane64 - lane is the wavefront's offset inside a 256-thread team)
        double re = 0.0, im = 0.0;
        for (int i = lane64; i < 2 * N; i += 64) {
          const double xv = view_sample(view, row, chunk, s0 + i) * win64[i];
          const int j = (int)(((int64_t)ks * i) % (2 * N));
          cx<double> w = tw64[j < N ? j : j - N];
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
      if (valid && lane64 == 0 && k <= N) brow[k >> 6] = word;
    }
    team_sync<SY>();
  }
}

// ---------------------------------------------------------------------------------------
// Apply + inverse: k_apply_istft (kernels.hpp).  mask = M (float field) or K16 * kscale (integer weight sums).
// ---------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 3) void k_apply_istft_mr(View view, Geom g, MrPlan pl, const cx<float>* __restrict__ tw_g,
                                                        const float* __restrict__ win_a, const float* __restrict__ win_s,
                                                        const float* __restrict__ M, float* __restrict__ seg,
                                                        const unsigned short* __restrict__ K16, float kscale, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<float>* tw = reinterpret_cast<cx<float>*>(smem);
  const int lane = threadIdx.x % NT, wave = threadIdx.x / NT;
  cx<float>* buf0 = tw + N + (size_t)(2 * wave) * N;
  cx<float>* buf1 = buf0 + N;
  mr_stage_twiddles(tw, tw_g, N, (int)threadIdx.x, (int)blockDim.x);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  for (int fi = 0; fi < fpw; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpw + fi) * teams + wave;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    const float* fp = valid ? frame_ptr_f32(view, row, chunk, s0, 2 * N) : nullptr;  // team-uniform
    if (fp) {
      for (int j = lane; j < N; j += NT) buf0[j] = {fp[2 * j] * win_a[2 * j], fp[2 * j + 1] * win_a[2 * j + 1]};
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<float> z = {0.f, 0.f};
        if (valid) {
          z.x = (float)view_sample(view, row, chunk, s0 + 2 * j) * win_a[2 * j];
          z.y = (float)view_sample(view, row, chunk, s0 + 2 * j + 1) * win_a[2 * j + 1];
        }
        buf0[j] = z;
      }
    }
    team_sync<SY>();
    cx<float>* Z = mr_fft<float, false, NT, SY>(buf0, buf1, tw, pl, lane);
    cx<float>* other = Z == buf0 ? buf1 : buf0;
    // split -> mask -> merge, pairwise in place: task k handles bins k and N - k
    if (valid) {
      const float* Mrow = M + (u * g.T + t) * g.FS;
      const unsigned short* Krow = K16 + (u * g.T + t) * g.FS;
      auto mask_at = [&](int k) -> float { return K16 ? (float)Krow[k] * kscale : Mrow[k]; };
      for (int k = lane; k <= N / 2; k += NT) {
        if (k == 0) {
          const cx<float> a = Z[0];
          const float y0 = (a.x + a.y) * mask_at(0);
          const float yN = (a.x - a.y) * mask_at(N);
          Z[0] = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
        } else {
          const cx<float> a = Z[k], b = Z[N - k];
          const cx<float> w = tw[k];
          const cx<float> E = {(a.x + b.x) * 0.5f, (a.y - b.y) * 0.5f};
          const cx<float> O = {(a.y + b.y) * 0.5f, (b.x - a.x) * 0.5f};
          const cx<float> wO = cmul(w, O);
          const float mk = mask_at(k), mn = mask_at(N - k);
          const cx<float> Yk = {(E.x + wO.x) * mk, (E.y + wO.y) * mk};
          const cx<float> Yn = {(E.x - wO.x) * mn, (-E.y + wO.y) * mn};  // X[N-k] * mn
          const cx<float> Ep = {(Yk.x + Yn.x) * 0.5f, (Yk.y - Yn.y) * 0.5f};
          const cx<float> D = {(Yk.x - Yn.x) * 0.5f, (Yk.y + Yn.y) * 0.5f};
          const cx<float> wc = {w.x, -w.y};
          const cx<float> Op = cmul(D, wc);
          Z[k] = {Ep.x - Op.y, Ep.y + Op.x};
          if (k != N - k) Z[N - k] = {Ep.x + Op.y, -Ep.y + Op.x};
        }
      }
    }
    team_sync<SY>();
    const cx<float>* Y = mr_fft<float, true, NT, SY>(Z, other, tw, pl, lane);
    if (valid) {
      float2* srow = reinterpret_cast<float2*>(seg + (u * g.T + t) * (int64_t)g.n);
      for (int j = lane; j < N; j += NT) {
        const cx<float> z = Y[j];
        srow[j] = make_float2(z.x * win_s[2 * j], z.y * win_s[2 * j + 1]);
      }
    }
    team_sync<SY>();
  }
}

}  // namespace sg
