// Kernels of the mixed-radix frame lengths (see mixed.hpp); included by mixed.hip only.
#pragma once
#include "mixed.hpp"

namespace sg {

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
__device__ __forceinline__ void mr_pass(const cx<T>* __restrict__ x, cx<T>* __restrict__ y, const cx<T>* tw,
                                        const cx<T>* __restrict__ pt /* this pass's twiddle table */, int N, int S, bool last,
                                        int lane) {
  const int NB = N / R;
  // i = p S + q: p from a float reciprocal ((i + 1/2) / S is at least 1 / (2 S) >= 2.4e-4 away from an integer, i, S <= 2048:
  // the rounding of the product cannot cross it) instead of an integer division per butterfly
  const float rS = 1.0f / (float)S;
  for (int i = lane; i < NB; i += NT) {
    const int gi = (int)(((float)i + 0.5f) * rS), base = gi * S, q = i - base;
    const cx<T>* wp = pt + gi * (R - 1) - 1;   // wp[k] = w_N^(base k), k = 1 .. R - 1
    cx<T> v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = x[mlp<T>(i + j * NB)];
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
        if (!last && k > 0) {
          cx<T> w = wp[k];
          if (INV) w.y = -w.y;
          acc = cmul(acc, w);
        }
        y[mlp<T>(o + S * k)] = acc;
      }
    } else {
      if constexpr (R == 8 || R == 4 || R == 2) dftR<R, INV>(v);
      else if constexpr (R == 3) dft3<INV>(v);
      else dft5<INV>(v);
      if (!last) {   // the last pass has p == 0: all twiddles are 1
#pragma unroll
        for (int k = 1; k < R; ++k) {
          cx<T> w = wp[k];
          if (INV) w.y = -w.y;
          v[k] = cmul(v[k], w);
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) y[mlp<T>(o + S * k)] = v[k];
    }
  }
}

// Complex transform of a[0 .. N) (unnormalised; INV: exp(+i ...)).  Returns the buffer that holds the result (a or b).
// SY as in fft_wave.hpp: 1 = the team is (part of) one wavefront and the buffers are its own, else a workgroup barrier.
template <typename T, bool INV, int NT, int SY>
__device__ __forceinline__ cx<T>* mr_fft(cx<T>* a, cx<T>* b, const cx<T>* tw, const cx<T>* ptw, const MrPlan& pl, int lane) {
  int S = 1;
  for (int p = 0; p < pl.np; ++p) {
    const int R = pl.R[p];
    const bool last = p + 1 == pl.np;
    const cx<T>* pt = ptw + pl.toff[p];
    switch (R) {
      case 8: mr_pass<8, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
      case 4: mr_pass<4, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
      case 2: mr_pass<2, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
      case 5: mr_pass<5, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
      case 3: mr_pass<3, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
      case 7: mr_pass<7, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
      case 11: mr_pass<11, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
      default: mr_pass<13, INV, T, NT>(a, b, tw, pt, pl.N, S, last, lane); break;
    }
    team_sync<SY>();
    S *= R;
    cx<T>* t = a; a = b; b = t;
  }
  return a;
}

template <typename T>
__device__ __forceinline__ void mr_stage_twiddles(cx<T>* tw, const cx<T>* __restrict__ tw_g, int N, int tid, int nthr) {
  if (N <= 0) return;
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

// window * frame of the N complex pairs of a frame whose samples are all readable float32 (fp != null), into the team's
// buffer; returns the lane's share of || window * frame ||^2.  EIGHT elements' loads (two samples + two window entries each)
// are in flight per lane before the first is used: written as `for (j = lane; j < N; j += NT) buf[j] = fp[2 j] * w[2 j] ...`
// the loop stays rolled and waits for every element in turn -- N / NT dependent global round trips per frame (eight at
// n_fft = 1000), which WAS these kernels' time (round 6: k_decide_mr 126 us per two minutes at n_fft = 1000, whatever the
// team size, frames in flight or twiddle scheme).  The frame's start has any parity: scalar loads, not 8-byte ones.
template <typename T, int NT>
__device__ __forceinline__ float mr_gather(cx<T>* buf, const float* __restrict__ fp, const T* __restrict__ win, int N, int lane) {
  float nrm2 = 0.f;
  for (int j0 = lane; j0 < N; j0 += 8 * NT) {
    float xa[8], xb[8];
    T wa[8], wb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int j = min(j0 + k * NT, N - 1);   // (clamped, not predicated)
      xa[k] = fp[2 * j];
      xb[k] = fp[2 * j + 1];
      wa[k] = win[2 * j];
      wb[k] = win[2 * j + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int j = j0 + k * NT;
      const cx<T> z = {(T)xa[k] * wa[k], (T)xb[k] * wb[k]};
      if (j < N) {
        buf[mlp<T>(j)] = z;
        nrm2 += (float)(z.x * z.x + z.y * z.y);
      }
    }
  }
  return nrm2;
}

constexpr int MR_MAXM = 17;   // bins per thread (k_stft_mr, k_stft_bits_mr): N + 1 <= 17 NT -- mr_team() in mixed.hip

// ---------------------------------------------------------------------------------------
// Forward STFT: k_stft (kernels.hpp) with the frame length at run time.  NT threads per frame (64: one wavefront; 256:
// the workgroup), blockDim.x / NT frames in flight per workgroup, FPW frames per team.
// ---------------------------------------------------------------------------------------
template <typename TC, int NT>
__global__ __launch_bounds__(256, sizeof(TC) == 8 ? 2 : 3) void k_stft_mr(View view, Geom g, MrPlan pl, const cx<TC>* __restrict__ tw_g,
                                                 const cx<TC>* __restrict__ pt_g, const TC* __restrict__ wfull, double* __restrict__ P_out,
                                                 float* __restrict__ mag_out, double* __restrict__ z_out, double z_scale,
                                                 unsigned long long* __restrict__ pmax_bits, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<TC>* tw = reinterpret_cast<cx<TC>*>(smem);
  cx<TC>* ptw = tw + N;
  const int lane = threadIdx.x % NT, wave = threadIdx.x / NT;
  cx<TC>* buf0 = ptw + pl.ptotal + (size_t)(2 * wave) * mlpn<TC>(N);   // (lpn is a constexpr function of its argument: fine at run time)
  cx<TC>* buf1 = buf0 + mlpn<TC>(N);
  mr_stage_twiddles(tw, tw_g, N, (int)threadIdx.x, (int)blockDim.x);
  mr_stage_twiddles(ptw, pt_g, pl.ptotal, (int)threadIdx.x, (int)blockDim.x);
  const int64_t u = blockIdx.y;
  const int64_t row = (view.unit0 + u) / view.n_chunks;
  const int64_t chunk = view.c0 + (view.unit0 + u) % view.n_chunks;
  __syncthreads();
  // running band maxima (pmax_bits): registers in the float64 kernel (the exact pipeline's pre-pass); the float32 kernel --
  // whose callers rarely ask for them -- goes to the atomics per frame instead of holding 34 registers for it
  constexpr bool VM = sizeof(TC) == 8;
  double vmax[VM ? MR_MAXM : 1];
#pragma unroll
  for (int m = 0; m < (VM ? MR_MAXM : 1); ++m) vmax[m] = 0.0;
  for (int fi = 0; fi < fpw; ++fi) {
    const int64_t t = ((int64_t)blockIdx.x * fpw + fi) * teams + wave;
    const bool valid = t < g.T;
    const int64_t s0 = t * g.H - g.padL;
    const float* fp = valid ? frame_ptr_f32(view, row, chunk, s0, 2 * N) : nullptr;  // team-uniform
    if (fp) {
      (void)mr_gather<TC, NT>(buf0, fp, wfull, N, lane);
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<TC> z = {(TC)0, (TC)0};
        if (valid) {
          z.x = (TC)view_sample(view, row, chunk, s0 + 2 * j) * wfull[2 * j];
          z.y = (TC)view_sample(view, row, chunk, s0 + 2 * j + 1) * wfull[2 * j + 1];
        }
        buf0[mlp<TC>(j)] = z;
      }
    }
    team_sync<SY>();
    const cx<TC>* Z = mr_fft<TC, false, NT, SY>(buf0, buf1, tw, ptw, pl, lane);
    if (valid) {
      const int64_t rowoff = (u * g.T + t) * g.FS;
#pragma unroll
      for (int m = 0; m < MR_MAXM; ++m) {
        const int k = lane + NT * m;
        if (k > N) continue;
        const cx<TC> a = Z[mlp<TC>(k == N ? 0 : k)];
        const cx<TC> b = Z[mlp<TC>((k == 0 || k == N) ? 0 : N - k)];
        const cx<TC> w = tw[k == N ? 0 : k];
        const cx<TC> X = rfft_bin(a, b, w, k, N);
        const double Pk = (double)X.x * (double)X.x + (double)X.y * (double)X.y;
        if constexpr (VM) vmax[m] = nanmax(vmax[m], Pk);
        else if (pmax_bits) atomicMax(&pmax_bits[u * g.FS + k], (unsigned long long)__double_as_longlong(nanmax(0.0, Pk)));
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
  if (VM && pmax_bits) {
#pragma unroll
    for (int m = 0; m < (VM ? MR_MAXM : 1); ++m) {
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
                                                      const cx<double>* __restrict__ pt_g, const double* __restrict__ wfull, ThreshConsts tc, double mag_scale,
                                                      double top_db, unsigned long long* __restrict__ pmax_bits,
                                                      unsigned long long* __restrict__ bits, int wpr, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<double>* tw = reinterpret_cast<cx<double>*>(smem);
  cx<double>* ptw = tw + N;
  double* sT2 = reinterpret_cast<double*>(ptw + pl.ptotal + (size_t)(2 * teams) * mlpn<double>(N));  // [N + 1] compare constants
  const int lane = threadIdx.x % NT, wave = threadIdx.x / NT;
  cx<double>* buf0 = ptw + pl.ptotal + (size_t)(2 * wave) * mlpn<double>(N);
  cx<double>* buf1 = buf0 + mlpn<double>(N);
  const int64_t u = blockIdx.y;
  const int need = need_of(tc, u);
  const bool floor_live = need == 1;
  if (MODE == 0 && !floor_live) return;  // whole block: uniform
  mr_stage_twiddles(tw, tw_g, N, (int)threadIdx.x, (int)blockDim.x);
  mr_stage_twiddles(ptw, pt_g, pl.ptotal, (int)threadIdx.x, (int)blockDim.x);
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
      buf0[mlp<double>(j)] = z;
    }
    team_sync<SY>();
    const cx<double>* Z = mr_fft<double, false, NT, SY>(buf0, buf1, tw, ptw, pl, lane);
    unsigned long long* brow = bits + ((u * g.T + t) * (int64_t)wpr);
#pragma unroll
    for (int m = 0; m < MR_MAXM; ++m) {
      const int k = lane + NT * m;
      if (NT * m > N) break;   // (team-uniform: the ballot below is the whole wavefront's)
      bool pred = false;
      if (k <= N) {
        const cx<double> a = Z[mlp<double>(k == N ? 0 : k)];
        const cx<double> b = Z[mlp<double>((k == 0 || k == N) ? 0 : N - k)];
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
                                                   const cx<float>* __restrict__ pt_g, const float* __restrict__ win32, const cx<double>* __restrict__ tw64,
                                                   const double* __restrict__ win64, ThreshConsts tc, double mag_scale,
                                                   double top_db, unsigned long long* __restrict__ bits, int wpr, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<float>* tw = reinterpret_cast<cx<float>*>(smem);
  cx<float>* ptw = tw + N;
  float* sT2 = reinterpret_cast<float*>(ptw + pl.ptotal + (size_t)(2 * teams) * mlpn<float>(N));  // [N + 1] compare constants (float32)
  float* s_red = sT2 + N + 1;                                               // [4] per-wave partial norms (NT = 256)
  const int lane64 = threadIdx.x & 63, lane = threadIdx.x % NT, team = threadIdx.x / NT;
  cx<float>* buf0 = ptw + pl.ptotal + (size_t)(2 * team) * mlpn<float>(N);
  cx<float>* buf1 = buf0 + mlpn<float>(N);
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
  mr_stage_twiddles(ptw, pt_g, pl.ptotal, (int)threadIdx.x, (int)blockDim.x);
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
      nrm2 = mr_gather<float, NT>(buf0, fp, win32, N, lane);
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<float> z = {0.f, 0.f};
        if (valid) {
          z.x = (float)view_sample(view, row, chunk, s0 + 2 * j) * win32[2 * j];
          z.y = (float)view_sample(view, row, chunk, s0 + 2 * j + 1) * win32[2 * j + 1];
        }
        nrm2 += z.x * z.x + z.y * z.y;
        buf0[mlp<float>(j)] = z;
      }
    }
    for (int off = (NT < 64 ? NT : 64) / 2; off > 0; off >>= 1) nrm2 += __shfl_xor(nrm2, off);
    if constexpr (NT == 256) {   // the frame's four wavefronts: partial norms through LDS (the pass sync below orders them)
      if (lane64 == 0) s_red[threadIdx.x >> 6] = nrm2;
      __syncthreads();
      nrm2 = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
    // 2 delta^2 = 2 * 2^-32 * nrm2; a silent frame (nrm2 == 0) has no ambiguous cells
    const float d2 = nrm2 > 0.f ? 2.0f * 2.3283064e-10f * nrm2 : -1.0f;
    team_sync<SY>();
    const cx<float>* Z = mr_fft<float, false, NT, SY>(buf0, buf1, tw, ptw, pl, lane);
    unsigned long long* brow = bits + ((u * g.T + t) * (int64_t)wpr);
    // teams below a wavefront: a ballot carries NT bins of each of the wavefront's 64 / NT frames; every team collects its own
    // (k_decide_lds, fused.hpp).  9 words = 576 bins >= N + 1 for the sizes such teams are used for (mr_team: N <= 543)
    constexpr int NWS = NT < 64 ? 9 : 1;
    [[maybe_unused]] unsigned long long acc[NWS];
    [[maybe_unused]] const int tq = lane64 / NT;      // team within the wavefront
#pragma unroll
    for (int w = 0; w < NWS; ++w) acc[w] = 0ull;
#pragma unroll 1
    for (int m = 0; NT * m <= N; ++m) {
      const int k = lane + NT * m;
      bool pred = false, amb = false;
      if (k <= N) {
        const cx<float> a = Z[mlp<float>(k == N ? 0 : k)];
        const cx<float> b = Z[mlp<float>((k == 0 || k == N) ? 0 : N - k)];
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
        // the cell's bin and frame: NT >= 64: the wavefront's 64 lanes are consecutive bins of one frame (lane64 - lane is
        // its offset inside a 256-thread team); NT < 64: lane src belongs to team src / NT of the wavefront = frame t - tq + ...
        const int ks = NT >= 64 ? k - lane64 + src : (src % NT) + NT * m;
        const int64_t s0s = NT >= 64 ? s0 : (t - tq + src / NT) * g.H - g.padL;
        double re = 0.0, im = 0.0;
        for (int i = lane64; i < 2 * N; i += 64) {
          const double xv = view_sample(view, row, chunk, s0s + i) * win64[i];
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
      if constexpr (NT >= 64) {
        if (valid && lane64 == 0 && k <= N) brow[k >> 6] = word;
      } else {
        const unsigned long long seg = (word >> (NT * tq)) & ((1ull << NT) - 1ull);
        const int w0 = (NT * m) >> 6;
#pragma unroll
        for (int w = 0; w < NWS; ++w) acc[w] |= w == w0 ? seg << ((NT * m) & 63) : 0ull;
      }
    }
    if constexpr (NT < 64) {
      if (valid && lane == 0) {
#pragma unroll
        for (int w = 0; w < NWS; ++w)
          if (w * 64 <= N) brow[w] = acc[w];
      }
    }
    team_sync<SY>();
  }
}

// ---------------------------------------------------------------------------------------
// Apply + inverse: k_apply_istft (kernels.hpp).  mask = M (float field) or K16 * kscale (integer weight sums).
// ---------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 3) void k_apply_istft_mr(View view, Geom g, MrPlan pl, const cx<float>* __restrict__ tw_g,
                                                        const cx<float>* __restrict__ pt_g, const float* __restrict__ win_a, const float* __restrict__ win_s,
                                                        const float* __restrict__ M, float* __restrict__ seg,
                                                        const unsigned short* __restrict__ K16, float kscale, int fpw) {
  constexpr int SY = NT <= 64 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = pl.N, teams = blockDim.x / NT;
  cx<float>* tw = reinterpret_cast<cx<float>*>(smem);
  const int lane = threadIdx.x % NT, wave = threadIdx.x / NT;
  cx<float>* ptw = tw + N;
  cx<float>* buf0 = ptw + pl.ptotal + (size_t)(2 * wave) * mlpn<float>(N);
  cx<float>* buf1 = buf0 + mlpn<float>(N);
  mr_stage_twiddles(tw, tw_g, N, (int)threadIdx.x, (int)blockDim.x);
  mr_stage_twiddles(ptw, pt_g, pl.ptotal, (int)threadIdx.x, (int)blockDim.x);
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
      (void)mr_gather<float, NT>(buf0, fp, win_a, N, lane);
    } else {
      for (int j = lane; j < N; j += NT) {
        cx<float> z = {0.f, 0.f};
        if (valid) {
          z.x = (float)view_sample(view, row, chunk, s0 + 2 * j) * win_a[2 * j];
          z.y = (float)view_sample(view, row, chunk, s0 + 2 * j + 1) * win_a[2 * j + 1];
        }
        buf0[mlp<float>(j)] = z;
      }
    }
    team_sync<SY>();
    cx<float>* Z = mr_fft<float, false, NT, SY>(buf0, buf1, tw, ptw, pl, lane);
    cx<float>* other = Z == buf0 ? buf1 : buf0;
    // split -> mask -> merge, pairwise in place: task k handles bins k and N - k
    if (valid) {
      const float* Mrow = M + (u * g.T + t) * g.FS;
      const unsigned short* Krow = K16 + (u * g.T + t) * g.FS;
      auto mask_at = [&](int k) -> float { return K16 ? (float)Krow[k] * kscale : Mrow[k]; };
      // (the mask entries of FOUR tasks -- bins k and N - k each -- are loaded before the first is used: N / 2 / NT dependent
      // global round trips otherwise, as in mr_gather)
      const int half = N / 2;
      for (int k0 = lane; k0 <= half; k0 += 4 * NT) {
        float mk4[4], mn4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = min(k0 + e * NT, half);
          mk4[e] = mask_at(k);
          mn4[e] = mask_at(N - k);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + e * NT;
          if (k > half) break;
          const float mk = mk4[e], mn = mn4[e];
          if (k == 0) {
            const cx<float> a = Z[0];
            const float y0 = (a.x + a.y) * mk;
            const float yN = (a.x - a.y) * mn;
            Z[0] = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
          } else {
            const cx<float> a = Z[mlp<float>(k)], b = Z[mlp<float>(N - k)];
            const cx<float> w = tw[k];
            const cx<float> E = {(a.x + b.x) * 0.5f, (a.y - b.y) * 0.5f};
            const cx<float> O = {(a.y + b.y) * 0.5f, (b.x - a.x) * 0.5f};
            const cx<float> wO = cmul(w, O);
            const cx<float> Yk = {(E.x + wO.x) * mk, (E.y + wO.y) * mk};
            const cx<float> Yn = {(E.x - wO.x) * mn, (-E.y + wO.y) * mn};  // X[N-k] * mn
            const cx<float> Ep = {(Yk.x + Yn.x) * 0.5f, (Yk.y - Yn.y) * 0.5f};
            const cx<float> D = {(Yk.x - Yn.x) * 0.5f, (Yk.y + Yn.y) * 0.5f};
            const cx<float> wc = {w.x, -w.y};
            const cx<float> Op = cmul(D, wc);
            Z[mlp<float>(k)] = {Ep.x - Op.y, Ep.y + Op.x};
            if (k != N - k) Z[mlp<float>(N - k)] = {Ep.x + Op.y, -Ep.y + Op.x};
          }
        }
      }
    }
    team_sync<SY>();
    const cx<float>* Y = mr_fft<float, true, NT, SY>(Z, other, tw, ptw, pl, lane);
    if (valid) {
      float2* srow = reinterpret_cast<float2*>(seg + (u * g.T + t) * (int64_t)g.n);
      for (int j0 = lane; j0 < N; j0 += 8 * NT) {   // (synthesis-window loads batched like mr_gather's)
        float2 w8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w8[e] = reinterpret_cast<const float2*>(win_s)[min(j0 + e * NT, N - 1)];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = j0 + e * NT;
          if (j < N) {
            const cx<float> z = Y[mlp<float>(j)];
            srow[j] = make_float2(z.x * w8[e].x, z.y * w8[e].y);
          }
        }
      }
    }
    team_sync<SY>();
  }
}

}  // namespace sg
