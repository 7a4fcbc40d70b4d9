// Per-wavefront Stockham FFT for gfx950 (wave64).
//
// One 64-lane wavefront transforms one complex sequence of length N (N = n_fft/2, the
// real frame packed as even/odd pairs) that lives in that wave's private LDS slice.
// Constant-geometry Stockham autosort passes of radix 8/4/2: in every pass lane l owns
// butterflies i = l, l+64, ... ; it reads x[i + j*N/R] (conflict-free, lane-contiguous),
// does the R-point DFT in registers, applies the twiddle w^(p*k) and writes
// y[(i-q)*R + q + s*k].  Reads of a pass complete before its writes start (WAVE_SYNC),
// so the transform is in place; the result comes out in natural order.
//
// Twiddles come from one master table tw[k] = exp(-2*pi*i*k/(2N)), k in [0, N), shared
// by the whole workgroup in LDS: it serves both the complex passes (w_N^t = tw[2t], with
// tw[k+N] = -tw[k]) and the real-FFT split/merge step (w_2N^k).
#pragma once
#include <hip/hip_runtime.h>

namespace sg {

template <typename T>
struct cx {
  T x, y;
};

template <typename T>
__device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// multiply by -i (forward) or +i (inverse)
template <bool INV, typename T>
__device__ __forceinline__ cx<T> rot90(cx<T> a) {
  if (INV) return {-a.y, a.x};
  return {a.y, -a.x};
}

// Workgroup-level sync between FFT passes.  All waves of a block run the same pass
// sequence, so a block barrier is always legal here.
#define SG_PASS_SYNC() __syncthreads()

// (round 5) Sync between the passes of ONE team's transform.  A team of fewer than 64 lanes lies inside one wavefront, whose
// LDS operations execute in order: its buffer needs no barrier at all, only the compiler pinned (the idiom of
// fast::wave_lds_sync).  Teams of 64 / 256 threads keep the workgroup barrier their kernels were written around.
template <int NT>
__device__ __forceinline__ void team_sync() {
  if constexpr (NT < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

template <bool INV, typename T>
__device__ __forceinline__ void dft2(cx<T>* v) {
  cx<T> a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}

template <bool INV, typename T>
__device__ __forceinline__ void dft4(cx<T>* v) {
  cx<T> s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
  cx<T> s13 = cadd(v[1], v[3]), d13 = rot90<INV>(csub(v[1], v[3]));
  v[0] = cadd(s02, s13);
  v[2] = csub(s02, s13);
  v[1] = cadd(d02, d13);
  v[3] = csub(d02, d13);
}

template <bool INV, typename T>
__device__ __forceinline__ void dft8(cx<T>* v) {
  cx<T> e[4] = {v[0], v[2], v[4], v[6]};
  cx<T> o[4] = {v[1], v[3], v[5], v[7]};
  dft4<INV>(e);
  dft4<INV>(o);
  const T h = (T)0.70710678118654752440;
  // o[k] *= w8^k,  w8 = exp(-+ i pi/4)
  {
    cx<T> t = o[1];
    if (INV) o[1] = {(t.x - t.y) * h, (t.x + t.y) * h};
    else     o[1] = {(t.x + t.y) * h, (t.y - t.x) * h};
    o[2] = rot90<INV>(o[2]);
    t = o[3];
    if (INV) o[3] = {(-t.x - t.y) * h, (t.x - t.y) * h};
    else     o[3] = {(t.y - t.x) * h, (-t.x - t.y) * h};
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = cadd(e[k], o[k]);
    v[k + 4] = csub(e[k], o[k]);
  }
}

template <int R, bool INV, typename T>
__device__ __forceinline__ void dftR(cx<T>* v) {
  if (R == 8) dft8<INV>(v);
  else if (R == 4) dft4<INV>(v);
  else dft2<INV>(v);
}

// LDS index padding of the float64 transform buffers: one spare element after every 8.  The
// Stockham passes scatter their outputs with strides of 8 / 64 elements; unpadded, the 16-byte
// float64 elements of a wavefront land on the same few banks (128-byte stride: an 8-way conflict
// per 16-lane pass).  With the padding a stride-8 access advances 9 elements = 36 banks per lane:
// conflict-free.  float32 buffers stay unpadded (measured: padding costs them the 16-byte alignment
// of element pairs and more than it gains).  Every user indexes its buffer through lp<T>().
template <typename T>
__device__ __forceinline__ int lp(int e) {
  return sizeof(T) == 8 ? e + (e >> 3) : e;
}
template <typename T>
constexpr int lpn(int n) {  // padded length of an n-element buffer
  return sizeof(T) == 8 ? n + (n >> 3) : n;
}

// w_N^t from the master table (w_2N^k, k < N);  conjugated for the inverse transform.
template <int N, bool INV, typename T>
__device__ __forceinline__ cx<T> twN(const cx<T>* tw, int t) {
  int k = 2 * t;
  cx<T> w;
  if (k < N) {
    w = tw[k];
  } else {
    w = tw[k - N];
    w.x = -w.x;
    w.y = -w.y;
  }
  if (INV) w.y = -w.y;
  return w;
}

// NT = threads that cooperate on one transform (64: one wavefront per frame; 256: a whole
// workgroup per frame, for the sizes whose butterflies would not fit one wave's registers).
// SY: what team_sync synchronises (default: the team size).  A kernel whose 64-lane teams ARE wavefronts and whose transform
// buffers are team-private passes SY = 1 (wave-level ordering, no workgroup barrier): k_stft, k_apply_istft, k_decide_lds.
template <typename T, int N, int S, bool INV, int NT = 64, int SY = NT>
struct FftPass {
  static __device__ __forceinline__ void run(cx<T>* buf, const cx<T>* tw, int lane) {
    constexpr int NR = N / S;  // current sub-transform length
    constexpr int R = (NR % 8 == 0) ? 8 : ((NR % 4 == 0) ? 4 : 2);
    constexpr int NB = N / R;  // butterflies in this pass
    constexpr int PER = (NB + NT - 1) / NT;
    cx<T> v[PER][R];
#pragma unroll
    for (int c = 0; c < PER; ++c) {
      int i = lane + NT * c;
      if (NB >= NT || i < NB) {
#pragma unroll
        for (int j = 0; j < R; ++j) v[c][j] = buf[lp<T>(i + j * NB)];
      }
    }
    team_sync<SY>();
#pragma unroll
    for (int c = 0; c < PER; ++c) {
      int i = lane + NT * c;
      if (NB >= NT || i < NB) {
        int q = i & (S - 1);
        int base = i - q;  // = p * S
        dftR<R, INV>(v[c]);
        if (NR != R) {  // the last pass has p == 0: all twiddles are 1
#pragma unroll
          for (int k = 1; k < R; ++k) v[c][k] = cmul(v[c][k], twN<N, INV>(tw, base * k));
        }
        int o = base * R + q;
#pragma unroll
        for (int k = 0; k < R; ++k) buf[lp<T>(o + S * k)] = v[c][k];
      }
    }
    team_sync<SY>();
    if constexpr (NR / R > 1) FftPass<T, N, S * R, INV, NT, SY>::run(buf, tw, lane);
  }
};

// (round 5) The workgroup's twiddle table into LDS with every load of a thread in flight before its first store: the rolled
// `for (i = tid; i < N; i += threads) tw[i] = tw_g[i]` waits for each load in turn -- sixteen dependent round trips per
// workgroup at n_fft = 4096 (N = 2048, 128 threads), per EIGHT frames of work.  At most 8 entries per thread at a time.
template <int NTHR, int N, typename T>
__device__ __forceinline__ void stage_twiddles(cx<T>* tw, const cx<T>* __restrict__ tw_g, int tid) {
  constexpr int K = (N + NTHR - 1) / NTHR, KB = K < 8 ? K : 8;
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += KB) {
    cx<T> t[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int i = tid + (k0 + k) * NTHR;
      t[k] = tw_g[i < N ? i : N - 1];
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int i = tid + (k0 + k) * NTHR;
      if (k0 + k < K && i < N) tw[i] = t[k];
    }
  }
}

// In-place complex FFT of buf[0..N) (unnormalised; INV uses exp(+i...)).
template <typename T, int N, bool INV, int NT = 64, int SY = NT>
__device__ __forceinline__ void wave_fft(cx<T>* buf, const cx<T>* tw, int lane) {
  FftPass<T, N, 1, INV, NT, SY>::run(buf, tw, lane);
}

// Real-FFT split: from Zc = FFT_N(x_even + i x_odd) compute bin k of the length-2N real
// transform, k in [0, N].  a = Zc[k], b = Zc[N-k] (a = b = Zc[0] for k = 0 and k = N).
template <typename T>
__device__ __forceinline__ cx<T> rfft_bin(cx<T> a, cx<T> b, cx<T> w, int k, int N) {
  if (k == 0) return {a.x + a.y, (T)0};
  if (k == N) return {a.x - a.y, (T)0};
  cx<T> E = {(a.x + b.x) * (T)0.5, (a.y - b.y) * (T)0.5};
  cx<T> O = {(a.y + b.y) * (T)0.5, (b.x - a.x) * (T)0.5};  // (a - conj b) / (2i)
  return cadd(E, cmul(w, O));
}

}  // namespace sg
