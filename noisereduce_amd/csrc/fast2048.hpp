// Fast path for n_fft = win_length = 2048, hop = 512, float32: register transform of 1024 complex points on 32 lanes.
//
// A real frame of 2048 samples = 1024 complex points z[m] = x[2m] + i x[2m+1]; lane (g, c) = (lane >> 5, lane & 31) holds
// the 32 points z[c + 32 r] of frame g (two frames per wavefront).  1024 = 32 x 32:
//     DFT32 over r (registers) -> twiddle w_1024^(c k1) -> exchange through LDS -> DFT32 over c (registers)
//     => lane c' holds row k1 = c':  v[k2] = Zc[c' + 32 k2]  -- natural bin order, bin k = c' + 32 k2.
// The exchange runs in two phases of 16 columns (lanes c < 16 write, everybody reads half a row; then lanes c >= 16),
// so a frame needs a 32 x 16 slice (4 KB): two frames fit the wave slice of the 1024-path kernels, with the same
// swizzle (16-byte chunk ^ ((row >> 1) & 7)).
// Real-FFT split / merge: bins k and 1024 - k sit in DIFFERENT lanes here (one row per lane: row 32 - c', register
// 31 - k2).  A lane fetches the partner value of each of its registers 0..15 (ds_bpermute), evaluates that pair once
// and hands the partner's half of the result back through a second shuffle -- the partner does the same for registers
// 16..31.  Lane 0 (rows 0: bins 32 k2) pairs its own registers i <-> 32 - i; DC / Nyquist and bin 512 are explicit.
// Tile = 4 waves = 8 frames, hop = 512 samples.  Tiles abut: the 3 hops that straddle two tiles are written as partial
// sums and combined by k_ola_seam2048 (overlapping tiles would redo 3 of every 8 transforms).
//
//   k_decide_fast2048   float32 decisions + exact float64 refinement -> bits [unit][frame][17 words]
//   k_mag_fast2048      |X| float32, natural bin order
//   k_apply_fast2048    x float mask -> inverse -> window -> overlap-add in four ordered rounds -> samples
#pragma once
#include "fastpath.hpp"

namespace sg {
namespace fast {

constexpr int F20_NC = 1024;          // complex points per frame
constexpr int F20_H = 512;            // hop
constexpr int F20_F = 1025;
constexpr int F20_FSL = 512 + 16;     // complex slots per frame slice (32 rows x 16 columns + skew)
constexpr int F20_XP = 512 + 32;      // floats between the 512-sample rows of the staged span
static_assert(2 * F20_FSL <= WAVE_CX_H, "two frame slices must fit a wave's region");

struct Fast20Args {
  View view;
  Geom g;
  const float* win;          // window float32 (2048)
  const double* win64;       // window float64 (2048)
  const cf* tw2048;          // w_2048^k, k < 1024 (float32)
  const cx<double>* tw64;    // w_2048^k, k < 1024 (float64)
  ThreshConsts tc;
  double mag_scale, top_db;
  unsigned long long* bits;  // [units][T][17]
  float* mag;                // [units][T][FS]
  const float* Mf;           // float mask [units][T][FS]
  const unsigned short* K;   // apply<KMASK>: integer weight sums of the smoothed bit mask [units][T][FS], natural order
  float inv_ktot;
  const float* wsq;          // window squared (2048)
  const float* invn;         // 1 / sum_q wsq[512 q + s], s < 512
  OutMap om;
  int64_t h_begin, h_end;
  int normalize;
  float* part;               // seam mode: [units][tiles][6][512] un-normalised partial hops (3 leading, 3 trailing), else nullptr
  int n_tiles;
  FloorLazy fl;              // decide: in-kernel floor test (thresh.hpp), alim == nullptr: flags computed a priori
  double iir_b;              // magnitude: the recurrence's b (non-stationary gate) ...
  double* sub;               // ... and its per-tile partials [units][tiles][2][FS] (fastpath.hpp: mag_sub_partials), or nullptr
};

// w_1024^e from the w_2048 table
__device__ __forceinline__ cf f20_w1024(const cf* tw2048, int e) {
  e &= 1023;
  cf w = tw2048[2 * (e & 511)];
  if (e >= 512) { w.x = -w.x; w.y = -w.y; }
  return w;
}

// Stage the twiddle table T[k1][c] = w_1024^(k1 c) and the tile's sample span; gather v[r] = (x[2m], x[2m+1]) * w,
// m = c + 32 r, of frame tf0 + 2 wave + g.
template <int WAVES, bool MX = false>
__device__ __forceinline__ unsigned f20_gather(const Fast20Args& A, cf* tw, cf* regions, int64_t row, int64_t chunk,
                                           int64_t tf0, cf* v, bool& valid) {   // returns (MX) the largest |sample| this thread staged, as a bit pattern
  unsigned mx_ = 0u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, c = lane & 31;
  constexpr int NF = 2 * WAVES, ROWS = NF - 1 + 4, SPAN = ROWS * F20_H;
  static_assert(ROWS * F20_XP * 4 <= WAVES * WAVE_CX_H * 8, "span must fit the exchange slices");
  {
    // (all table loads issued before the first store: see stage_tables in fastpath.hpp)
    constexpr int K = (1024 + WAVES * 64 - 1) / (WAVES * 64);
    cf t[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = min(tid + k * WAVES * 64, 1023);
      t[k] = f20_w1024(A.tw2048, (i >> 5) * (i & 31));
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = tid + k * WAVES * 64;
      if (i < 1024) tw[i] = t[k];
    }
  }
  const Geom& G = A.g;
  const int64_t s0b = tf0 * F20_H - G.padL;
  const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
  const float* sp = (const float*)A.view.x + row * A.view.stride + gb;
  const bool vec = A.view.dtype == 0 && tf0 >= 0 && tf0 + NF <= G.T && s0b >= 0 && s0b + SPAN <= A.view.Lp &&
                   gb >= A.view.lo && gb + SPAN <= A.view.hi && (reinterpret_cast<uintptr_t>(sp) & 15) == 0;
  float* xs = reinterpret_cast<float*>(regions);
  if (vec) {
    const unsigned m = stage_span_vec<WAVES * 64, SPAN, F20_XP, 512, MX>(xs, sp, tid);
    if constexpr (MX) mx_ = m;
  } else {
    unsigned m = 0u;
    for (int i = tid; i < SPAN; i += WAVES * 64) {
      const float xv = (float)view_sample(A.view, row, chunk, s0b + i);
      xs[(i >> 9) * F20_XP + (i & 511)] = xv;
      m = max(m, __float_as_uint(xv) & 0x7fffffffu);
    }
    if constexpr (MX) mx_ = m;
  }
  __syncthreads();
  const int f = 2 * wave + g;
  const int64_t t = tf0 + f;
  valid = t >= 0 && t < G.T;
  const float* xa = xs + f * F20_XP + 2 * c;
  const float2* wl = reinterpret_cast<const float2*>(A.win + 2 * c);
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    // sample index 2 c + 64 r of the frame: row r / 8 of the span, column 2 c + 64 (r % 8)
    float2 x2 = *reinterpret_cast<const float2*>(xa + (r >> 3) * F20_XP + 64 * (r & 7));
    if (!valid) x2 = make_float2(0.f, 0.f);
    const float2 w2 = wl[32 * r];
    v[r] = {x2.x * w2.x, x2.y * w2.y};
  }
  __syncthreads();
  return mx_;
}

// forward: v[r] = z[c + 32 r]  ->  v[k2] = Zc[c + 32 k2]
__device__ __forceinline__ void fft1k_fwd(cf* v, cf* fb, const cf* tw, int c) {
  __builtin_amdgcn_sched_barrier(0);
  dft_reg<32, false>(v);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k1 = 1; k1 < 32; ++k1) v[k1] = cmul(v[k1], tw[k1 * 32 + c]);
  cf o[32];
  const int cc = c & 15;
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    if ((c >> 4) == ph) {
#pragma unroll
      for (int k1 = 0; k1 < 32; ++k1) fb[k1 * 16 + (cc ^ (2 * ((k1 >> 1) & 7)))] = v[k1];
    }
    wave_lds_sync();
    xchg_read_row(fb, c, o + 16 * ph);
    wave_lds_sync();
  }
  __builtin_amdgcn_sched_barrier(0);
  dft_reg<32, false>(o);
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = o[i];
  __builtin_amdgcn_sched_barrier(0);
}

// inverse (unnormalised): v[k2] = Zc[c + 32 k2]  ->  v[r] = 1024 z[c + 32 r]
__device__ __forceinline__ void fft1k_inv(cf* v, cf* fb, const cf* tw, int c) {
  __builtin_amdgcn_sched_barrier(0);
  dft_reg<32, true>(v);
  cf o[32];
  const int cc = c & 15;
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    xchg_write_row(fb, c, v + 16 * ph);
    wave_lds_sync();
    if ((c >> 4) == ph) {
#pragma unroll
      for (int k1 = 0; k1 < 32; ++k1) o[k1] = fb[k1 * 16 + (cc ^ (2 * ((k1 >> 1) & 7)))];
    }
    wave_lds_sync();
  }
#pragma unroll
  for (int k1 = 1; k1 < 32; ++k1) {
    cf w = tw[k1 * 32 + c];
    w.y = -w.y;
    o[k1] = cmul(o[k1], w);
  }
  __builtin_amdgcn_sched_barrier(0);
  dft_reg<32, true>(o);
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = o[i];
  __builtin_amdgcn_sched_barrier(0);
}

// w_2048^(c + 32 k2) = w_2048^c * w_64^k2   (cos / sin of 2 pi k2 / 64: compile-time constants after unrolling)
__device__ constexpr float F20_C64[32] = {1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 0.0f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f};
__device__ constexpr float F20_S64[32] = {0.000000000e+00f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f, 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f};
__device__ __forceinline__ cf f20_wk(cf wl, int k2) {
  const float cs = F20_C64[k2], sn = -F20_S64[k2];      // w_64^k2 = cos - i sin
  return {wl.x * cs - wl.y * sn, wl.x * sn + wl.y * cs};
}

__device__ __forceinline__ double f20_exact_power(const Fast20Args& A, int64_t row, int64_t chunk, int64_t t, int f, int lane) {
  const int64_t s0 = t * F20_H - A.g.padL;
  double re = 0.0, im = 0.0;
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    const int m = lane + 64 * i;
    const double xv = view_sample(A.view, row, chunk, s0 + m) * A.win64[m];
    const int j = (f * m) & 2047;
    cx<double> w = A.tw64[j & 1023];
    if (j >= 1024) { w.x = -w.x; w.y = -w.y; }
    re += xv * w.x;
    im += xv * w.y;
  }
  for (int off = 32; off > 0; off >>= 1) {
    re += __shfl_xor(re, off);
    im += __shfl_xor(im, off);
  }
  return re * re + im * im;
}

// ---------------------------------------------------------------------------------------------------------------
// REDO: the second launch of a call with the in-kernel floor test (thresh.hpp: FloorLazy): only the units whose test fired.
template <int WAVES, bool REDO = false>
__global__ __launch_bounds__(WAVES * 64, 3) void k_decide_fast2048(Fast20Args A) {
  if (REDO && A.fl.alim[1] != A.tc.need_tag) return;   // no unit of this call reported (the common case)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw = reinterpret_cast<cf*>(smem);                 // [32][32] w_1024^(k1 c)
  cf* regions = tw + 1024;
  float* s_t2 = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);   // [1025] compare constants x4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, c = lane & 31;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  const bool lazy = A.fl.alim != nullptr;
  const int need = (lazy && !REDO) ? 0 : need_of(A.tc, u);
  if (REDO && need == 0) return;   // whole workgroup
  const unsigned fl_bound = REDO ? 0xffffffffu : floor_lazy_bound(A.fl, lane);
  const bool floor_live = need == 1;
  auto t2eff = [&](int f) -> double {
    double v = A.tc.T2[f];
    if (floor_live) {
      const double fl = cell_db(A.tc.pmax[u * G.FS + f], A.mag_scale) - A.top_db;
      if (fl > A.tc.thresh[f]) v = -1.0;
    }
    if (need == 2) v = T2_NEVER;
    return v;
  };
  stage_t2_plain<WAVES * 64, F20_F>(s_t2, A.tc.T2, need, 4.0, tid, t2eff);
  constexpr int NF = 2 * WAVES;
  const int64_t tf0 = (int64_t)blockIdx.x * NF;
  cf v[32];
  bool valid;
  const unsigned fl_mx = f20_gather<WAVES, true>(A, tw, regions, row, chunk, tf0, v, valid);
  if (!REDO) floor_lazy_report(A.tc, A.fl, fl_bound, fl_mx, u, G.FS, lane);
  const int64_t tq = tf0 + 2 * wave;
  if (tq >= G.T) return;   // wave-uniform
  float nrm2 = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) nrm2 += v[r].x * v[r].x + v[r].y * v[r].y;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) nrm2 += __shfl_xor(nrm2, o);
  cf* fb = regions + wave * WAVE_CX_H + g * F20_FSL;
  {
    int z0 = 0;
    asm volatile("" : "+v"(z0));
    fft1k_fwd(v, fb, tw + z0, c);
  }
  const float d2 = nrm2 > 0.f ? 8.0f * 2.3283064e-10f * nrm2 : -1.0f;
  const cf wl = A.tw2048[c];
  unsigned pred = 0, amb = 0;
  bool predN = false, ambN = false;     // bin 1024 (lane 0)
  {
    // One pair per iteration (see k_apply_fast2048): the split of (Zc[k], Zc[1024 - k]) yields 2 X[k] for the lane's own
    // register i AND 2 X[1024 - k], whose power belongs to the partner's register 31 - i -- the powers are swapped
    // through one shuffle instead of both lanes evaluating both pairs.  Lane 0 pairs its own registers i <-> 32 - i.
    const int src = (lane & 32) | ((32 - c) & 31);
    const bool l0 = c == 0;
    auto decide = [&](float P, float T, int q) {
      const float diff = P - T;
      pred |= (diff > 0.f ? 1u : 0u) << q;
      amb |= ((diff * diff <= d2 * (P + T)) ? 1u : 0u) << q;
    };
    float Pp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const cf ob = v[31 - i], own = v[(32 - i) & 31];
      cf ta;
      ta.x = __shfl(ob.x, src); ta.y = __shfl(ob.y, src);
      const cf ba = {l0 ? own.x : ta.x, l0 ? own.y : ta.y};
      cf xa, xb;
      split_pair(v[i], ba, f20_wk(wl, i), xa, xb);
      decide(xa.x * xa.x + xa.y * xa.y, s_t2[c + 32 * i], i);
      Pp[i] = xb.x * xb.x + xb.y * xb.y;
    }
    {   // lane 0, i = 0: the pair (Zc[0], Zc[0]) also yields bin 1024
      const float PN = Pp[0], TN = s_t2[1024], dN = PN - TN;
      predN = l0 && dN > 0.f;
      ambN = l0 && dN * dN <= d2 * (PN + TN);
    }
    const float P512 = 4.f * (v[16].x * v[16].x + v[16].y * v[16].y);   // lane 0: Zc[512] is its own partner, X = conj Zc
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float pu = __shfl(Pp[i], src);                       // generic: the partner evaluated OUR register 31 - i
      const float p0 = i < 15 ? Pp[i + 1] : P512;               // lane 0: register 31 - i = 32 - (i + 1)
      decide(l0 ? p0 : pu, s_t2[c + 32 * (31 - i)], 31 - i);
    }
  }
  if (need == 2) { pred = 0; amb = 0; predN = false; ambN = false; }
  if (!valid) { pred = 0; amb = 0; predN = false; ambN = false; }
  while (true) {
    const unsigned long long pending = __ballot(amb != 0 || ambN);
    if (pending == 0) break;
    const int src = __ffsll((long long)pending) - 1;
    const unsigned amb_s = (unsigned)__shfl((int)amb, src);
    const int q = amb_s ? (__ffs((int)amb_s) - 1) : 32;
    const int cs = src & 31, gs = src >> 5;
    const int f = q < 32 ? cs + 32 * q : 1024;
    const Fast20Args& L = *late_args<Fast20Args>();     // (cold path: arguments re-read here, not kept live from the entry)
    const double P = f20_exact_power(L, row, chunk, tq + gs, f, lane);
    double t2 = L.tc.T2[f];
    if (floor_live) {
      const double fl = cell_db(L.tc.pmax[u * (int64_t)L.g.FS + f], L.mag_scale) - L.top_db;
      if (fl > L.tc.thresh[f]) t2 = -1.0;
    }
    if (need == 2) t2 = T2_NEVER;
    const bool pass = P > t2;
    if (lane == src) {
      if (q < 32) {
        pred = (pred & ~(1u << q)) | ((pass ? 1u : 0u) << q);
        amb &= ~(1u << q);
      } else {
        predN = pass;
        ambN = false;
      }
    }
  }
  // pack: the ballot of register k2 is, per frame, the 32 bins 32 k2 .. 32 k2 + 31 in natural order
  unsigned long long myword = 0ull;
  const int sh = 32 * g;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const unsigned long long b0 = __ballot((pred >> (2 * w)) & 1u), b1 = __ballot((pred >> (2 * w + 1)) & 1u);
    const unsigned long long word = ((b0 >> sh) & 0xffffffffull) | (((b1 >> sh) & 0xffffffffull) << 32);
    if (c == w) myword = word;
  }
  {
    const unsigned long long bN = __ballot(predN);
    if (c == 16) myword = (bN >> sh) & 1ull;
  }
  const int64_t t = tq + g;
  if (valid && c < 17) A.bits[(u * G.T + t) * 17 + c] = myword;
}

// ---------------------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 3) void k_mag_fast2048(Fast20Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw = reinterpret_cast<cf*>(smem);
  cf* regions = tw + 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, c = lane & 31;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  constexpr int NF = 2 * WAVES;
  const int64_t tf0 = (int64_t)blockIdx.x * NF;
  cf v[32];
  bool valid;
  f20_gather<WAVES>(A, tw, regions, row, chunk, tf0, v, valid);
  const int64_t tq = tf0 + 2 * wave;
  const bool with_sub = A.sub != nullptr;
  constexpr int TP = 1028;   // floats between the rows of the |X| tile (with_sub): 2 rows per wave in its own exchange slice
  static_assert(2 * TP * 4 <= WAVE_CX_H * 8, "a wave's |X| rows fit its exchange slice");
  if (tq < G.T) {   // (wave-uniform)
    cf* fb = regions + wave * WAVE_CX_H + g * F20_FSL;
    fft1k_fwd(v, fb, tw, c);
    const cf wl = A.tw2048[c];
    float* mrow = A.mag + (u * G.T + (valid ? tq + g : 0)) * (int64_t)G.FS;
    float* trow = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + g * TP;
    const int src = (lane & 32) | ((32 - c) & 31);
    const bool l0 = c == 0;
    if (with_sub) wave_lds_sync();   // both lane groups are past their exchange reads: the slice becomes the wave's two |X| rows
    auto put = [&](int k, float m) {
      if (valid) mrow[k] = m;
      if (with_sub) trow[k] = m;
    };
    float Pp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // one pair per iteration, the partner's power handed over (see k_decide_fast2048)
      const cf ob = v[31 - i], own = v[(32 - i) & 31];
      cf ta;
      ta.x = __shfl(ob.x, src); ta.y = __shfl(ob.y, src);
      const cf ba = {l0 ? own.x : ta.x, l0 ? own.y : ta.y};
      cf xa, xb;
      split_pair(v[i], ba, f20_wk(wl, i), xa, xb);
      put(c + 32 * i, half_sqrt(xa.x * xa.x + xa.y * xa.y));
      Pp[i] = xb.x * xb.x + xb.y * xb.y;
    }
    if (l0) put(1024, half_sqrt(Pp[0]));
    const float P512 = 4.f * (v[16].x * v[16].x + v[16].y * v[16].y);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float pu = __shfl(Pp[i], src);
      const float p0 = i < 15 ? Pp[i + 1] : P512;
      put(c + 32 * (31 - i), half_sqrt(l0 ? p0 : pu));
    }
  }
  if (!with_sub) return;
  __syncthreads();
  mag_sub_partials<WAVES * 64, NF, 2, TP, F20_F>(regions, (int)min<int64_t>((int64_t)NF, G.T - tf0), A.iir_b,
                                                A.sub + ((u * gridDim.x + blockIdx.x) * 2) * (int64_t)G.FS, G.FS, tid);
}

// ---------------------------------------------------------------------------------------------------------------
template <int WAVES, bool KMASK>
__global__ __launch_bounds__(WAVES * 64, 3) void k_apply_fast2048(Fast20Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw = reinterpret_cast<cf*>(smem);
  cf* regions = tw + 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, c = lane & 31;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  constexpr int NF = 2 * WAVES, NH = NF - 3;
  const bool seam = A.part != nullptr;        // abutting tiles + k_ola_seam2048, else overlapping tiles
  const int64_t tf0 = A.h_begin - 3 + (int64_t)blockIdx.x * (seam ? NF : NH);
  cf v[32];
  bool valid;
  f20_gather<WAVES>(A, tw, regions, row, chunk, tf0, v, valid);
  const int f = 2 * wave + g;                 // tile-local frame
  const int64_t t = tf0 + f;
  cf* fb = regions + wave * WAVE_CX_H + g * F20_FSL;
  const bool wave_live = tf0 + 2 * wave + 1 >= 0 && tf0 + 2 * wave < G.T;
  if (wave_live) {
    {
      int z0 = 0;
      asm volatile("" : "+v"(z0));
      fft1k_fwd(v, fb, tw + z0, c);
    }
    const cf wl = A.tw2048[c];
    const int64_t moff = (u * G.T + (valid ? t : 0)) * (int64_t)G.FS;
    // pair_mask leaves out four 1/2 factors; the inverse transform a factor 1024; K / ktot for the integer sums
    const float ks = (KMASK ? A.inv_ktot : 1.0f) * (0.25f / 1024.0f);
    auto mval = [&](int k) -> float {
      if constexpr (KMASK) return (float)A.K[moff + k] * ks;
      else return A.Mf[moff + k] * ks;
    };
    {
      // One pair per iteration: the lane fetches the partner value of its register i (lane 32 - c, register 31 - i),
      // runs split -> mask -> merge ONCE, keeps the merged value of its own bin and hands the other one -- the new
      // Zc'[1024 - k], which belongs in the partner's register 31 - i -- back through a second shuffle (the partner
      // does the same for us): half the pair arithmetic and half the mask loads of evaluating every bin separately.
      // Lane 0 pairs its own registers i <-> 32 - i: the partner VALUE it needs was overwritten one iteration ago
      // (carried along), and the merged partner value it produces belongs one register higher than where the generic
      // hand-back puts it (shifted after the loop); its self-paired bin 512 and DC / Nyquist are set explicitly.
      const int src = (lane & 32) | ((32 - c) & 31);
      const bool l0 = c == 0;
      const cf a0 = v[0], a16 = v[16];
      cf carry = v[0];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const cf oa = v[i], ob = v[31 - i];
        cf ta;
        ta.x = __shfl(ob.x, src); ta.y = __shfl(ob.y, src);     // Zc[1024 - k]: lane 32 - c, register 31 - i
        cf ba = {l0 ? carry.x : ta.x, l0 ? carry.y : ta.y};      // lane 0: original register 32 - i (i = 0: itself)
        carry = ob;
        const int ka = c + 32 * i;
        cf xa = oa;
        pair_mask(xa, ba, f20_wk(wl, i), mval(ka), mval(1024 - ka));          // (ka = 0: the second mask is bin 1024's)
        v[i] = xa;
        v[31 - i].x = __shfl(ba.x, src);                          // the partner's merged value for OUR register 31 - i
        v[31 - i].y = __shfl(ba.y, src);
      }
      if (l0) {
        // lane 0 received its own hand-backs: register 31 - i holds Zc'[32 (32 - i)], which belongs in register 32 - i
#pragma unroll
        for (int j = 31; j >= 17; --j) v[j] = v[j - 1];
        // bin 512 = Zc[512] is its own partner: Zc'[512] = Zc[512] m (irfft side: conj conj); DC / Nyquist:
        // Zc'[0] = ((X0 m0 + XN mN) / 2, (X0 m0 - XN mN) / 2), X0 = Re + Im, XN = Re - Im
        const float m16 = mval(512) * 4.0f;
        v[16] = {a16.x * m16, a16.y * m16};
        const float y0 = (a0.x + a0.y) * mval(0) * 4.0f;
        const float yN = (a0.x - a0.y) * mval(1024) * 4.0f;
        v[0] = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
      }
    }
    {
      int zi = 0, ci = c;
      asm volatile("" : "+v"(zi), "+v"(ci));
      fft1k_inv(v, fb + zi, tw + zi, ci);
    }
  }
  __syncthreads();   // every wave is past its exchanges: the regions become the tile's hop buffer
  // overlap-add in four ordered rounds: in round j every frame adds its quarter j to tile hop f + j -- no two frames
  // meet in a round, and a hop receives its quarters in the fixed order j = 0..3.  Hop buffer: (NF + 3) x 512 floats.
  float* hop = reinterpret_cast<float*>(regions);
  static_assert((NF + 3) * F20_XP * 4 <= WAVES * WAVE_CX_H * 8, "hop buffer must fit the regions");
  const float2* ws = reinterpret_cast<const float2*>(A.win + 2 * c);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool first = (j == 0) || (f == NF - 1);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = 8 * j + rr;
      const float2 w2 = ws[32 * r];
      float2* dst = reinterpret_cast<float2*>(hop + (f + j) * F20_XP + 2 * c + 64 * rr);
      float2 nw = {v[r].x * w2.x, v[r].y * w2.y};
      if (!first) { const float2 old = *dst; nw.x += old.x; nw.y += old.y; }
      *dst = nw;
    }
    __syncthreads();
  }
  const int s4 = (tid & 127) * 4;
  for (int jj = (seam ? 0 : 3) + (tid >> 7); jj < (seam ? NF + 3 : NF); jj += (WAVES * 64) >> 7) {
    const int64_t h = tf0 + jj;
    if (h < A.h_begin || h >= A.h_end) continue;
    float4 a4 = *reinterpret_cast<const float4*>(&hop[jj * F20_XP + s4]);
    if (seam && (jj < 3 || jj >= NF)) {   // straddling hop: partial sum only; slots 0..2 leading, 3..5 trailing
      const int slot = jj < 3 ? jj : 3 + (jj - NF);
      *reinterpret_cast<float4*>(A.part + (((u * A.n_tiles + blockIdx.x) * 6 + slot) * 512 + s4)) = a4;
      continue;
    }
    bool all_valid = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t ti = h - q;
      if (ti < 0 || ti >= G.T) all_valid = false;
    }
    if (!A.normalize) {
    } else if (all_valid) {
      const float4 n4 = *reinterpret_cast<const float4*>(&A.invn[s4]);
      a4.x *= n4.x; a4.y *= n4.y; a4.z *= n4.z; a4.w *= n4.w;
    } else {
      float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ti = h - q;
        if (ti >= 0 && ti < G.T) {
          const float4 w4 = *reinterpret_cast<const float4*>(&A.wsq[F20_H * q + s4]);
          nrm.x += w4.x; nrm.y += w4.y; nrm.z += w4.z; nrm.w += w4.w;
        }
      }
      a4.x /= (nrm.x > 1e-10f ? nrm.x : 1.f);
      a4.y /= (nrm.y > 1e-10f ? nrm.y : 1.f);
      a4.z /= (nrm.z > 1e-10f ? nrm.z : 1.f);
      a4.w /= (nrm.w > 1e-10f ? nrm.w : 1.f);
    }
    {
      const int64_t pb = h * F20_H - G.padL;
      const int64_t gi0 = chunk * A.om.g_step + (pb - A.om.p0);
      if (A.om.dtype == 0 && pb >= A.om.p0 && pb + F20_H <= A.om.p1 && pb + F20_H <= G.Lout && gi0 >= A.om.g_lo &&
          gi0 + F20_H <= A.om.g_hi) {
        float* dst = (float*)A.om.out + (row * A.om.stride + gi0 - A.om.g0 + s4);
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = a4;
          continue;
        }
      }
    }
    const float vals[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t p = h * F20_H + s4 + e - G.padL;
      if (p < A.om.p0 || p >= A.om.p1) continue;
      const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
      if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
      store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? vals[e] : 0.f);
    }
  }
}

// Seam hops of abutting tiles: hop tf0(b + 1) + k (k = 0..2) = trailing partial k of tile b + leading partial k of tile
// b + 1 (fixed order), normalised and stored.
template <int NF>
__global__ __launch_bounds__(512) void k_ola_seam2048(Fast20Args A) {
  const Geom& G = A.g;
  const int64_t u = blockIdx.y, b = blockIdx.x;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  const int s = threadIdx.x;
  const float* pa = A.part + ((u * A.n_tiles + b) * 6 + 3) * 512;
  const float* pb = A.part + ((u * A.n_tiles + b + 1) * 6 + 0) * 512;
  // (round 6) all six partials and the envelope in flight before the first use: the three hops were three dependent round
  // trips (8.3 us for a kernel that moves 1.8 MB)
  float va[3], vb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { va[k] = pa[k * 512 + s]; vb[k] = pb[k * 512 + s]; }
  const float inv = A.invn[s];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int64_t h = A.h_begin - 3 + (int64_t)NF * (b + 1) + k;
    if (h < A.h_begin || h >= A.h_end) continue;
    float val = va[k] + vb[k];
    if (A.normalize) {
      if (h - 3 >= 0 && h < G.T) {
        val *= inv;
      } else {
        float nrm = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t ti = h - q;
          if (ti >= 0 && ti < G.T) nrm += A.wsq[F20_H * q + s];
        }
        val /= (nrm > 1e-10f ? nrm : 1.f);
      }
    }
    const int64_t p = h * F20_H + s - G.padL;
    if (p < A.om.p0 || p >= A.om.p1) continue;
    const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
    if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
    store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? val : 0.f);
  }
}

}  // namespace fast
}  // namespace sg
