// Fast path for n_fft = win_length = 512, hop = 128 (the usual frame of 16 kHz speech front ends), float32, on the
// SAME register transform as the 1024 path (fastpath.hpp: 512 complex points, one lane group of 16 lanes, 32 points
// per lane) -- by the two-real-sequences trick: a lane group transforms TWO consecutive real frames A, B at once as
//     z[m] = a[m] + i b[m],  Z = FFT_512(z):   A[k] = (Z[k] + conj Z[512-k]) / 2,   B[k] = (Z[k] - conj Z[512-k]) / (2i)
// The conjugate pairs (k, 512 - k) are the pairs the 1024 path already keeps in one lane, and E = a + conj b,
// O = (a - conj b) / i of its split ARE 2A[k] and 2B[k]: no twiddle, no cross-lane traffic.  Inverse: Z'[k] = Ya[k] +
// i Yb[k], Z'[512-k] = conj Ya[k] + i conj Yb[k]; the real / imaginary parts of IFFT(Z') are the two output frames.
// One wavefront = 8 frames, one workgroup (4 waves) = a tile of 32 frames -> 29 complete hops of 128 samples (tiles
// overlap by 3 frames: every tile finishes its hops alone, 9 % redundant transforms, no hand-off between workgroups).
//
//   k_decide_fast512   frames -> FFT -> |A|^2, |B|^2 against the compare constants (float32 + the exact float64
//                      refinement of k_decide_fast) -> mask bits [unit][frame][5 words]
//   k_mag_fast512      frames -> FFT -> |A|, |B| (float32, natural bin order) for the non-stationary masks
//   k_apply_fast512    frames -> FFT -> x float mask (natural bin order) -> IFFT -> window -> overlap-add -> samples
// Lane (g, c) of a wave owns, for both of its frames, the 16 bins  { c + 32 e, (32 - c) + 32 e : e < 8 }  (lane 0:
// 0, 32, .., 224, 256 and 16 + 32 e) -- slot sl of the pair loop below yields bin bin5(c, sl).
#pragma once
#include "fastpath.hpp"

namespace sg {
namespace fast {

constexpr int F5_N = 512, F5_H = 128, F5_F = 257;
constexpr int F5_FPW = 8;                  // frames per wave
constexpr int F5_XP = 136;                 // floats between the 128-sample rows of the staged span (bank spread)
constexpr int F5_HP = 136;                 // floats between a wave's hop accumulators

__host__ __device__ inline int bin5(int c, int sl) {   // bin of pair slot sl (0..15) in lane c
  if (c != 0) return sl < 8 ? c + 32 * sl : (32 - c) + 32 * (15 - sl);
  return sl < 8 ? 32 * sl : 16 + 32 * (sl - 8);        // (lane 0, slot 0 = bin 0; bin 256 rides separately)
}

struct Fast5Args {
  View view;
  Geom g;
  const float* win;        // window, float32 (512)
  const double* win64;     // window, float64 (512): exact refinement
  const cf* tw512;         // w_512^j (512)
  const cx<double>* tw64;  // w_512^j float64 (256 entries: j < 256; w^(j+256) = -w^j)
  ThreshConsts tc;
  double mag_scale, top_db;
  unsigned long long* bits;  // decide: [units][T][5]
  float* mag;                // magnitude: [units][T][FS]
  const float* Mf;           // apply: float mask [units][T][FS], natural bin order
  const unsigned short* K;   // apply<KMASK>: integer weight sums of the smoothed bit mask [units][T][FS] (mask = K / ktot)
  float inv_ktot;
  const float* wsq;          // apply: window squared (512)
  const float* invn;         // apply: 1 / sum_q wsq[128 q + s], s < 128
  OutMap om;
  int64_t h_begin, h_end;    // apply: ext hops (128-sample blocks, ext = unit sample + padL) to produce
  int normalize;
  FloorLazy fl;              // decide: in-kernel floor test (thresh.hpp), alim == nullptr: flags computed a priori
  float* part;               // apply / one-pass gate, seam mode: [units][tiles][6][hop] un-normalised partial hops (3 leading, 3 trailing: k_ola_seam), else nullptr
  int n_tiles;
  double iir_b;              // magnitude: the recurrence's b (non-stationary gate) ...
  double* sub;               // ... and its per-tile partials [units][tiles][2][FS] (fastpath.hpp: mag_sub_partials), or nullptr
};

// stage tables + the tile's sample span, gather the lane's 32 complex points of its frame pair:
// v[r] = (a[m], b[m]) * w[m], m = c + 16 r; frame A = tq + 2 g, frame B = A + 1.  `tf0`: first frame of the tile.
// Returns with the span consumed (the exchange slices may be overwritten).
template <int WAVES, bool MX = false>
__device__ __forceinline__ unsigned f5_gather(const Fast5Args& A, cf* tw512, cf* regions, float* swin, int64_t row,
                                          int64_t chunk, int64_t tf0, int64_t t_lim, cf* v, bool& validA, bool& validB) {   // returns (MX) the largest |sample| this thread staged, as a bit pattern
  unsigned mx_ = 0u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  constexpr int NF = F5_FPW * WAVES, ROWS = NF - 1 + 4, SPAN = ROWS * F5_H;
  static_assert(ROWS * F5_XP * 4 <= WAVES * WAVE_CX_H * 8, "span must fit the exchange slices");
  stage_tables<WAVES * 64, 128>(tw512, A.tw512, swin, A.win, tid);
  const Geom& G = A.g;
  const int64_t s0b = tf0 * F5_H - G.padL;
  const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
  const float* sp = (const float*)A.view.x + row * A.view.stride + gb;
  const bool vec = A.view.dtype == 0 && tf0 >= 0 && tf0 + NF <= G.T && s0b >= 0 && s0b + SPAN <= A.view.Lp &&
                   gb >= A.view.lo && gb + SPAN <= A.view.hi && (reinterpret_cast<uintptr_t>(sp) & 15) == 0;
  float* xs = reinterpret_cast<float*>(regions);
  if (vec) {
    const unsigned m = stage_span_vec<WAVES * 64, SPAN, F5_XP, 128, MX>(xs, sp, tid);
    if constexpr (MX) mx_ = m;
  } else {
    unsigned m = 0u;
    for (int i = tid; i < SPAN; i += WAVES * 64) {
      const float xv = (float)view_sample(A.view, row, chunk, s0b + i);
      xs[(i >> 7) * F5_XP + (i & 127)] = xv;
      m = max(m, __float_as_uint(xv) & 0x7fffffffu);
    }
    if constexpr (MX) mx_ = m;
  }
  __syncthreads();
  const int fa = F5_FPW * wave + 2 * g;                       // tile-local index of frame A
  const int64_t tA = tf0 + fa;
  validA = tA >= 0 && tA < t_lim && tA < G.T;
  validB = tA + 1 >= 0 && tA + 1 < t_lim && tA + 1 < G.T;
  const float* xa = xs + fa * F5_XP + c;
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int off = (r >> 3) * F5_XP + 16 * (r & 7);
    const float w = swin[c + 16 * r];
    float a = xa[off], b = xa[off + F5_XP];
    if (!validA) a = 0.f;                                      // frames before / past the unit: zeros
    if (!validB) b = 0.f;
    v[r] = {a * w, b * w};
  }
  __syncthreads();
  return mx_;
}

// the 16 conjugate pairs of a lane in slot order: (a, b) = (Z[k], Z[512 - k]); lane 0 pairs its self-conjugate rows
// differently (fastpath.hpp).  Slot 0 of lane 0 is NOT a pair: v[0] = Z[0], v[8] = Z[256] are handled by the callers.
// (round 6: the selects are v_cndmask_b32_e64 on a constant SGPR-pair lane mask -- sel_s, fastpath.hpp -- instead of `l0 ? :`
// next to its compare: 4 cycles of the vector pipe instead of 20 each, and opaque to the SLP vectoriser, which turned the select
// chains of the one-pass kernel into <7 x i32> operations on a stack copy of five spectrum values)
constexpr unsigned long long F5_L0 = 0x0001000100010001ull;   // lanes with c == 0
__device__ __forceinline__ void f5_pair(const cf* v, int sl, bool l0, cf& a, cf& b) {
  (void)l0;
  auto sel = [&](cf a0, cf a1) -> cf { return {sel_s(F5_L0, a0.x, a1.x), sel_s(F5_L0, a0.y, a1.y)}; };
  if (sl == 0) { a = v[0]; b = v[31]; return; }
  a = sl < 8 ? v[sl] : sel(v[8 + sl], v[sl]);
  b = sl < 8 ? sel(v[16 - sl], v[31 - sl]) : sel(v[39 - sl], v[31 - sl]);
}

__device__ __forceinline__ double f5_exact_power(const Fast5Args& A, int64_t row, int64_t chunk, int64_t t, int f, int lane) {
  const int64_t s0 = t * F5_H - A.g.padL;
  double re = 0.0, im = 0.0;
#pragma unroll 4
  for (int i = 0; i < 8; ++i) {
    const int m = lane + 64 * i;
    const double xv = view_sample(A.view, row, chunk, s0 + m) * A.win64[m];
    const int j = (f * m) & 511;
    cx<double> w = A.tw64[j & 255];
    if (j >= 256) { w.x = -w.x; w.y = -w.y; }
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
__global__ __launch_bounds__(WAVES * 64, 3) void k_decide_fast512(Fast5Args A) {
  if (REDO && A.fl.alim[1] != A.tc.need_tag) return;   // no unit of this call reported (the common case)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  float* s_t2 = swin + F5_N;                 // [257] float32 compare constants x4 (the split works on 2 X)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  // lazy (A.fl.alim set): the first launch assumes "no floor live" and tests the samples it stages; REDO serves the flagged units
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
  stage_t2_plain<WAVES * 64, F5_F>(s_t2, A.tc.T2, need, 4.0, tid, t2eff);
  constexpr int NF = F5_FPW * WAVES;
  const int64_t tf0 = (int64_t)blockIdx.x * NF;
  cf v[32];
  bool validA, validB;
  const unsigned fl_mx = f5_gather<WAVES, true>(A, tw512, regions, swin, row, chunk, tf0, G.T, v, validA, validB);
  if (!REDO) floor_lazy_report(A.tc, A.fl, fl_bound, fl_mx, u, G.FS, lane);
  const int64_t tq = tf0 + F5_FPW * wave;
  if (tq >= G.T) return;   // wave-uniform; no barrier below
  // delta^2 = 2^-32 ||x w||^2 per frame (see k_decide_fast): the two frames' norms from the real / imaginary parts
  float nA = 0.f, nB = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) { nA += v[r].x * v[r].x; nB += v[r].y * v[r].y; }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { nA += __shfl_xor(nA, o); nB += __shfl_xor(nB, o); }
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
  {
    int z0 = 0;
    asm volatile("" : "+v"(z0));
    fft512_fwd_half(v, fb, tw512 + z0, c);
  }
  const bool l0 = c == 0;
  // (one transform carries both frames: the rounding error in either spectrum scales with the norm of the PAIR)
  const float nAB = nA + nB;
  const float dA = nAB > 0.f ? 8.0f * 2.3283064e-10f * nAB : -1.0f;
  const float dB = dA;
  unsigned pA = 0, pB = 0, aA = 0, aB = 0;          // bit sl: decision / ambiguity of slot sl's bin, frames A and B
  bool p256A = false, p256B = false, a256A = false, a256B = false;
  auto decide = [&](float P, float T, float d2, unsigned& pr, unsigned& am, int q) {
    const float diff = P - T;
    pr |= (diff > 0.f ? 1u : 0u) << q;
    am |= ((diff * diff <= d2 * (P + T)) ? 1u : 0u) << q;
  };
#pragma unroll
  for (int sl = 0; sl < 16; ++sl) {
    cf a, b;
    f5_pair(v, sl, l0, a, b);
    const cf E = {a.x + b.x, a.y - b.y}, O = {a.y + b.y, b.x - a.x};   // 2 A[k], 2 B[k]
    float PA = E.x * E.x + E.y * E.y, PB = O.x * O.x + O.y * O.y;
    if (sl == 0) {   // lane 0: Z[0] is its own partner: A[0] = Re Z[0], B[0] = Im Z[0] (x4 like the others)
      const float xa = 2.f * v[0].x, xb = 2.f * v[0].y;
      PA = l0 ? xa * xa : PA;
      PB = l0 ? xb * xb : PB;
    }
    const float T = s_t2[bin5(c, sl)];
    decide(PA, T, dA, pA, aA, sl);
    decide(PB, T, dB, pB, aB, sl);
  }
  {  // bin 256 (lane 0: v[8] = Z[256] is its own partner)
    const float xa = 2.f * v[8].x, xb = 2.f * v[8].y, T = s_t2[256];
    const float da = xa * xa - T, db = xb * xb - T;
    p256A = l0 && da > 0.f;
    p256B = l0 && db > 0.f;
    a256A = l0 && da * da <= dA * (xa * xa + T);
    a256B = l0 && db * db <= dB * (xb * xb + T);
  }
  if (need == 2) { pA = pB = 0; aA = aB = 0; p256A = p256B = a256A = a256B = false; }   // NaN powers have no sign
  if (!validA) { pA = 0; aA = 0; p256A = a256A = false; }
  if (!validB) { pB = 0; aB = 0; p256B = a256B = false; }
  // exact re-evaluation of ambiguous cells, one at a time, whole wave cooperating
  while (true) {
    const unsigned long long pending = __ballot(aA != 0 || aB != 0 || a256A || a256B);
    if (pending == 0) break;
    const int src = __ffsll((long long)pending) - 1;
    const unsigned sA = (unsigned)__shfl((int)aA, src), sB = (unsigned)__shfl((int)aB, src);
    const int s256A = __shfl((int)a256A, src);
    const int cs = src & 15, gs = src >> 4;
    int which, f;   // 0: A slot, 1: B slot, 2: A bin 256, 3: B bin 256
    int q = 0;
    if (sA) { which = 0; q = __ffs((int)sA) - 1; f = bin5(cs, q); }
    else if (sB) { which = 1; q = __ffs((int)sB) - 1; f = bin5(cs, q); }
    else if (s256A) { which = 2; f = 256; }
    else { which = 3; f = 256; }
    const int64_t t = tq + 2 * gs + (which & 1);
    const Fast5Args& L = *late_args<Fast5Args>();       // (cold path: arguments re-read here, not kept live from the entry)
    const double P = f5_exact_power(L, row, chunk, t, f, lane);
    double t2 = L.tc.T2[f];
    if (floor_live) {
      const double fl = cell_db(L.tc.pmax[u * (int64_t)L.g.FS + f], L.mag_scale) - L.top_db;
      if (fl > L.tc.thresh[f]) t2 = -1.0;
    }
    if (need == 2) t2 = T2_NEVER;
    const bool pass = P > t2;
    if (lane == src) {
      if (which == 0) { pA = (pA & ~(1u << q)) | ((pass ? 1u : 0u) << q); aA &= ~(1u << q); }
      else if (which == 1) { pB = (pB & ~(1u << q)) | ((pass ? 1u : 0u) << q); aB &= ~(1u << q); }
      else if (which == 2) { p256A = pass; a256A = false; }
      else { p256B = pass; a256B = false; }
    }
  }
  // Pack.  A 16 x 16 bit transpose across the lane group (k_gate_onepass: four xor-shuffle steps on both 16-bit halves
  // at once) leaves in lane k the FIELD of slot k: bit c = the decision of lane c, low half frame A, high half frame
  // B.  Field of slot sl < 8: bins 32 sl + c (c = 0..15); sl >= 8: lanes c >= 1 hold bins 32 j + 32 - c (j = 15 - sl:
  // bit c -> position 16 - c of block j's upper half) and lane 0 holds bin 16 + 32 (sl - 8) (position 0 of block
  // sl - 8's upper half).  Word w = blocks 2 w, 2 w + 1; lane c < 4 of the group assembles word c.
  unsigned tr = (pA & 0xffffu) | (pB << 16);
  auto tstep = [&](int sft, unsigned msk) {
    const unsigned y = (unsigned)__shfl_xor((int)tr, sft);
    const bool up = (c & sft) != 0;
    const unsigned ysh = up ? (y >> sft) : (y << sft);
    const unsigned mk = up ? msk : ~msk;
    tr = (tr & ~mk) | (ysh & mk);
  };
  tstep(8, 0x00ff00ffu);
  tstep(4, 0x0f0f0f0fu);
  tstep(2, 0x33333333u);
  tstep(1, 0x55555555u);
  const int gl = lane & 48, w = c & 3;
  unsigned long long wA = 0ull, wB = 0ull;
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int j = 2 * w + h2;                                            // 32-bin block
    const unsigned lo = (unsigned)__shfl((int)tr, gl | j);               // slot j:      bins 32 j + 0..15
    const unsigned up = (unsigned)__shfl((int)tr, gl | (15 - j));        // slot 15 - j: lanes c >= 1 -> bins 32 j + 32 - c
    const unsigned z0 = (unsigned)__shfl((int)tr, gl | (8 + j));         // slot 8 + j:  lane 0 -> bin 32 j + 16
    const unsigned upA = up & 0xfffeu, upB = (up >> 16) & 0xfffeu;
    const unsigned hiA = (((__brev(upA) >> 16) << 1) & 0xffffu) | (z0 & 1u);
    const unsigned hiB = (((__brev(upB) >> 16) << 1) & 0xffffu) | ((z0 >> 16) & 1u);
    wA |= (unsigned long long)((lo & 0xffffu) | (hiA << 16)) << (32 * h2);
    wB |= (unsigned long long)((lo >> 16) | (hiB << 16)) << (32 * h2);
  }
  {
    const int sh = 16 * g;
    const unsigned long long bA = __ballot(p256A), bB = __ballot(p256B);   // lane 0 of every group
    if (c == 4) { wA = (bA >> sh) & 1ull; wB = (bB >> sh) & 1ull; }
  }
  const int64_t tA = tq + 2 * g;
  if (c < 5) {
    if (validA) A.bits[(u * G.T + tA) * 5 + c] = wA;
    if (validB) A.bits[(u * G.T + tA + 1) * 5 + c] = wB;
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 3) void k_mag_fast512(Fast5Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  constexpr int NF = F5_FPW * WAVES;
  const int64_t tf0 = (int64_t)blockIdx.x * NF;
  cf v[32];
  bool validA, validB;
  f5_gather<WAVES>(A, tw512, regions, swin, row, chunk, tf0, G.T, v, validA, validB);
  const int64_t tq = tf0 + F5_FPW * wave;
  const bool with_sub = A.sub != nullptr;
  constexpr int TP = 260;   // floats between the rows of the |X| tile (with_sub): 8 rows per wave in its own exchange slice
  static_assert(F5_FPW * TP * 4 <= WAVE_CX_H * 8, "a wave's |X| rows fit its exchange slice");
  if (tq < G.T) {   // (wave-uniform)
    cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
    fft512_fwd_half(v, fb, tw512, c);
    const bool l0 = c == 0;
    const int64_t tA = tq + 2 * g;
    float* mA = A.mag + (u * G.T + (validA ? tA : 0)) * (int64_t)G.FS;
    float* mB = A.mag + (u * G.T + (validB ? tA + 1 : 0)) * (int64_t)G.FS;
    float* tA_ = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + (2 * g) * TP;   // (the wave's transform is done)
    float* tB_ = tA_ + TP;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) {
      cf a, b;
      f5_pair(v, sl, l0, a, b);
      const cf E = {a.x + b.x, a.y - b.y}, O = {a.y + b.y, b.x - a.x};
      float PA = E.x * E.x + E.y * E.y, PB = O.x * O.x + O.y * O.y;
      if (sl == 0) {
        const float xa = 2.f * v[0].x, xb = 2.f * v[0].y;
        PA = l0 ? xa * xa : PA;
        PB = l0 ? xb * xb : PB;
      }
      const int f = bin5(c, sl);
      const float ma = half_sqrt(PA), mb = half_sqrt(PB);
      if (validA) mA[f] = ma;
      if (validB) mB[f] = mb;
      if (with_sub) { tA_[f] = ma; tB_[f] = mb; }
    }
    if (l0) {
      if (validA) mA[256] = fabsf(v[8].x);
      if (validB) mB[256] = fabsf(v[8].y);
      if (with_sub) { tA_[256] = fabsf(v[8].x); tB_[256] = fabsf(v[8].y); }
    }
  }
  if (!with_sub) return;
  __syncthreads();
  mag_sub_partials<WAVES * 64, NF, F5_FPW, TP, F5_F>(regions, (int)min<int64_t>((int64_t)NF, G.T - tf0), A.iir_b,
                                                    A.sub + ((u * gridDim.x + blockIdx.x) * 2) * (int64_t)G.FS, G.FS, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// Apply: FFT -> x mask -> IFFT -> window -> overlap-add -> samples.  Tiles overlap by 3 frames: a tile of NF frames
// completes NF - 3 hops on its own.
template <int WAVES, bool KMASK>
__global__ __launch_bounds__(WAVES * 64, 3) void k_apply_fast512(Fast5Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  constexpr int NF = F5_FPW * WAVES, NH = NF - 3;
  const bool seam = A.part != nullptr;        // abutting tiles + k_ola_seam (fastpath.hpp), else tiles that overlap by 3 frames
  const int64_t tf0 = A.h_begin - 3 + (int64_t)blockIdx.x * (seam ? NF : NH);   // first frame of the tile
  cf v[32];
  bool validA, validB;
  f5_gather<WAVES>(A, tw512, regions, swin, row, chunk, tf0, G.T, v, validA, validB);
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
  const bool l0 = c == 0;
  const int64_t tA = tf0 + F5_FPW * wave + 2 * g;
  const bool wave_live = tf0 + F5_FPW * wave + F5_FPW - 1 >= 0 && tf0 + F5_FPW * wave < G.T;
  if (wave_live) {
    {
      int z0 = 0;
      asm volatile("" : "+v"(z0));
      fft512_fwd_half(v, fb, tw512 + z0, c);
    }
    // X_A = E / 2, X_B = O / 2; Y = X * mask; Z'[k] = Ya + i Yb, Z'[512 - k] = conj Ya + i conj Yb.  The 1/2 of the
    // split and the 1/512 of the inverse transform ride in the mask scale.
    const float ks = (KMASK ? A.inv_ktot : 1.0f) * (0.5f / 512.0f);
    const int64_t offA = (u * G.T + (validA ? tA : 0)) * (int64_t)G.FS, offB = (u * G.T + (validB ? tA + 1 : 0)) * (int64_t)G.FS;
    float ma[16], mb[16];
    float m256a, m256b;
    // (round 6) A frame outside [0, T) shares its transform with a real one: its spectrum comes out of the split as rounding
    // residue of the partner's (1e-8), and its mask is ZERO (a zero SCALE: one select per frame -- a select per entry made the
    // compiler branch around the mask loads, 52 -> 75 us) -- not whatever row the address clamp lands on.
    const float ksA = validA ? ks : 0.f, ksB = validB ? ks : 0.f;
    if constexpr (KMASK) {
      const unsigned short *KA = A.K + offA, *KB = A.K + offB;
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) {
        const int f = bin5(c, sl);
        ma[sl] = (float)KA[f] * ksA;
        mb[sl] = (float)KB[f] * ksB;
      }
      m256a = (float)KA[256] * (2.f * ksA);
      m256b = (float)KB[256] * (2.f * ksB);
    } else {
      const float *MA = A.Mf + offA, *MB = A.Mf + offB;
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) {
        const int f = bin5(c, sl);
        ma[sl] = MA[f] * ksA;
        mb[sl] = MB[f] * ksB;
      }
      m256a = MA[256] * (2.f * ksA);
      m256b = MB[256] * (2.f * ksB);
    }
    cf na[16], nb[16];
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) {
      cf a, b;
      f5_pair(v, sl, l0, a, b);
      const cf E = {a.x + b.x, a.y - b.y}, O = {a.y + b.y, b.x - a.x};
      const cf Ya = {E.x * ma[sl], E.y * ma[sl]}, Yb = {O.x * mb[sl], O.y * mb[sl]};
      na[sl] = {Ya.x - Yb.y, Ya.y + Yb.x};
      nb[sl] = {Ya.x + Yb.y, Yb.x - Ya.y};
    }
    // scatter back (the inverse of f5_pair): lanes >= 1: register sl <- na[sl], register 31 - sl <- nb[sl]
    auto sel = [&](cf a0, cf a1) -> cf { return {sel_s(F5_L0, a0.x, a1.x), sel_s(F5_L0, a0.y, a1.y)}; };
    cf nv[32];
    {
      // lane 0, slot 0: Z[0] and Z[256] are their own partners: Z' = (Re Z ma, Im Z mb) with the masks of bins 0 / 256
      const cf z0 = {v[0].x * (2.f * ma[0]), v[0].y * (2.f * mb[0])};
      const cf z8 = {v[8].x * m256a, v[8].y * m256b};
      nv[0] = sel(z0, na[0]);
      nv[8] = z8;           // lane 0 only; lanes >= 1 overwrite below
      nv[31] = nb[0];       // lanes >= 1 only; lane 0 overwrites below
    }
#pragma unroll
    for (int i = 1; i < 8; ++i) nv[i] = na[i];
    {
      const cf keep8 = nv[8];
      nv[8] = sel(keep8, na[8]);
    }
#pragma unroll
    for (int i = 9; i < 16; ++i) nv[i] = sel(nb[16 - i], na[i]);
#pragma unroll
    for (int i = 16; i < 24; ++i) nv[i] = sel(na[i - 8], nb[31 - i]);
#pragma unroll
    for (int i = 24; i < 31; ++i) nv[i] = sel(nb[39 - i], nb[31 - i]);
    {
      const cf keep31 = nv[31];
      nv[31] = sel(nb[8], keep31);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = nv[i];
    {
      int zi = 0, ci = c;
      asm volatile("" : "+v"(zi), "+v"(ci));
      fft512_inv_half(v, fb + zi, tw512 + zi, ci);
    }
  }
  // wave-private overlap-add of the wave's 8 frames into 11 hop accumulators (reusing the exchange slices).  Step j:
  // every frame adds its quarter j: frame f touches hop f + j -- within a step no two frames touch the same hop, and a
  // hop receives its quarters in the fixed order j = 0..3 (LDS operations of a wave execute in order).
  float* acc = reinterpret_cast<float*>(regions + wave * WAVE_CX_H);
  static_assert((F5_FPW + 3) * F5_HP * 4 <= WAVE_CX_H * 8, "hop accumulators must fit the wave's slice");
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int fA = 2 * g, fB = 2 * g + 1;
    const bool firstA = j == 0, firstB = (j == 0) || (g == 3);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = 8 * j + rr;
      const float ws = swin[c + 16 * r];
      float* dA = acc + (fA + j) * F5_HP + c + 16 * rr;
      float* dB = acc + (fB + j) * F5_HP + c + 16 * rr;
      float ya = v[r].x * ws, yb = v[r].y * ws;
      if (!firstA) ya += *dA;
      *dA = ya;
      if (!firstB) yb += *dB;
      *dB = yb;
    }
    wave_lds_sync();
  }
  __syncthreads();
  // cross-wave combine: tile hop jj = wave (jj >> 3)'s local hop jj & 7 plus, for jj & 7 <= 2, the previous wave's
  // local hop (jj & 7) + 8 (fixed order: earlier wave first); 32 threads x float4 per hop
  const float* fr = reinterpret_cast<const float*>(regions);
  const int s4 = (tid & 31) * 4;
  for (int jj = (seam ? 0 : 3) + (tid >> 5); jj < (seam ? NF + 3 : NF); jj += (WAVES * 64) >> 5) {
    const int64_t h = tf0 + jj;
    if (h < A.h_begin || h >= A.h_end) continue;
    const int wv = jj < NF ? jj >> 3 : WAVES - 1, lh = jj < NF ? jj & 7 : 8 + (jj - NF);   // (jj >= NF: the last wave's overflow rows)
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wv >= 1 && lh <= 2) a4 = *reinterpret_cast<const float4*>(&fr[(wv - 1) * WAVE_CX_H * 2 + (lh + 8) * F5_HP + s4]);
    {
      const float4 f4 = *reinterpret_cast<const float4*>(&fr[wv * WAVE_CX_H * 2 + lh * F5_HP + s4]);
      a4.x += f4.x; a4.y += f4.y; a4.z += f4.z; a4.w += f4.w;
    }
    if (seam && (jj < 3 || jj >= NF)) {   // straddling hop: partial sum only; slots 0..2 leading, 3..5 trailing
      const int slot = jj < 3 ? jj : 3 + (jj - NF);
      *reinterpret_cast<float4*>(A.part + ((((size_t)u * A.n_tiles + blockIdx.x) * 6 + slot) * F5_H + s4)) = a4;
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
          const float4 w4 = *reinterpret_cast<const float4*>(&A.wsq[F5_H * q + s4]);
          nrm.x += w4.x; nrm.y += w4.y; nrm.z += w4.z; nrm.w += w4.w;
        }
      }
      a4.x /= (nrm.x > 1e-10f ? nrm.x : 1.f);
      a4.y /= (nrm.y > 1e-10f ? nrm.y : 1.f);
      a4.z /= (nrm.z > 1e-10f ? nrm.z : 1.f);
      a4.w /= (nrm.w > 1e-10f ? nrm.w : 1.f);
    }
    {
      const int64_t pb = h * F5_H - G.padL;
      const int64_t gi0 = chunk * A.om.g_step + (pb - A.om.p0);
      if (A.om.dtype == 0 && pb >= A.om.p0 && pb + F5_H <= A.om.p1 && pb + F5_H <= G.Lout && gi0 >= A.om.g_lo &&
          gi0 + F5_H <= A.om.g_hi) {
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
      const int64_t p = h * F5_H + s4 + e - G.padL;
      if (p < A.om.p0 || p >= A.om.p1) continue;
      const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
      if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
      store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? vals[e] : 0.f);
    }
  }
}

}  // namespace fast
}  // namespace sg
