// Fast path for n_fft = win_length = 256, hop = 64 (8 kHz telephony, short frames at 16 kHz), float32, on the SAME register
// transform as the 1024 / 512 paths (fastpath.hpp: a lane group of 16 lanes, 32 complex points per lane, DFT32 in registers
// -> twiddle -> one swizzled LDS exchange -> second stage in registers) -- a lane group carries FOUR consecutive real frames:
//
//   * two real frames ride in one complex sequence (fast512.hpp): p1 = a + i b, p2 = c + i d, 256 points each;
//   * the two sequences are dealt to the lanes by parity: lane c holds p[cp + 8 r], r < 32, cp = c >> 1, of sequence c & 1.
//     256 = 32 x 8:  P[k1 + 32 k2] = sum_cp w_8^(cp k2) [ w_256^(cp k1) sum_r p[cp + 8 r] w_32^(r k1) ]
//     -- stage 1 is the 1024 path's DFT32 with the twiddle column c & ~1 (w_512^(2 cp k1) = w_256^(cp k1)), the exchange is
//     unchanged, stage 2 is TWO DFT8 over the even / odd columns of a row instead of one DFT16: the sequences never mix.
//   After the forward transform lane c' holds rows k1 = c' and 32 - c' (lane 0: rows 0 and 16) of BOTH sequences:
//     v[j] = P1[row1 + 32 j], v[8 + j] = P2[row1 + 32 j], v[16 + j] = P1[row2 + 32 j], v[24 + j] = P2[row2 + 32 j],  j < 8
//   and the conjugate pair (k, 256 - k) of the two-real split sits in one lane: 2 A[k] = Z[k] + conj Z[256 - k],
//   2 B[k] = (Z[k] - conj Z[256 - k]) / i -- no twiddle, no cross-lane traffic.
// One wavefront = 16 frames, one workgroup (4 waves) = a tile of 64 frames -> 61 complete hops of 64 samples (tiles overlap
// by 3 frames: 5 % redundant transforms, no hand-off between workgroups).
//
//   k_decide_fast256   frames -> FFT -> |X|^2 of the four frames against the compare constants (float32 + the exact float64
//                      refinement of k_decide_fast) -> mask bits [unit][frame][3 words]
//   k_mag_fast256      frames -> FFT -> |X| (float32, natural bin order) for the non-stationary masks
//   k_apply_fast256    frames -> FFT -> x mask (float, or the K counts of the bit path) -> IFFT -> window -> overlap-add
// Slot e < 8 of lane c pairs (Z[k], Z[256 - k]) with  k = c + 32 e  (lanes >= 1; the REAL bin is min(k, 256 - k) = bin6(c, e));
// lane 0 pairs inside its rows: slots 1..3 = bins 32 e (row 0), slots 4..7 = bins 16 + 32 (e - 4) (row 16), slot 0 = DC, and
// bin 128 (row 0, j = 4) rides separately.
#pragma once
#include "fastpath.hpp"

namespace sg {
namespace fast {

constexpr int F25_N = 256, F25_H = 64, F25_F = 129;
constexpr int F25_FPW = 16;                 // frames per wave
constexpr int F25_XP = 68;                  // floats between the 64-sample rows of the staged span: 2 XP = 8 (mod 64), so the
constexpr int F25_HP = 68;                  // 8 (group, parity) pairs of a wave land 8 banks apart -- conflict-free
constexpr int F25_T2 = 136;                 // floats of the compare-constant table

__host__ __device__ inline int bin6(int c, int e) {   // real bin of pair slot e (0..7) in lane c
  if (c != 0) return e < 4 ? c + 32 * e : (32 - c) + 32 * (7 - e);
  return e < 4 ? 32 * e : 16 + 32 * (e - 4);
}

struct Fast25Args {
  View view;
  Geom g;
  const float* win;        // window, float32 (256)
  const double* win64;     // window, float64 (256): exact refinement
  const cf* tw512;         // w_512^j (512)
  const cx<double>* tw64;  // w_256^j float64 (128 entries; w^(j+128) = -w^j)
  ThreshConsts tc;
  double mag_scale, top_db;
  unsigned long long* bits;  // decide: [units][T][3]
  float* mag;                // magnitude: [units][T][FS]
  const float* Mf;           // apply: float mask [units][T][FS], natural bin order
  const unsigned short* K;   // apply<KMASK>: integer weight sums of the smoothed bit mask [units][T][FS] (mask = K / ktot)
  float inv_ktot;
  const float* wsq;          // apply: window squared (256)
  const float* invn;         // apply: 1 / sum_q wsq[64 q + s], s < 64
  OutMap om;
  int64_t h_begin, h_end;    // apply: ext hops (64-sample blocks, ext = unit sample + padL) to produce
  int normalize;
  FloorLazy fl;              // decide: in-kernel floor test (thresh.hpp), alim == nullptr: flags computed a priori
  float* part;               // apply / one-pass gate, seam mode: [units][tiles][6][hop] un-normalised partial hops (3 leading, 3 trailing: k_ola_seam), else nullptr
  int n_tiles;
  double iir_b;              // magnitude: the recurrence's b (non-stationary gate) ...
  double* sub;               // ... and its per-tile partials [units][tiles][2][FS] (fastpath.hpp: mag_sub_partials), or nullptr
};

// second stage: two DFT8 per row (even columns = sequence 1, odd columns = sequence 2)
template <bool INV>
__device__ __forceinline__ void f25_stage2(cf* v) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    cf e[8], o[8];
    if (!INV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { e[j] = v[16 * h + 2 * j]; o[j] = v[16 * h + 2 * j + 1]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { e[j] = v[16 * h + j]; o[j] = v[16 * h + 8 + j]; }
    }
    dft_reg<8, INV>(e);
    dft_reg<8, INV>(o);
    if (!INV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[16 * h + j] = e[j]; v[16 * h + 8 + j] = o[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[16 * h + 2 * j] = e[j]; v[16 * h + 2 * j + 1] = o[j]; }
    }
  }
}

// forward, half-size exchange slice (fft512_fwd_half with the 256-point stages): v[r] = p[cp + 8 r] -> the row layout above
__device__ __forceinline__ void f25_fwd_half(cf* v, cf* fb, const cf* tw512, int c) {
  __builtin_amdgcn_sched_barrier(0);
  stage_dft32_tw_fwd(v, tw512, c & ~1);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) fb[k1 * 16 + (c ^ (2 * ((k1 >> 1) & 7)))] = v[k1];
  wave_lds_sync();
  xchg_read_row(fb, row1(c), v);
  wave_lds_sync();
#pragma unroll
  for (int k1 = 16; k1 < 32; ++k1) fb[(k1 - 16) * 16 + (c ^ (2 * ((k1 >> 1) & 7)))] = v[k1];
  wave_lds_sync();
  xchg_read_row(fb, row2(c) - 16, v + 16);
  wave_lds_sync();
  f25_stage2<false>(v);
  __builtin_amdgcn_sched_barrier(0);
}

// inverse (unnormalised): the row layout -> v[r] = 256 p[cp + 8 r]
__device__ __forceinline__ void f25_inv_half(cf* v, cf* fb, const cf* tw512, int c) {
  __builtin_amdgcn_sched_barrier(0);
  f25_stage2<true>(v);
  xchg_write_row(fb, row1(c), v);
  wave_lds_sync();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) v[k1] = fb[k1 * 16 + (c ^ (2 * ((k1 >> 1) & 7)))];
  wave_lds_sync();
  xchg_write_row(fb, row2(c) - 16, v + 16);
  wave_lds_sync();
#pragma unroll
  for (int k1 = 16; k1 < 32; ++k1) v[k1] = fb[(k1 - 16) * 16 + (c ^ (2 * ((k1 >> 1) & 7)))];
  wave_lds_sync();
  stage_tw_dft32_inv(v, tw512, c & ~1);
  __builtin_amdgcn_sched_barrier(0);
}

// stage tables + the tile's sample span, gather the lane's 32 complex points: v[r] = (x[m], y[m]) * w[m], m = cp + 8 r, of the
// lane's frame pair X = tf0 + 16 wave + 4 g + 2 (c & 1), Y = X + 1.  Returns with the span consumed.
template <int WAVES, bool MX = false>
__device__ __forceinline__ unsigned f25_gather(const Fast25Args& A, cf* tw512, cf* regions, float* swin, int64_t row,
                                           int64_t chunk, int64_t tf0, cf* v, bool& validX, bool& validY) {   // returns (MX) the largest |sample| this thread staged, as a bit pattern
  unsigned mx_ = 0u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15, cp = c >> 1;
  constexpr int NF = F25_FPW * WAVES, ROWS = NF - 1 + 4, SPAN = ROWS * F25_H;
  static_assert(ROWS * F25_XP * 4 <= WAVES * WAVE_CX_H * 8, "span must fit the exchange slices");
  stage_tables<WAVES * 64, 64>(tw512, A.tw512, swin, A.win, tid);
  const Geom& G = A.g;
  const int64_t s0b = tf0 * F25_H - G.padL;
  const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
  const float* sp = (const float*)A.view.x + row * A.view.stride + gb;
  const bool vec = A.view.dtype == 0 && tf0 >= 0 && tf0 + NF <= G.T && s0b >= 0 && s0b + SPAN <= A.view.Lp &&
                   gb >= A.view.lo && gb + SPAN <= A.view.hi && (reinterpret_cast<uintptr_t>(sp) & 15) == 0;
  float* xs = reinterpret_cast<float*>(regions);
  if (vec) {
    const unsigned m = stage_span_vec<WAVES * 64, SPAN, F25_XP, 64, MX>(xs, sp, tid);
    if constexpr (MX) mx_ = m;
  } else {
    unsigned m = 0u;
    for (int i = tid; i < SPAN; i += WAVES * 64) {
      const float xv = (float)view_sample(A.view, row, chunk, s0b + i);
      xs[(i >> 6) * F25_XP + (i & 63)] = xv;
      m = max(m, __float_as_uint(xv) & 0x7fffffffu);
    }
    if constexpr (MX) mx_ = m;
  }
  __syncthreads();
  const int fx = F25_FPW * wave + 4 * g + 2 * (c & 1);         // tile-local index of frame X
  const int64_t tX = tf0 + fx;
  validX = tX >= 0 && tX < G.T;
  validY = tX + 1 >= 0 && tX + 1 < G.T;
  const float* xa = xs + fx * F25_XP + cp;
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int off = (r >> 3) * F25_XP + 8 * (r & 7);
    const float w = swin[cp + 8 * r];
    float a = xa[off], b = xa[off + F25_XP];
    if (!validX) a = 0.f;                                      // frames before / past the unit: zeros
    if (!validY) b = 0.f;
    v[r] = {a * w, b * w};
  }
  __syncthreads();
  return mx_;
}

// the conjugate pair of slot e of sequence `off` (0: frames A, B; 8: frames C, D): (a, b) = (Z[k], Z[256 - k]).  Slot 0 of
// lane 0 is NOT a pair (callers handle DC and bin 128).
constexpr unsigned long long F25_L0 = 0x0001000100010001ull;   // lanes with c == 0 (sel_s: fastpath.hpp; see fast512.hpp)
__device__ __forceinline__ void f25_pair(const cf* v, int off, int e, bool l0, cf& a, cf& b) {
  (void)l0;
  auto sel = [&](cf a0, cf a1) -> cf { return {sel_s(F25_L0, a0.x, a1.x), sel_s(F25_L0, a0.y, a1.y)}; };
  const cf* pa = v + off;
  const cf* pb = v + 16 + off;
  a = e < 4 ? pa[e] : sel(pb[e - 4], pa[e]);
  if (e == 0) b = pb[7];
  else if (e < 4) b = sel(pa[8 - e], pb[7 - e]);
  else b = sel(pb[11 - e], pb[7 - e]);
}

__device__ __forceinline__ double f25_exact_power(const Fast25Args& A, int64_t row, int64_t chunk, int64_t t, int f, int lane) {
  const int64_t s0 = t * F25_H - A.g.padL;
  double re = 0.0, im = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = lane + 64 * i;
    const double xv = view_sample(A.view, row, chunk, s0 + m) * A.win64[m];
    const int j = (f * m) & 255;
    cx<double> w = A.tw64[j & 127];
    if (j >= 128) { w.x = -w.x; w.y = -w.y; }
    re += xv * w.x;
    im += xv * w.y;
  }
  for (int off = 32; off > 0; off >>= 1) {
    re += __shfl_xor(re, off);
    im += __shfl_xor(im, off);
  }
  return re * re + im * im;
}

// 4 |X|^2 of the lane's 8 slots for the four frames: P[fr][e]; lane 0: slot 0 = DC, P128[fr] = bin 128
__device__ __forceinline__ void f25_powers(const cf* v, bool l0, float (&P)[4][8], float (&P128)[4]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int off = 8 * s;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cf a, b;
      f25_pair(v, off, e, l0, a, b);
      const cf E = {a.x + b.x, a.y - b.y}, O = {a.y + b.y, b.x - a.x};   // 2 X[k] of the sequence's two frames
      float PX = E.x * E.x + E.y * E.y, PY = O.x * O.x + O.y * O.y;
      if (e == 0) {   // lane 0: Z[0] is its own partner: X[0] = Re Z[0], Y[0] = Im Z[0] (x4 like the others)
        const float xa = 2.f * v[off].x, xb = 2.f * v[off].y;
        PX = l0 ? xa * xa : PX;
        PY = l0 ? xb * xb : PY;
      }
      P[2 * s][e] = PX;
      P[2 * s + 1][e] = PY;
    }
    const float xa = 2.f * v[off + 4].x, xb = 2.f * v[off + 4].y;   // lane 0: Z[128] is its own partner
    P128[2 * s] = xa * xa;
    P128[2 * s + 1] = xb * xb;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// REDO: the second launch of a call with the in-kernel floor test (thresh.hpp: FloorLazy): only the units whose test fired.
template <int WAVES, bool REDO = false>
__global__ __launch_bounds__(WAVES * 64, 3) void k_decide_fast256(Fast25Args A) {
  if (REDO && A.fl.alim[1] != A.tc.need_tag) return;   // no unit of this call reported (the common case)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  float* s_t2 = swin + F25_N;                // [129] float32 compare constants x4 (the split works on 2 X)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
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
  stage_t2_plain<WAVES * 64, F25_F>(s_t2, A.tc.T2, need, 4.0, tid, t2eff);
  constexpr int NF = F25_FPW * WAVES;
  const int64_t tf0 = (int64_t)blockIdx.x * NF;
  cf v[32];
  bool validX, validY;
  const unsigned fl_mx = f25_gather<WAVES, true>(A, tw512, regions, swin, row, chunk, tf0, v, validX, validY);
  if (!REDO) floor_lazy_report(A.tc, A.fl, fl_bound, fl_mx, u, G.FS, lane);
  const int64_t tq = tf0 + F25_FPW * wave;
  if (tq >= G.T) return;   // wave-uniform; no barrier below
  // delta^2 = 2^-32 ||x w||^2 (see k_decide_fast): one sequence carries two frames, the rounding error in either spectrum
  // scales with the norm of the PAIR; the two sequences of a lane group never mix.  Norm over the 8 lanes of equal parity.
  float nXY = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) nXY += v[r].x * v[r].x + v[r].y * v[r].y;
#pragma unroll
  for (int o = 2; o < 16; o <<= 1) nXY += __shfl_xor(nXY, o);
  const float n1 = __shfl(nXY, lane & 48), n2 = __shfl(nXY, (lane & 48) | 1);   // sequence 1 (even lanes), sequence 2
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
  {
    int z0 = 0;
    asm volatile("" : "+v"(z0));
    f25_fwd_half(v, fb, tw512 + z0, c);
  }
  const bool l0 = c == 0;
  float d2[2];
  d2[0] = n1 > 0.f ? 8.0f * 2.3283064e-10f * n1 : -1.0f;
  d2[1] = n2 > 0.f ? 8.0f * 2.3283064e-10f * n2 : -1.0f;
  // frames of the group: t = tq + 4 g + fr; which of them exist
  const int64_t tg = tq + 4 * g;
  float P[4][8], P128[4];
  f25_powers(v, l0, P, P128);
  unsigned pr = 0, am = 0;            // bit 8 fr + e: decision / ambiguity of slot e, frame fr
  unsigned p128 = 0, a128 = 0;        // bit fr: bin 128 (lane 0)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float T = s_t2[bin6(c, e)];
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      const float diff = P[fr][e] - T;
      pr |= (diff > 0.f ? 1u : 0u) << (8 * fr + e);
      am |= ((diff * diff <= d2[fr >> 1] * (P[fr][e] + T)) ? 1u : 0u) << (8 * fr + e);
    }
  }
  {
    const float T = s_t2[128];
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      const float diff = P128[fr] - T;
      p128 |= ((l0 && diff > 0.f) ? 1u : 0u) << fr;
      a128 |= ((l0 && diff * diff <= d2[fr >> 1] * (P128[fr] + T)) ? 1u : 0u) << fr;
    }
  }
  if (need == 2) { pr = 0; am = 0; p128 = 0; a128 = 0; }   // NaN powers have no sign
#pragma unroll
  for (int fr = 0; fr < 4; ++fr) {
    const bool ok = tg + fr >= 0 && tg + fr < G.T;
    if (!ok) { pr &= ~(0xffu << (8 * fr)); am &= ~(0xffu << (8 * fr)); p128 &= ~(1u << fr); a128 &= ~(1u << fr); }
  }
  // exact re-evaluation of ambiguous cells, one at a time, whole wave cooperating
  while (true) {
    const unsigned long long pending = __ballot(am != 0 || a128 != 0);
    if (pending == 0) break;
    const int src = __ffsll((long long)pending) - 1;
    const unsigned sam = (unsigned)__shfl((int)am, src), s128 = (unsigned)__shfl((int)a128, src);
    const int cs = src & 15, gs = src >> 4;
    int fr, f, q;
    if (sam) { q = __ffs((int)sam) - 1; fr = q >> 3; f = bin6(cs, q & 7); }
    else { q = -1; fr = __ffs((int)s128) - 1; f = 128; }
    const int64_t t = tq + 4 * gs + fr;
    const Fast25Args& L = *late_args<Fast25Args>();     // (cold path: arguments re-read here, not kept live from the entry)
    const double Pe = f25_exact_power(L, row, chunk, t, f, lane);
    double t2 = L.tc.T2[f];
    if (floor_live) {
      const double fl = cell_db(L.tc.pmax[u * (int64_t)L.g.FS + f], L.mag_scale) - L.top_db;
      if (fl > L.tc.thresh[f]) t2 = -1.0;
    }
    if (need == 2) t2 = T2_NEVER;
    const bool pass = Pe > t2;
    if (lane == src) {
      if (q >= 0) { pr = (pr & ~(1u << q)) | ((pass ? 1u : 0u) << q); am &= ~(1u << q); }
      else { p128 = (p128 & ~(1u << fr)) | ((pass ? 1u : 0u) << fr); a128 &= ~(1u << fr); }
    }
  }
  // Pack.  Block j (bins 32 j .. 32 j + 31) of a frame: bits 0..15 = the ballot of slot j (lane c -> bin 32 j + c; lane 0's
  // slot j IS bin 32 j), bits 16..31 = the ballot of slot 7 - j of lanes >= 1 (lane c -> bin 32 j + 32 - c: reversed) with
  // lane 0's slot 4 + j (bin 32 j + 16) in position 16.  Lane c = 4 fr + w (w < 3) of a group stores word w of frame fr.
  unsigned long long mine = 0ull;
  const int sh = 16 * g;
#pragma unroll
  for (int fr = 0; fr < 4; ++fr) {
    unsigned blk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned long long b1 = __ballot(((pr >> (8 * fr + j)) & 1u) != 0);
      const unsigned sb = l0 ? (pr >> (8 * fr + 4 + j)) & 1u : (pr >> (8 * fr + 7 - j)) & 1u;
      const unsigned long long b2 = __ballot(sb != 0);
      const unsigned lo = (unsigned)(b1 >> sh) & 0xffffu, up = (unsigned)(b2 >> sh) & 0xffffu;
      const unsigned hi = (((__brev(up & 0xfffeu) >> 16) << 1) & 0xffffu) | (up & 1u);
      blk[j] = lo | (hi << 16);
    }
    const unsigned long long bN = __ballot(((p128 >> fr) & 1u) != 0);
    const unsigned long long w0 = (unsigned long long)blk[0] | ((unsigned long long)blk[1] << 32);
    const unsigned long long w1 = (unsigned long long)blk[2] | ((unsigned long long)blk[3] << 32);
    const unsigned long long w2 = (bN >> sh) & 1ull;
    if ((c >> 2) == fr) mine = (c & 3) == 0 ? w0 : ((c & 3) == 1 ? w1 : w2);
  }
  {
    const int fr = c >> 2, w = c & 3;
    const int64_t t = tg + fr;
    if (w < 3 && t >= 0 && t < G.T) A.bits[(u * G.T + t) * 3 + w] = mine;
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 3) void k_mag_fast256(Fast25Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  constexpr int NF = F25_FPW * WAVES;
  const int64_t tf0 = (int64_t)blockIdx.x * NF;
  cf v[32];
  bool validX, validY;
  f25_gather<WAVES>(A, tw512, regions, swin, row, chunk, tf0, v, validX, validY);
  const int64_t tq = tf0 + F25_FPW * wave;
  const bool with_sub = A.sub != nullptr;
  constexpr int TP = 132;   // floats between the rows of the |X| tile (with_sub): 16 rows per wave in its own exchange slice
  static_assert(F25_FPW * TP * 4 <= WAVE_CX_H * 8, "a wave's |X| rows fit its exchange slice");
  if (tq < G.T) {   // (wave-uniform)
    cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
    f25_fwd_half(v, fb, tw512, c);
    const bool l0 = c == 0;
    const int64_t tg = tq + 4 * g;
    float P[4][8], P128[4];
    f25_powers(v, l0, P, P128);
    float* trow = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + (4 * g) * TP;   // (the wave's transform is done)
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      const int64_t t = tg + fr;
      const bool ok = t >= 0 && t < G.T;
      float* m = A.mag + (u * G.T + (ok ? t : 0)) * (int64_t)G.FS;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mv = half_sqrt(P[fr][e]);
        if (ok) m[bin6(c, e)] = mv;
        if (with_sub) trow[fr * TP + bin6(c, e)] = mv;
      }
      if (l0) {
        const float mv = half_sqrt(P128[fr]);
        if (ok) m[128] = mv;
        if (with_sub) trow[fr * TP + 128] = mv;
      }
    }
  }
  if (!with_sub) return;
  __syncthreads();
  mag_sub_partials<WAVES * 64, NF, F25_FPW, TP, F25_F>(regions, (int)min<int64_t>((int64_t)NF, G.T - tf0), A.iir_b,
                                                      A.sub + ((u * gridDim.x + blockIdx.x) * 2) * (int64_t)G.FS, G.FS, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// Apply: FFT -> x mask -> IFFT -> window -> overlap-add -> samples.  Tiles overlap by 3 frames: a tile of NF frames
// completes NF - 3 hops on its own.
template <int WAVES, bool KMASK>
__global__ __launch_bounds__(WAVES * 64, 3) void k_apply_fast256(Fast25Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15, cp = c >> 1;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  constexpr int NF = F25_FPW * WAVES, NH = NF - 3;
  const bool seam = A.part != nullptr;        // abutting tiles + k_ola_seam (fastpath.hpp), else tiles that overlap by 3 frames
  const int64_t tf0 = A.h_begin - 3 + (int64_t)blockIdx.x * (seam ? NF : NH);   // first frame of the tile
  cf v[32];
  bool validX, validY;
  f25_gather<WAVES>(A, tw512, regions, swin, row, chunk, tf0, v, validX, validY);
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
  const bool l0 = c == 0;
  const int64_t tg = tf0 + F25_FPW * wave + 4 * g;
  const bool wave_live = tf0 + F25_FPW * wave + F25_FPW - 1 >= 0 && tf0 + F25_FPW * wave < G.T;
  if (wave_live) {
    {
      int z0 = 0;
      asm volatile("" : "+v"(z0));
      f25_fwd_half(v, fb, tw512 + z0, c);
    }
    // X = E / 2 (O / 2); Y = X * mask; Z'[k] = Yx + i Yy, Z'[256 - k] = conj Yx + i conj Yy.  The 1/2 of the split and the
    // 1/256 of the inverse transform ride in the mask scale.
    const float ks = (KMASK ? A.inv_ktot : 1.0f) * (0.5f / 256.0f);
    float mk[4][8], m128[4];
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      const int64_t t = tg + fr;
      const bool ok = t >= 0 && t < G.T;
      const int64_t off = (u * G.T + (ok ? t : 0)) * (int64_t)G.FS;
      // (round 6) frames outside [0, T): a zero mask SCALE (fast512.hpp: k_apply_fast512)
      const float kf = ok ? ks : 0.f;
      if constexpr (KMASK) {
        const unsigned short* Kr = A.K + off;
#pragma unroll
        for (int e = 0; e < 8; ++e) mk[fr][e] = (float)Kr[bin6(c, e)] * kf;
        m128[fr] = (float)Kr[128] * (2.f * kf);
      } else {
        const float* Mr = A.Mf + off;
#pragma unroll
        for (int e = 0; e < 8; ++e) mk[fr][e] = Mr[bin6(c, e)] * kf;
        m128[fr] = Mr[128] * (2.f * kf);
      }
    }
    auto sel = [&](cf a0, cf a1) -> cf { return {sel_s(F25_L0, a0.x, a1.x), sel_s(F25_L0, a0.y, a1.y)}; };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int off = 8 * s;
      cf na[8], nb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        cf a, b;
        f25_pair(v, off, e, l0, a, b);
        const cf E = {a.x + b.x, a.y - b.y}, O = {a.y + b.y, b.x - a.x};
        const float mx = mk[2 * s][e], my = mk[2 * s + 1][e];
        const cf Yx = {E.x * mx, E.y * mx}, Yy = {O.x * my, O.y * my};
        na[e] = {Yx.x - Yy.y, Yx.y + Yy.x};
        nb[e] = {Yx.x + Yy.y, Yy.x - Yx.y};
      }
      // lane 0: Z[0] and Z[128] are their own partners: Z' = (Re Z mx, Im Z my) with the masks of bins 0 / 128
      const cf z0 = {v[off].x * (2.f * mk[2 * s][0]), v[off].y * (2.f * mk[2 * s + 1][0])};
      const cf z4 = {v[off + 4].x * m128[2 * s], v[off + 4].y * m128[2 * s + 1]};
      // scatter back (the inverse of f25_pair).  lanes >= 1: pa[i] <- na[i], pb[i] <- nb[7 - i];
      // lane 0: pa[0] <- z0, pa[4] <- z4, pa[1..3] <- na[1..3], pa[5..7] <- nb[3..1], pb[0..3] <- na[4..7], pb[4..7] <- nb[7..4]
      cf npa[8], npb[8];
      npa[0] = sel(z0, na[0]);
#pragma unroll
      for (int i = 1; i < 4; ++i) npa[i] = na[i];
      npa[4] = sel(z4, na[4]);
#pragma unroll
      for (int i = 5; i < 8; ++i) npa[i] = sel(nb[8 - i], na[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) npb[i] = sel(na[i + 4], nb[7 - i]);
#pragma unroll
      for (int i = 4; i < 8; ++i) npb[i] = sel(nb[11 - i], nb[7 - i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) { v[off + i] = npa[i]; v[16 + off + i] = npb[i]; }
    }
    {
      int zi = 0, ci = c;
      asm volatile("" : "+v"(zi), "+v"(ci));
      f25_inv_half(v, fb + zi, tw512 + zi, ci);
    }
  }
  // wave-private overlap-add of the wave's 16 frames into 19 hop accumulators (reusing the exchange slices).  Step j: every
  // frame adds its quarter j: frame f touches hop f + j -- within a step no two frames touch the same hop, and a hop receives
  // its quarters in the fixed order j = 0..3 (LDS operations of a wave execute in order).  Lane (g, c): frames
  // X = 4 g + 2 (c & 1) (real parts) and X + 1 (imaginary parts), samples cp + 8 r.
  float* acc = reinterpret_cast<float*>(regions + wave * WAVE_CX_H);
  static_assert((F25_FPW + 3) * F25_HP * 4 <= WAVE_CX_H * 8, "hop accumulators must fit the wave's slice");
  {
    const int fX = 4 * g + 2 * (c & 1), fY = fX + 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool firstX = j == 0, firstY = (j == 0) || (fY == F25_FPW - 1);
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * j + rr;
        const float ws = swin[cp + 8 * r];
        float* dX = acc + (fX + j) * F25_HP + cp + 8 * rr;
        float* dY = acc + (fY + j) * F25_HP + cp + 8 * rr;
        float yx = v[r].x * ws, yy = v[r].y * ws;
        if (!firstX) yx += *dX;
        *dX = yx;
        if (!firstY) yy += *dY;
        *dY = yy;
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  // cross-wave combine: tile hop jj = wave (jj >> 4)'s local hop jj & 15 plus, for jj & 15 <= 2, the previous wave's local
  // hop (jj & 15) + 16 (fixed order: earlier wave first); 16 threads x float4 per hop
  const float* fr = reinterpret_cast<const float*>(regions);
  const int s4 = (tid & 15) * 4;
  for (int jj = (seam ? 0 : 3) + (tid >> 4); jj < (seam ? NF + 3 : NF); jj += (WAVES * 64) >> 4) {
    const int64_t h = tf0 + jj;
    if (h < A.h_begin || h >= A.h_end) continue;
    const int wv = jj < NF ? jj >> 4 : WAVES - 1, lh = jj < NF ? jj & 15 : 16 + (jj - NF);   // (jj >= NF: the last wave's overflow rows)
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wv >= 1 && lh <= 2) a4 = *reinterpret_cast<const float4*>(&fr[(wv - 1) * WAVE_CX_H * 2 + (lh + 16) * F25_HP + s4]);
    {
      const float4 f4 = *reinterpret_cast<const float4*>(&fr[wv * WAVE_CX_H * 2 + lh * F25_HP + s4]);
      a4.x += f4.x; a4.y += f4.y; a4.z += f4.z; a4.w += f4.w;
    }
    if (seam && (jj < 3 || jj >= NF)) {   // straddling hop: partial sum only; slots 0..2 leading, 3..5 trailing
      const int slot = jj < 3 ? jj : 3 + (jj - NF);
      *reinterpret_cast<float4*>(A.part + ((((size_t)u * A.n_tiles + blockIdx.x) * 6 + slot) * F25_H + s4)) = a4;
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
          const float4 w4 = *reinterpret_cast<const float4*>(&A.wsq[F25_H * q + s4]);
          nrm.x += w4.x; nrm.y += w4.y; nrm.z += w4.z; nrm.w += w4.w;
        }
      }
      a4.x /= (nrm.x > 1e-10f ? nrm.x : 1.f);
      a4.y /= (nrm.y > 1e-10f ? nrm.y : 1.f);
      a4.z /= (nrm.z > 1e-10f ? nrm.z : 1.f);
      a4.w /= (nrm.w > 1e-10f ? nrm.w : 1.f);
    }
    {
      const int64_t pb = h * F25_H - G.padL;
      const int64_t gi0 = chunk * A.om.g_step + (pb - A.om.p0);
      if (A.om.dtype == 0 && pb >= A.om.p0 && pb + F25_H <= A.om.p1 && pb + F25_H <= G.Lout && gi0 >= A.om.g_lo &&
          gi0 + F25_H <= A.om.g_hi) {
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
      const int64_t p = h * F25_H + s4 + e - G.padL;
      if (p < A.om.p0 || p >= A.om.p1) continue;
      const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
      if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
      store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? vals[e] : 0.f);
    }
  }
}

}  // namespace fast
}  // namespace sg
