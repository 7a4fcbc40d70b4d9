// float64 fused apply at the default geometry (n_fft = win = 1024, hop 256): frames -> FFT -> x mask -> IFFT -> window ->
// overlap-add -> output samples, every value in double precision -- k_apply_fast<KMASK> (fastpath.hpp) on the float64
// register transform of fast64.hpp.
//
// Why it exists (round 5): integer recordings must come out as the TRUNCATED float64 result of the reference
// (base.py:217-226 casts a float64 array), and precision="float64" promises float64-accurate samples.  The mask of the
// stationary gate is already exact on the fused path -- the decisions of k_decide_fast are bit-identical to float64
// decisions and the smoothing is integer arithmetic on bits (K = sum vf vt bit, mask = K / ktot) -- so only the two
// transforms, the mask multiply and the overlap-add need float64.  The materialised pipeline of exact.hpp moves 32 bytes
// per time-frequency cell through HBM five times (4.6 ms for ten minutes of 48 kHz audio); this kernel reads the
// recording and the uint16 K field once and writes the output once.
//
// A float32 result cannot be "fixed up" instead: its error (~3e-7 of peak, 0.006 LSB of an int16 recording) puts ~1 % of
// the samples within reach of an integer boundary, i.e. some sample of EVERY 256-sample hop, and a hop's exact value needs
// the float64 transforms of its four frames -- everything would be recomputed.
//
// One workgroup = WAVES wavefronts = 4 WAVES consecutive frames -> 4 WAVES - 3 finished hops (tiles overlap by three
// frames: no hand-off between workgroups).  One wavefront = 4 frames, lane (g, c) holds the 32 packed complex points
// z[c + 16 r] of frame g (128 VGPRs); conjugate bins k / 512 - k live in the same lane after lane 0 has permuted its
// registers once (rg_cyc), so split -> x mask -> merge run in place.
#pragma once
#include "exact.hpp"
#include "fast64.hpp"

namespace sg {
namespace fast {

// dft_inplace_d (fast64.hpp) with a direction: INV conjugates the twiddles (unnormalised inverse)
template <int R, bool INV, int LEN = 2>
__device__ __forceinline__ void dft_inplace_dx(cd* v) {
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
          const cd t = rot90<INV>(b);
          v[base + j] = cadd(a, t);
          v[base + j + H] = csub(a, t);
        } else {
          // p = a + w b, q = 2 a - p;  w = w_LEN^j = c - i s (forward), c + i s (inverse)
          const double c = twcd<32>(j * (32 / LEN)), s = INV ? -twsd<32>(j * (32 / LEN)) : twsd<32>(j * (32 / LEN));
          cd p;
          p.x = fma(b.x, c, fma(b.y, s, a.x));
          p.y = fma(b.y, c, fma(-b.x, s, a.y));
          v[base + j] = p;
          v[base + j + H] = {fma(2.0, a.x, -p.x), fma(2.0, a.y, -p.y)};
        }
      }
    }
    dft_inplace_dx<R, INV, LEN * 2>(v);
  }
}

// Mirror image of fft512_fwd_half_d: v[k2] = Y[row1 + 32 k2], v[16 + k2] = Y[row2 + 32 k2]  ->  v[r] = 512 y[c + 16 r]
// (unnormalised).  Same exchange slots and swizzle as the forward transform (rows written, columns read).
__device__ __forceinline__ void fft512_inv_half_d(cd* v, cd* fb, const cd* tw512, int c) {
  __builtin_amdgcn_sched_barrier(0);
  cd w[32];
  {
    cd a[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) a[brev<16>(k2)] = v[k2];
    dft_inplace_dx<16, true>(a);          // a[h] = element (row1, column h)
    const int row = row1(c);
#pragma unroll
    for (int h = 0; h < 16; ++h) fb[row * 16 + (h ^ row)] = a[h];
  }
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {   // (four at a time: see fft512_fwd_half_d)
    const cd e = fb[k1 * 16 + (c ^ k1)];
    if (k1 == 0) {
      w[brev<32>(k1)] = e;
    } else {
      cd t = tw512[k1 * 16 + c];
      t.y = -t.y;
      w[brev<32>(k1)] = cmul(e, t);
    }
    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
  {
    cd a[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) a[brev<16>(k2)] = v[16 + k2];
    dft_inplace_dx<16, true>(a);
    const int row = row2(c) - 16;
#pragma unroll
    for (int h = 0; h < 16; ++h) fb[row * 16 + (h ^ row)] = a[h];
  }
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k1 = 16; k1 < 32; ++k1) {
    const cd e = fb[(k1 - 16) * 16 + (c ^ (k1 - 16))];
    cd t = tw512[k1 * 16 + c];
    t.y = -t.y;
    w[brev<32>(k1)] = cmul(e, t);
    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
  dft_inplace_dx<32, true>(w);
#pragma unroll
  for (int r = 0; r < 32; ++r) v[r] = w[r];
  __builtin_amdgcn_sched_barrier(0);
}

// split / merge of one conjugate pair in double (split_pair / merge_pair of fastpath.hpp; the four 1/2 factors are folded
// into the mask scale by the caller)
__device__ __forceinline__ void split_pair_d(cd a, cd b, cd w, cd& xa, cd& xb) {
  const cd E = {a.x + b.x, a.y - b.y};
  const cd O = {a.y + b.y, b.x - a.x};
  xa.x = fma(w.x, O.x, fma(-w.y, O.y, E.x));
  xa.y = fma(w.x, O.y, fma(w.y, O.x, E.y));
  xb.x = fma(2.0, E.x, -xa.x);
  xb.y = fma(2.0, E.y, -xa.y);
}
__device__ __forceinline__ void merge_pair_d(cd& xa, cd& xb, cd w, double mk, double mn) {
  const double nx = xb.x * mn, ny = -xb.y * mn;
  const cd Ep = {fma(xa.x, mk, nx), fma(xa.y, mk, -ny)};
  const cd D = {fma(xa.x, mk, -nx), fma(xa.y, mk, ny)};
  const double ax = fma(D.x, w.y, fma(-D.y, w.x, Ep.x));
  const double ay = fma(D.x, w.x, fma(D.y, w.y, Ep.y));
  xa = {ax, ay};
  xb = {fma(2.0, Ep.x, -ax), fma(-2.0, Ep.y, ay)};
}
// rg_lane0_to_entries / rg_lane0_from_entries (fastpath.hpp) on double values
__device__ __forceinline__ void lane0_to_entries_d(cd* v, bool l0) {
  const cd t = v[rg_cyc(0)];
#pragma unroll
  for (int i = 0; i < 23; ++i) {
    const cd s = v[rg_cyc(i + 1)], d = v[rg_cyc(i)];
    v[rg_cyc(i)] = {l0 ? s.x : d.x, l0 ? s.y : d.y};
  }
  const cd d = v[rg_cyc(23)];
  v[rg_cyc(23)] = {l0 ? t.x : d.x, l0 ? t.y : d.y};
}
__device__ __forceinline__ void lane0_from_entries_d(cd* v, bool l0) {
  const cd t = v[rg_cyc(23)];
#pragma unroll
  for (int i = 23; i >= 1; --i) {
    const cd s = v[rg_cyc(i - 1)], d = v[rg_cyc(i)];
    v[rg_cyc(i)] = {l0 ? s.x : d.x, l0 ? s.y : d.y};
  }
  const cd d = v[rg_cyc(0)];
  v[rg_cyc(0)] = {l0 ? t.x : d.x, l0 ? t.y : d.y};
}

struct Apply64Args {
  View view;
  Geom g;
  OutMap om;
  const unsigned short* K;   // KMASK: permuted mask counts [units][T][FSK] (k_smooth_bits2's layout for k_apply_fast<KMASK>)
  const double* Mf;          // !KMASK: float64 mask field [units][T][FS], natural bin order (exact.hpp's xM: the non-stationary gate)
  const double* win;         // analysis == synthesis window, double[1024]
  const double* norm;        // sum_q win^2[256 q + s], s < 256 (interior hops)
  const cd* tw1024;          // w_1024^k, k = 0..511
  double kscale;             // KMASK: 1 / (ktot * 512); !KMASK: 1 / 512
  // KMASK, prop_decrease < 1 (stationary.py:108-114 applies it BEFORE the zero-padded smoothing): mask = (p K + (1 - p) edge) / ktot,
  // edge = the smoothing filter's integer weight inside the spectrogram, ef(f) * et(t) -- (nf+1)^2 (nt+1)^2 = ktot except within
  // nf bins / nt frames of its borders.  prop = 1: mask = K / ktot.
  double prop;
  int nf, nt;
  int64_t h_begin, h_end;    // ext hops (256-sample blocks, ext = unit sample + 512) to produce
};

constexpr int A64_XP = 264;   // doubles between the staged hop rows / the hop accumulators of a wave

template <int WAVES, bool KMASK = true>
__global__ __launch_bounds__(WAVES * 64, 2) void k_apply_fast64(Apply64Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* tw512 = reinterpret_cast<cd*>(smem);   // [32][16]: w_512^(k1 c)
  cd* regions = tw512 + FN;
  cd* tw_lo = regions + WAVES * 4 * FSLOTS_D;   // w_1024^0..16
  constexpr int NF = 4 * WAVES, NH = NF - 3;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  {
    constexpr int K = (FN + WAVES * 64 - 1) / (WAVES * 64);
    cd t[K];
    const cd tl = A.tw1024[min(tid, 16)];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = min(tid + k * WAVES * 64, FN - 1);
      const int idx = 2 * (i >> 4) * (i & 15);   // w_512^j = w_1024^(2 j)
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
  const int64_t tf_tile = A.h_begin - 3 + (int64_t)blockIdx.x * NH;   // first frame of the tile
  const int64_t t = tf_tile + 4 * wave + g;
  const bool fvalid = t >= 0 && t < G.T;
  // the tile's sample span, as doubles (any sample type: view_sample; zero outside the readable range)
  constexpr int SPAN = (NF - 1) * 256 + 1024;
  static_assert((SPAN / 256) * A64_XP * 8 <= WAVES * 4 * FSLOTS_D * 16, "span fits the exchange slices");
  double* xs = reinterpret_cast<double*>(regions);
  {
    const int64_t s0b = tf_tile * 256 - G.padL;
    const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
    if (s0b >= 0 && s0b + SPAN <= A.view.Lp && gb >= A.view.lo && gb + SPAN <= A.view.hi) {
      // interior tile: every sample of the span is readable -- one unchecked load per sample, the type switch outside the loop
      const int64_t b0 = row * A.view.stride + gb;
      auto fill = [&](auto* p) {
#pragma unroll 4
        for (int i = tid; i < SPAN; i += WAVES * 64) xs[(i >> 8) * A64_XP + (i & 255)] = (double)p[b0 + i];
      };
      switch (A.view.dtype) {
        case 0: fill((const float*)A.view.x); break;
        case 1: fill((const double*)A.view.x); break;
        case 2: fill((const int16_t*)A.view.x); break;
        default: fill((const int32_t*)A.view.x); break;
      }
    } else {
      for (int i = tid; i < SPAN; i += WAVES * 64) xs[(i >> 8) * A64_XP + (i & 255)] = view_sample(A.view, row, chunk, s0b + i);
    }
  }
  __syncthreads();
  cd v[32];
  {
    const double* xl = xs + (4 * wave + g) * A64_XP + 2 * c;
    const double2* wsrc = reinterpret_cast<const double2*>(A.win + 2 * c);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      double2 x2 = *reinterpret_cast<const double2*>(xl + (r >> 3) * A64_XP + 32 * (r & 7));
      if (!fvalid) x2 = make_double2(0.0, 0.0);   // frames before / past the row do not exist in the reference
      const double2 w2 = wsrc[16 * r];
      v[brev<32>(r)] = {x2.x * w2.x, x2.y * w2.y};
    }
  }
  __syncthreads();   // every lane has its samples: the span becomes the exchange slices
  cd* fb = regions + (wave * 4 + g) * FSLOTS_D;
  const bool wave_live = tf_tile + 4 * wave + 3 >= 0 && tf_tile + 4 * wave < G.T;
  if (wave_live) {
    fft512_fwd_half_d(v, fb, tw512, c);
    // KMASK: mask counts of this lane's 32 entries (+ bin 512), 16 registers; !KMASK: the float64 mask row, read where used
    unsigned kw[KMASK ? 16 : 1];
    const double* Mrow = nullptr;
    double k512;
    if constexpr (KMASK) {
      const unsigned short* Krow = A.K + ((u * G.T + (fvalid ? t : 0)) * (int64_t)FSK);
      const uint4* p4 = reinterpret_cast<const uint4*>(Krow + c * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 w4 = p4[q];
        kw[4 * q] = w4.x; kw[4 * q + 1] = w4.y; kw[4 * q + 2] = w4.z; kw[4 * q + 3] = w4.w;
      }
      k512 = (double)Krow[512] * A.kscale;
    } else {
      Mrow = A.Mf + (u * G.T + (fvalid ? t : 0)) * (int64_t)G.FS;
      k512 = Mrow[512] * A.kscale;
    }
    // integer weight of the valid taps along one axis at position i of n (closed form of the triangle's tails)
    auto edge1 = [](int m, int64_t i, int64_t n) -> double {
      const int64_t l = i < m ? m - i : 0, r = (n - 1 - i) < m ? m - (n - 1 - i) : 0;
      return (double)((int64_t)(m + 1) * (m + 1) - l * (l + 1) / 2 - r * (r + 1) / 2);
    };
    const bool with_prop = KMASK && A.prop != 1.0;
    const double q_et = with_prop ? (1.0 - A.prop) * edge1(A.nt, fvalid ? t : 0, G.T) : 0.0;
    if constexpr (KMASK) {
      if (with_prop) k512 = (A.prop * (double)(A.K + ((u * G.T + (fvalid ? t : 0)) * (int64_t)FSK))[512] + q_et * edge1(A.nf, 512, G.F)) * A.kscale;
    }
    auto mval = [&](int e, double scale) -> double {
      if constexpr (KMASK) {
        const unsigned wv = kw[e >> 1];
        const double kv = (double)((e & 1) ? (wv >> 16) : (wv & 0xffffu));
        if (with_prop) return (A.prop * kv + q_et * edge1(A.nf, bin_of_entry(c, e), G.F)) * scale;
        return kv * scale;
      } else {
        return Mrow[bin_of_entry(c, e)] * scale;
      }
    };
    const bool l0 = c == 0;
    const cd wlo = tw_lo[c];
    cd whi = wlo;
    {
      const cd w16 = tw_lo[16];
      if (l0) whi = {-w16.y, w16.x};   // i * w_1024^16
    }
    lane0_to_entries_d(v, l0);
    const double ks = A.kscale * 0.25;   // split + merge leave out four 1/2 factors
    {
      // slot 0: lanes >= 1 the pair (v[0], v[31]); lane 0: bins 0 / 512 from v[0], bin 256 = v[31] scaled
      const cd r0 = v[0], r31 = v[31];
      cd xa, xb;
      split_pair_d(r0, r31, wlo, xa, xb);
      merge_pair_d(xa, xb, wlo, mval(0, ks), mval(31, ks));
      const double y0 = (r0.x + r0.y) * mval(0, A.kscale);
      const double yN = (r0.x - r0.y) * k512;
      const cd z0 = {0.5 * (y0 + yN), 0.5 * (y0 - yN)};
      const double m8 = mval(31, A.kscale);
      const cd z8 = {r31.x * m8, r31.y * m8};
      v[0] = {l0 ? z0.x : xa.x, l0 ? z0.y : xa.y};
      v[31] = {l0 ? z8.x : xb.x, l0 ? z8.y : xb.y};
    }
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      const cd ws = sl < 8 ? wlo : whi;
      const double cc = twcd<32>(sl), ss = twsd<32>(sl);
      const cd w = {ws.x * cc + ws.y * ss, ws.y * cc - ws.x * ss};   // ws * w_32^sl
      cd xa, xb;
      split_pair_d(v[sl], v[31 - sl], w, xa, xb);
      merge_pair_d(xa, xb, w, mval(sl, ks), mval(31 - sl, ks));
      v[sl] = xa;
      v[31 - sl] = xb;
      __builtin_amdgcn_sched_barrier(0);
    }
    lane0_from_entries_d(v, l0);
    {
      // fresh (opaque) lane arithmetic for the inverse exchange: shared with the forward transform (CSE), the 32 swizzled
      // exchange addresses stay live across the pair stage -- and were spilled (68 VGPRs, 1 GB of scratch traffic per ten
      // minutes of audio: profiles/r05_v1_traffic_detail.json)
      int zi = 0, ci = c;
      asm volatile("" : "+v"(zi), "+v"(ci));
      fft512_inv_half_d(v, fb + zi, tw512 + zi, ci);
    }
  }
  // synthesis window, wave-private overlap-add of this wave's 4 frames into 7 hop accumulators (k_apply_fast<LEAN>):
  // step j: frame g adds its quarter j to hop g + j -- the four lane groups never collide within a step and a hop receives
  // its quarters in the fixed order j = 0..3
  double* acc = reinterpret_cast<double*>(regions + wave * 4 * FSLOTS_D);
  static_assert(7 * A64_XP * 8 <= 4 * FSLOTS_D * 16, "hop accumulators fit the wave's exchange slices");
  {
    const double2* wsrc = reinterpret_cast<const double2*>(A.win + 2 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * j + rr;
        double2* dst = reinterpret_cast<double2*>(acc + (g + j) * A64_XP + 2 * c + 32 * rr);
        const double2 w2 = wsrc[16 * r];
        double2 nw = make_double2(v[r].x * w2.x, v[r].y * w2.y);
        if (!wave_live) nw = make_double2(0.0, 0.0);
        if (j != 0) {   // (frame g == 3 is the first to touch hops 4..6: plain store; sel_s: fastpath.hpp)
          const double2 old = *dst;
          nw.x = sel_s(OLA_KEEP, nw.x + old.x, nw.x);
          nw.y = sel_s(OLA_KEEP, nw.y + old.y, nw.y);
        }
        *dst = nw;
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  // tile hop jj (ext hop tf_tile + jj), jj = 3 .. NF - 1: the accumulators of wave jj / 4 (local hop jj % 4) and, for
  // jj % 4 <= 2, of the wave before it (local hop jj % 4 + 4); fixed order: earlier wave first
  const double* fr = reinterpret_cast<const double*>(regions);
  constexpr int WSTRIDE = 4 * FSLOTS_D * 2;   // doubles between the waves' regions
  const int s4 = lane * 4;
  for (int jj = 3 + wave; jj < NF; jj += WAVES) {
    const int64_t h = tf_tile + jj;
    if (h >= A.h_end || h < A.h_begin) continue;
    const int wh = jj >> 2, lh = jj & 3;
    double a4[4] = {0.0, 0.0, 0.0, 0.0};
    if (wh >= 1 && lh <= 2) {
      const double2 p = *reinterpret_cast<const double2*>(&fr[(wh - 1) * WSTRIDE + (lh + 4) * A64_XP + s4]);
      const double2 q = *reinterpret_cast<const double2*>(&fr[(wh - 1) * WSTRIDE + (lh + 4) * A64_XP + s4 + 2]);
      a4[0] = p.x; a4[1] = p.y; a4[2] = q.x; a4[3] = q.y;
    }
    if (wh < WAVES) {
      const double2 p = *reinterpret_cast<const double2*>(&fr[wh * WSTRIDE + lh * A64_XP + s4]);
      const double2 q = *reinterpret_cast<const double2*>(&fr[wh * WSTRIDE + lh * A64_XP + s4 + 2]);
      a4[0] += p.x; a4[1] += p.y; a4[2] += q.x; a4[3] += q.y;
    }
    // window envelope (scipy/_spectral_py.py:1708-1725): sum of w^2 over the frames that cover the sample
    bool all_valid = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t ti = h - q;
      if (ti < 0 || ti >= G.T) all_valid = false;
    }
    double n4[4];
    if (all_valid) {
      const double2 p = *reinterpret_cast<const double2*>(&A.norm[s4]);
      const double2 q = *reinterpret_cast<const double2*>(&A.norm[s4 + 2]);
      n4[0] = p.x; n4[1] = p.y; n4[2] = q.x; n4[3] = q.y;
    } else {
      n4[0] = n4[1] = n4[2] = n4[3] = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ti = h - q;
        if (ti >= 0 && ti < G.T) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double wv = A.win[256 * q + s4 + e];
            n4[e] += wv * wv;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a4[e] = a4[e] / (n4[e] > 1e-10 ? n4[e] : 1.0);
    const int64_t pb = h * 256 - G.padL;                       // unit-local position of the hop's first sample
    const int64_t gi0 = chunk * A.om.g_step + (pb - A.om.p0);
    if (pb >= A.om.p0 && pb + 256 <= A.om.p1 && pb + 256 <= G.Lout && gi0 >= A.om.g_lo && gi0 + 256 <= A.om.g_hi) {
      // whole hop inside the kept range: one vector store per lane when the destination is aligned
      const int64_t di = row * A.om.stride + gi0 - A.om.g0 + s4;
      if (A.om.dtype == 2) {
        int16_t* dst = (int16_t*)A.om.out + di;
        if ((reinterpret_cast<uintptr_t>(dst) & 7) == 0) {
          const unsigned lo = (unsigned)(unsigned short)(int16_t)a4[0] | ((unsigned)(unsigned short)(int16_t)a4[1] << 16);
          const unsigned hi = (unsigned)(unsigned short)(int16_t)a4[2] | ((unsigned)(unsigned short)(int16_t)a4[3] << 16);
          *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
          continue;
        }
      } else if (A.om.dtype == 1) {
        double* dst = (double*)A.om.out + di;
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          *reinterpret_cast<double2*>(dst) = make_double2(a4[0], a4[1]);
          *reinterpret_cast<double2*>(dst + 2) = make_double2(a4[2], a4[3]);
          continue;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t p = pb + s4 + e;
      if (p < A.om.p0 || p >= A.om.p1) continue;
      const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
      if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
      exact::store_sample_f64(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? a4[e] : 0.0);
    }
  }
}

}  // namespace fast
}  // namespace sg
