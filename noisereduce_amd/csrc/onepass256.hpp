// One-pass stationary gate for n_fft = win = 256, hop = 64 (round 6): k_gate_onepass512 (onepass512.hpp) on the transforms of
// fast256.hpp -- four real frames per 512-point register transform, a tile of 64 frames (abutting tiles + k_ola_seam by default,
// 61 complete hops per tile with SG_OPT_FORCE_NOSEAM: onepass512.hpp), 129 bins = 3 bit words per frame.
//
//   k_decide_fast256 + k_smooth_bits2 + k_apply_fast256<K>   ->   k_gate_onepass256   (no bit field, no K field in HBM)
//
// Integer smoothing on the matrix cores: 9 blocks of 16 bit rows (64 + 2 nt <= 144) x 9 blocks of 16 bins; per bin block
// H = bits x band matrix for the nine row blocks, then K = time weights x H for the four 16-frame blocks of the tile, each
// reaching 16 + 2 nt <= 96 rows = three k-blocks of 32 (the third only when nt > 24: at 48 kHz the 50 ms window is 37 frames).
// Everything else -- tickets, tagged granules, bounded polls / NaN-poisoned output, in-kernel floor test + REDO, spectra
// parked during the exact re-evaluation -- as onepass512.hpp / onepass.hpp.
#pragma once
#include "fast256.hpp"

namespace sg {
namespace fast {

constexpr int O25_NF = 64, O25_NH = 61;               // frames / complete hops per tile
constexpr int O25_XW = 3;                             // 64-bit words per bit row (129 bins)
constexpr int O25_TILE_WORDS = O25_NF * O25_XW * 2;   // payload of one tile: 384 tagged 8-byte halves = 3072 B
constexpr int O25_BW = O25_XW + 2;                    // bit row pitch in LDS: one zero word on each side
constexpr int O25_ROWS = 144;                         // bit rows in LDS: nine 16-row blocks (64 + 2 nt <= 144)
constexpr int O25_KP = 144;                           // K row pitch (entries): 9 blocks of 16 bins
constexpr int O25_MAX_NT = 40;                        // 16 + 2 nt <= 96 rows = three k-blocks of the time product
constexpr int O25_MAX_NF = 8;
// constant operands (device table `tab`, 64-bit entries): [0, 64) Bf, [64, 128) At1, [128, 192) At2, [192, 256) At3,
// [256, 512) byte -> eight 0 / 1 bytes

#ifndef O25_OCC
#define O25_OCC 3
#endif
struct OnePass25Args {
  Fast25Args A;                // FIRST (late_args)
  unsigned long long* xbits;   // [units][n_tiles + 2][64][3][2] published mask bits: granules {32 bits, epoch}
  unsigned* ticket;
  unsigned ticket_base;
  unsigned epoch;
  unsigned poll_epoch;         // = epoch; tests (SG_OPT_INJECT_HANDOFF_FAULT bits 3..4): a tag no producer writes, with spin_max = 0
  int spin_max;                // polls per hand-off before the tile gives up (OP_SPIN_MAX)
  unsigned* err;
  int nf, nt, n_tiles;
  int scan_q;
  float prop;                  // prop_decrease (onepass512.hpp)
  const unsigned long long* tab;
};

template <int WAVES, bool REDO = false>
__global__ __launch_bounds__(WAVES * 64, O25_OCC) void k_gate_onepass256(OnePass25Args P) {
  static_assert(WAVES == 4, "tile = 64 frames");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  float* s_t2 = swin + F25_N;                // [129] float32 compare constants x4
  unsigned* s_misc = reinterpret_cast<unsigned*>(s_t2 + F25_T2);   // [0] ticket, [1] lost hand-off
  unsigned long long* s_exp = reinterpret_cast<unsigned long long*>(s_misc + 4);   // [256] byte -> eight 0 / 1 bytes
  const Fast25Args& A = P.A;
  if (REDO && A.fl.alim[1] != A.tc.need_tag) return;   // no unit of this call reported (the common case)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15, cp = c >> 1;
  const Geom& G = A.g;
  if (tid == 0) {
    s_misc[0] = atomicAdd(P.ticket, 1u) - P.ticket_base;
    s_misc[1] = 0u;
  }
  s_exp[tid] = P.tab[256 + tid];
  const unsigned fl_bound = REDO ? 0xffffffffu : floor_lazy_bound(A.fl, lane);
  __syncthreads();
  const int ntt = P.n_tiles + 2;
  const unsigned ticket = s_misc[0];
  const int64_t u = ticket / (unsigned)ntt;
  const int jt = (int)(ticket % (unsigned)ntt) - 1;
  const bool halo_tile = jt < 0 || jt >= P.n_tiles;
  const unsigned gu = (unsigned)(A.view.unit0 + u), nch = (unsigned)A.view.n_chunks;
  const int64_t row = gu / nch;
  const int64_t chunk = A.view.c0 + gu % nch;
  const bool lazy = A.fl.alim != nullptr;
  const int need = (lazy && !REDO) ? 0 : need_of(A.tc, u);
  if (REDO && need == 0) return;   // whole workgroup
  if (!REDO && lazy && ticket == 0u && tid == 0) P.ticket[8] = 0u;
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
  constexpr int NF = O25_NF, NH = O25_NH;
  const bool seam = A.part != nullptr;   // abutting tiles + k_ola_seam (fastpath.hpp); else tiles that overlap by 3 frames
  const int step = seam ? NF : NH;       // frames from one tile to the next
  const int64_t tf0 = A.h_begin - 3 + (int64_t)jt * step;   // first frame of the tile
  cf v[32];
  bool validX, validY;
  unsigned fl_mx = f25_gather<WAVES, true>(A, tw512, regions, swin, row, chunk, tf0, v, validX, validY);
  if (!REDO && lazy) {   // the unit window's samples no tile stages: dealt to the unit's tiles in slices (onepass512.hpp)
    constexpr int SPAN = (NF - 1 + 4) * F25_H;
    const int64_t g0 = chunk * A.view.cs - A.view.pad;
    const int64_t s_lo = max<int64_t>(0, A.view.lo - g0), s_hi = min<int64_t>(A.view.Lp, A.view.hi - g0);
    const int64_t sp0 = (A.h_begin - 3 - step) * F25_H - G.padL;
    const int64_t sp1 = (A.h_begin - 3 + (int64_t)P.n_tiles * step) * F25_H - G.padL + SPAN;
    const int64_t first = min(s_hi, max(s_lo, sp0)), last = max(s_lo, min(s_hi, sp1));
    const int64_t lenA = first - s_lo;
    const int64_t c0 = (int64_t)(jt + 1) * P.scan_q, c1 = min(c0 + P.scan_q, lenA + (s_hi - last));
    for (int64_t i = c0 + tid; i < c1; i += WAVES * 64)
      fl_mx = max(fl_mx, __float_as_uint((float)view_sample(A.view, row, chunk, i < lenA ? s_lo + i : last + (i - lenA))) & 0x7fffffffu);
  }
  if (!REDO) floor_lazy_report(A.tc, A.fl, fl_bound, fl_mx, u, G.FS, lane);
  const int64_t tq = tf0 + F25_FPW * wave;
  // ---- forward transform + decisions (k_decide_fast256) -------------------------------------------------------
  float nXY = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) nXY += v[r].x * v[r].x + v[r].y * v[r].y;
#pragma unroll
  for (int o = 2; o < 16; o <<= 1) nXY += __shfl_xor(nXY, o);
  const float n1 = __shfl(nXY, lane & 48), n2 = __shfl(nXY, (lane & 48) | 1);
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
  {
    int z0 = 0;
    asm volatile("" : "+v"(z0));
    f25_fwd_half(v, fb, tw512 + z0, c);
  }
  const bool l0 = c == 0;
  const int64_t tg = tq + 4 * g;
  unsigned long long mine = 0ull;   // lane c = 4 fr + w (w < 3) of a group: word w of frame tg + fr
  {
    float d2[2];
    d2[0] = n1 > 0.f ? 8.0f * 2.3283064e-10f * n1 : -1.0f;
    d2[1] = n2 > 0.f ? 8.0f * 2.3283064e-10f * n2 : -1.0f;
    unsigned pr = 0, am = 0;
    unsigned p128 = 0, a128 = 0;
    {
      float Pw[4][8], P128[4];
      f25_powers(v, l0, Pw, P128);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float T = s_t2[bin6(c, e)];
#pragma unroll
        for (int fr = 0; fr < 4; ++fr) {
          const float diff = Pw[fr][e] - T;
          pr |= (diff > 0.f ? 1u : 0u) << (8 * fr + e);
          am |= ((diff * diff <= d2[fr >> 1] * (Pw[fr][e] + T)) ? 1u : 0u) << (8 * fr + e);
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
    }
    if (need == 2) { pr = 0; am = 0; p128 = 0; a128 = 0; }
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      const bool ok = tg + fr >= 0 && tg + fr < G.T;
      if (!ok) { pr &= ~(0xffu << (8 * fr)); am &= ~(0xffu << (8 * fr)); p128 &= ~(1u << fr); a128 &= ~(1u << fr); }
    }
    // exact re-evaluation of ambiguous cells; half of the spectra parked in the wave's idle exchange slice meanwhile
    if (__ballot(am != 0 || a128 != 0) != 0ull) {
      float* park = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + lane;
      static_assert(64 * 32 * 4 <= WAVE_CX_H * 8, "parked registers must fit the wave's slice");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        park[(2 * i) * 64] = v[16 + i].x;
        park[(2 * i + 1) * 64] = v[16 + i].y;
      }
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
        const Fast25Args& L = *late_args<Fast25Args>();     // (A is the FIRST member of the kernel's argument)
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
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[16 + i].x = park[(2 * i) * 64];
        v[16 + i].y = park[(2 * i + 1) * 64];
      }
      wave_lds_sync();
    }
    // pack (k_decide_fast256)
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
  }
  // ---- publish this tile's bits; the spectra stay in v[] --------------------------------------------------------
  const int fq = F25_FPW * wave + 4 * g;     // tile row of the group's first frame
  const int my_row = fq + (c >> 2), my_w = c & 3;
  unsigned long long* xb_mine = P.xbits + ((size_t)u * ntt + (jt + 1)) * O25_TILE_WORDS;
  if (my_w < O25_XW) {
    const op_v4u gr = {(unsigned)mine, P.epoch, (unsigned)(mine >> 32), P.epoch};
    op_st16_sc1(&xb_mine[(my_row * O25_XW + my_w) * 2], gr);
  }
  __syncthreads();   // every wave is past its forward exchange: the slices are idle from here
  if (halo_tile) return;

  // ---- integer smoothing on the matrix cores --------------------------------------------------------------------
  const int nt = P.nt;
  char* arena = reinterpret_cast<char*>(regions);
  unsigned long long* brow = reinterpret_cast<unsigned long long*>(arena);                           // [O25_ROWS][O25_BW]
  unsigned short* Ks = reinterpret_cast<unsigned short*>(arena + (size_t)O25_ROWS * O25_BW * 8);     // [64][O25_KP]
  static_assert(O25_ROWS * O25_BW * 8 + O25_NF * O25_KP * 2 <= WAVES * WAVE_CX_H * 8, "bits + K tile must fit the exchange slices");
  if (my_w < O25_XW) brow[(nt + my_row) * O25_BW + 1 + my_w] = mine;
  for (int r = tid; r < O25_ROWS; r += WAVES * 64) {
    brow[r * O25_BW] = 0ull;
    brow[r * O25_BW + O25_BW - 1] = 0ull;
    if (r >= NF + 2 * nt) {
#pragma unroll
      for (int w = 1; w <= O25_XW; ++w) brow[r * O25_BW + w] = 0ull;
    }
  }
  for (int i = tid; i < 2 * nt * O25_XW; i += WAVES * 64) {
    const int side = i >= nt * O25_XW;
    const int rem = i - side * nt * O25_XW;
    const int rr = rem / O25_XW, w = rem - rr * O25_XW;
    // tile j - 1 holds frames tf0 - step ..: frame tf0 - nt + rr is its row step - nt + rr; tile j + 1: frame tf0 + 64 + rr is its row 64 - step + rr
    const unsigned long long* src = side ? xb_mine + O25_TILE_WORDS + ((NF - step + rr) * O25_XW + w) * 2
                                         : xb_mine - O25_TILE_WORDS + ((step - nt + rr) * O25_XW + w) * 2;
    op_v4u gr = op_ld16_sc1(src);
    for (int spin = 0; gr[1] != P.poll_epoch || gr[3] != P.poll_epoch; ++spin) {
      if (spin >= P.spin_max) {
        atomicOr_system(P.err, 1u);
        s_misc[1] = 1u;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      gr = op_ld16_sc1(src);
    }
    brow[(side ? nt + NF + rr : rr) * O25_BW + 1 + w] = (unsigned long long)gr[0] | ((unsigned long long)gr[2] << 32);
  }
  const int q4 = lane >> 4, j16 = lane & 15;
  const long Bf = (long)P.tab[lane], At1 = (long)P.tab[64 + lane], At2 = (long)P.tab[128 + lane], At3 = (long)P.tab[192 + lane];
  const bool three = 2 * nt > 48;      // a third k-block of rows (wave-uniform)
  __syncthreads();
  {
    typedef int o25_v4i __attribute__((ext_vector_type(4)));
    const o25_v4i zero4 = {0, 0, 0, 0};
    const unsigned char* wbb = reinterpret_cast<const unsigned char*>(brow);
    constexpr int WPB = O25_BW * 8;
    for (int b = wave; b < 9; b += WAVES) {
      unsigned hp[10];
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const long a = (long)s_exp[wbb[(16 * m + j16) * WPB + 7 + q4 + 2 * b]];
        const o25_v4i hv = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, Bf, zero4, 0, 0, 0);
        hp[m] = (unsigned)hv[0] | ((unsigned)hv[1] << 8) | ((unsigned)hv[2] << 16) | ((unsigned)hv[3] << 24);
      }
      hp[9] = 0u;
#pragma unroll
      for (int o = 0; o < 4; ++o) {   // output frames 16 o + j at bit row nt + 16 o + j: row blocks o .. o + 5
        const long bt1 = (long)(((unsigned long long)hp[o + 1] << 32) | (unsigned long long)hp[o]);
        const long bt2 = (long)(((unsigned long long)hp[o + 3] << 32) | (unsigned long long)hp[o + 2]);
        o25_v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8(At1, bt1, zero4, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_16x16x32_i8(At2, bt2, d, 0, 0, 0);
        if (three) {
          const long bt3 = (long)(((unsigned long long)hp[o + 5 < 10 ? o + 5 : 9] << 32) | (unsigned long long)hp[o + 4]);
          d = __builtin_amdgcn_mfma_i32_16x16x32_i8(At3, bt3, d, 0, 0, 0);
        }
        unsigned short* kd = Ks + (16 * o + 4 * q4) * O25_KP + 16 * b + j16;
        kd[0] = (unsigned short)d[0];
        kd[O25_KP] = (unsigned short)d[1];
        kd[2 * O25_KP] = (unsigned short)d[2];
        kd[3 * O25_KP] = (unsigned short)d[3];
      }
    }
  }
  __syncthreads();
  float mk[4][8], m128[4];
  {
    const float ks = A.inv_ktot * (0.5f / 256.0f) * P.prop;
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      const unsigned short* Kr = Ks + (fq + fr) * O25_KP;
      const int64_t t = tf0 + fq + fr;
      const float kf = (t >= 0 && t < G.T) ? ks : 0.f;   // frames outside [0, T): mask zero (k_apply_fast256)
#pragma unroll
      for (int e = 0; e < 8; ++e) mk[fr][e] = (float)Kr[bin6(c, e)] * kf;
      m128[fr] = (float)Kr[128] * (2.f * kf);
    }
    if (P.prop != 1.0f) {   // + (1 - p) E / ktot (thresh.hpp: tri_valid)
      const float kq = (1.0f - P.prop) * A.inv_ktot * (0.5f / 256.0f);
#pragma unroll
      for (int fr = 0; fr < 4; ++fr) {
        const int64_t t = tf0 + fq + fr;
        const float tt = (t >= 0 && t < G.T) ? kq * tri_valid(nt, t, G.T) : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) mk[fr][e] = fmaf(tri_valid(P.nf, bin6(c, e), F25_F), tt, mk[fr][e]);
        m128[fr] = fmaf(2.f * tri_valid(P.nf, 128, F25_F), tt, m128[fr]);
      }
    }
  }
  __syncthreads();   // every lane has its mask entries: the slices are free for the inverse transform

  // ---- x mask, merge, inverse transform, window, overlap-add, store (k_apply_fast256<K>) ------------------------
  const bool wave_live = tf0 + F25_FPW * wave + F25_FPW - 1 >= 0 && tf0 + F25_FPW * wave < G.T;
  if (wave_live) {
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
      const cf z0 = {v[off].x * (2.f * mk[2 * s][0]), v[off].y * (2.f * mk[2 * s + 1][0])};
      const cf z4 = {v[off + 4].x * m128[2 * s], v[off + 4].y * m128[2 * s + 1]};
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
        float yx = wave_live ? v[r].x * ws : 0.f, yy = wave_live ? v[r].y * ws : 0.f;
        if (!firstX) yx += *dX;
        *dX = yx;
        if (!firstY) yy += *dY;
        *dY = yy;
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  const float poison = s_misc[1] != 0u ? __uint_as_float(0x7fc00000u) : 0.f;
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
    a4.x += poison; a4.y += poison; a4.z += poison; a4.w += poison;
    if (seam && (jj < 3 || jj >= NF)) {   // straddling hop: partial sum only (poisoned with the tile); slots 0..2 leading, 3..5 trailing
      const int slot = jj < 3 ? jj : 3 + (jj - NF);
      *reinterpret_cast<float4*>(A.part + ((((size_t)u * A.n_tiles + jt) * 6 + slot) * F25_H + s4)) = a4;
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
