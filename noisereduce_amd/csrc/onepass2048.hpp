// One-pass stationary gate for n_fft = win = 2048, hop = 512 (round 6): k_gate_onepass512 (onepass512.hpp) on the transforms of
// fast2048.hpp -- one real frame per 32 lanes, a tile of 8 frames (4 wavefronts x 2), 1025 bins = 17 bit words per frame.
//
//   k_decide_fast2048 + k_smooth_bits2 + k_apply_fast2048<K> + k_ola_seam2048   ->   k_gate_onepass2048 + k_ola_seam2048
//
// Tiles ABUT, as k_apply_fast2048's (overlapping by 3 frames would redo 3 of every 8 transforms): the 3 hops that straddle two
// tiles leave as partial sums and k_ola_seam2048 combines them -- the bits are the only exchange inside the launch.
// Integer smoothing on the matrix cores, with two differences from the 512 / 256 kernels:
//   * the frequency half-width is 10 bins at 48 kHz, 23 at 22.05 kHz (base.py:100: 500 Hz / (sr / 1024)): a 16-bin output block
//     reads 16 + 2 nf <= 64 bins = TWO 32-bin k-blocks (band matrices Bf_lo: bins 16 b - 24 .., Bf_hi: bins 16 b + 8 ..), nf <= 24;
//   * H = bits x band reaches (nf + 1)^2 = 576 > 127 (the int8 range of the second product's operand): it enters the time
//     product as two base-128 digits, K = At x (H & 127) + 128 At x (H >> 7)   (exact: integers).
// 8 + 2 nt <= 32 bit rows (nt <= 8) = two 16-row blocks = ONE k-block of the time product.  Six MFMAs per bin block, 65 blocks.
// The K tile (8 x 1025 uint16) and the bits live in the exchange slices, idle between the transforms; the pair stage reads its
// mask entries straight from there (k_apply_fast2048<K> reads them from HBM), one barrier before the inverse exchange.
// Everything else -- tickets, tagged granules, bounded polls / NaN-poisoned output, in-kernel floor test + REDO, spectra
// parked during the exact re-evaluation -- as onepass512.hpp / onepass.hpp.
#pragma once
#include "fast2048.hpp"

namespace sg {
namespace fast {

constexpr int O20_NF = 8;                             // frames per tile = tile step
constexpr int O20_XW = 17;                            // 64-bit words per bit row (1025 bins)
constexpr int O20_TILE_WORDS = O20_NF * O20_XW * 2;   // payload of one tile: 272 tagged 8-byte halves = 2176 B
constexpr int O20_BW = O20_XW + 2;                    // bit row pitch in LDS: one zero word on each side
constexpr int O20_ROWS = 32;                          // bit rows in LDS: two 16-row blocks (8 + 2 nt <= 24)
constexpr int O20_KP = 1048;                          // K row pitch (entries): 65 blocks of 16 bins + 8 (rows 4 apart: 16 banks apart)
constexpr int O20_MAX_NT = 8;
constexpr int O20_MAX_NF = 24;
// constant operands (device table `tab`, 64-bit entries): [0, 64) Bf_lo, [64, 128) Bf_hi, [128, 192) At, [192, 448) byte ->
// eight 0 / 1 bytes

#ifndef O20_OCC
#define O20_OCC 3
#endif
struct OnePass20Args {
  Fast20Args A;                // FIRST (late_args); part / n_tiles: the seam partials
  unsigned long long* xbits;   // [units][n_tiles + 2][8][17][2] published mask bits: granules {32 bits, epoch}
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
__global__ __launch_bounds__(WAVES * 64, O20_OCC) void k_gate_onepass2048(OnePass20Args P) {
  static_assert(WAVES == 4, "tile = 8 frames");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw = reinterpret_cast<cf*>(smem);                 // [32][32] w_1024^(k1 c)
  cf* regions = tw + 1024;
  float* s_t2 = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);   // [1025] compare constants x4
  unsigned* s_misc = reinterpret_cast<unsigned*>(s_t2 + 1028);           // [0] ticket, [1] lost hand-off
  unsigned long long* s_exp = reinterpret_cast<unsigned long long*>(s_misc + 4);   // [256] byte -> eight 0 / 1 bytes
  const Fast20Args& A = P.A;
  if (REDO && A.fl.alim[1] != A.tc.need_tag) return;   // no unit of this call reported (the common case)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, c = lane & 31;
  const Geom& G = A.g;
  if (tid == 0) {
    s_misc[0] = atomicAdd(P.ticket, 1u) - P.ticket_base;
    s_misc[1] = 0u;
  }
  s_exp[tid] = P.tab[192 + tid];
  const unsigned fl_bound = REDO ? 0xffffffffu : floor_lazy_bound(A.fl, lane);
  __syncthreads();
  const int ntt = P.n_tiles + 2;                   // tiles per unit incl. one decide-only halo tile per side
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
  if (!REDO && lazy && ticket == 0u && tid == 0) P.ticket[8] = 0u;   // (the second launch's counter starts from zero)
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
  constexpr int NF = O20_NF;
  const int64_t tf0 = A.h_begin - 3 + (int64_t)jt * NF;   // first frame of the tile
  cf v[32];
  bool valid;
  unsigned fl_mx = f20_gather<WAVES, true>(A, tw, regions, row, chunk, tf0, v, valid);
  if (!REDO && lazy) {   // the unit window's samples no tile stages: dealt to the unit's tiles in slices (onepass512.hpp)
    constexpr int SPAN = (NF - 1 + 4) * F20_H;
    const int64_t g0 = chunk * A.view.cs - A.view.pad;
    const int64_t s_lo = max<int64_t>(0, A.view.lo - g0), s_hi = min<int64_t>(A.view.Lp, A.view.hi - g0);
    const int64_t sp0 = (A.h_begin - 3 - NF) * F20_H - G.padL;
    const int64_t sp1 = (A.h_begin - 3 + (int64_t)P.n_tiles * NF) * F20_H - G.padL + SPAN;
    const int64_t first = min(s_hi, max(s_lo, sp0)), last = max(s_lo, min(s_hi, sp1));
    const int64_t lenA = first - s_lo;
    const int64_t c0 = (int64_t)(jt + 1) * P.scan_q, c1 = min(c0 + P.scan_q, lenA + (s_hi - last));
    for (int64_t i = c0 + tid; i < c1; i += WAVES * 64)
      fl_mx = max(fl_mx, __float_as_uint((float)view_sample(A.view, row, chunk, i < lenA ? s_lo + i : last + (i - lenA))) & 0x7fffffffu);
  }
  if (!REDO) floor_lazy_report(A.tc, A.fl, fl_bound, fl_mx, u, G.FS, lane);
  const int64_t tq = tf0 + 2 * wave;
  // ---- forward transform + decisions (k_decide_fast2048) ------------------------------------------------------
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
  const cf wl = A.tw2048[c];
  const int src = (lane & 32) | ((32 - c) & 31);
  const bool l0 = c == 0;
  unsigned long long myword = 0ull;   // lane c < 17 of group g: word c of frame tq + g
  {
    const float d2 = nrm2 > 0.f ? 8.0f * 2.3283064e-10f * nrm2 : -1.0f;
    unsigned pred = 0, amb = 0;
    bool predN = false, ambN = false;     // bin 1024 (lane 0)
    {
      auto decide = [&](float Pw, float T, int q) {
        const float diff = Pw - T;
        pred |= (diff > 0.f ? 1u : 0u) << q;
        amb |= ((diff * diff <= d2 * (Pw + T)) ? 1u : 0u) << q;
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
      const float P512 = 4.f * (v[16].x * v[16].x + v[16].y * v[16].y);   // lane 0: Zc[512] is its own partner
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pu = __shfl(Pp[i], src);
        const float p0 = i < 15 ? Pp[i + 1] : P512;
        decide(l0 ? p0 : pu, s_t2[c + 32 * (31 - i)], 31 - i);
      }
    }
    if (need == 2 || !valid) { pred = 0; amb = 0; predN = false; ambN = false; }
    // exact re-evaluation of ambiguous cells, one at a time, whole wave cooperating; half of the spectra parked in the wave's
    // idle exchange slice for the duration (onepass512.hpp)
    if (__ballot(amb != 0 || ambN) != 0ull) {
      float* park = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + lane;
      static_assert(64 * 32 * 4 <= WAVE_CX_H * 8, "parked registers must fit the wave's slice");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        park[(2 * i) * 64] = v[16 + i].x;
        park[(2 * i + 1) * 64] = v[16 + i].y;
      }
      while (true) {
        const unsigned long long pending = __ballot(amb != 0 || ambN);
        if (pending == 0) break;
        const int sl = __ffsll((long long)pending) - 1;
        const unsigned amb_s = (unsigned)__shfl((int)amb, sl);
        const int q = amb_s ? (__ffs((int)amb_s) - 1) : 32;
        const int cs = sl & 31, gs = sl >> 5;
        const int f = q < 32 ? cs + 32 * q : 1024;
        const Fast20Args& L = *late_args<Fast20Args>();     // (cold path: arguments re-read here; A is the FIRST member)
        const double Pe = f20_exact_power(L, row, chunk, tq + gs, f, lane);
        double t2 = L.tc.T2[f];
        if (floor_live) {
          const double fl = cell_db(L.tc.pmax[u * (int64_t)L.g.FS + f], L.mag_scale) - L.top_db;
          if (fl > L.tc.thresh[f]) t2 = -1.0;
        }
        if (need == 2) t2 = T2_NEVER;
        const bool pass = Pe > t2;
        if (lane == sl) {
          if (q < 32) {
            pred = (pred & ~(1u << q)) | ((pass ? 1u : 0u) << q);
            amb &= ~(1u << q);
          } else {
            predN = pass;
            ambN = false;
          }
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
    // pack (k_decide_fast2048): the ballot of register k2 is, per frame, the 32 bins 32 k2 .. 32 k2 + 31 in natural order
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
  }
  // ---- publish this tile's bits; the spectra stay in v[] --------------------------------------------------------
  const int fr = 2 * wave + g;      // tile row of this lane group's frame
  unsigned long long* xb_mine = P.xbits + ((size_t)u * ntt + (jt + 1)) * O20_TILE_WORDS;
  if (c < O20_XW) {
    const op_v4u ga = {(unsigned)myword, P.epoch, (unsigned)(myword >> 32), P.epoch};
    op_st16_sc1(&xb_mine[(fr * O20_XW + c) * 2], ga);
  }
  __syncthreads();   // every wave is past its forward exchange: the slices are idle from here
  if (halo_tile) return;

  // ---- integer smoothing on the matrix cores ---------------------------------------------------------------------
  const int nt = P.nt;
  char* arena = reinterpret_cast<char*>(regions);
  unsigned long long* brow = reinterpret_cast<unsigned long long*>(arena);                           // [O20_ROWS][O20_BW]
  unsigned short* Ks = reinterpret_cast<unsigned short*>(arena + (size_t)O20_ROWS * O20_BW * 8);     // [8][O20_KP]
  static_assert(O20_ROWS * O20_BW * 8 + O20_NF * O20_KP * 2 <= WAVES * WAVE_CX_H * 8, "bits + K tile must fit the exchange slices");
  if (c < O20_XW) brow[(nt + fr) * O20_BW + 1 + c] = myword;
  for (int r = tid; r < O20_ROWS; r += WAVES * 64) {
    brow[r * O20_BW] = 0ull;
    brow[r * O20_BW + O20_BW - 1] = 0ull;
    if (r >= NF + 2 * nt) {       // rows past the neighbours': zero (the k-block reads them)
#pragma unroll
      for (int w = 1; w <= O20_XW; ++w) brow[r * O20_BW + w] = 0ull;
    }
  }
  // neighbour rows: one 16-byte load per 64-bit word (2 nt x 17 words <= 272), polled until both tags are current
  for (int i = tid; i < 2 * nt * O20_XW; i += WAVES * 64) {
    const int side = i >= nt * O20_XW;
    const int rem = i - side * nt * O20_XW;
    const int rr = rem / O20_XW, w = rem - rr * O20_XW;
    // tile j - 1 holds frames tf0 - 8 ..: frame tf0 - nt + rr is its row 8 - nt + rr; tile j + 1: frame tf0 + 8 + rr is its row rr
    const unsigned long long* sp = side ? xb_mine + O20_TILE_WORDS + (rr * O20_XW + w) * 2
                                        : xb_mine - O20_TILE_WORDS + ((NF - nt + rr) * O20_XW + w) * 2;
    op_v4u gr = op_ld16_sc1(sp);
    for (int spin = 0; gr[1] != P.poll_epoch || gr[3] != P.poll_epoch; ++spin) {
      if (spin >= P.spin_max) {   // bounded: report instead of hanging the device
        atomicOr_system(P.err, 1u);
        s_misc[1] = 1u;            // the tile's mask is unknown: every hop it touches becomes NaN
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      gr = op_ld16_sc1(sp);
    }
    brow[(side ? nt + NF + rr : rr) * O20_BW + 1 + w] = (unsigned long long)gr[0] | ((unsigned long long)gr[2] << 32);
  }
  const int q4 = lane >> 4, j16 = lane & 15;
  const long Bl = (long)P.tab[lane], Bh = (long)P.tab[64 + lane], At = (long)P.tab[128 + lane];
  __syncthreads();
  {
    // Per 16-bin block b:  H[row][16 b + j] = sum_k bit[row][16 b - 24 + k] vf[k - 24 - j] + sum_k bit[row][16 b + 8 + k] vf[k + 8 - j]
    // for the two row blocks (result: lane = bin column, 4 consecutive rows per lane group = the B layout of the time product);
    //   K[frame][bin] = sum_slot vt[row(slot) - nt - frame] H[row(slot)][bin],   k-slot 8 q + e = row 4 q + e of block 0 (e < 4) / 1
    // on the two base-128 digits of H.  Output frames 0..7 = lane groups q4 < 2.
    typedef int o20_v4i __attribute__((ext_vector_type(4)));
    const o20_v4i zero4 = {0, 0, 0, 0};
    const unsigned char* wbb = reinterpret_cast<const unsigned char*>(brow);
    constexpr int WPB = O20_BW * 8;
    for (int b = wave; b < 65; b += WAVES) {
      unsigned lo[2], hi[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const unsigned char* rp = wbb + (16 * m + j16) * WPB + q4 + 2 * b;
        const long a0 = (long)s_exp[rp[5]], a1 = (long)s_exp[rp[9]];
        o20_v4i hv = __builtin_amdgcn_mfma_i32_16x16x32_i8(a0, Bl, zero4, 0, 0, 0);
        hv = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, Bh, hv, 0, 0, 0);
        lo[m] = ((unsigned)hv[0] & 127u) | (((unsigned)hv[1] & 127u) << 8) | (((unsigned)hv[2] & 127u) << 16) | (((unsigned)hv[3] & 127u) << 24);
        hi[m] = ((unsigned)hv[0] >> 7) | (((unsigned)hv[1] >> 7) << 8) | (((unsigned)hv[2] >> 7) << 16) | (((unsigned)hv[3] >> 7) << 24);
      }
      const long bt0 = (long)(((unsigned long long)lo[1] << 32) | (unsigned long long)lo[0]);
      const long bt1 = (long)(((unsigned long long)hi[1] << 32) | (unsigned long long)hi[0]);
      const o20_v4i d0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(At, bt0, zero4, 0, 0, 0);
      const o20_v4i d1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(At, bt1, zero4, 0, 0, 0);
      if (q4 < 2) {
        unsigned short* kd = Ks + (4 * q4) * O20_KP + 16 * b + j16;
        kd[0] = (unsigned short)(d0[0] + (d1[0] << 7));
        kd[O20_KP] = (unsigned short)(d0[1] + (d1[1] << 7));
        kd[2 * O20_KP] = (unsigned short)(d0[2] + (d1[2] << 7));
        kd[3 * O20_KP] = (unsigned short)(d0[3] + (d1[3] << 7));
      }
    }
  }
  __syncthreads();

  // ---- x mask, merge (k_apply_fast2048<K>, the mask entries read from the K tile in LDS) --------------------------
  const bool wave_live = tf0 + 2 * wave + 1 >= 0 && tf0 + 2 * wave < G.T;
  if (wave_live) {
    // pair_mask leaves out four 1/2 factors; the inverse transform a factor 1024; K / ktot for the integer sums
    const float ks = A.inv_ktot * (0.25f / 1024.0f) * P.prop;
    const bool propn = P.prop != 1.0f;   // + (1 - p) E / ktot (thresh.hpp: tri_valid)
    const float tq_ = propn ? (1.0f - P.prop) * A.inv_ktot * (0.25f / 1024.0f) * tri_valid(nt, tf0 + fr, G.T) : 0.f;
    const int nfw = P.nf;
    const unsigned short* Kf = Ks + fr * O20_KP;
    auto mval = [&](int k) -> float {
      float m = (float)Kf[k] * ks;
      if (propn) m = fmaf(tri_valid(nfw, k, F20_F), tq_, m);
      return m;
    };
    // (opaque copy: the 16 pair twiddles w_2048^c w_64^i are recomputed here -- kept from the decision stage they are 32 live
    // registers across the smoothing, 14 of which went to scratch)
    cf wl2 = wl;
    asm volatile("" : "+v"(wl2.x), "+v"(wl2.y));
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
      pair_mask(xa, ba, f20_wk(wl2, i), mval(ka), mval(1024 - ka));          // (ka = 0: the second mask is bin 1024's)
      v[i] = xa;
      v[31 - i].x = __shfl(ba.x, src);                          // the partner's merged value for OUR register 31 - i
      v[31 - i].y = __shfl(ba.y, src);
    }
    if (l0) {
#pragma unroll
      for (int j = 31; j >= 17; --j) v[j] = v[j - 1];
      const float m16 = mval(512) * 4.0f;
      v[16] = {a16.x * m16, a16.y * m16};
      const float y0 = (a0.x + a0.y) * mval(0) * 4.0f;
      const float yN = (a0.x - a0.y) * mval(1024) * 4.0f;
      v[0] = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
    }
  }
  __syncthreads();   // every lane has used its mask entries: the slices are free for the inverse transform
  if (wave_live) {
    int zi = 0, ci = c;
    asm volatile("" : "+v"(zi), "+v"(ci));
    fft1k_inv(v, fb + zi, tw + zi, ci);
  }
  __syncthreads();   // every wave is past its exchanges: the regions become the tile's hop buffer
  // ---- window, overlap-add in four ordered rounds, store; straddling hops as partial sums (k_apply_fast2048, seam mode) ----
  float* hop = reinterpret_cast<float*>(regions);
  static_assert((NF + 3) * F20_XP * 4 <= WAVES * WAVE_CX_H * 8, "hop buffer must fit the regions");
  const float2* ws = reinterpret_cast<const float2*>(A.win + 2 * c);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool first = (j == 0) || (fr == NF - 1);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = 8 * j + rr;
      const float2 w2 = ws[32 * r];
      float2* dst = reinterpret_cast<float2*>(hop + (fr + j) * F20_XP + 2 * c + 64 * rr);
      float2 nw = {wave_live ? v[r].x * w2.x : 0.f, wave_live ? v[r].y * w2.y : 0.f};
      if (!first) { const float2 old = *dst; nw.x += old.x; nw.y += old.y; }
      *dst = nw;
    }
    __syncthreads();
  }
  const float poison = s_misc[1] != 0u ? __uint_as_float(0x7fc00000u) : 0.f;
  const int s4 = (tid & 127) * 4;
  for (int jj = (tid >> 7); jj < NF + 3; jj += (WAVES * 64) >> 7) {
    const int64_t h = tf0 + jj;
    if (h < A.h_begin || h >= A.h_end) continue;
    float4 a4 = *reinterpret_cast<const float4*>(&hop[jj * F20_XP + s4]);
    a4.x += poison; a4.y += poison; a4.z += poison; a4.w += poison;
    if (jj < 3 || jj >= NF) {   // straddling hop: partial sum only; slots 0..2 leading, 3..5 trailing
      const int slot = jj < 3 ? jj : 3 + (jj - NF);
      *reinterpret_cast<float4*>(A.part + ((((size_t)u * A.n_tiles + jt) * 6 + slot) * 512 + s4)) = a4;
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

}  // namespace fast
}  // namespace sg
