// One-pass stationary gate for n_fft = win = 512, hop = 128 (round 6; VERDICT r3 / r4 / r5 "one-pass gates for n_fft != 1024").
//
//   k_decide_fast512 + k_smooth_bits2 + k_apply_fast512<K>      (3 transforms per frame pair, bit field and K field in HBM)
//     ->  k_gate_onepass512                                      (2 transforms per frame pair, neither field)
//
// The structure of k_gate_onepass (onepass.hpp) on the transforms of fast512.hpp.  A workgroup takes a TICKET and owns one tile
// of 32 consecutive frames of one unit (4 wavefronts x 4 lane groups x one frame PAIR per 512-point register transform).  After
// the forward transform it decides its 32 x 257 cells (float32 + exact float64 refinement: the bits of k_decide_fast512), keeps
// the spectra in registers and
//   1. publishes the tile's bits (32 rows x 5 words) to its neighbours as data-tagged granules {32 bits, launch epoch};
//   2. polls the nt adjacent rows of tiles j - 1 and j + 1 (nt <= 24: the time half-width of the smoothing filter);
//   3. smooths the (32 + 2 nt) x 257 bit tile with the exact integer separable triangle filter ON THE MATRIX CORES
//      (v_mfma_i32_16x16x32_i8, the formulation of onepass.hpp): per 16-bin block, H = bits x band matrix for the five 16-row
//      blocks of the tile (nf <= 8), then K = time weights x H for the two 16-frame halves (two 32-row k-blocks each:
//      16 + 2 nt <= 64 rows); H never leaves the registers, bits and the K tile (uint16) live in the exchange slices, which are
//      idle between the two transforms: 3 workgroups per CU as k_apply_fast512.  (A first build ran both directions as
//      sliding-window recurrences in LDS -- popcounts along f, two running boxcars along t: 203 us per two minutes where the three
//      kernels it replaces take 103.)
// then x mask -> inverse transform -> window -> overlap-add -> store exactly as k_apply_fast512<K>.  Tiles ABUT by default (A.part
// set: the 3 hops that straddle two tiles leave as partial sums, k_ola_seam -- fastpath.hpp -- combines them after the launch: 1490
// workgroups per two minutes instead of 1640, no redundant transforms) or, SG_OPT_FORCE_NOSEAM, overlap by 3 frames (29 complete
// hops per tile); either way the bits are the only exchange INSIDE the launch.  Inter-workgroup protocol, deadlock freedom (tickets; publish before wait), bounded polls and NaN-poisoned
// output of a tile that lost a hand-off: onepass.hpp.  The -top_db floor test runs on the staged samples (thresh.hpp:
// FloorLazy); REDO = the second launch for the units whose test fired.  Any prop_decrease (a scale and an offset on the
// mask entries).
#pragma once
#include "fast512.hpp"

namespace sg {
namespace fast {

constexpr int O5_NF = 32, O5_NH = 29;              // frames / complete hops per tile
constexpr int O5_XW = 5;                           // 64-bit words per bit row (257 bins)
constexpr int O5_TILE_WORDS = O5_NF * O5_XW * 2;   // payload of one tile: 320 tagged 8-byte halves = 2560 B
constexpr int O5_BW = O5_XW + 2;                   // bit row pitch in LDS: one zero word on each side
constexpr int O5_ROWS = 80;                        // bit rows in LDS: five 16-row blocks (32 + 2 nt <= 80)
constexpr int O5_KP = 272;                         // K row pitch (entries): 17 blocks of 16 bins
constexpr int O5_MAX_NT = 24;                      // 16 + 2 nt <= 64 rows = two k-blocks of the time product
constexpr int O5_MAX_NF = 8;                       // the 32 x 16 band matrix of the frequency product
// constant operands (device table `tab`, 64-bit entries): [0, 64) Bf (band matrix, per lane), [64, 128) At1, [128, 192) At2 (time
// weights of the two k-blocks), [192, 448) byte -> eight 0 / 1 bytes

#ifndef O5_OCC
#define O5_OCC 3   // workgroups per CU the register budget is set for
#endif
struct OnePass5Args {
  Fast5Args A;                 // view, geometry, tables, compare constants, output map, hop range, floor test (FIRST: late_args)
  unsigned long long* xbits;   // [units][n_tiles + 2][32][5][2] published mask bits: granules {32 bits, epoch}
  unsigned* ticket;            // work counter: never reset, a launch takes exactly units * (n_tiles + 2) tickets
  unsigned ticket_base;
  unsigned epoch;
  unsigned poll_epoch;         // = epoch; tests (SG_OPT_INJECT_HANDOFF_FAULT bits 3..4): a tag no producer writes, with spin_max = 0
  int spin_max;                // polls per hand-off before the tile gives up (OP_SPIN_MAX)
  unsigned* err;               // host-mapped word: bit 0 = a bit hand-off timed out
  int nf, nt, n_tiles;
  int scan_q;                  // in-kernel floor test: samples of the unit window's unstaged part that each tile scans
  float prop;                  // prop_decrease: mask = prop K / ktot + (1 - prop)   (stationary.py:116-119: after the smoothing)
  const unsigned long long* tab;   // MFMA operands + byte expansion (see above)
};

template <int WAVES, bool REDO = false>
__global__ __launch_bounds__(WAVES * 64, O5_OCC) void k_gate_onepass512(OnePass5Args P) {
  static_assert(WAVES == 4, "tile = 32 frames");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  float* s_t2 = swin + F5_N;                 // [257] float32 compare constants x4 (the split works on 2 X)
  unsigned* s_misc = reinterpret_cast<unsigned*>(s_t2 + 264);   // [0] ticket, [1] lost hand-off
  unsigned long long* s_exp = reinterpret_cast<unsigned long long*>(s_misc + 4);   // [256] byte -> eight 0 / 1 bytes
  const Fast5Args& A = P.A;
  if (REDO && A.fl.alim[1] != A.tc.need_tag) return;   // no unit of this call reported (the common case)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
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
  stage_t2_plain<WAVES * 64, F5_F>(s_t2, A.tc.T2, need, 4.0, tid, t2eff);
  constexpr int NF = O5_NF, NH = O5_NH;
  const bool seam = A.part != nullptr;   // abutting tiles + k_ola_seam (fastpath.hpp); else tiles that overlap by 3 frames
  const int step = seam ? NF : NH;       // frames from one tile to the next
  const int64_t tf0 = A.h_begin - 3 + (int64_t)jt * step;   // first frame of the tile
  cf v[32];
  bool validA, validB;
  unsigned fl_mx = f5_gather<WAVES, true>(A, tw512, regions, swin, row, chunk, tf0, G.T, v, validA, validB);
  if (!REDO && lazy) {
    // The tiles of a unit stage the frames its kept hops need (+- nt), not the whole window: the rest -- the chunk's padding
    // beyond them, A = [s_lo, first span) and B = [last span's end, s_hi) -- is dealt to the unit's tiles in slices of scan_q
    // samples of A ++ B (onepass.hpp "floor test"; a few hundred samples per tile at the default chunking)
    constexpr int SPAN = (NF - 1 + 4) * F5_H;
    const int64_t g0 = chunk * A.view.cs - A.view.pad;
    const int64_t s_lo = max<int64_t>(0, A.view.lo - g0), s_hi = min<int64_t>(A.view.Lp, A.view.hi - g0);
    const int64_t sp0 = (A.h_begin - 3 - step) * F5_H - G.padL;
    const int64_t sp1 = (A.h_begin - 3 + (int64_t)P.n_tiles * step) * F5_H - G.padL + SPAN;
    const int64_t first = min(s_hi, max(s_lo, sp0)), last = max(s_lo, min(s_hi, sp1));
    const int64_t lenA = first - s_lo;
    const int64_t c0 = (int64_t)(jt + 1) * P.scan_q, c1 = min(c0 + P.scan_q, lenA + (s_hi - last));
    for (int64_t i = c0 + tid; i < c1; i += WAVES * 64)
      fl_mx = max(fl_mx, __float_as_uint((float)view_sample(A.view, row, chunk, i < lenA ? s_lo + i : last + (i - lenA))) & 0x7fffffffu);
  }
  if (!REDO) floor_lazy_report(A.tc, A.fl, fl_bound, fl_mx, u, G.FS, lane);
  const int64_t tq = tf0 + F5_FPW * wave;
  // ---- forward transform + decisions (k_decide_fast512) -------------------------------------------------------
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
  unsigned long long wA = 0ull, wB = 0ull;   // lane c < 5 of group g: word c of frames tq + 2 g, tq + 2 g + 1
  {
    const float nAB = nA + nB;
    const float dA = nAB > 0.f ? 8.0f * 2.3283064e-10f * nAB : -1.0f;
    const float dB = dA;
    unsigned pA = 0, pB = 0, aA = 0, aB = 0;
    bool p256A = false, p256B = false, a256A = false, a256B = false;
    auto decide = [&](float Pw, float T, float d2, unsigned& pr, unsigned& am, int q) {
      const float diff = Pw - T;
      pr |= (diff > 0.f ? 1u : 0u) << q;
      am |= ((diff * diff <= d2 * (Pw + T)) ? 1u : 0u) << q;
    };
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) {
      cf a, b;
      f5_pair(v, sl, l0, a, b);
      const cf E = {a.x + b.x, a.y - b.y}, O = {a.y + b.y, b.x - a.x};   // 2 A[k], 2 B[k]
      float PA = E.x * E.x + E.y * E.y, PB = O.x * O.x + O.y * O.y;
      if (sl == 0) {
        const float xa = 2.f * v[0].x, xb = 2.f * v[0].y;
        PA = l0 ? xa * xa : PA;
        PB = l0 ? xb * xb : PB;
      }
      const float T = s_t2[bin5(c, sl)];
      decide(PA, T, dA, pA, aA, sl);
      decide(PB, T, dB, pB, aB, sl);
    }
    {
      const float xa = 2.f * v[8].x, xb = 2.f * v[8].y, T = s_t2[256];
      const float da = xa * xa - T, db = xb * xb - T;
      p256A = l0 && da > 0.f;
      p256B = l0 && db > 0.f;
      a256A = l0 && da * da <= dA * (xa * xa + T);
      a256B = l0 && db * db <= dB * (xb * xb + T);
    }
    if (need == 2) { pA = pB = 0; aA = aB = 0; p256A = p256B = a256A = a256B = false; }
    if (!validA) { pA = 0; aA = 0; p256A = a256A = false; }
    if (!validB) { pB = 0; aB = 0; p256B = a256B = false; }
    // exact re-evaluation of ambiguous cells, one at a time, whole wave cooperating.  Rare (about one wave in fifty has an
    // ambiguous cell), but its float64 temporaries do not fit next to the 64 registers of the spectra at three waves per SIMD
    // (the first build kept 29 of them in scratch through the hot path): the wave parks half of the spectra in its idle
    // exchange slice for the duration (onepass.hpp does the same).
    if (__ballot(aA != 0 || aB != 0 || a256A || a256B) != 0ull) {
      float* park = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + lane;
      static_assert(64 * 32 * 4 <= WAVE_CX_H * 8, "parked registers must fit the wave's slice");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        park[(2 * i) * 64] = v[16 + i].x;
        park[(2 * i + 1) * 64] = v[16 + i].y;
      }
      while (true) {   // exact re-evaluation of ambiguous cells, one at a time, whole wave cooperating
        const unsigned long long pending = __ballot(aA != 0 || aB != 0 || a256A || a256B);
        if (pending == 0) break;
        const int src = __ffsll((long long)pending) - 1;
        const unsigned sA = (unsigned)__shfl((int)aA, src), sB = (unsigned)__shfl((int)aB, src);
        const int s256A = __shfl((int)a256A, src);
        const int cs = src & 15, gs = src >> 4;
        int which, f;
        int q = 0;
        if (sA) { which = 0; q = __ffs((int)sA) - 1; f = bin5(cs, q); }
        else if (sB) { which = 1; q = __ffs((int)sB) - 1; f = bin5(cs, q); }
        else if (s256A) { which = 2; f = 256; }
        else { which = 3; f = 256; }
        const int64_t t = tq + 2 * gs + (which & 1);
        const Fast5Args& L = *late_args<Fast5Args>();       // (cold path: arguments re-read here; A is the FIRST member)
        const double Pe = f5_exact_power(L, row, chunk, t, f, lane);
        double t2 = L.tc.T2[f];
        if (floor_live) {
          const double fl = cell_db(L.tc.pmax[u * (int64_t)L.g.FS + f], L.mag_scale) - L.top_db;
          if (fl > L.tc.thresh[f]) t2 = -1.0;
        }
        if (need == 2) t2 = T2_NEVER;
        const bool pass = Pe > t2;
        if (lane == src) {
          if (which == 0) { pA = (pA & ~(1u << q)) | ((pass ? 1u : 0u) << q); aA &= ~(1u << q); }
          else if (which == 1) { pB = (pB & ~(1u << q)) | ((pass ? 1u : 0u) << q); aB &= ~(1u << q); }
          else if (which == 2) { p256A = pass; a256A = false; }
          else { p256B = pass; a256B = false; }
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
    // pack (k_decide_fast512): 16 x 16 bit transpose across the lane group, then lane c < 4 assembles word c, lane 4 bin 256
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
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int j = 2 * w + h2;
      const unsigned lo = (unsigned)__shfl((int)tr, gl | j);
      const unsigned up = (unsigned)__shfl((int)tr, gl | (15 - j));
      const unsigned z0 = (unsigned)__shfl((int)tr, gl | (8 + j));
      const unsigned upA = up & 0xfffeu, upB = (up >> 16) & 0xfffeu;
      const unsigned hiA = (((__brev(upA) >> 16) << 1) & 0xffffu) | (z0 & 1u);
      const unsigned hiB = (((__brev(upB) >> 16) << 1) & 0xffffu) | ((z0 >> 16) & 1u);
      wA |= (unsigned long long)((lo & 0xffffu) | (hiA << 16)) << (32 * h2);
      wB |= (unsigned long long)((lo >> 16) | (hiB << 16)) << (32 * h2);
    }
    {
      const int sh = 16 * g;
      const unsigned long long bA = __ballot(p256A), bB = __ballot(p256B);
      if (c == 4) { wA = (bA >> sh) & 1ull; wB = (bB >> sh) & 1ull; }
    }
  }
  // ---- publish this tile's bits; the spectra stay in v[] --------------------------------------------------------
  const int fa = F5_FPW * wave + 2 * g;      // tile row of frame A (B: fa + 1)
  unsigned long long* xb_mine = P.xbits + ((size_t)u * ntt + (jt + 1)) * O5_TILE_WORDS;
  if (c < O5_XW) {
    const op_v4u ga = {(unsigned)wA, P.epoch, (unsigned)(wA >> 32), P.epoch};
    const op_v4u gb = {(unsigned)wB, P.epoch, (unsigned)(wB >> 32), P.epoch};
    op_st16_sc1(&xb_mine[((fa)*O5_XW + c) * 2], ga);
    op_st16_sc1(&xb_mine[((fa + 1) * O5_XW + c) * 2], gb);
  }
  __syncthreads();   // every wave is past its forward exchange: the slices are idle from here
  if (halo_tile) return;

  // ---- integer smoothing on the matrix cores (onepass.hpp) ----------------------------------------------------------
  const int nt = P.nt;
  char* arena = reinterpret_cast<char*>(regions);
  unsigned long long* brow = reinterpret_cast<unsigned long long*>(arena);                           // [O5_ROWS][O5_BW]
  unsigned short* Ks = reinterpret_cast<unsigned short*>(arena + (size_t)O5_ROWS * O5_BW * 8);       // [32][O5_KP]
  static_assert(O5_ROWS * O5_BW * 8 + O5_NF * O5_KP * 2 <= WAVES * WAVE_CX_H * 8, "bits + K tile must fit the exchange slices");
  if (c < O5_XW) {
    brow[(nt + fa) * O5_BW + 1 + c] = wA;
    brow[(nt + fa + 1) * O5_BW + 1 + c] = wB;
  }
  for (int r = tid; r < O5_ROWS; r += WAVES * 64) {
    brow[r * O5_BW] = 0ull;
    brow[r * O5_BW + O5_BW - 1] = 0ull;
    if (r >= NF + 2 * nt) {       // rows past the neighbours': zero (the last k-block reads them)
#pragma unroll
      for (int w = 1; w <= O5_XW; ++w) brow[r * O5_BW + w] = 0ull;
    }
  }
  // neighbour rows: one 16-byte load per 64-bit word (2 nt x 5 words <= 240: one per thread), polled until both tags are current
  for (int i = tid; i < 2 * nt * O5_XW; i += WAVES * 64) {
    const int side = i >= nt * O5_XW;
    const int rem = i - side * nt * O5_XW;
    const int rr = rem / O5_XW, w = rem - rr * O5_XW;
    // tile j - 1 holds frames tf0 - step ..: frame tf0 - nt + rr is its row step - nt + rr; tile j + 1: frame tf0 + 32 + rr is its row 32 - step + rr
    const unsigned long long* src = side ? xb_mine + O5_TILE_WORDS + ((NF - step + rr) * O5_XW + w) * 2
                                         : xb_mine - O5_TILE_WORDS + ((step - nt + rr) * O5_XW + w) * 2;
    op_v4u gr = op_ld16_sc1(src);
    for (int spin = 0; gr[1] != P.poll_epoch || gr[3] != P.poll_epoch; ++spin) {
      if (spin >= P.spin_max) {   // bounded: report instead of hanging the device
        atomicOr_system(P.err, 1u);
        s_misc[1] = 1u;            // the tile's mask is unknown: every hop it finalises becomes NaN
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      gr = op_ld16_sc1(src);
    }
    brow[(side ? nt + NF + rr : rr) * O5_BW + 1 + w] = (unsigned long long)gr[0] | ((unsigned long long)gr[2] << 32);
  }
  const int q4 = lane >> 4, j16 = lane & 15;
  const long Bf = (long)P.tab[lane], At1 = (long)P.tab[64 + lane], At2 = (long)P.tab[128 + lane];
  __syncthreads();
  {
    // Per 16-bin block b:  H[row][bin] = sum_k bit[row][16 b - 8 + k] vf[k - 8 - j]   (A = 16 rows x 32 bins of 0 / 1 bytes, B = Bf)
    // for the five row blocks; the result layout (lane = bin column, 4 consecutive rows per lane group) IS the B layout of
    //   K[frame][bin] = sum_slot vt[row(slot) - nt - frame] H[row(slot)][bin]       (A = At1 / At2, B = two packed H blocks)
    // Output frames 16 hh + j (hh = 0, 1) sit at bit row nt + 16 hh + j and reach rows 16 hh + j .. 16 hh + j + 2 nt: row blocks
    // hh .. hh + 3; k-slot 8 q + e of the first product = row 4 q + e of block hh (e < 4) / block hh + 1 (e >= 4), of the second
    // = blocks hh + 2 / hh + 3 -- the same weights for both halves.
    typedef int o5_v4i __attribute__((ext_vector_type(4)));
    const o5_v4i zero4 = {0, 0, 0, 0};
    const unsigned char* wbb = reinterpret_cast<const unsigned char*>(brow);
    constexpr int WPB = O5_BW * 8;
    for (int b = wave; b < 17; b += WAVES) {
      unsigned hp[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const long a = (long)s_exp[wbb[(16 * m + j16) * WPB + 7 + q4 + 2 * b]];
        const o5_v4i hv = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, Bf, zero4, 0, 0, 0);
        hp[m] = (unsigned)hv[0] | ((unsigned)hv[1] << 8) | ((unsigned)hv[2] << 16) | ((unsigned)hv[3] << 24);
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const long bt1 = (long)(((unsigned long long)hp[hh + 1] << 32) | (unsigned long long)hp[hh]);
        const long bt2 = (long)(((unsigned long long)hp[hh + 3] << 32) | (unsigned long long)hp[hh + 2]);
        o5_v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8(At1, bt1, zero4, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_16x16x32_i8(At2, bt2, d, 0, 0, 0);
        unsigned short* kd = Ks + (16 * hh + 4 * q4) * O5_KP + 16 * b + j16;   // lane group q4: output frames 4 q4 .. 4 q4 + 3
        kd[0] = (unsigned short)d[0];
        kd[O5_KP] = (unsigned short)d[1];
        kd[2 * O5_KP] = (unsigned short)d[2];
        kd[3 * O5_KP] = (unsigned short)d[3];
      }
    }
  }
  __syncthreads();
  float ma[16], mb[16], m256a, m256b;
  {
    const float ks = A.inv_ktot * (0.5f / 512.0f) * P.prop;   // p K / ktot, the 1/2 of the split and the 1/512 of the inverse transform
    const unsigned short* KA = Ks + fa * O5_KP;
    const unsigned short* KB = KA + O5_KP;
    const float ksA = validA ? ks : 0.f, ksB = validB ? ks : 0.f;   // frames outside [0, T): mask zero (k_apply_fast512)
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) {
      const int f = bin5(c, sl);
      ma[sl] = (float)KA[f] * ksA;
      mb[sl] = (float)KB[f] * ksB;
    }
    m256a = (float)KA[256] * (2.f * ksA);
    m256b = (float)KB[256] * (2.f * ksB);
    if (P.prop != 1.0f) {   // + (1 - p) E / ktot (thresh.hpp: tri_valid)
      const float kq = (1.0f - P.prop) * A.inv_ktot * (0.5f / 512.0f);
      const float tA = validA ? kq * tri_valid(nt, tf0 + fa, G.T) : 0.f, tB = validB ? kq * tri_valid(nt, tf0 + fa + 1, G.T) : 0.f;
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) {
        const float wf = tri_valid(P.nf, bin5(c, sl), F5_F);
        ma[sl] = fmaf(wf, tA, ma[sl]);
        mb[sl] = fmaf(wf, tB, mb[sl]);
      }
      const float w256 = 2.f * tri_valid(P.nf, 256, F5_F);
      m256a = fmaf(w256, tA, m256a);
      m256b = fmaf(w256, tB, m256b);
    }
  }
  __syncthreads();   // every lane has its mask entries: the slices are free for the inverse transform

  // ---- x mask, merge, inverse transform, window, overlap-add, store (k_apply_fast512<K>) --------------------------
  const bool wave_live = tf0 + F5_FPW * wave + F5_FPW - 1 >= 0 && tf0 + F5_FPW * wave < G.T;
  if (wave_live) {
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
    auto sel = [&](cf a0, cf a1) -> cf { return {sel_s(F5_L0, a0.x, a1.x), sel_s(F5_L0, a0.y, a1.y)}; };
    cf nv[32];
    {
      const cf z0 = {v[0].x * (2.f * ma[0]), v[0].y * (2.f * mb[0])};
      const cf z8 = {v[8].x * m256a, v[8].y * m256b};
      nv[0] = sel(z0, na[0]);
      nv[8] = z8;
      nv[31] = nb[0];
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
      float ya = wave_live ? v[r].x * ws : 0.f, yb = wave_live ? v[r].y * ws : 0.f;
      if (!firstA) ya += *dA;
      *dA = ya;
      if (!firstB) yb += *dB;
      *dB = yb;
    }
    wave_lds_sync();
  }
  __syncthreads();
  const float poison = s_misc[1] != 0u ? __uint_as_float(0x7fc00000u) : 0.f;
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
    a4.x += poison; a4.y += poison; a4.z += poison; a4.w += poison;
    if (seam && (jj < 3 || jj >= NF)) {   // straddling hop: partial sum only (poisoned with the tile); slots 0..2 leading, 3..5 trailing
      const int slot = jj < 3 ? jj : 3 + (jj - NF);
      *reinterpret_cast<float4*>(A.part + ((((size_t)u * A.n_tiles + jt) * 6 + slot) * F5_H + s4)) = a4;
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
