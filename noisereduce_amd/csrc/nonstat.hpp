// Non-stationary mask of variant S (nonstationary.py:59-87) in two passes over the magnitude field:
//
//   S = filtfilt(one-pole)(|X|) along time   (get_time_smoothed_representation, nonstationary.py:106-115;
//                                             scipy filtfilt(padtype=None): forward seeded with A[0], backward
//                                             seeded with the forward pass's last value)
//   raw = sigmoid(((|X| - S) / S - thresh) * slope),   M = p * smooth(raw) + (1 - p)
//
// k_iir_sigmoid_seg + k_smooth_tiled moved the field 8 times (three sweeps of the recurrence over |X| and the
// raw mask, then the smoothing pass).  The recurrence is linear, so a time tile's contribution is fixed by two
// numbers per band -- its zero-state forward end value and the backward zero-state sum of its forward values --
// and the state entering any tile follows from a short chain over tiles:
//
//   k_iir_part   (reads |X| once)      per (tile, band): e_f = sum_t b c^(end-1-t) A[t],
//                                      E0 = sum_t b c^(t-start) s0[t]   (s0 = zero-state forward response)
//   k_iir_chain  (tiny)                forward state before / backward state after every tile
//   k_iir_mask   (reads |X| once more, writes M) per (time tile + nt halo rows, bin block incl. nf halo columns):
//                                      a column of the tile in registers; forward sweep; backward sweep that REGENERATES
//                                      the forward values in reverse (s_f[t-1] = (s_f[t] - b A[t]) / c: error growth
//                                      c^-rows, 1.25 at the default 2 s time constant) and writes the sigmoid in place;
//                                      separable triangle smoothing; p * . + (1 - p); store.
//
// The chain only knows the states at TILE boundaries; the states at the outer edges of a tile's halo rows follow
// from them inside k_iir_mask by the same inversion over the nt halo rows it has loaded anyway (forward state:
// s_f[t-1] = (s_f[t] - b A[t]) / c down the leading halo; backward state: S[t+1] = (S[t] - b s_f[t]) / c up the
// trailing halo, beside the forward sweep).  Rounds 2-3 split every tile at ts + nt and te - nt instead: three
// (e_f, E0) pairs per tile and band, 49 MB of partials and as many carries per 10-minute call, and a chain three
// times as long.  All recurrences in float64 (the reference's precision).
#pragma once
#include "kernels.hpp"
#include "nonstat_mask.hpp"

namespace sg {

// partials [unit][tile][2][FS].  One thread = FOUR adjacent bins of a tile (16-byte loads of the row-major field, four
// independent recurrences in flight; a bin per thread moved 4 bytes per lane and load -- 65 us for the 254 MB of a
// 10-minute call, 3.9 TB/s); threads are dealt over (tile, bin quad) pairs so that every lane of the last wave works.
// Columns F .. FS - 1 (padding of the field) are computed too and never read.
__global__ __launch_bounds__(256) void k_iir_part(const float* __restrict__ A, Geom g, NsTiling tl, double b,
                                                  double* __restrict__ part) {
  const unsigned C4 = (unsigned)g.FS / 4u;          // FS is a multiple of 16
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const int64_t nk = tl.n_tiles();
  if (idx >= (unsigned)nk * C4) return;
  const int64_t k = idx / C4;
  const int f0 = (int)(idx % C4) * 4;
  const int64_t u = blockIdx.y;
  const double c = 1.0 - b;
  const float* a = A + u * g.T * g.FS + f0;
  const int64_t ts = k * NS_TT, te = ts + NS_TT < g.T ? ts + NS_TT : g.T;
  double e[4] = {0.0, 0.0, 0.0, 0.0}, E0[4] = {0.0, 0.0, 0.0, 0.0}, pw = b;
  for (int64_t t = ts; t < te; t += 8) {   // 8 rows in flight (the recurrences are serial, their operands are not)
    float4 av[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) av[q] = *reinterpret_cast<const float4*>(a + (t + q < te ? t + q : te - 1) * g.FS);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (t + q < te) {
        const float x[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          e[j] = b * (double)x[j] + c * e[j];
          E0[j] += pw * e[j];
        }
        pw *= c;
      }
  }
  double* o = part + ((u * nk + k) * 2) * (int64_t)g.FS + f0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] = e[j];
    o[g.FS + j] = E0[j];
  }
}

// The same partials from k_mag_fast's 16-frame sub-tiles (fastpath.hpp: MagArgs::sub), four per time tile.  For
// consecutive pieces A (entering state 0, length L_A) and B: the state entering B is e_A, so
//   e_AB = e_B + c^L_B e_A,    E0_AB = E0_A + c^L_A (E0_B + e_A b c (1 - c^(2 L_B)) / (1 - c^2))
// (the bracket is k_iir_chain's E_b with s_in = e_A).  One thread per (tile, bin); every sub-tile has 16 frames but the
// unit's last.
__global__ __launch_bounds__(256) void k_iir_comb(const double* __restrict__ sub, Geom g, NsTiling tl, double b,
                                                  double* __restrict__ part, int nsub) {
  constexpr int SUBS = NS_TT / 16;
  const unsigned FSu = (unsigned)g.FS;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const int64_t nk = tl.n_tiles();
  if (idx >= (unsigned)nk * FSu) return;
  const int64_t k = idx / FSu;
  const int f = (int)(idx % FSu);
  if (f >= g.F) return;
  const int64_t u = blockIdx.y;
  const double c = 1.0 - b, gq = b * c / (1.0 - c * c);
  const double c16 = pow(c, 16.0), q16 = 1.0 - pow(c, 32.0);
  const double* sp = sub + ((u * nsub + k * SUBS) * 2) * (int64_t)g.FS + f;
  const int ns = (int)min<int64_t>(SUBS, nsub - k * SUBS);
  double ev[SUBS], Ev[SUBS];
#pragma unroll
  for (int j = 0; j < SUBS; ++j) {
    const int jj = j < ns ? j : ns - 1;
    ev[j] = sp[jj * 2 * (int64_t)g.FS];
    Ev[j] = sp[jj * 2 * (int64_t)g.FS + g.FS];
  }
  auto clen = [&](int j, double& cl, double& ql) {   // c^len and 1 - c^(2 len) of sub-tile j of this tile
    const int64_t len = min<int64_t>(16, g.T - (k * SUBS + j) * 16);
    if (len == 16) { cl = c16; ql = q16; } else { cl = pow(c, (double)len); ql = 1.0 - pow(c, 2.0 * (double)len); }
  };
  double e = ev[0], E0 = Ev[0], cL, q0;
  clen(0, cL, q0);
#pragma unroll
  for (int j = 1; j < SUBS; ++j)
    if (j < ns) {
      double cl, ql;
      clen(j, cl, ql);
      E0 += cL * (Ev[j] + e * gq * ql);
      e = ev[j] + cl * e;
      cL *= cl;
    }
  double* o = part + ((u * nk + k) * 2) * (int64_t)g.FS + f;
  o[0] = e;
  o[g.FS] = E0;
}

// carries [unit][tile][2][FS]: [0] forward state before the tile's first frame (s_f[ts - 1]; s_f[-1] := A[0], the
// lfilter_zi steady state), [1] backward state at its end (S[te]; S[T] := s_f[T - 1], the seed of the backward pass)
// TA / SQ: the first frame's magnitude as it lies in memory -- float |X| (the fused pipeline) or double |X|^2 (SQ: the
// float64 pipeline's power field; its tiles are tl.tt = 32 frames)
template <typename TA, bool SQ>
__global__ __launch_bounds__(64) void k_iir_chain(const TA* __restrict__ A, const double* __restrict__ part,
                                                  Geom g, NsTiling tl, double b, double* __restrict__ carry,
                                                  int64_t n_units) {
  const int64_t nk = tl.n_tiles();
  const int nj = (int)nk;
  const double c = 1.0 - b;
  // c^len and 1 - c^(2 len): every tile has NS_TT frames but the last (no table: an hour in one window has 10 k tiles)
  const double len_last = (double)(g.T - (nk - 1) * tl.tt);
  const double pf1 = pow(c, (double)tl.tt), pf2 = 1.0 - pow(c, 2.0 * tl.tt);
  const double pl1 = pow(c, len_last), pl2 = 1.0 - pow(c, 2.0 * len_last);
  auto pw1 = [&](int j) { return j == nj - 1 ? pl1 : pf1; };
  auto pw2 = [&](int j) { return j == nj - 1 ? pl2 : pf2; };
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_units * g.FS) return;
  const int64_t u = i / g.FS;
  const int f = (int)(i % g.FS);
  if (f >= g.F) return;
  const double* pb = part + (u * nk * 2) * (int64_t)g.FS + f;
  double* cb = carry + (u * nk * 2) * (int64_t)g.FS + f;
  double s = (double)A[u * g.T * g.FS + f];  // s[-1] = A[0]  (lfilter_zi steady state)
  if constexpr (SQ) s = sqrt(s);
  // the chain is serial, its operands are not: the partials of the NEXT 16 tiles are in flight while this batch's
  // recurrence runs (double-buffered by hand; a batch is one dependent round trip to memory otherwise)
  const int64_t st2 = 2 * (int64_t)g.FS;
  constexpr int CB = 16;
  {
    double e[2][CB];
    auto fetch = [&](int j0, double* d) {
#pragma unroll
      for (int q = 0; q < CB; ++q) d[q] = j0 + q < nj ? pb[(j0 + q) * st2] : 0.0;
    };
    auto run = [&](int j0, const double* d) {
#pragma unroll
      for (int q = 0; q < CB; ++q)
        if (j0 + q < nj) {
          cb[(j0 + q) * st2] = s;
          s = d[q] + pw1(j0 + q) * s;
        }
    };
    fetch(0, e[0]);
    for (int j0 = 0; j0 < nj; j0 += 2 * CB) {
      fetch(j0 + CB, e[1]);
      run(j0, e[0]);
      fetch(j0 + 2 * CB, e[0]);
      run(j0 + CB, e[1]);
    }
  }
  // backward: S[end_j] given; S[start_j] = E_b + c^len S[end_j],
  // E_b = sum_t b c^(t-start) s_f[t] = E0 + s_in * b c (1 - c^(2 len)) / (1 - c^2)   (s_f = s0 + c^(t-start+1) s_in)
  double S = s;  // seed: the forward pass's last value
  const double gq = b * c / (1.0 - c * c);
  {
    double e0[2][CB], sin[2][CB];
    auto fetch = [&](int j1, double* d0, double* d1) {
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const int j = j1 - q;
        d0[q] = j >= 0 ? pb[j * st2 + g.FS] : 0.0;
        d1[q] = j >= 0 ? cb[j * st2] : 0.0;
      }
    };
    auto run = [&](int j1, const double* d0, const double* d1) {
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const int j = j1 - q;
        if (j >= 0) {
          cb[j * st2 + g.FS] = S;
          S = (d0[q] + d1[q] * gq * pw2(j)) + pw1(j) * S;
        }
      }
    };
    fetch(nj - 1, e0[0], sin[0]);
    for (int j1 = nj - 1; j1 >= 0; j1 -= 2 * CB) {
      fetch(j1 - CB, e0[1], sin[1]);
      run(j1, e0[0], sin[0]);
      fetch(j1 - 2 * CB, e0[0], sin[0]);
      run(j1 - CB, e0[1], sin[1]);
    }
  }
}

// (round 5) k_iir_comb + k_iir_chain in ONE kernel with the chain cut into 16 runs of tiles per band.  The serial form walks a
// band's tiles one dependent multiply-add at a time, forward and back (41 tiles at the default geometry, 162 at a hop of 64
// samples: 17 - 33 us of latency with a few hundred wavefronts on the chip), after k_iir_comb has already read the 16-frame
// partials once (19 us).  The recurrences are affine maps (s' = e + c^len s;  S_start = E_b + c^len S_end), and maps compose:
//   1. wavefront w of a workgroup (64 bands x 16 wavefronts) composes the forward maps of its run of pieces;
//   2. through LDS every wavefront learns the state entering its run (at most 15 compositions from s[-1] = A[0]);
//   3. it replays its run with that state: the forward carries of its tiles, the backward terms E_b of its pieces (they need
//      the entering states) and the composed BACKWARD map of the run, accumulated in the same forward order;
//   4. through LDS again: the state at the end of its run from the runs behind it and the seed S[T] = s_f[T - 1];
//   5. it replays its run backwards: the backward carries.
// PER: pieces per 64-frame tile -- 4 (k_mag_fast's 16-frame sub-tiles, no k_iir_comb) or 1 (k_iir_part's tile partials).
// A run is at most NSP_MAXP pieces (they live in registers): longer windows keep the serial kernels (nsp_ok).
// carry: [unit][tile][2][FS] exactly as k_iir_chain writes it.
constexpr int NSP_WAVES = 16, NSP_MAXP = 16;
__host__ __device__ inline bool nsp_ok(int64_t n_tiles, int per) {
  const int64_t G = (n_tiles + NSP_WAVES - 1) / NSP_WAVES;
  return n_tiles >= 1 && G * per <= NSP_MAXP;
}

template <int PER>
__global__ __launch_bounds__(64 * NSP_WAVES) void k_iir_chain_par(const float* __restrict__ A, const double* __restrict__ pieces,
                                                                   Geom g, NsTiling tl, double b, double* __restrict__ carry,
                                                                   int np, double pf1, double pf2, double pl1, double pl2) {
  __shared__ double sA[NSP_WAVES][64], sB[NSP_WAVES][64];   // forward maps of the runs, then their backward maps
  __shared__ double sSeed[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane;
  const bool live = f < g.F;
  const int fc = live ? f : g.F - 1;
  const int64_t u = blockIdx.y;
  const int nk = (int)tl.n_tiles();
  const double c = 1.0 - b, gq = b * c / (1.0 - c * c);
  // pf1 = c^PLEN, pf2 = 1 - c^(2 PLEN) for full pieces, pl1 / pl2 the same for the unit's last piece: from the HOST (four
  // float64 pow() per thread were most of this kernel's instructions: 442 k threads per ten minutes where k_iir_chain has 25 k)
  const int G = (nk + NSP_WAVES - 1) / NSP_WAVES;          // tiles per run
  const int k0 = w * G;
  const int j0 = k0 * PER;
  const int j1 = min(np, (k0 + G) * PER);
  const int cnt = j1 > j0 ? j1 - j0 : 0;                    // pieces of this run (0: a wavefront beyond the last tile)
  const int64_t st2 = 2 * (int64_t)g.FS;
  const double* pb = pieces + (u * np * 2) * (int64_t)g.FS + fc;
  double e[NSP_MAXP], E[NSP_MAXP];
#pragma unroll
  for (int q = 0; q < NSP_MAXP; ++q) {
    // (clamped, not predicated: with `if (q < cnt)` around the loads the call took 1.65 ms instead of 0.49 -- measured)
    const int j = q < cnt ? j0 + q : (np - 1);
    e[q] = pb[j * st2];
    E[q] = pb[j * st2 + g.FS];
  }
  double s = (double)A[u * g.T * g.FS + fc];               // s[-1] = A[0]  (lfilter_zi steady state)
  // 1. forward map of the run: s_out = Bm + Am s_in
  {
    double Am = 1.0, Bm = 0.0;
#pragma unroll
    for (int q = 0; q < NSP_MAXP; ++q)
      if (q < cnt) {
        const double cl = (j0 + q == np - 1) ? pl1 : pf1;
        Bm = e[q] + cl * Bm;
        Am *= cl;
      }
    sA[w][lane] = Am;
    sB[w][lane] = Bm;
  }
  __syncthreads();
  // 2. the state entering the run
  for (int v = 0; v < w; ++v) s = sB[v][lane] + sA[v][lane] * s;
  __syncthreads();                                          // (sA / sB are reused for the backward maps)
  // 3. forward replay: carries, backward terms, the run's backward map S_start = Qm + Pm S_end
  double* cb = carry + (u * nk * 2) * (int64_t)g.FS + fc;
  {
    double Pm = 1.0, Qm = 0.0;
#pragma unroll
    for (int q = 0; q < NSP_MAXP; ++q)
      if (q < cnt) {
        const bool last = j0 + q == np - 1;
        const double cl = last ? pl1 : pf1, ql = last ? pl2 : pf2;
        if (q % PER == 0 && live) cb[(k0 + q / PER) * st2] = s;
        const double Eb = E[q] + s * gq * ql;               // sum_t b c^(t - start) s_f[t] of the piece
        E[q] = Eb;
        Qm += Pm * Eb;
        Pm *= cl;
        s = e[q] + cl * s;
      }
    sA[w][lane] = Pm;
    sB[w][lane] = Qm;
    if (cnt > 0 && j1 == np) sSeed[lane] = s;               // the forward pass's last value seeds the backward pass
  }
  __syncthreads();
  // 4. the state at the end of the run
  double S = sSeed[lane];
  for (int v = NSP_WAVES - 1; v > w; --v) S = sB[v][lane] + sA[v][lane] * S;
  // 5. backward replay
#pragma unroll
  for (int q = NSP_MAXP - 1; q >= 0; --q)
    if (q < cnt) {
      const bool last = j0 + q == np - 1;
      const double cl = last ? pl1 : pf1;
      if ((q % PER == PER - 1 || last) && live) cb[(k0 + q / PER) * st2 + g.FS] = S;
      S = E[q] + cl * S;
    }
}


}  // namespace sg
