// TorchGate.forward (variant T, stationary, statistics from the row itself) at the default geometry in ONE kernel:
// one workgroup = one batch row (clip) of at most 64 frames.
//
//   k_power_fast64 (float64 transform of every frame) -> k_row_decide (68 MB of float64 powers re-read)
//   -> k_smooth_bits2 -> k_apply_fast (forward transform again, K field through HBM)        4 launches, 8.0 x algorithmic
//     ->  k_row_gate                                                                        1 launch, samples in / out
//
// 16 wavefronts, wave w = frames 4w .. 4w+3 (the register FFT of fastpath.hpp).  The float32 spectra of the WHOLE row
// stay in registers (64 per lane) from the forward transform to the inverse one; everything in between lives in LDS:
//
//   1. stage the row's samples, gather + window, forward transform, real-FFT split (once)
//   2. |2X|^2 (float32) of all 64 x 513 cells -> LDS tile; one thread pair per band walks its column: maximum, then
//      floored dB values relative to the maximum (torchgate/utils.py:5-23: top_db = 40 under the band's maximum over time),
//      mean and standard deviation (ddof = 1, torchgate.py:158-160) accumulated in float64, threshold -> compare constant
//      in the power domain -- together with a BOUND on its error (below)
//   3. every lane decides its own 33 cells against the band constants; a cell closer to the threshold than the bound
//      makes its (row, band) AMBIGUOUS
//   4. ambiguous bands (a few per row) are re-evaluated EXACTLY: the 63 float64 DFT sums of that band (1024 terms each),
//      float64 statistics with the reference's formulas, float64 compare -- their bits replace the float32 ones
//   5. exact integer separable triangle smoothing of the bit tile in LDS (the arithmetic of k_smooth_bits2)
//   6. x mask -> merge -> inverse transform -> window -> overlap-add inside the workgroup -> output.  No tiles, no
//      hand-offs between workgroups, no tickets: a row is one workgroup.
//
// Error bound of step 2 (why float32 is enough).  Measured on this kernel's own transform (tools/rowgate_margin.py,
// profiles/r04_rowgate_margin.json: 7 signal families, 7 M cells): the float32 |X| is off by 0.0068 (RMS) / 0.047 (max) x
// 2^-16 ||x w||_2 on noise-like bins, and by <= 3 ulp of |X| itself on the bins of a strong tone.  The kernel assumes
//   | |2X|_f32 - |2X| |  <=  d_t + RG_REL |2X|,   d_t = 2 * 2^-18 ||x w||_2 (5 x the largest error seen, 37 x RMS),
//                                                  RG_REL = 2^-21 (8 ulp)
// In dB a cell is off by at most  e_i = (20 / ln 10) x (1 + 2 x),  x = d_t / |2X_i| + RG_REL  (x <= 1/2; deeper nulls that
// are not surely floored make the band ambiguous at once); cells surely below the floor contribute -top_db exactly (no
// error), the band maximum is off by e_M.  Mean and standard deviation are Lipschitz in their inputs:
//   |d thresh| <= sum(e) / n + |n_std| sqrt(sum(e^2) / (n - ddof))            (worst case: all errors conspire)
// plus 2e-5 dB for the float32 logarithm / exponential and the effect of the reference's eps = 2.2e-16 inside the
// logarithm (computed per band; bands whose maximum is below 1e-8 are ambiguous unless the row is digital silence).
// Sums run in float64, so their own rounding does not count.  Ambiguous: ~0.3 % of the (row, band) pairs of noise + tone.
#pragma once
#include "fastpath.hpp"
#include "fused.hpp"   // funnel_r

namespace sg {
namespace fast {

// Two shapes of the workgroup (template parameters WAVES x QUADS, 64 frames per row at most either way):
//   16 x 1: every wave transforms one quad of frames; 4 waves per SIMD, 128 VGPRs (the compiler spills ~30 around the
//           transforms), the exchange slices alias the staged samples;
//    8 x 2: every wave transforms two quads one after the other; 2 waves per SIMD, 256 VGPRs, no scratch, the slices sit
//           behind the samples.  Measured on 256 x 16000: 16 x 1 is the faster one (tools/rowgate_scale.py).
constexpr int RG_FRAMES = 64;
constexpr int RG_NTMAX = 16;             // time half-width of the smoothing filter
constexpr int RG_NFMAX = 30;             // frequency half-width (128-bit sliding window)
constexpr int RG_PP = 528;               // float pitch of the power tile: rows 4w+g of a wave land on disjoint bank quarters
constexpr int RG_WP = 11;                // 64-bit words per bit row: 9 + one zero word on each side
constexpr int RG_FP = 584;               // uint16 pitch of the count / K tile: column(f) = f + 4 (f / 32)  (smooth2_pitch(513))
constexpr int RG_ROWS_MAX = 64 + 2 * RG_NTMAX;
constexpr float RG_REL = 4.7683716e-7f;  // 2^-21: relative part of the float32 transform's error bound
// The tile region (RG_REGION bytes) over time: samples (67 hops of 256) + exchange slices behind them | power tile |
// samples + float64 window + w_1024^j + exact powers | samples + slices | count / K tile | slices + hop accumulators
constexpr int RG_REGION = 16 * WAVE_CX_H * 8;          // 139264
constexpr int RG_SPAN_BYTES = 67 * 256 * 4;            // 68608: hop pitch 256 (8 x 2: the slices must fit behind the samples)
constexpr int RG_SPAN_BYTES_P = 67 * 288 * 4;          // 77184: hop pitch 288 (16 x 1: conflict-free gather, see k_apply_fast)
__host__ __device__ constexpr int rg_slices(int waves) { return RG_REGION - waves * WAVE_CX_H * 8; }   // the slices sit at the END of the region
constexpr int RG_EX_W64 = 77312, RG_EX_TW = RG_EX_W64 + 8192, RG_EX_PW = RG_EX_TW + 16384, RG_EX_BANDS = 64;
static_assert(RG_SPAN_BYTES <= rg_slices(8) && RG_SPAN_BYTES_P <= RG_EX_W64 && RG_EX_PW + RG_EX_BANDS * 64 * 8 <= RG_REGION,
              "tile-region LDS map");

#ifndef RG_TRACE
#define RG_TRACE 0   // development only: per-phase shader-clock stamps (tools/rowgate_trace.sh)
#endif
struct RowGateArgs {
  View view;
  Geom g;
  OutMap om;
  const float* win;          // analysis == synthesis window (1024)
  const float* wsq;          // window squared (1024)
  const float* invn;         // 1 / sum_q wsq[256 q + s]
  const cf* tw512;
  const cf* tw1024;
  const double* win64;       // exact path
  const cx<double>* tw64;    // w_1024^j float64
  double mag_scale, top_db, n_std;
  int ddof;
  int nf, nt;
  float kscale;              // 1 / (ktot * 512)
  float inv_ktot;
  float* mask_out;           // optional [rows][T][FS] float mask (natural bin order) for the backward pass
  unsigned long long* bits_out;   // optional [rows][T][9] decisions (stage tap)
  unsigned* n_exact;         // optional counter: (row, band) pairs that took the exact path
  float* ptile_out;          // optional [rows][64][RG_PP] float32 powers (4x) of pass 1 (stage tap: error-margin measurements)
#if RG_TRACE
  unsigned long long* trace; // [rows][16] shader-clock stamps of wave 0 at the phase boundaries (development builds)
#endif
};

#if RG_TRACE
#define RG_STAMP(i) do { if (tid == 0) A.trace[(size_t)blockIdx.x * 16 + (i)] = (unsigned long long)clock64(); } while (0)
#else
#define RG_STAMP(i) do { } while (0)
#endif

__host__ __device__ constexpr size_t rowgate_lds_bytes() {
  return (size_t)FN * 8 + (size_t)RG_REGION + 1024 * 4 + (size_t)RG_ROWS_MAX * RG_WP * 8 +
         2 * T2_FLOATS * 4 + 2 * 64 * 4 + 64 * 8 + 520 + 520 * 2 + 64 + 144;
}

// exact float64 |X[f]|^2 (UNSCALED transform, like k_power_fast64's) of frame t of a row: the whole wave cooperates
__device__ __forceinline__ double rg_exact_power(const RowGateArgs& A, int64_t row, int64_t t, int f, int lane) {
  const int64_t s0 = t * 256 - A.g.padL;
  double re = 0.0, im = 0.0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int m = lane + 64 * i;
    const double xv = view_sample(A.view, row, 0, s0 + m) * A.win64[m];
    const int j = (f * m) & 1023;
    cx<double> w = A.tw64[j & 511];
    if (j >= 512) { w.x = -w.x; w.y = -w.y; }
    re += xv * w.x;
    im += xv * w.y;
  }
  for (int off = 32; off > 0; off >>= 1) {
    re += __shfl_xor(re, off);
    im += __shfl_xor(im, off);
  }
  return re * re + im * im;
}

template <int RG_WAVES, int RG_QUADS>
__global__ __launch_bounds__(RG_WAVES * 64, 1) void k_row_gate(RowGateArgs A) {
  static_assert(RG_WAVES * RG_QUADS * 4 == RG_FRAMES, "64 frames per row");
  constexpr int RG_THREADS = RG_WAVES * 64;
  constexpr int RG_SLICES = rg_slices(RG_WAVES);
  constexpr bool ALIASED = RG_SLICES < RG_SPAN_BYTES;    // (16 x 1) the exchanges overwrite the staged samples
  constexpr int XP = ALIASED ? 288 : 256;                // floats between hops of the staged samples
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  char* region = reinterpret_cast<char*>(tw512 + FN);
  float* swin = reinterpret_cast<float*>(region + RG_REGION);
  unsigned long long* wb = reinterpret_cast<unsigned long long*>(swin + 1024);          // bit rows [2 nt + 64][RG_WP]
  float* s_t2 = reinterpret_cast<float*>(wb + RG_ROWS_MAX * RG_WP);                      // [513] compare constants (4x power)
  float* s_cb = s_t2 + T2_FLOATS;                                                        // [513] ambiguity widths 4 eta^2 T^2
  float* s_d2 = s_cb + T2_FLOATS;                                                        // [64] per frame: 2 (2 delta_t)^2
  float* s_dl = s_d2 + 64;                                                               // [64] per frame: 2 delta_t
  double* s_ex = reinterpret_cast<double*>(s_dl + 64);                                   // [64] exact powers of one band (slow path)
  unsigned char* s_flag = reinterpret_cast<unsigned char*>(s_ex + 64);                   // [520] band is ambiguous
  unsigned short* s_list = reinterpret_cast<unsigned short*>(s_flag + 520);              // [520] compacted
  unsigned* s_misc = reinterpret_cast<unsigned*>(s_list + 520);                          // [0] list length  [1] row has a sample
  cf* s_tw1024 = reinterpret_cast<cf*>(s_misc + 16);                                     // [17] w_1024^0..16 (the split's twiddles)
  const Geom& G = A.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: an SGPR)
  const int g = lane >> 4, c = lane & 15;
  const int nt = A.nt, nf = A.nf;
  const int T = (int)G.T;
  const int64_t row = A.view.unit0 + blockIdx.x;
  const bool l0 = c == 0;
  float* tile = reinterpret_cast<float*>(region);                    // power tile [64][RG_PP] (between the transforms)
  float* xs = reinterpret_cast<float*>(region);                      // the row's samples: hop h at xs + 256 h
  cf* slices = reinterpret_cast<cf*>(region + RG_SLICES);            // 8 exchange slices, behind the samples
  const int span = (T - 1) * 256 + 1024;       // samples covered by the row's frames, from position -padL
  // frame of this lane group in quad q: wave w transforms frames 4 (w + 8 q) .. + 3
  auto frame_of = [&](int q) { return 4 * (wave + RG_WAVES * q) + g; };

  // the row's samples -> LDS, zero outside the row.  float32 rows: every 16-byte load of the thread is issued before the
  // first LDS store (span_issue / span_commit; a rolled load-store loop is serialised by the compiler: s_waitcnt vmcnt(0)
  // per iteration, four to five dependent round trips with nothing else on the CU to hide them), indices clamped into
  // the row, the zero padding and the row's ragged end patched at the store.
  constexpr int SPAN_K = (((RG_FRAMES - 1) * 256 + 1024) / 4 + RG_THREADS - 1) / RG_THREADS;
  const float* sp_row = (const float*)A.view.x + row * A.view.stride;
  const bool span_vec = A.view.dtype == 0 && (reinterpret_cast<uintptr_t>(sp_row) & 15) == 0 && A.view.lo <= 0 &&
                        A.view.hi >= A.view.Lp && A.view.Lp >= 8;
  auto span_issue = [&](float4 (&q)[SPAN_K]) __attribute__((always_inline)) {
    if (!span_vec) {
#pragma unroll
      for (int k = 0; k < SPAN_K; ++k) q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    const int64_t last4 = (A.view.Lp & ~(int64_t)3) - 4;
#pragma unroll
    for (int k = 0; k < SPAN_K; ++k) {
      const int64_t s = (int64_t)4 * (tid + k * RG_THREADS) - G.padL;      // multiple of 4 (padL = 512)
      q[k] = *reinterpret_cast<const float4*>(sp_row + min(max(s, (int64_t)0), last4));
    }
  };
  auto span_commit = [&](float4 (&q)[SPAN_K]) __attribute__((always_inline)) -> bool {
    bool any = false;
    if (span_vec) {
#pragma unroll
      for (int k = 0; k < SPAN_K; ++k) {
        const int i4 = tid + k * RG_THREADS;
        if (i4 < span / 4) {
          const int e = 4 * i4;
          const int64_t s = (int64_t)e - G.padL;
          float4 v4 = q[k];
          if (!(s >= 0 && s + 4 <= A.view.Lp)) {   // zero padding before / behind the row, or the row's ragged last samples
            auto at = [&](int64_t sj) -> float { return (sj >= 0 && sj < A.view.Lp) ? sp_row[sj] : 0.f; };
            v4 = make_float4(at(s), at(s + 1), at(s + 2), at(s + 3));
          }
          any = any || v4.x != 0.f || v4.y != 0.f || v4.z != 0.f || v4.w != 0.f;
          *reinterpret_cast<float4*>(&xs[(e >> 8) * XP + (e & 255)]) = v4;
        }
      }
    } else {
      for (int i4 = tid; i4 < span / 4; i4 += RG_THREADS) {
        const int e = 4 * i4;
        const int64_t s = (int64_t)e - G.padL;
        float4 v4;
        v4.x = (float)view_sample(A.view, row, 0, s);
        v4.y = (float)view_sample(A.view, row, 0, s + 1);
        v4.z = (float)view_sample(A.view, row, 0, s + 2);
        v4.w = (float)view_sample(A.view, row, 0, s + 3);
        any = any || v4.x != 0.f || v4.y != 0.f || v4.z != 0.f || v4.w != 0.f;
        *reinterpret_cast<float4*>(&xs[(e >> 8) * XP + (e & 255)]) = v4;
      }
    }
    return any;     // this thread staged a non-zero sample (NaN counts as one)
  };
  auto stage_span = [&]() __attribute__((always_inline)) -> bool {
    float4 q[SPAN_K];
    span_issue(q);
    return span_commit(q);
  };
  // gather (window x frame t) -> forward transform -> split in place: v[e] = 2 X[bin_of_entry(c, e)]; lane 0 keeps its two
  // unpaired registers raw: v[0] = Zc[0] (bins 0 / 512), v[31] = Zc[256] (bin 256).  Returns 2 (2 delta_t)^2.
  // The exchange slices lie behind the samples, so the transforms of one quad do not disturb the gather of the other.
  auto forward = [&](cf* v, int t) -> float {
    cf wlo, whi;
    // fresh (opaque) lane arithmetic per call: shared between the calls (CSE), the 32 swizzled exchange addresses and the
    // gather addresses would stay live across everything in between
    int c = lane & 15, zo = 0;
    asm volatile("" : "+v"(c), "+v"(zo));
    const bool l0 = c == 0;
    const bool fvalid = t < T;
    cf* fb = slices + wave * WAVE_CX_H + frame_base_h(g) + zo;
    {
      const float* xp = xs + t * XP + 2 * c + zo;
      const float2* wl = reinterpret_cast<const float2*>(swin + 2 * c + zo);
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        float2 x2 = *reinterpret_cast<const float2*>(xp + (r >> 3) * XP + 32 * (r & 7));
        if (!fvalid) x2 = make_float2(0.f, 0.f);
        const float2 w2 = wl[16 * r];
        v[r] = {x2.x * w2.x, x2.y * w2.y};
      }
    }
    if constexpr (ALIASED) __syncthreads();   // every lane has its samples: the exchanges may overwrite them
    float nrm2 = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) nrm2 += v[r].x * v[r].x + v[r].y * v[r].y;
    nrm2 += __shfl_xor(nrm2, 1);
    nrm2 += __shfl_xor(nrm2, 2);
    nrm2 += __shfl_xor(nrm2, 4);
    nrm2 += __shfl_xor(nrm2, 8);
    fft512_fwd_half(v, fb, tw512 + zo, c);
    wlo = s_tw1024[c];
    asm volatile("" : "+v"(wlo.x), "+v"(wlo.y));
    whi = wlo;
    {
      const cf w16 = s_tw1024[16];
      if (l0) whi = {-w16.y, w16.x};  // i * w_1024^16
    }
    rg_lane0_to_entries(v, l0);
    {
      const cf r0 = v[0], r31 = v[31];
      cf xa, xb;
      split_pair(r0, r31, wlo, xa, xb);
      v[0] = {l0 ? r0.x : xa.x, l0 ? r0.y : xa.y};
      v[31] = {l0 ? r31.x : xb.x, l0 ? r31.y : xb.y};
#pragma unroll
      for (int sl = 1; sl < 16; ++sl) {
        const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
        cf ya, yb;
        split_pair(v[sl], v[31 - sl], w, ya, yb);
        v[sl] = ya;
        v[31 - sl] = yb;
      }
    }
    return 8.0f * 1.4551915e-11f * nrm2;   // 2 (2 delta_t)^2 with delta_t = 2^-18 ||x w||_2 (see RG_REL)
  };

  // ---- tables, zero-filled bit rows, the row's samples ------------------------------------------------------------------
  // (all global loads of the prologue -- tables and span -- in flight before the first store)
  static_assert(RG_THREADS >= FN && RG_THREADS >= 256, "one table load per thread");
  bool any_sample;
  {
    const int it = min(tid, FN - 1);
    const cf tw_v = A.tw512[(it >> 4) * (it & 15)];
    const float4 w4 = reinterpret_cast<const float4*>(A.win)[min(tid, 255)];
    const cf t10 = A.tw1024[min(tid, 16)];
    float4 q[SPAN_K];
    span_issue(q);
    for (int i = tid; i < (T + 2 * nt) * RG_WP; i += RG_THREADS) wb[i] = 0ull;
    for (int i = tid; i < 520; i += RG_THREADS) s_flag[i] = 0;
    unsigned* s_maxu0 = reinterpret_cast<unsigned*>(s_cb);
    for (int i = tid; i < 513; i += RG_THREADS) s_maxu0[i] = 0u;
    if (tid == 0) { s_misc[0] = 0u; s_misc[1] = 0u; s_misc[2] = 0u; }
    if (tid < FN) tw512[tid] = tw_v;
    if (tid < 256) reinterpret_cast<float4*>(swin)[tid] = w4;
    if (tid < 17) s_tw1024[tid] = t10;
    RG_STAMP(0);
    any_sample = span_commit(q);
  }
  unsigned* s_maxu = reinterpret_cast<unsigned*>(s_cb);   // band maxima (bit patterns) until the statistics replace them by cb
  __syncthreads();
  if (any_sample) s_misc[1] = 1u;     // (every writer stores 1)  0: digital silence
  RG_STAMP(1);   // tables + span

  // ---- pass 1: powers (4x) of all cells -> LDS tile --------------------------------------------------------------------
  {
    float pw[RG_QUADS][32], p512[RG_QUADS], d2q[RG_QUADS];
#pragma unroll
    for (int q = 0; q < RG_QUADS; ++q) {
      cf v[32];
      d2q[q] = forward(v, frame_of(q));
#pragma unroll
      for (int e = 0; e < 32; ++e) pw[q][e] = v[e].x * v[e].x + v[e].y * v[e].y;
      const float x0 = 2.f * (v[0].x + v[0].y), xN = 2.f * (v[0].x - v[0].y);   // lane 0: 2 X[0], 2 X[512]
      pw[q][0] = l0 ? x0 * x0 : pw[q][0];
      pw[q][31] = l0 ? 4.f * pw[q][31] : pw[q][31];
      p512[q] = xN * xN;
    }
    __syncthreads();   // all gathers and forward exchanges done: samples and slices become the power tile
    // tile columns of this lane's entries: lanes c >= 1: entry e < 16 = bin c + 32 e, entry e >= 16 = bin (32 - c) + 32 (e - 16)
#pragma unroll
    for (int q = 0; q < RG_QUADS; ++q) {
      const int t = frame_of(q);
      if (t < T) {
        float* trow = tile + t * RG_PP;
        float* t_lo = trow + c;
        float* t_hi = trow + (32 - c);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          const int b0 = bin_of_entry(0, e);   // lane 0 (compile-time)
          float* dst = l0 ? trow + b0 : (e < 16 ? t_lo + 32 * e : t_hi + 32 * (e - 16));
          *dst = pw[q][e];
          // band maximum over the frames: LDS atomic on the bit pattern (powers are >= 0; a NaN pattern beats every
          // number, so the maximum is NaN-sticky like torch.max)
          atomicMax(s_maxu + (dst - trow), __float_as_uint(pw[q][e]));
        }
        if (l0) {
          trow[512] = p512[q];
          atomicMax(s_maxu + 512, __float_as_uint(p512[q]));
          s_d2[t] = d2q[q];
          const float dl = __builtin_amdgcn_sqrtf(0.5f * d2q[q]);
          s_dl[t] = dl;
          atomicMax(&s_misc[2], __float_as_uint(dl));     // largest 2 delta_t of the row (NaN-sticky too)
        }
      }
    }
  }
  __syncthreads();
  RG_STAMP(2);   // pass 1: gather, forward transform, power tile
  if (A.ptile_out) {
    float* po = A.ptile_out + (size_t)blockIdx.x * 64 * RG_PP;
    for (int i = tid; i < T * RG_PP; i += RG_THREADS) po[i] = tile[i];
  }

  // ---- band statistics: thread pair (2j, 2j+1) = band j (frames of equal parity), 256 bands per round ---------------------
  {
  const float kDb = 3.01029995663981195f;      // 10 log10(2)
  const float E_LOG = 2e-5f;                   // float32 log2 / exp2 evaluation (see header)
  const float top = (float)A.top_db;
  const bool has_sample = s_misc[1] != 0u;
  const int par = tid & 1;
  for (int f = tid >> 1; f < 513; f += RG_THREADS / 2) {
    // the band's maximum was gathered while the tile was written; its error: the row's largest 2 delta_t (conservative)
    const float M = __uint_as_float(s_maxu[f]);
    const float dM = __uint_as_float(s_misc[2]);
    // amplitude of the maximum and its absolute error
    const float aM = __builtin_amdgcn_sqrtf(M);
    const float delM = dM + RG_REL * aM;
    bool exact = false;
    float t2 = 3.0e38f, cb = 0.f;              // default: no cell passes (NaN in the band)
    if (M != M) {
      // keep the default
    } else if (!(M < 3.0e38f)) {
      exact = true;                            // overflow of the float32 power: float64 decides
    } else if (!(aM * 0.5f * (float)A.mag_scale >= 1e-8f)) {
      // very quiet band: the reference's eps matters.  Digital silence (every sample 0): all cells equal, `>` is
      // False everywhere (t2 = 0: P > 0 never holds for P = 0); anything else is left to float64
      if (has_sample) exact = true;
      else { t2 = 0.f; cb = 0.f; }
    } else {
      const float xM = delM / aM;
      if (xM > 0.25f) exact = true;
      const float eM = 8.6858896f * xM * (1.f + 2.f * xM);
      // the reference's eps inside the logarithm moves an unfloored cell (>= 10^(-top/20) of the maximum) by at most
      const float E_FIX = E_LOG + 8.6858896f * 2.220446e-16f / (0.0099f * aM * 0.5f * (float)A.mag_scale);
      // surely floored:  |2X_i| (1 + rel) + 2 delta_i < 10^(-top/20) (aM - delM) (1 - 2^-20)
      const float Famp = __builtin_amdgcn_exp2f(-top * (1.f / 6.0205999f)) * (aM - delM) * 0.999999f;
      const float rM = 1.0f / M;
      // pass 2: floored dB relative to the maximum, float64 sums; error sums in float32
      double S1 = 0.0, S2 = 0.0;
      float E1 = 0.f, E2 = 0.f;
      for (int tt = par; tt < T; tt += 2) {
        const float p = tile[tt * RG_PP + f];
        const float dl = s_dl[tt];                          // 2 delta_t
        const float rs = __builtin_amdgcn_rsqf(p);
        const float a = p * rs;                             // |2X| (p = 0: NaN, handled by the floor test below)
        float d = kDb * __builtin_amdgcn_logf(p * rM);      // <= 0 (up to rounding)
        float e = 0.f;
        if (!(p > 0.f) || fmaf(a, RG_REL, a) + dl < Famp) {
          d = -top;                                         // surely floored: exact
        } else {
          const float x = fmaf(dl, rs, RG_REL);
          if (x > 0.5f) exact = true;                       // a deep null that may or may not be floored
          e = 8.6858896f * x * (1.f + 2.f * x) + eM + E_FIX;
          d = fmaxf(d, -top);
          d = fminf(d, 0.f);
        }
        S1 += (double)d;
        S2 += (double)d * (double)d;
        E1 += e;
        E2 = fmaf(e, e, E2);
      }
      S1 += __shfl_xor(S1, 1);
      S2 += __shfl_xor(S2, 1);
      E1 += __shfl_xor(E1, 1);
      E2 += __shfl_xor(E2, 1);
      exact = exact || (__shfl_xor((int)exact, 1) != 0);
      const double Tn = (double)T;
      const double mean_d = S1 / Tn;
      double var = (S2 - S1 * S1 / Tn) / (Tn - (double)A.ddof);
      if (var < 0.0) var = 0.0;
      const float th = (float)(mean_d + sqrt(var) * A.n_std);            // threshold relative to the maximum, dB
      // (+ eM outside the sums: th is relative to the ESTIMATED maximum and t2 below multiplies by it again -- in absolute dB
      // the floored cells, exact relative to the maximum, carry the maximum's own error; a band where almost every frame is
      // floored would otherwise see only the unfloored cells' share of it)
      const float Eth = E1 / (float)T + fabsf((float)A.n_std) * __builtin_amdgcn_sqrtf(E2 / (float)(T - A.ddof)) + E_FIX + eM;
      if (-top > th + Eth) {
        t2 = -3.0e38f;                           // the floor lifts every cell above the threshold: all pass
      } else if (-top > th - Eth || !(Eth < 1.0f) || !(th == th)) {
        exact = true;
      } else {
        // compare constant in the (4x) power domain: T^2 = M * 2^(th / (10 log10 2)); ambiguity width 4 eta^2 T^2,
        // eta = 10^(Eth / 20) - 1 (relative error of the amplitude threshold) + the cell's own relative error
        t2 = M * __builtin_amdgcn_exp2f(th * (1.f / kDb));
        const float eta = expm1f(Eth * 0.11512925f) + 1.01f * RG_REL;   // 10^(Eth / 20) - 1, not its first-order term (Eth < 1 dB)
        cb = 4.f * eta * eta * t2;
      }
    }
    if (par == 0) {
      s_t2[f] = t2;
      s_cb[f] = cb;
      if (exact) s_flag[f] = 1;
    }
  }
  }
  __syncthreads();
  RG_STAMP(3);   // band statistics

  // ---- decisions (float32) from the tile: one wave per (frame, 64-bin word), the ballot IS the word ----------------------
  //   passes     <=>  P > T                                   (False for a NaN power, like the reference's compare)
  //   ambiguous  <=>  (P - T)^2 <= (2 d2_t + cb)(P + T)       (implied by | |2X| - T | <= 2 delta_t + eta T)
  // group q = (word w, third of the rows): the band constants of a lane do not depend on the row
  {
    const int rpp = (T + 2) / 3;
    for (int q = wave; q < 27; q += RG_WAVES) {
      const int w = q % 9, r0 = (q / 9) * rpp, r1 = min(T, r0 + rpp);
      const int f = 64 * w + lane;
      const bool on = f < 513;
      const float Tv = on ? s_t2[f] : 3.0e38f;
      const float cbv = on ? s_cb[f] : 0.f;
      bool ambacc = false;
#pragma unroll 4
      for (int r = r0; r < r1; ++r) {
        const float P = on ? tile[r * RG_PP + f] : 0.f;
        const float wd = 2.f * s_d2[r] + cbv;
        const float nd = Tv - P;
        ambacc = ambacc || (wd > 0.f && nd * nd <= wd * (P + Tv));   // "never" / "always" constants: nd^2 overflows -> false
        const unsigned long long word = __ballot(on && P > Tv);
        if (lane == 0) wb[(nt + r) * RG_WP + 1 + w] = word;
      }
      if (on && ambacc) s_flag[f] = 1;
    }
  }
  __syncthreads();
  RG_STAMP(4);   // decisions

  // ---- exact re-evaluation of the ambiguous bands ---------------------------------------------------------------------
  for (int f = tid; f < 513; f += RG_THREADS) {
    if (s_flag[f]) {
      const unsigned idx = atomicAdd(&s_misc[0], 1u);
      s_list[idx] = (unsigned short)f;
    }
  }
  __syncthreads();
  const int n_amb = (int)s_misc[0];
  // float32-exact sample types (float32, int16): the samples are staged in LDS and 16 lanes share one (frame, band) sum
  const bool lds_x = A.view.dtype == 0 || A.view.dtype == 2;
  if (n_amb > 0) {
    if (tid == 0 && A.n_exact) atomicAdd(A.n_exact, (unsigned)n_amb);
    // float64 statistics of one band with the reference's formulas (k_row_decide), lane = frame; patches the bit rows
    auto band_exact = [&](int f, double Pe, bool on) {
      const double eps = 2.220446049250313e-16;
      double m = Pe;
      for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off);
        m = (m != m || o != o) ? (double)NAN : fmax(m, o);
      }
      const double mdb = cell_db(m, A.mag_scale);
      double d = 0.0, dsq = 0.0;
      if (on) {
        d = cell_db(Pe, A.mag_scale) - mdb;
        d = (d != d) ? d : fmax(d, -A.top_db);
        dsq = d * d;
      }
      for (int off = 32; off > 0; off >>= 1) {
        d += __shfl_xor(d, off);
        dsq += __shfl_xor(dsq, off);
      }
      const double Tn = (double)T;
      double var = (dsq - d * d / Tn) / (Tn - (double)A.ddof);
      if (var < 0.0) var = 0.0;
      const double th = (mdb + d / Tn) + sqrt(var) * A.n_std;
      const double fl = mdb - A.top_db;
      double t2e;
      if (th != th || fl != fl) {
        t2e = 1e300;
      } else if (fl > th || 20.0 * log10(eps) > th) {
        t2e = -1.0;
      } else {
        const double tm = (exp10(th / 20.0) - eps) / A.mag_scale;
        t2e = tm > 0.0 ? tm * tm : 0.0;
      }
      if (on) {   // (bands that share a 64-bit word may be patched by different waves at once: LDS atomics)
        unsigned long long* wp = &wb[(nt + lane) * RG_WP + 1 + (f >> 6)];
        const unsigned long long bit = 1ull << (f & 63);
        if (Pe > t2e) atomicOr(wp, bit);
        else atomicAnd(wp, ~bit);
      }
    };
    if (lds_x) {
      double* w64s = reinterpret_cast<double*>(region + RG_EX_W64);
      cx<double>* tws = reinterpret_cast<cx<double>*>(region + RG_EX_TW);
      double* exb = reinterpret_cast<double*>(region + RG_EX_PW);
      (void)stage_span();
      for (int i = tid; i < 1024; i += RG_THREADS) w64s[i] = A.win64[i];
      for (int i = tid; i < 1024; i += RG_THREADS) {     // w_1024^i for every i: no sign logic in the inner loop
        cx<double> w = A.tw64[i & 511];
        if (i & 512) { w.x = -w.x; w.y = -w.y; }
        tws[i] = w;
      }
      __syncthreads();
      const int p = tid & 15;                       // 16 lanes per frame, lane p sums the terms m = p + 16 i
      const double* wq = w64s + p;
      for (int k0 = 0; k0 < n_amb; k0 += RG_EX_BANDS) {
        const int nbk = min(RG_EX_BANDS, n_amb - k0);
        for (int k = 0; k < nbk; ++k) {
          const int f = (int)s_list[k0 + k];
          const int dj = (16 * f) & 1023;
          // radix-4 split of the 1024-term sum: with y_k[m] = (x w)[m + 256 k], m < 256, and w^(256 f) = (-i)^f
          //   X[f] = sum_m ( (y0 + s y2) + (-i)^f (y1 + s y3) ) w^(f m),  s = (-1)^f
          // -- a quarter of the twiddle reads (the LDS pipe bounds this loop) and 4 instead of 8 multiply-adds per 4 terms
          const double sg2 = (f & 1) ? -1.0 : 1.0;
          const double cr = (f & 1) ? 0.0 : ((f & 2) ? -1.0 : 1.0);      // (-i)^f = cr + i ci
          const double ci = (f & 1) ? ((f & 2) ? 1.0 : -1.0) : 0.0;
#pragma unroll 1
          for (int hq = 0; hq < RG_FRAMES / (RG_THREADS / 16); ++hq) {
            const int tq = (tid >> 4) + (RG_THREADS / 16) * hq;
            double re = 0.0, im = 0.0;
            if (tq < T) {
              const float* xq = xs + tq * XP + p;
              int j = (f * p) & 1023;
#pragma unroll 4
              for (int i = 0; i < 16; ++i) {
                const double y0 = (double)xq[16 * i] * wq[16 * i];
                const double y1 = (double)xq[XP + 16 * i] * wq[256 + 16 * i];
                const double y2 = (double)xq[2 * XP + 16 * i] * wq[512 + 16 * i];
                const double y3 = (double)xq[3 * XP + 16 * i] * wq[768 + 16 * i];
                const double a = fma(sg2, y2, y0), b = fma(sg2, y3, y1);
                const double Cr = fma(cr, b, a), Ci = ci * b;
                const cx<double> w = tws[j];
                re = fma(Cr, w.x, fma(-Ci, w.y, re));
                im = fma(Cr, w.y, fma(Ci, w.x, im));
                j = (j + dj) & 1023;
              }
            }
            re += __shfl_xor(re, 1); im += __shfl_xor(im, 1);
            re += __shfl_xor(re, 2); im += __shfl_xor(im, 2);
            re += __shfl_xor(re, 4); im += __shfl_xor(im, 4);
            re += __shfl_xor(re, 8); im += __shfl_xor(im, 8);
            if (p == 0 && tq < T) exb[k * 64 + tq] = re * re + im * im;
          }
        }
        __syncthreads();
        for (int k = wave; k < nbk; k += RG_WAVES) {
          const bool on = lane < T;
          band_exact((int)s_list[k0 + k], on ? exb[k * 64 + lane] : 0.0, on);
        }
        __syncthreads();
      }
    } else {
      // other sample types (float64, int32: not exact in float32): one wave per (frame, band) sum, samples from memory
      for (int k = 0; k < n_amb; ++k) {
        const int f = (int)s_list[k];
#pragma unroll 1
        for (int q = 0; q < 64 / RG_WAVES; ++q) {
          const int tq = (64 / RG_WAVES) * wave + q;
          if (tq < T) {
            const double Pe = rg_exact_power(A, row, tq, f, lane);
            if (lane == 0) s_ex[tq] = Pe;
          }
        }
        __syncthreads();
        if (wave == 0) {
          const bool on = lane < T;
          band_exact(f, on ? s_ex[lane] : 0.0, on);
        }
        __syncthreads();
      }
    }
  }
  if (A.bits_out) {
    for (int i = tid; i < T * 9; i += RG_THREADS) {
      const int r = i / 9, w = i - 9 * r;
      A.bits_out[((int64_t)blockIdx.x * T + r) * 9 + w] = wb[(nt + r) * RG_WP + 1 + w];
    }
  }
  RG_STAMP(5);   // exact re-evaluation (+ stage tap)

  // ---- pass 2: the row's spectra again (they stay in registers from here to the inverse transforms) ----------------------
  if (!(n_amb > 0 && lds_x)) (void)stage_span();   // (the exact phase has staged the samples already)
  __syncthreads();
  cf v[RG_QUADS][32];
#pragma unroll
  for (int q = 0; q < RG_QUADS; ++q) (void)forward(v[q], frame_of(q));
  __syncthreads();   // all gathers and forward exchanges done: the region becomes the count / K tile
  RG_STAMP(6);   // pass 2: span, gather, forward transform, split

  // ---- smoothing, exact integer arithmetic (k_smooth_bits2's): K[t][f] = sum_a sum_b vf[a] vt[b] bit[t+b][f+a] ------------
  unsigned short* cfp = reinterpret_cast<unsigned short*>(region);   // [rows + 2][RG_FP] counts along f, then K in place
  const int rows = T + 2 * nt;
  RG_STAMP(14);
  {
    // rows outside the spectrogram are zero: [0, nt) in front, [nt + T, nt + T + nt + 2) behind (two more than the tile:
    // the walk along t reads them)
    unsigned long long* z0 = reinterpret_cast<unsigned long long*>(cfp);
    unsigned long long* z1 = reinterpret_cast<unsigned long long*>(cfp + (size_t)(nt + T) * RG_FP);
    for (int i = tid; i < nt * (RG_FP / 4); i += RG_THREADS) z0[i] = 0ull;
    for (int i = tid; i < (nt + 2) * (RG_FP / 4); i += RG_THREADS) z1[i] = 0ull;
  }
  RG_STAMP(12);
  {
    // one thread = one (frame, 64-bin word), 63 x 8 = 504 tasks: c[f] = sum_a (nf + 1 - |a|) bit[f + a] by the recurrence
    // c += R - L of k_smooth_bits2 (R / L = set bits in the nf + 1 bins right of / left of and including f).  The three bit
    // streams the recurrence reads -- bit(f + 1), bit(f + nf + 2), bit(f - nf) -- are the 128-bit window shifted ONCE per
    // task (64-bit shifts are quarter rate); word 7's thread also leaves the count of bin 512, the recurrence's next value.
    const unsigned long long m1 = (1ull << (nf + 1)) - 1ull;
    for (int task = tid; task < T * 8; task += RG_THREADS) {
      const int r = nt + (task >> 3), w = task & 7;
      const unsigned long long* rb = wb + (size_t)r * RG_WP + 1 + w;
      const unsigned long long lo = funnel_r(rb[-1], rb[0], 64 - nf);      // bins 64 w - nf ...
      const unsigned long long hi = funnel_r(rb[0], rb[1], 64 - nf);       // bins 64 w - nf + 64 ...
      int cnt = 0;
      for (int i = 0; i <= 2 * nf; ++i) cnt += (nf + 1 - (i < nf ? nf - i : i - nf)) * (int)((lo >> i) & 1ull);
      int R = __popcll((lo >> (nf + 1)) & m1);
      int L = __popcll(lo & m1);
      const unsigned long long sA = funnel_r(lo, hi, nf + 1);              // bit i = bit(f + 1),      f = 64 w + i
      const unsigned long long sB = funnel_r(lo, hi, 2 * nf + 2);          // bit i = bit(f + nf + 2)  (2 nf + 2 <= 62)
      const unsigned a0 = (unsigned)sA, a1 = (unsigned)(sA >> 32), b0 = (unsigned)sB, b1 = (unsigned)(sB >> 32);
      const unsigned c0 = (unsigned)lo, c1 = (unsigned)(lo >> 32);         // bit i = bit(f - nf)
      unsigned short* out = cfp + (size_t)r * RG_FP + 72 * w;   // column(f) = f + 4 (f / 32)
#pragma unroll 1
      for (int i4 = 0; i4 < 64; i4 += 4) {
        const unsigned av = i4 & 32 ? a1 : a0, bv = i4 & 32 ? b1 : b0, cv = i4 & 32 ? c1 : c0;
        unsigned v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = (i4 & 31) + e;
          v4[e] = (unsigned)cnt;
          cnt += R - L;
          const int bA = (int)((av >> i) & 1u);
          R += (int)((bv >> i) & 1u) - bA;
          L += bA - (int)((cv >> i) & 1u);
        }
        const unsigned long long pk = (unsigned long long)(v4[0] | (v4[1] << 16)) | ((unsigned long long)(v4[2] | (v4[3] << 16)) << 32);
        *reinterpret_cast<unsigned long long*>(out + i4 + ((i4 >> 5) << 2)) = pk;
      }
      if (w == 7) out[72] = (unsigned short)cnt;    // bin 512 = column 576
    }
  }
  RG_STAMP(13);
  __syncthreads();
  RG_STAMP(7);   // smoothing along f
  if (tid < 512) {
    // along t, one column per thread, IN PLACE: output row i overwrites count row i (dead once it has been read)
    const int f = tid;    // columns 0..511
    unsigned short* col = cfp + f + ((f >> 5) << 2);
    int cnt = 0, R = 0, L = 0;
    for (int b = -nt; b <= nt + 1; ++b) {
      const int x = (int)col[(size_t)(nt + b) * RG_FP];
      if (b <= nt) cnt += (nt + 1 - (b < 0 ? -b : b)) * x;
      if (b >= 1) R += x;
      if (b <= 0) L += x;
    }
    const unsigned short* pa2 = col + (size_t)(2 * nt + 2) * RG_FP;
    const unsigned short* pb2 = col + (size_t)(nt + 1) * RG_FP;
    unsigned short* pc2 = col;
#pragma unroll 4
    for (int i = 0; i < T; ++i) {
      const int xa = (int)*pa2, xb = (int)*pb2, xc = (int)*pc2;
      *pc2 = (unsigned short)cnt;
      pa2 += RG_FP; pb2 += RG_FP; pc2 += RG_FP;
      cnt += R - L;
      R += xa - xb;
      L += xb - xc;
    }
  }
  if (wave == 0) {
    // column 512 (bin 512): lane = frame, direct sum over the 2 nt + 1 rows; every lane reads before any lane writes
    const unsigned short* col = cfp + 576;
    int acc = 0;
    if (lane < T)
      for (int b = -nt; b <= nt; ++b) acc += (nt + 1 - (b < 0 ? -b : b)) * (int)col[(size_t)(lane + nt + b) * RG_FP];
    wave_lds_sync();
    if (lane < T) cfp[(size_t)lane * RG_FP + 576] = (unsigned short)acc;
  }
  __syncthreads();
  RG_STAMP(8);   // smoothing along t
  if (A.mask_out) {
    // four bins per thread and store (the K tile keeps 4 adjacent bins in one 8-byte word: column(f) = f + 4 (f / 32)); bin 512
    // of every row by the first T threads
    float* mo = A.mask_out + (int64_t)blockIdx.x * T * G.FS;
    for (int i = tid; i < T * 128; i += RG_THREADS) {
      const int r = i >> 7, f = (i & 127) * 4;
      const uint2 k4 = *reinterpret_cast<const uint2*>(&cfp[(size_t)r * RG_FP + f + ((f >> 5) << 2)]);
      *reinterpret_cast<float4*>(&mo[(int64_t)r * G.FS + f]) =
          make_float4((float)(k4.x & 0xffffu) * A.inv_ktot, (float)(k4.x >> 16) * A.inv_ktot, (float)(k4.y & 0xffffu) * A.inv_ktot,
                      (float)(k4.y >> 16) * A.inv_ktot);
    }
    if (tid < T) mo[(int64_t)tid * G.FS + 512] = (float)cfp[(size_t)tid * RG_FP + 576] * A.inv_ktot;
  }

  // ---- x mask -> merge (second half of pair_mask), in place, both quads ------------------------------------------------
#pragma unroll
  for (int q = 0; q < RG_QUADS; ++q) {
    const int t = frame_of(q);
    const bool fvalid = t < T;
    cf* vq = v[q];
    const unsigned short* krow = cfp + (size_t)(fvalid ? t : 0) * RG_FP;
    const unsigned short* k_lo = krow + c;                 // lanes c >= 1: entry e < 16 = bin c + 32 e -> column c + 36 e
    const unsigned short* k_hi = krow + (32 - c);          // entry e >= 16 = bin (32 - c) + 32 (e - 16)
    auto mval = [&](int e, float scale) -> float {
      const int b0 = bin_of_entry(0, e);                   // lane 0 (compile-time)
      const unsigned short kv = c == 0 ? krow[b0 + ((b0 >> 5) << 2)] : (e < 16 ? k_lo[36 * e] : k_hi[36 * (e - 16)]);
      return (float)kv * scale;
    };
    const float k512 = (float)krow[512 + 64] * A.kscale;
    const float ks = A.kscale * 0.25f;
    // (the split's twiddles again from LDS -- not the forward transform's copies kept alive across the smoothing phases)
    cf wlo = s_tw1024[c];
    asm volatile("" : "+v"(wlo.x), "+v"(wlo.y));
    cf whi = wlo;
    {
      const cf w16 = s_tw1024[16];
      if (l0) whi = {-w16.y, w16.x};  // i * w_1024^16
    }
    {
      // slot 0: lanes >= 1 merge the pair (v[0], v[31]); lane 0: bins 0 / 512 from v[0], bin 256 = v[31] scaled
      const cf r0 = vq[0], r31 = vq[31];
      cf xa = r0, xb = r31;
      merge_pair(xa, xb, wlo, mval(0, ks), mval(31, ks));
      const float y0 = (r0.x + r0.y) * mval(0, A.kscale);
      const float yN = (r0.x - r0.y) * k512;
      const cf z0 = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
      const float m8 = mval(31, A.kscale);  // entry 31 of lane 0 = bin 256
      const cf z8 = {r31.x * m8, r31.y * m8};
      vq[0] = {l0 ? z0.x : xa.x, l0 ? z0.y : xa.y};
      vq[31] = {l0 ? z8.x : xb.x, l0 ? z8.y : xb.y};
    }
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
      merge_pair(vq[sl], vq[31 - sl], w, mval(sl, ks), mval(31 - sl, ks));
    }
    rg_lane0_from_entries(vq, l0);   // back to the transform's register order
    if (!fvalid) {
#pragma unroll
      for (int i = 0; i < 32; ++i) vq[i] = {0.f, 0.f};
    }
  }
  __syncthreads();   // every lane has read its K values: the region is reused by the inverse transforms
  RG_STAMP(9);   // mask + merge

  // ---- per quad: inverse transform, synthesis window, wave-private overlap-add (k_apply_fast<LEAN>), then the cross-wave
  // combine of the 32 frames' hops.  Quad 0 = frames 0..31 -> hops 0..34; hops 32..34 also receive frames 32..34 of quad 1:
  // their partial sums wait in `carry` (the dead compare-constant tables).  Quad 1 = frames 32..63 -> hops 32..66.
  float* carry = s_t2;    // [3][256]
  static_assert(2 * T2_FLOATS >= 3 * 256, "carry buffer");
  // (the lane index from v_mbcnt, not from threadIdx.x: the workitem id need not survive -- in a spill slot -- from the
  // entry block to the epilogue of a kernel at its 128-VGPR cap)
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int s4 = lane_e * 4;
  const int64_t h_begin = (A.om.p0 + G.padL) / 256, h_end = (A.om.p1 - 1 + G.padL) / 256 + 1;
#pragma unroll
  for (int q = 0; q < RG_QUADS; ++q) {
    cf* vq = v[q];
    {
      int zi = 0, ci = c;
      asm volatile("" : "+v"(zi), "+v"(ci));
      fft512_inv_half(vq, slices + wave * WAVE_CX_H + frame_base_h(g) + zi, tw512 + zi, ci);
    }
    float* acc = reinterpret_cast<float*>(slices + wave * WAVE_CX_H);
    {
      const float2* wsrc2 = reinterpret_cast<const float2*>(swin + 2 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (first contribution to hop g + j -- j == 0, or frame g == 3 -- is a plain store; sel_s: fastpath.hpp)
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int r = 8 * j + rr;
          float2* dst = reinterpret_cast<float2*>(acc + (g + j) * HPITCH + 2 * c + 32 * rr);
          const float2 ws = wsrc2[16 * r];
          float2 nw = {vq[r].x * ws.x, vq[r].y * ws.y};
          if (j != 0) {
            const float2 old = *dst;
            nw.x = sel_s(OLA_KEEP, nw.x + old.x, nw.x);
            nw.y = sel_s(OLA_KEEP, nw.y + old.y, nw.y);
          }
          *dst = nw;
        }
        wave_lds_sync();
      }
    }
    // (1 / envelope of this lane's four sample phases: loaded HERE, behind the transforms -- four registers that do not
    // stay live across the inverse transform of a kernel at its 128-VGPR cap)
    const float4 n4 = *reinterpret_cast<const float4*>(&A.invn[s4]);
    __syncthreads();
    // combine: local hop lj = 0..34 of this quad (ext hop jj = 32 q + lj): wave lj / 4 (hop lj % 4) + wave lj / 4 - 1 (hop lj % 4 + 4)
    const float* fr = reinterpret_cast<const float*>(slices);
    for (int lj = wave; lj < 4 * RG_WAVES + 3; lj += RG_WAVES) {
      const int jj = 4 * RG_WAVES * q + lj;
      float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const int wh = lj >> 2, lh = lj & 3;
      if (wh >= 1 && lh <= 2) a4 = *reinterpret_cast<const float4*>(&fr[(wh - 1) * WAVE_CX_H * 2 + (lh + 4) * HPITCH + s4]);
      if (wh < RG_WAVES) {
        const float4 f4 = *reinterpret_cast<const float4*>(&fr[wh * WAVE_CX_H * 2 + lh * HPITCH + s4]);
        a4.x += f4.x; a4.y += f4.y; a4.z += f4.z; a4.w += f4.w;
      }
      if (RG_QUADS == 2 && q == 0 && lj >= 4 * RG_WAVES) {          // frames 32.. have not contributed yet
        *reinterpret_cast<float4*>(&carry[(lj - 4 * RG_WAVES) * 256 + s4]) = a4;
        continue;
      }
      if (RG_QUADS == 2 && q == 1 && lj < 3) {
        const float4 f4 = *reinterpret_cast<const float4*>(&carry[lj * 256 + s4]);
        a4.x += f4.x; a4.y += f4.y; a4.z += f4.z; a4.w += f4.w;
      }
      if (jj < (int)h_begin || jj >= (int)h_end) continue;
      bool all_valid = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ti = jj - k;
        if (ti < 0 || ti >= T) all_valid = false;
      }
      if (all_valid) {
        a4.x *= n4.x; a4.y *= n4.y; a4.z *= n4.z; a4.w *= n4.w;
      } else {
        float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ti = jj - k;
          if (ti >= 0 && ti < T) {
            const float4 w4 = *reinterpret_cast<const float4*>(&A.wsq[256 * k + s4]);
            nrm.x += w4.x; nrm.y += w4.y; nrm.z += w4.z; nrm.w += w4.w;
          }
        }
        a4.x /= (nrm.x > 1e-10f ? nrm.x : 1.f);
        a4.y /= (nrm.y > 1e-10f ? nrm.y : 1.f);
        a4.z /= (nrm.z > 1e-10f ? nrm.z : 1.f);
        a4.w /= (nrm.w > 1e-10f ? nrm.w : 1.f);
      }
      const int64_t pbs = (int64_t)jj * 256 - G.padL;
      const int64_t gi0 = pbs - A.om.p0;
      if (A.om.dtype == 0 && pbs >= A.om.p0 && pbs + 256 <= A.om.p1 && pbs + 256 <= G.Lout && gi0 >= A.om.g_lo &&
          gi0 + 256 <= A.om.g_hi) {
        float* dst = (float*)A.om.out + (row * A.om.stride + gi0 - A.om.g0 + s4);
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = a4;
          continue;
        }
      }
      const float vals[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t p = (int64_t)jj * 256 + s4 + e - G.padL;
        if (p < A.om.p0 || p >= A.om.p1) continue;
        const int64_t gi = p - A.om.p0;
        if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
        store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? vals[e] : 0.f);
      }
    }
    __syncthreads();   // the accumulators are consumed: the next quad's inverse exchange may overwrite them
    RG_STAMP(10 + q);  // inverse transform, window, overlap-add, combine, store
  }
}

}  // namespace fast
}  // namespace sg
