// Fast path for the default STFT geometry (n_fft = win_length = 1024, hop = 256), float32.
//
// FFT core: one wavefront transforms FOUR frames at once.  A 1024-sample real frame is packed
// as 512 complex points z[m] = x[2m] + i x[2m+1]; lane (g, c) = (lane >> 4, lane & 15) holds the 32
// points z[c + 16 r], r = 0..31, of frame g in registers.  512 = 32 x 16:
//
//   forward:  DFT32 over r (registers)  ->  twiddle w512^(c k1)  ->  ONE exchange through LDS
//             ->  DFT16 over c (registers)   =>  lane c' holds rows k1 = c' and k1 = 32 - c'
//             (lane 0: rows 0 and 16) of Zc[k1 + 32 k2], k2 = 0..15.
//   inverse:  the mirror image (DFT16 over k2 -> exchange -> twiddle -> DFT32 over k1).
//
// Rows are dealt so that the two bins k and 512-k that the real-FFT split/merge couples
// always sit in the SAME lane: the mask multiply needs no cross-lane traffic and no LDS.
// The exchange layout is XOR-swizzled (16-byte chunk index ^ ((row >> 1) & 7)) and odd frames
// are skewed by 128 B so that every ds_write_b64 / ds_read_b128 is bank-conflict free.
#pragma once
#include "kernels.hpp"
#include "thresh.hpp"

namespace sg {

// (round 5) The float32 compare constants of a decision kernel into LDS, COUNT bands by NTHR threads.  The common unit (no
// live floor, no non-finite sample: need == 0) takes its constants straight from T2: every load of a thread is issued before
// the first conversion -- the rolled form `for (i = tid; ...) s[i] = f(T2[i])` waits for each load in turn (the prologue-load
// story of stage_tables): five dependent round trips per 8-frame tile at n_fft = 2048, nine / seventeen at 4096 / 8192.
// `slow(i)`: the exact constant of band i for the other units.
template <int NTHR, int COUNT, typename SLOW>
__device__ __forceinline__ void stage_t2_plain(float* s_t2, const double* __restrict__ T2, int need, double scale, int tid,
                                               SLOW slow) {
  if (need == 0) {   // (workgroup-uniform)
    constexpr int K = (COUNT + NTHR - 1) / NTHR;
    double t[K];
#pragma unroll
    for (int k = 0; k < K; ++k) t[k] = T2[min(tid + k * NTHR, COUNT - 1)];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = tid + k * NTHR;
      if ((k + 1) * NTHR <= COUNT || i < COUNT) s_t2[i] = t2_to_f32(t[k], scale);
    }
  } else {
    for (int i = tid; i < COUNT; i += NTHR) s_t2[i] = t2_to_f32(slow(i), scale);
  }
}

// (round 5) The kernel's own argument struct re-read from the kernel-argument segment through an OPAQUE pointer, for the COLD
// parts of a long kernel (the exact re-evaluation of an ambiguous cell needs the view's dozen fields, the float64 tables and
// the threshold pointers).  Taken from the by-value argument they stay live in SGPRs from the kernel's entry through every
// phase (k_decide_fast512 / 2048 / 256: 83 - 92 spilled SGPRs); read through this pointer they are loaded where they are
// used.  ARGS must be the kernel's FIRST (only) parameter.  (k_gate_onepass does the same per field, onepass.hpp.)
template <typename ARGS>
__device__ __forceinline__ const ARGS* late_args() {
  const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return reinterpret_cast<const ARGS*>((const char*)kp);
}

namespace fast {

constexpr int FN = 512;           // complex points per frame
constexpr int FPITCH = FN + 16;    // LDS complex slots per frame slice: 512 + a 128-byte skew
constexpr int WAVE_CX = 4 * FPITCH;  // LDS complex slots per wave
constexpr int FSK = 528;          // mask row pitch (entries) of the permuted K layout

typedef cx<float> cf;

// Prologue loads of the register-FFT kernels: twiddle table, window table and the tile's sample span go to LDS.  Written
// as rolled loops (load, store, next index) the compiler keeps them rolled and waits for every load before its store --
// s_waitcnt vmcnt(0) per iteration: eight dependent memory round trips per tile in k_mag_fast / k_apply_fast /
// k_decide_fast, whose tiles live 13 - 30 us (round 4, ISA of those loops).  Here every load of a group is issued before
// the first store; indices are CLAMPED, not predicated (a predicated load is a branch and a copy of its result that
// waits for all loads in flight).
template <int NTHR, int WIN4>   // WIN4: float4 entries of the window table (0: the kernel keeps no window in LDS)
__device__ __forceinline__ void stage_tables(cf* tw512, const cf* __restrict__ tw_src, float* swin,
                                             const float* __restrict__ win_src, int tid) {
  constexpr bool WIN = WIN4 > 0;
  constexpr int K = (FN + NTHR - 1) / NTHR, KW = WIN ? (WIN4 + NTHR - 1) / NTHR : 1;
  cf t[K];
  float4 w[KW];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = min(tid + k * NTHR, FN - 1);
    t[k] = tw_src[(i >> 4) * (i & 15)];
  }
  if constexpr (WIN) {
#pragma unroll
    for (int k = 0; k < KW; ++k) w[k] = reinterpret_cast<const float4*>(win_src)[min(tid + k * NTHR, WIN4 - 1)];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = tid + k * NTHR;
    if ((k + 1) * NTHR <= FN || i < FN) tw512[i] = t[k];
  }
  if constexpr (WIN) {
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int i = tid + k * NTHR;
      if ((k + 1) * NTHR <= WIN4 || i < WIN4) reinterpret_cast<float4*>(swin)[i] = w[k];
    }
  }
}
// SPAN contiguous float32 samples at sp (16-byte aligned) -> xs, rows of ROW samples at a pitch of XPITCH floats
// (MX: also return the largest |sample| this thread staged, as a bit pattern with the sign cleared -- the in-kernel floor test)
template <int NTHR, int SPAN, int XPITCH, int ROW = 256, bool MX = false>
__device__ __forceinline__ unsigned stage_span_vec(float* xs, const float* __restrict__ sp, int tid) {
  static_assert(SPAN % 4 == 0, "16-byte loads");
  constexpr int N4 = SPAN / 4, NQ = (N4 + NTHR - 1) / NTHR;
  float4 q[NQ];
  unsigned mi = 0u;
#pragma unroll
  for (int k = 0; k < NQ; ++k) q[k] = reinterpret_cast<const float4*>(sp)[min(tid + k * NTHR, N4 - 1)];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int e = 4 * (tid + k * NTHR);
    if ((k + 1) * NTHR <= N4 || e < SPAN) *reinterpret_cast<float4*>(&xs[(e / ROW) * XPITCH + (e % ROW)]) = q[k];
    if constexpr (MX) {
      auto ab = [](float x) -> unsigned { return __float_as_uint(x) & 0x7fffffffu; };
      mi = max(max(mi, max(ab(q[k].x), ab(q[k].y))), max(ab(q[k].z), ab(q[k].w)));
    }
  }
  return mi;
}

// The same from int16 samples (8-byte aligned): four samples per load, converted on the way into LDS.  Integer
// recordings on the float64 pipeline stage their decision transforms through this instead of 19 checked per-sample loads
// per thread (k_decide_fast: 181 -> 113 us of a ten-minute call).
template <int NTHR, int SPAN, int XPITCH, int ROW = 256>
__device__ __forceinline__ void stage_span_vec_i16(float* xs, const int16_t* __restrict__ sp, int tid) {
  static_assert(SPAN % 4 == 0, "8-byte loads");
  constexpr int N4 = SPAN / 4, NQ = (N4 + NTHR - 1) / NTHR;
  uint2 q[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) q[k] = reinterpret_cast<const uint2*>(sp)[min(tid + k * NTHR, N4 - 1)];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int e = 4 * (tid + k * NTHR);
    const float4 f = {(float)(short)(q[k].x & 0xffffu), (float)((int)q[k].x >> 16), (float)(short)(q[k].y & 0xffffu),
                      (float)((int)q[k].y >> 16)};
    if ((k + 1) * NTHR <= N4 || e < SPAN) *reinterpret_cast<float4*>(&xs[(e / ROW) * XPITCH + (e % ROW)]) = f;
  }
}

// cos(2 pi k / 32), k = 0..8
__device__ constexpr float C32[9] = {1.0f,
                                     0.98078528040323044913f,
                                     0.92387953251128675613f,
                                     0.83146961230254523708f,
                                     0.70710678118654752440f,
                                     0.55557023301960222474f,
                                     0.38268343236508977173f,
                                     0.19509032201612826785f,
                                     0.0f};

// w_R^k = exp(-2 pi i k / R) for R in {8, 16, 32}, 0 <= k < R/2  (compile-time after unrolling)
template <int R>
__device__ __forceinline__ constexpr float twc(int k) {
  int j = k * (32 / R);  // index in 32nds of a turn, 0..15
  return j <= 8 ? C32[j] : -C32[16 - j];
}
template <int R>
__device__ __forceinline__ constexpr float tws(int k) {  // sin(2 pi k / R) >= 0 for k < R/2
  int j = k * (32 / R);
  return j <= 8 ? C32[8 - j] : C32[j - 8];
}

// ---------------------------------------------------------------------------------------------------------------
// Packed float32 butterflies (SG_PK_FFT, v_pk_*_f32 through inline asm: the compiler's own packing shuffles register
// pairs).  A complex value is one aligned VGPR pair (re, im); op_sel / neg modifiers do the swaps and sign flips of a
// complex product inside the instruction, twiddle constants ride in SGPR pairs (cos, sin):
//     p = e + w o      2 x v_pk_fma_f32        (scalar form: 4 x v_fma_f32)
//     q = 2 e - p      1 x v_pk_fma_f32        (2)
//     e +- o, e +- i o 1 x v_pk_add_f32 each   (2)
//     a * w            v_pk_mul_f32 + v_pk_fma_f32   (4)
// Every half of a packed operation is the IEEE operation the scalar form performs, in the same order: the results are
// bit-identical (tools/ubench/pk_check.hip).  A packed instruction occupies the VALU for twice as long as a scalar one
// (profiles/r03_valu_rate2.txt: the float32 rate of the pipe is the same either way) -- what is saved is ISSUE SLOTS: a
// wave issues one instruction per ~4 cycles whatever it is, and the one-pass gate is bound by that.
// ---------------------------------------------------------------------------------------------------------------
#ifndef SG_PK_FFT
#define SG_PK_FFT 0   // measured: no gain (profiles/r03_packed_fft_ab.txt); needs the packed-fp32-ops target feature (tools/ab_build.sh)
#endif
typedef float pf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pf2 pk_of(cf a) { return pf2{a.x, a.y}; }
__device__ __forceinline__ cf pk_to(pf2 a) { return cf{a.x, a.y}; }
__device__ __forceinline__ pf2 pk_add(pf2 a, pf2 b) {
  pf2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ pf2 pk_sub(pf2 a, pf2 b) {
  pf2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ pf2 pk_mul(pf2 a, pf2 b) {
  pf2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a + (b.y, -b.x)  /  a + (-b.y, b.x)
__device__ __forceinline__ pf2 pk_add_mi(pf2 a, pf2 b) {   // a + (-i) b
  pf2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ pf2 pk_add_pi(pf2 a, pf2 b) {   // a + (+i) b
  pf2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// butterfly with the twiddle folded in: p = a + w b, q = 2 a - p;  w = c - i s (forward), c + i s (inverse); cs = (c, s)
// uniform (SGPR pair).  p.x = fma(b.x, c, fma(b.y, +-s, a.x)), p.y = fma(b.y, c, fma(-+b.x, s, a.y)): as dft_reg's scalar form
template <bool INV>
__device__ __forceinline__ void pk_bfly_tw(pf2 a, pf2 b, pf2 cs, pf2& p, pf2& q) {
  pf2 t;
  if (INV) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(t) : "v"(b), "s"(cs), "v"(a));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(t) : "v"(b), "s"(cs), "v"(a));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(p) : "v"(b), "s"(cs), "v"(t));
  asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(p));
}
// a * w (CONJ: a * conj w), w a VGPR pair
template <bool CONJ>
__device__ __forceinline__ pf2 pk_cmul(pf2 a, pf2 w) {
  pf2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  if (CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}

// In-register DFT of R points on packed values, natural order in and out (decimation in time): dft_reg below, packed.
template <int R, bool INV>
__device__ __forceinline__ void dft_reg_pk(pf2* v) {
  if constexpr (R == 2) {
    const pf2 a = v[0], b = v[1];
    v[0] = pk_add(a, b);
    v[1] = pk_sub(a, b);
  } else if constexpr (R == 4) {
    const pf2 s02 = pk_add(v[0], v[2]), d02 = pk_sub(v[0], v[2]);
    const pf2 s13 = pk_add(v[1], v[3]), d13 = pk_sub(v[1], v[3]);
    v[0] = pk_add(s02, s13);
    v[2] = pk_sub(s02, s13);
    v[1] = INV ? pk_add_pi(d02, d13) : pk_add_mi(d02, d13);   // d02 + rot90(d13)
    v[3] = INV ? pk_add_mi(d02, d13) : pk_add_pi(d02, d13);   // d02 - rot90(d13)
  } else {
    pf2 e[R / 2], o[R / 2];
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      e[k] = v[2 * k];
      o[k] = v[2 * k + 1];
    }
    dft_reg_pk<R / 2, INV>(e);
    dft_reg_pk<R / 2, INV>(o);
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      if (k == 0) {
        v[k] = pk_add(e[k], o[k]);
        v[k + R / 2] = pk_sub(e[k], o[k]);
      } else if (k == R / 4) {
        v[k] = INV ? pk_add_pi(e[k], o[k]) : pk_add_mi(e[k], o[k]);
        v[k + R / 2] = INV ? pk_add_mi(e[k], o[k]) : pk_add_pi(e[k], o[k]);
      } else {
        const pf2 cs = {twc<R>(k), tws<R>(k)};
        pk_bfly_tw<INV>(e[k], o[k], cs, v[k], v[k + R / 2]);
      }
    }
  }
}

template <bool INV>
__device__ __forceinline__ cf mul_tw(cf a, float c, float s) {
  // a * (c - i s) forward, a * (c + i s) inverse
  if (INV) return {a.x * c - a.y * s, a.y * c + a.x * s};
  return {a.x * c + a.y * s, a.y * c - a.x * s};
}

// In-register DFT of R points, natural order in and out (decimation in time).
template <int R, bool INV>
__device__ __forceinline__ void dft_reg(cf* v) {
  if constexpr (R == 2) {
    cf a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if constexpr (R == 4) {
    dft4<INV>(v);
  } else {
    cf e[R / 2], o[R / 2];
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      e[k] = v[2 * k];
      o[k] = v[2 * k + 1];
    }
    dft_reg<R / 2, INV>(e);
    dft_reg<R / 2, INV>(o);
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      if (k == 0 || k == R / 4) {
        const cf t = k == 0 ? o[k] : rot90<INV>(o[k]);
        v[k] = cadd(e[k], t);
        v[k + R / 2] = csub(e[k], t);
      } else {
        // butterfly with the twiddle folded into fused multiply-adds: p = e + w o (4 FMAs), q = e - w o = 2 e - p
        // (2 FMAs) -- 6 instructions instead of 4 (complex product) + 4 (add, subtract)
        const float c = twc<R>(k), s = INV ? -tws<R>(k) : tws<R>(k);   // w = c - i s (forward), c + i s (inverse)
        const cf a = e[k], b = o[k];
        cf p;
        p.x = fmaf(b.x, c, fmaf(b.y, s, a.x));
        p.y = fmaf(b.y, c, fmaf(-b.x, s, a.y));
        v[k] = p;
        v[k + R / 2] = {fmaf(2.0f, a.x, -p.x), fmaf(2.0f, a.y, -p.y)};
      }
    }
  }
}

// The three register stages of the 512-point transform, shared by every variant below (packed or scalar butterflies:
// the same arithmetic in every kernel that uses them, so the kernels agree to the bit).
//   forward: DFT32 over r, then the twiddle w_512^(c k1)
__device__ __forceinline__ void stage_dft32_tw_fwd(cf* v, const cf* tw512, int c) {
#if SG_PK_FFT
  pf2 p[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) p[r] = pk_of(v[r]);
  dft_reg_pk<32, false>(p);
  __builtin_amdgcn_sched_barrier(0);
  v[0] = pk_to(p[0]);
#pragma unroll
  for (int k1 = 1; k1 < 32; ++k1) v[k1] = pk_to(pk_cmul<false>(p[k1], pk_of(tw512[k1 * 16 + c])));
#else
  dft_reg<32, false>(v);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k1 = 1; k1 < 32; ++k1) v[k1] = cmul(v[k1], tw512[k1 * 16 + c]);
#endif
}
//   inverse: conjugate twiddle, then DFT32 over k1
__device__ __forceinline__ void stage_tw_dft32_inv(cf* v, const cf* tw512, int c) {
#if SG_PK_FFT
  pf2 p[32];
  p[0] = pk_of(v[0]);
#pragma unroll
  for (int k1 = 1; k1 < 32; ++k1) p[k1] = pk_cmul<true>(pk_of(v[k1]), pk_of(tw512[k1 * 16 + c]));
  __builtin_amdgcn_sched_barrier(0);
  dft_reg_pk<32, true>(p);
#pragma unroll
  for (int r = 0; r < 32; ++r) v[r] = pk_to(p[r]);
#else
#pragma unroll
  for (int k1 = 1; k1 < 32; ++k1) {
    cf w = tw512[k1 * 16 + c];
    w.y = -w.y;
    v[k1] = cmul(v[k1], w);
  }
  __builtin_amdgcn_sched_barrier(0);
  dft_reg<32, true>(v);
#endif
}
//   both directions: the two DFT16 over the lane's rows
template <bool INV>
__device__ __forceinline__ void stage_dft16x2(cf* v) {
#if SG_PK_FFT
  pf2 p[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) p[r] = pk_of(v[r]);
  dft_reg_pk<16, INV>(p);
  dft_reg_pk<16, INV>(p + 16);
#pragma unroll
  for (int r = 0; r < 32; ++r) v[r] = pk_to(p[r]);
#else
  dft_reg<16, INV>(v);
  dft_reg<16, INV>(v + 16);
#endif
}

// Per-lane select with the condition in an SGPR PAIR (v_cndmask_b32_e64): bit l of `m` set -> a, else b.  A v_cndmask_b32
// that reads VCC -- what the compiler emits for `c ? a : b` next to its compare -- holds the vector pipe of a gfx950 SIMD
// for ~20 cycles, the SGPR-pair form for 4 (tools/ubench/valu_classes.hip, profiles/r05_valu_classes.txt).  The
// wave-private overlap-add below selects "first contribution to this hop: store, else add" per float: 48 selects per wave
// and tile.  OLA_KEEP: lanes of frames g = 0..2 add to what is there (frame 3 is the first to touch hops 4..6).
constexpr unsigned long long OLA_KEEP = 0x0000ffffffffffffull;
__device__ __forceinline__ float sel_s(unsigned long long m, float a, float b) {
  float d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m));
  return d;
}
__device__ __forceinline__ unsigned sel_s(unsigned long long m, unsigned a, unsigned b) {
  unsigned d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m));
  return d;
}
__device__ __forceinline__ double sel_s(unsigned long long m, double a, double b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  unsigned lo, hi;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)ub), "v"((unsigned)ua), "s"(m));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((unsigned)(ub >> 32)), "v"((unsigned)(ua >> 32)), "s"(m));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wavefront execute in order; this only pins the compiler.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS complex-slot offset of frame g inside a wave's region: a pitch of 512+16 slots shifts odd
// frames by 128 B relative to the 256-byte bank row (conflict-free 32-lane b64 column reads).
__device__ __forceinline__ int frame_base(int g) { return g * FPITCH; }

// the two rows of Zc[k1 + 32 k2] owned by lane c
__device__ __forceinline__ int row1(int c) { return c; }
__device__ __forceinline__ int row2(int c) { return c == 0 ? 16 : 32 - c; }

// exchange, "column" side: element (row k1, column c) for all 32 rows, one b64 access each
__device__ __forceinline__ void xchg_write_cols(cf* fb, int c, const cf* A) {
#pragma unroll
  for (int k1 = 0; k1 < 32; ++k1) fb[k1 * 16 + (c ^ (2 * ((k1 >> 1) & 7)))] = A[k1];
}
__device__ __forceinline__ void xchg_read_cols(const cf* fb, int c, cf* A) {
#pragma unroll
  for (int k1 = 0; k1 < 32; ++k1) A[k1] = fb[k1 * 16 + (c ^ (2 * ((k1 >> 1) & 7)))];
}
// exchange, "row" side: one whole row (16 columns) as eight b128 accesses
__device__ __forceinline__ void xchg_read_row(const cf* fb, int row, cf* B) {
  const int s = (row >> 1) & 7;
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    float4 q = *reinterpret_cast<const float4*>(&fb[row * 16 + ((h ^ s) << 1)]);
    B[2 * h] = {q.x, q.y};
    B[2 * h + 1] = {q.z, q.w};
  }
}
__device__ __forceinline__ void xchg_write_row(cf* fb, int row, const cf* B) {
  const int s = (row >> 1) & 7;
#pragma unroll
  for (int h = 0; h < 8; ++h)
    *reinterpret_cast<float4*>(&fb[row * 16 + ((h ^ s) << 1)]) =
        make_float4(B[2 * h].x, B[2 * h].y, B[2 * h + 1].x, B[2 * h + 1].y);
}

// Forward: v[r] = z[c + 16 r]  ->  v[k2] = Zc[row1 + 32 k2], v[16 + k2] = Zc[row2 + 32 k2].
// fb: this frame's LDS slice; tw512: LDS table T[k1][c] = w_512^(k1 c) (row of 16 lanes contiguous:
// conflict-free, the four frames of a wave read the same row -> broadcast).
__device__ __forceinline__ void fft512_fwd(cf* v, cf* fb, const cf* tw512, int c) {
  // sched_barriers keep the phases apart: left alone, the scheduler overlaps the loads of one
  // phase with the arithmetic of the previous one and the live range balloons past 256 VGPRs.
  __builtin_amdgcn_sched_barrier(0);
  stage_dft32_tw_fwd(v, tw512, c);
  xchg_write_cols(fb, c, v);
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
  xchg_read_row(fb, row1(c), v);
  xchg_read_row(fb, row2(c), v + 16);
  wave_lds_sync();
  stage_dft16x2<false>(v);
  __builtin_amdgcn_sched_barrier(0);
}

// Inverse (unnormalised): v[k2], v[16 + k2] as above  ->  v[r] = 512 * z[c + 16 r].
__device__ __forceinline__ void fft512_inv(cf* v, cf* fb, const cf* tw512, int c) {
  __builtin_amdgcn_sched_barrier(0);
  stage_dft16x2<true>(v);
  xchg_write_row(fb, row1(c), v);
  xchg_write_row(fb, row2(c), v + 16);
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
  xchg_read_cols(fb, c, v);
  wave_lds_sync();
  stage_tw_dft32_inv(v, tw512, c);
  __builtin_amdgcn_sched_barrier(0);
}

// Forward transform with a HALF-size exchange slice (16 rows x 16 columns per frame): rows 0..15 go
// through LDS first (every lane's row1 lies there), then rows 16..31 reuse the same slice (row2).
// Two more wave-level syncs, half the LDS per wave -> a third wave per SIMD for the kernels that do
// not need the slice afterwards (decision and magnitude kernels).
constexpr int FPITCH_H = 256 + 16;
constexpr int WAVE_CX_H = 4 * FPITCH_H;

// (round 6) Abutting tiles for the packed-transform geometries (n_fft = 512 / 256) as at n_fft = 2048 (fast2048.hpp): a tile of NF frames
// completes NF - 3 hops; the 3 hops that straddle two tiles leave as partial sums -- part[unit][tile][6][HOP]: slots 0..2 the tile's
// leading hops, 3..5 its trailing ones -- and k_ola_seam combines them: hop tf0(b + 1) + k = trailing k of tile b + leading k of tile
// b + 1 (fixed order), normalised and stored.  Overlapping tiles redo 3 of every 32 / 64 transforms AND make 10 % more tiles: on two
// minutes of audio 1640 / 1560 workgroups for 768 resident slots (a third round for 104 / 24 of them) against 1490.
struct SeamArgs {
  View view;
  Geom g;
  OutMap om;
  int64_t h_begin, h_end;
  int normalize;
  const float* invn;   // 1 / sum_q wsq[HOP q + s]
  const float* wsq;    // window squared
  const float* part;
  int n_tiles;
};
template <int HOP, int NF>
__global__ __launch_bounds__(HOP) void k_ola_seam(SeamArgs A) {
  const Geom& G = A.g;
  const int64_t u = blockIdx.y, b = blockIdx.x;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  const int s = threadIdx.x;
  const float* pa = A.part + ((u * A.n_tiles + b) * 6 + 3) * HOP;
  const float* pb = A.part + ((u * A.n_tiles + b + 1) * 6 + 0) * HOP;
  float va[3], vb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { va[k] = pa[k * HOP + s]; vb[k] = pb[k * HOP + s]; }
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
          if (ti >= 0 && ti < G.T) nrm += A.wsq[HOP * q + s];
        }
        val /= (nrm > 1e-10f ? nrm : 1.f);
      }
    }
    const int64_t p = h * HOP + s - G.padL;
    if (p < A.om.p0 || p >= A.om.p1) continue;
    const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
    if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
    store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? val : 0.f);
  }
}

// (round 6) The |X| tile of a magnitude kernel as ONE piece of the non-stationary gate's time recurrence (nonstat.hpp: what
// k_iir_part computes from the field in a pass of its own): per band  e = sum_t b c^(end-1-t) A[t]  and  E0 = sum_t b c^(t-start) s0[t]
// (s0: zero-state forward response), float64, over the tile's first n frames.  Row r of the tile lives in the exchange slice of the
// wave that transformed it: slice r / FPW, row r % FPW, PITCH floats apart.  o: [2][FS] of this (unit, tile).
template <int NTHR, int NFR, int FPW, int PITCH, int F>
__device__ __forceinline__ void mag_sub_partials(const void* regions_, int n, double b, double* __restrict__ o, int FS, int tid) {
  const char* base = reinterpret_cast<const char*>(regions_);
  constexpr size_t SLICE = (size_t)WAVE_CX_H * 8;
  const double cc = 1.0 - b;
  for (int f = tid; f < F; f += NTHR) {
    double e = 0.0, E0 = 0.0, pw = b;
    auto step = [&](int r) {
      const double a = (double)reinterpret_cast<const float*>(base + (size_t)(r / FPW) * SLICE)[(r % FPW) * PITCH + f];
      e = b * a + cc * e;
      E0 += pw * e;
      pw *= cc;
    };
    if (n == NFR) {   // every tile but a unit's last: straight-line (the LDS reads up front)
#pragma unroll
      for (int r = 0; r < NFR; ++r) step(r);
    } else {
      for (int r = 0; r < n; ++r) step(r);
    }
    o[f] = e;
    o[FS + f] = E0;
  }
}
constexpr int HPITCH = 288;  // floats between the hop accumulators of a wave (k_apply_fast<LEAN>)
__device__ __forceinline__ int frame_base_h(int g) { return g * FPITCH_H; }

__device__ __forceinline__ void fft512_fwd_half(cf* v, cf* fb, const cf* tw512, int c) {
  __builtin_amdgcn_sched_barrier(0);
  stage_dft32_tw_fwd(v, tw512, c);
  // phase A: rows 0..15
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) fb[k1 * 16 + (c ^ (2 * ((k1 >> 1) & 7)))] = v[k1];
  wave_lds_sync();
  xchg_read_row(fb, row1(c), v);
  wave_lds_sync();
  // phase B: rows 16..31 stored at row - 16 (same swizzle: ((row - 16) >> 1) & 7 == (row >> 1) & 7)
#pragma unroll
  for (int k1 = 16; k1 < 32; ++k1) fb[(k1 - 16) * 16 + (c ^ (2 * ((k1 >> 1) & 7)))] = v[k1];
  wave_lds_sync();
  xchg_read_row(fb, row2(c) - 16, v + 16);
  wave_lds_sync();
  stage_dft16x2<false>(v);
  __builtin_amdgcn_sched_barrier(0);
}

// Inverse transform with the half-size slice: row1 rows (0..15) travel first, then row2 rows.
__device__ __forceinline__ void fft512_inv_half(cf* v, cf* fb, const cf* tw512, int c) {
  __builtin_amdgcn_sched_barrier(0);
  stage_dft16x2<true>(v);
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
  stage_tw_dft32_inv(v, tw512, c);
  __builtin_amdgcn_sched_barrier(0);
}

// One conjugate pair of the real-FFT split -> mask -> merge (see k_apply_istft in kernels.hpp):
// a = Zc[k], b = Zc[N-k], w = w_1024^k, mk / mn = mask of bin k / N-k.  Returns Zc'[k], Zc'[N-k].
// The four 1/2 factors of split and merge are NOT applied here: the caller folds 1/4 into the masks.
// The two halves are shared by every kernel that splits or merges (k_apply_fast, k_gate_onepass, the decision and
// magnitude kernels): the same fused multiply-adds in the same order, so the kernels agree to the bit.
//   split: E = a + conj b, O = (a - conj b) / i;  xa = E + w O = 2 X[k],  xb = E - w O = 2 E - xa (conj-pair value)
__device__ __forceinline__ void split_pair(cf a, cf b, cf w, cf& xa, cf& xb) {
  const cf E = {a.x + b.x, a.y - b.y};
  const cf O = {a.y + b.y, b.x - a.x};
  xa.x = fmaf(w.x, O.x, fmaf(-w.y, O.y, E.x));
  xa.y = fmaf(w.x, O.y, fmaf(w.y, O.x, E.y));
  xb.x = fmaf(2.0f, E.x, -xa.x);
  xb.y = fmaf(2.0f, E.y, -xa.y);
}
//   merge: Yk = xa mk, Yn = conj(xb) mn;  Ep = Yk + conj Yn, D = Yk - conj Yn, Op = D conj(w);
//          a' = (Ep.x - Op.y, Ep.y + Op.x),  b' = (Ep.x + Op.y, Op.x - Ep.y) = (2 Ep.x - a'.x, a'.y - 2 Ep.y)
__device__ __forceinline__ void merge_pair(cf& xa, cf& xb, cf w, float mk, float mn) {
  const float nx = xb.x * mn, ny = -xb.y * mn;
  const cf Ep = {fmaf(xa.x, mk, nx), fmaf(xa.y, mk, -ny)};
  const cf D = {fmaf(xa.x, mk, -nx), fmaf(xa.y, mk, ny)};
  const float ax = fmaf(D.x, w.y, fmaf(-D.y, w.x, Ep.x));
  const float ay = fmaf(D.x, w.x, fmaf(D.y, w.y, Ep.y));
  xa = {ax, ay};
  xb = {fmaf(2.0f, Ep.x, -ax), fmaf(-2.0f, Ep.y, ay)};
}
__device__ __forceinline__ void pair_mask(cf& a, cf& b, cf w, float mk, float mn) {
  cf xa, xb;
  split_pair(a, b, w, xa, xb);
  merge_pair(xa, xb, w, mk, mn);
  a = xa;
  b = xb;
}

// "Entries": lane c works on 16 conjugate-pair slots s = 0..15; entry e = s is the first bin of
// slot s, entry e = 31 - s its partner.  Mask / threshold tables for the fast kernels are stored
// in this order (position c*32 + e, plus position 512 = bin 512), so every lane indexes them the
// same way although lane 0 pairs its bins differently:
//   lanes c >= 1: slot s = (bin c + 32 s , bin (32-c) + 32 (15-s))         registers (v[s], v[31-s])
//   lane 0      : slot 0 = bins 0 / 512 (special); entry 31 = bin 256 (self-paired)
//                 s = 1..7 : (bin 32 s, bin 32 (16-s))                      registers (v[s], v[16-s])
//                 s = 8..15: (bin 16 + 32 (s-8), bin 16 + 32 (23-s))        registers (v[8+s], v[39-s])
__host__ __device__ inline int bin_of_entry(int c, int e) {
  if (c != 0) return e < 16 ? c + 32 * e : (32 - c) + 32 * (e - 16);
  if (e == 0) return 0;
  if (e < 8) return 32 * e;
  if (e < 24) return 16 + 32 * (e - 8);
  if (e < 31) return 32 * (e - 15);
  return 256;
}
__host__ __device__ inline int perm_inv(int pos) {  // table position -> bin
  if (pos >= 512) return 512;
  return bin_of_entry(pos >> 5, pos & 31);
}
__host__ __device__ inline int perm_pos(int f) {    // bin -> table position
  if (f >= 512) return 512;
  const int rho = f & 31, k = f >> 5;
  if (rho != 0 && rho != 16) return rho <= 15 ? rho * 32 + k : (32 - rho) * 32 + 16 + k;
  if (rho == 16) return 8 + k;          // lane 0, entries 8..23
  if (k < 8) return k;                  // lane 0, entries 0..7
  return k == 8 ? 31 : 15 + k;          // bin 256 -> entry 31; bins 32 k (k = 9..15) -> entries 24..30
}
// register of lane 0 that holds entry e (lanes c >= 1: register == entry)
__host__ __device__ constexpr int reg0_of_entry(int e) {
  return e < 8 ? e : (e < 24 ? e + 8 : (e < 31 ? e - 15 : 8));
}

// Lane 0 of a frame owns the self-paired rows 0 and 16 of the transform (fastpath.hpp): its conjugate pairs sit in other
// registers than the (s, 31 - s) pairs of lanes 1..15.  Instead of selecting operands per pair (which keeps both the
// transform's 64 registers and the 64 split values alive),
// lane 0 PERMUTES its registers once -- one 24-cycle walked in place with a single temporary -- so that afterwards
// register index == entry index in every lane (bin_of_entry) and split / merge run in place on (v[s], v[31 - s]):
//   new[1..7] = old[1..7], new[8..15] = old[16..23], new[16..23] = old[24..31], new[24..30] = old[9..15],
//   new[31] = old[8] (bin 256), new[0] = old[0] (bins 0 / 512)
__host__ __device__ constexpr int rg_cyc(int i) {   // position i of the cycle 8 <- 16 <- 24 <- 9 <- 17 <- 25 <- 10 ...
  return (i % 3 == 0) ? 8 + i / 3 : ((i % 3 == 1) ? 16 + i / 3 : 24 + i / 3);
}
__device__ __forceinline__ void rg_lane0_to_entries(cf* v, bool l0) {
  const cf t = v[rg_cyc(0)];
#pragma unroll
  for (int i = 0; i < 23; ++i) {
    const cf s = v[rg_cyc(i + 1)], d = v[rg_cyc(i)];
    v[rg_cyc(i)] = {l0 ? s.x : d.x, l0 ? s.y : d.y};
  }
  const cf d = v[rg_cyc(23)];
  v[rg_cyc(23)] = {l0 ? t.x : d.x, l0 ? t.y : d.y};
}
__device__ __forceinline__ void rg_lane0_from_entries(cf* v, bool l0) {
  const cf t = v[rg_cyc(23)];
#pragma unroll
  for (int i = 23; i >= 1; --i) {
    const cf s = v[rg_cyc(i - 1)], d = v[rg_cyc(i)];
    v[rg_cyc(i)] = {l0 ? s.x : d.x, l0 ? s.y : d.y};
  }
  const cf d = v[rg_cyc(0)];
  v[rg_cyc(0)] = {l0 ? t.x : d.x, l0 ? t.y : d.y};
}

constexpr int OP_SPIN_MAX = 1 << 20;     // polls before a hand-off is declared lost (~1 s): no unbounded spin
typedef unsigned short op_us2 __attribute__((ext_vector_type(2)));
typedef unsigned op_v4u __attribute__((ext_vector_type(4)));

// 16-byte write-through store / L1-bypassing load (the sc1 forms the 8-byte agent-scope atomics compile to):
// two tagged 8-byte granules per instruction.  Each 8-byte half carries its own tag, so the pair needs no
// atomicity beyond the 8-byte granule.
__device__ __forceinline__ void op_st16_sc1(void* p, op_v4u v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ op_v4u op_ld16_sc1(const void* p) {
  op_v4u v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct ApplyArgs {
  View view;
  Geom g;
  OutMap om;
  const unsigned short* K;  // permuted mask counts [units][T][FSK]            (KMASK = true)
  const float* Mf;          // float mask, natural bin order [units][T][FSK]   (KMASK = false)
  const float* win;         // analysis == synthesis window (1024)
  const float* wsq;         // window squared (1024)
  const float* invn;        // 1 / sum_q wsq[256 q + s], s < 256 (interior hops)
  const cf* tw512;          // w_512^j (512)
  const cf* tw1024;         // w_1024^j (512)
  float kscale;             // prop_decrease-free mask scale: 1 / (ktot * 512)
  int64_t h_begin;          // first ext hop (256-sample block, ext = unit sample + 512) to produce
  int64_t h_end;
  int normalize;            // 1: divide by the window envelope (ISTFT); 0: plain overlap-add (adjoint)
  float* part;              // seam mode: [units][tiles][6][256] un-normalised partial hops, else nullptr
  int n_tiles;
  // seam mode of the LEAN kernel: the three hops that straddle two tiles are handed over INSIDE the launch (tile j
  // publishes its trailing partial hops as tagged granules {float, epoch}, tile j + 1 adds its leading partials
  // and finalises them): no k_ola_seam launch.  Placement-independent like k_gate_onepass (onepass.hpp): the grid is
  // one-dimensional and a workgroup works on tile number `ticket` (one atomic per workgroup on a counter that is
  // never reset), NOT on its blockIdx -- HIP promises nothing about dispatch order.  A tile publishes before it
  // waits, and it only ever waits for the tile one ticket earlier, whose workgroup is therefore already running.
  // One counter PER UNIT (blockIdx.y picks the unit -- units are independent --, the ticket picks the tile inside
  // it; 64 bytes apart): a single counter serialises the atomics of a short launch at one memory channel (TorchGate,
  // 1280 workgroups: +11 us of 52).  The workgroup that draws a unit's last ticket puts the counter back to zero.
  unsigned long long* part2;   // [units][tiles][3][256] granules, or nullptr (then `part` + k_ola_seam)
  unsigned epoch;              // tag of this launch
  unsigned* err;               // host-mapped error word (bounded polls)
  unsigned* ticket;            // [units][16] work counters (in-kernel hand-off only), zero between launches
};

// ---------------------------------------------------------------------------------------
// Fused apply: frames -> FFT -> x mask -> IFFT -> window -> overlap-add -> output samples.
// One workgroup = WAVES wavefronts = 4*WAVES consecutive frames of one unit -> 4*WAVES-3 hops.
// ---------------------------------------------------------------------------------------
// LEAN: half-size exchange slices (two-phase exchange) and wave-private hop accumulators instead of
// 4 KB of stored frame per frame: 38 KB of LDS per workgroup and <= 168 VGPRs -> 3 waves per SIMD.
// SG_ABLATE (development only, default 0): bit mask that removes one ingredient of k_apply_fast to
// measure what it costs (results are wrong): 1 mask loads, 2 window loads, 4 output stores,
// 8 both transforms, 16 input loads, 32 pair stage, 64 wave-private overlap-add.  tools/ablate.sh builds and times the variants.
#ifndef SG_ABLATE
#define SG_ABLATE 0
#endif
// LOSE (tests only, SG_OPT_INJECT_HANDOFF_FAULT bit 5): the in-launch hand-off polls give up at once (see k_gate_onepass).
template <int WAVES, bool KMASK, bool LEAN, bool LOSE = false>
__global__ __launch_bounds__(WAVES * 64, LEAN ? 3 : 2) void k_apply_fast(ApplyArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  constexpr int NF = 4 * WAVES;   // frames per tile
  constexpr int NH = NF - 3;      // hops per tile
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  // LEAN: the window table (4 KB) lives in LDS behind the exchange slices: both window passes read it
  // with ds_read_b64 instead of 64 global loads per lane
  float* swin = reinterpret_cast<float*>(regions + WAVES * (LEAN ? WAVE_CX_H : WAVE_CX));
  stage_tables<WAVES * 64, LEAN ? 256 : 0>(tw512, A.tw512, swin, A.win, tid);
  const Geom& G = A.g;
  // which tile: the grid position, or -- when hops are handed from tile to tile inside the launch -- a ticket
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if constexpr (LEAN) {
    if (A.ticket != nullptr) {   // (uniform) the table loads above and the atomic share one memory round trip
      unsigned* s_tk = reinterpret_cast<unsigned*>(swin + 1024);
      if (tid == 0) {
        unsigned* ctr = A.ticket + (size_t)by * 16;
        const unsigned tk = atomicAdd(ctr, 1u);
        if (tk + 1u == (unsigned)A.n_tiles) atomicExch(ctr, 0u);   // all of this unit's tickets are out
        *s_tk = tk;
      }
      __syncthreads();
      bx = *s_tk;
    }
  }
  const int64_t u = by;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  // Tiles either overlap by 3 frames (each tile completes its NH hops on its own) or, in seam mode,
  // abut: then the 3 hops that straddle two tiles are written as un-normalised partial sums and
  // combined by k_ola_seam -- 3/16 fewer transforms.
  const bool seam = A.n_tiles > 0;
  const int64_t tf_tile = A.h_begin - 3 + (int64_t)bx * (seam ? NF : NH);  // first frame of the tile
  const int64_t t = tf_tile + 4 * wave + g;                  // this lane group's frame
  const bool fvalid = t >= 0 && t < G.T;
  // 1 / window envelope of this thread's four sample phases, used by the OLA epilogue: loaded at entry
  // when registers allow, just before the final barrier in the LEAN kernel (else it is spilled)
  float4 inv4;
  if constexpr (!LEAN) inv4 = *reinterpret_cast<const float4*>(&A.invn[(tid & 63) * 4]);

  // mask of this lane's 32 bins (+ bin 512 for lane c == 0), permuted layout.  KMASK: uint16 counts
  // (mask = K / ktot).  Issued before the forward transform when registers allow (!LEAN: latency
  // hidden behind the FFT), after it in the LEAN kernel (168-VGPR budget, 3 waves hide the latency).
  unsigned short kk[KMASK ? 32 : 1];
  float mf[KMASK ? 1 : 32];
  float k512 = 0.f;
  auto load_mask = [&]() {
  if constexpr ((SG_ABLATE & 1) != 0) {
#pragma unroll
    for (int q = 0; q < (KMASK ? 32 : 1); ++q) kk[q] = (unsigned short)(c + q);
#pragma unroll
    for (int q = 0; q < (KMASK ? 1 : 32); ++q) mf[q] = 0.5f + c;
    k512 = 1.f;
  } else if constexpr (KMASK) {
      const unsigned short* Krow = A.K + ((u * G.T + (fvalid ? t : 0)) * (int64_t)FSK);
      const uint4* p4 = reinterpret_cast<const uint4*>(Krow + c * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint4 w4 = p4[q];
        unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kk[q * 8 + 2 * j] = (unsigned short)(ws[j] & 0xffffu);
          kk[q * 8 + 2 * j + 1] = (unsigned short)(ws[j] >> 16);
        }
      }
      k512 = (float)Krow[512] * A.kscale;
    } else if constexpr (LEAN) {
      // (round 5) natural bin order, through the wave's idle exchange slice: the four mask rows of a wave are ONE contiguous
      // 8448-byte block of the field (frames tq .. tq + 3, pitch FSK) -- nine 16-byte loads per lane, then the 33 entries
      // from LDS (pitch 528 floats: the four lane groups read disjoint banks) instead of 33 four-byte global loads per
      // lane in 64-byte runs.  Called between the two transforms: the slice is free.
      float* mt = reinterpret_cast<float*>(regions + wave * WAVE_CX_H);
      static_assert(4 * FSK * 4 <= WAVE_CX_H * 8, "four mask rows fit the wave's exchange slice");
      const int64_t tq0 = tf_tile + 4 * wave;
      const float* Mu = A.Mf + (u * G.T) * (int64_t)FSK;
      constexpr int R4 = FSK / 4;                    // float4 per row
      auto ldrow = [&](int i) -> float4 {
        const int r = i / R4, j = i - r * R4;
        int64_t tr = tq0 + r;
        tr = tr < 0 ? 0 : (tr >= G.T ? G.T - 1 : tr);   // frames outside the unit: any row (their spectra are zero)
        return reinterpret_cast<const float4*>(Mu + tr * (int64_t)FSK)[j];
      };
      static_assert(4 * R4 == 8 * 64 + 16, "eight full passes of the wave + 16 lanes");
      const float4 q0 = ldrow(lane), q1 = ldrow(lane + 64), q2 = ldrow(lane + 128), q3 = ldrow(lane + 192), q4 = ldrow(lane + 256),
                   q5 = ldrow(lane + 320), q6 = ldrow(lane + 384), q7 = ldrow(lane + 448), q8 = ldrow(512 + (lane & 15));
      float4* mt4 = reinterpret_cast<float4*>(mt);
      mt4[lane] = q0; mt4[lane + 64] = q1; mt4[lane + 128] = q2; mt4[lane + 192] = q3; mt4[lane + 256] = q4;
      mt4[lane + 320] = q5; mt4[lane + 384] = q6; mt4[lane + 448] = q7;
      if (lane < 16) mt4[512 + lane] = q8;
      wave_lds_sync();
      const float* Mrow = mt + g * FSK;
#pragma unroll
      for (int e = 0; e < 32; ++e) mf[e] = Mrow[bin_of_entry(c, e)];
      k512 = Mrow[512] * A.kscale;
      wave_lds_sync();   // every lane has its entries: the inverse transform may overwrite the slice
    } else {
      // natural bin order: the 16 lanes of a frame read one 64-byte run per slot
      const float* Mrow = A.Mf + ((u * G.T + (fvalid ? t : 0)) * (int64_t)FSK);
#pragma unroll
      for (int e = 0; e < 32; ++e) mf[e] = Mrow[bin_of_entry(c, e)];
      k512 = Mrow[512] * A.kscale;
    }
  };
  if constexpr (!LEAN) load_mask();
  auto mval = [&](int q, float scale) -> float {
    if constexpr (KMASK) return (float)kk[q] * scale;
    else return mf[q] * scale;
  };
  cf* fb = regions + wave * (LEAN ? WAVE_CX_H : WAVE_CX) + (LEAN ? frame_base_h(g) : frame_base(g));
  // LEAN, interior tiles of float32 input: the tile's NF frames cover one contiguous span of
  // (NF-1)*256 + 1024 samples; the workgroup stages it once in LDS with coalesced 16-byte loads (the
  // exchange slices are idle until the first transform) instead of every lane gathering 32 float2
  // from global memory.  Hops are stored with a pitch of 288 floats so that the two lane groups of a
  // ds_read_b64 pass (frames g, g+1: 256 samples apart) hit disjoint bank halves.
  constexpr int SPAN = (NF - 1) * 256 + 1024, XPITCH = 288;
  static_assert(!LEAN || (SPAN / 256) * XPITCH <= WAVES * WAVE_CX_H * 2, "span must fit the exchange slices");
  // (LEAN) every tile stages its span: interior tiles of float32 input with 16-byte loads, the others (row edges,
  // other sample types) sample by sample through view_sample -- zero outside the readable range.  Short rows
  // (TorchGate: 4 tiles per row, 2 of them at an edge) run the same code as long ones.
  bool blk_vec = false;
  if constexpr (LEAN) {
    const int64_t s0b = tf_tile * 256 - G.padL;
    const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
    const float* sp = (const float*)A.view.x + row * A.view.stride + gb;
    blk_vec = A.view.dtype == 0 && tf_tile >= 0 && tf_tile + NF <= G.T && s0b >= 0 && s0b + SPAN <= A.view.Lp &&
              gb >= A.view.lo && gb + SPAN <= A.view.hi && !(SG_ABLATE & 16) &&
              (reinterpret_cast<uintptr_t>(sp) & 15) == 0;
    float* xs = reinterpret_cast<float*>(regions);
    if (blk_vec) {
      stage_span_vec<WAVES * 64, SPAN, XPITCH>(xs, sp, tid);
    } else {
      for (int i = tid; i < SPAN; i += WAVES * 64)
        xs[(i >> 8) * XPITCH + (i & 255)] = (float)view_sample(A.view, row, chunk, s0b + i);
    }
    __syncthreads();  // twiddles, window and span staged
  }
  // gather the frame: v[r] = (x[2c + 32r], x[2c + 32r + 1]) * window
  cf v[32];
  if constexpr (LEAN) {
    const float* xs = reinterpret_cast<const float*>(regions) + (4 * wave + g) * XPITCH + 2 * c;
    const float2* wl = reinterpret_cast<const float2*>(swin + 2 * c);
    if (blk_vec) {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const float2 x2 = *reinterpret_cast<const float2*>(xs + (r >> 3) * XPITCH + 32 * (r & 7));
        const float2 w2 = wl[16 * r];
        v[r] = {x2.x * w2.x, x2.y * w2.y};
      }
    } else {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        float2 x2 = *reinterpret_cast<const float2*>(xs + (r >> 3) * XPITCH + 32 * (r & 7));
        if (!fvalid) x2 = make_float2(0.f, 0.f);   // frames before / past the row: zeros
        const float2 w2 = wl[16 * r];
        v[r] = {x2.x * w2.x, x2.y * w2.y};
      }
    }
    __syncthreads();  // every lane has its samples: the span may be overwritten by the exchanges
  } else {
    const int64_t s0 = t * 256 - G.padL;  // unit-local index of frame sample 0
    const int64_t gbase = chunk * A.view.cs - A.view.pad + s0;
    const bool inside = fvalid && s0 >= 0 && s0 + 1024 <= A.view.Lp && gbase >= A.view.lo &&
                        gbase + 1024 <= A.view.hi && A.view.dtype == 0;
    const float* src = (const float*)A.view.x + row * A.view.stride + gbase + 2 * c;
    const bool aligned = (reinterpret_cast<uintptr_t>(src) & 7) == 0;
    const float2* wsrc = LEAN ? reinterpret_cast<const float2*>(swin + 2 * c)
                              : reinterpret_cast<const float2*>(A.win + 2 * c);
    if (inside && aligned) {
      const float2* s2 = reinterpret_cast<const float2*>(src);
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        float2 x2 = (SG_ABLATE & 16) ? make_float2(0.001f * r + c, 0.5f) : s2[16 * r];
        float2 w2 = (SG_ABLATE & 2) ? make_float2(0.5f, 0.25f + r) : wsrc[16 * r];
        v[r] = {x2.x * w2.x, x2.y * w2.y};
      }
    } else {
      // edge / non-float32 / unaligned frames: rolled gather staged through this frame's LDS slice
      // (LEAN: the slice holds 512 floats, so the frame is staged as two halves)
      float* fl = reinterpret_cast<float*>(fb);
      constexpr int NHALF = LEAN ? 2 : 1, RPH = 32 / NHALF;
#pragma unroll
      for (int hh = 0; hh < NHALF; ++hh) {
#pragma unroll 1
        for (int r = 0; r < RPH; ++r) {
          float a = 0.f, b = 0.f;
          if (fvalid) {
            a = (float)view_sample(A.view, row, chunk, s0 + 2 * c + 32 * (r + RPH * hh));
            b = (float)view_sample(A.view, row, chunk, s0 + 2 * c + 32 * (r + RPH * hh) + 1);
          }
          fl[2 * c + 32 * r] = a;
          fl[2 * c + 32 * r + 1] = b;
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < RPH; ++r) {
          float2 w2 = wsrc[16 * (r + RPH * hh)];
          cf x2 = fb[c + 16 * r];
          v[r + RPH * hh] = {x2.x * w2.x, x2.y * w2.y};
        }
        wave_lds_sync();
      }
    }
  }
  if constexpr (!LEAN) __syncthreads();  // twiddle table staged
  // (LEAN) a wave whose four frames all lie outside the row holds zeros: its transforms, masks and merge are skipped
  // (zeros in, zeros out), only the overlap-add below runs.  Short rows end with such waves: a TorchGate row of 63
  // frames fills 5 tiles of 16.
  const bool wave_live = !LEAN || (tf_tile + 4 * wave + 3 >= 0 && tf_tile + 4 * wave < G.T);
  if constexpr (LEAN) {
    if (wave_live) {
      if constexpr ((SG_ABLATE & 8) == 0) fft512_fwd_half(v, fb, tw512, c);
      load_mask();
    }
  } else {
    fft512_fwd(v, fb, tw512, c);
  }

  // synthesis window: (!LEAN) issued now so that it arrives while the inverse transform runs;
  // (LEAN) loaded after the inverse transform -- three waves per SIMD hide the latency and the
  // 64 registers stay free during the transforms
  float2 wsyn[32];
  const float2* wsrc2 = LEAN ? reinterpret_cast<const float2*>(swin + 2 * c)
                             : reinterpret_cast<const float2*>(A.win + 2 * c);
  if constexpr (!LEAN) asm volatile("" : "+v"(wsrc2));  // opaque: a separate load, not a CSE of the analysis window
  if constexpr (!LEAN) {
#pragma unroll
    for (int r = 0; r < 32; ++r) wsyn[r] = wsrc2[16 * r];
  }
  // split -> mask -> merge on conjugate pairs, all in this lane.  One instruction stream for all
  // lanes: lane 0 (self-paired rows 0 and 16) only differs in WHICH registers form a pair, handled
  // with v_cndmask selects on the way in and out (a divergent branch would run the stage twice).
  if (((SG_ABLATE & 32) == 0) && wave_live) {
    const float ks = A.kscale * 0.25f;  // pair_mask leaves out four 1/2 factors
    const bool l0 = c == 0;
    const cf wlo = A.tw1024[c];                         // w_1024^c   (lane 0: 1)
    cf whi = wlo;                                       // slots >= 8: lane 0 uses i * w_1024^16
    {
      const cf w16 = A.tw1024[16];
      if (l0) whi = {-w16.y, w16.x};
    }
    auto sel = [&](cf a0, cf a1) -> cf { return {l0 ? a0.x : a1.x, l0 ? a0.y : a1.y}; };
    cf nv[32];
    // slot 0: general pair (v[0], v[31]); lane 0: bins 0 / 512 from v[0], bin 256 = v[8] scaled
    cf s0a = v[0], s0b = v[31];
    pair_mask(s0a, s0b, wlo, mval(0, ks), mval(31, ks));
    {
      const cf a = v[0];
      const float y0 = (a.x + a.y) * mval(0, A.kscale);
      const float yN = (a.x - a.y) * k512;
      const cf z0 = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
      const float m8 = mval(31, A.kscale);  // entry 31 of lane 0 = bin 256
      const cf z8 = {v[8].x * m8, v[8].y * m8};
      nv[0] = sel(z0, s0a);
      nv[8] = z8;      // lane 0 only; lanes >= 1 overwrite nv[8] below (slot 8's first bin)
      nv[31] = s0b;    // lanes >= 1 only; lane 0 overwrites nv[31] below (slot 8's partner)
    }
    cf pa[16], pb[16];
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      // operands: lanes >= 1 (v[sl], v[31-sl]); lane 0 (v[sl], v[16-sl]) or (v[8+sl], v[39-sl])
      cf a = sl < 8 ? v[sl] : sel(v[8 + sl], v[sl]);
      cf b = sl < 8 ? sel(v[16 - sl], v[31 - sl]) : sel(v[39 - sl], v[31 - sl]);
      const cf wl = sl < 8 ? wlo : whi;
      const cf w = mul_tw<false>(wl, twc<32>(sl), tws<32>(sl));
      pair_mask(a, b, w, mval(sl, ks), mval(31 - sl, ks));
      pa[sl] = a;
      pb[sl] = b;
    }
    // scatter back: register i receives, for lanes >= 1, entry i; for lane 0, the entry that
    // lives in register i (reg0_of_entry)
#pragma unroll
    for (int i = 1; i < 8; ++i) nv[i] = pa[i];                                   // both
    {
      const cf keep8 = nv[8];
      nv[8] = sel(keep8, pa[8]);
    }
#pragma unroll
    for (int i = 9; i < 16; ++i) nv[i] = sel(pb[16 - i], pa[i]);                 // lane 0: partner of slot 16-i
#pragma unroll
    for (int i = 16; i < 24; ++i) nv[i] = sel(pa[i - 8], pb[31 - i]);            // lane 0: first bin of slot i-8
#pragma unroll
    for (int i = 24; i < 31; ++i) nv[i] = sel(pb[39 - i], pb[31 - i]);           // lane 0: partner of slot 39-i
    {
      const cf keep31 = nv[31];
      nv[31] = sel(pb[8], keep31);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = nv[i];
  }
  if constexpr (LEAN) {
    if (((SG_ABLATE & 8) == 0) && wave_live) fft512_inv_half(v, fb, tw512, c);
#pragma unroll
    for (int r = 0; r < 32; ++r) wsyn[r] = (SG_ABLATE & 2) ? make_float2(0.5f, 0.25f + r) : wsrc2[16 * r];
    // wave-private overlap-add of this wave's 4 frames into 7 hop accumulators (7 KB, reusing the
    // exchange slices).  Step j: every frame adds its quarter j -> frame g touches hop g + j: the four
    // lane groups never collide within a step, and a hop receives its quarters in the fixed order
    // j = 0, 1, 2, 3 (LDS operations of a wave execute in order) -> deterministic sums.
    // (hop pitch 288 floats: the two lane groups of a ds_read_b64 pass, frames g and g + 1, then hit
    // disjoint bank halves)
    float* acc = reinterpret_cast<float*>(regions + wave * WAVE_CX_H);
    static_assert(7 * HPITCH <= WAVE_CX_H * 2, "hop accumulators must fit the wave's exchange slices");
#pragma unroll
    for (int j = 0; j < ((SG_ABLATE & 64) ? 0 : 4); ++j) {
      // (first contribution to hop g + j -- j == 0, or frame g == 3 -- is a plain store)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * j + rr;
        float2* dst = reinterpret_cast<float2*>(acc + (g + j) * HPITCH + 2 * c + 32 * rr);
        float2 nw = {v[r].x * wsyn[r].x, v[r].y * wsyn[r].y};
        if (j != 0) {
          const float2 old = *dst;
          nw.x = sel_s(OLA_KEEP, nw.x + old.x, nw.x);
          nw.y = sel_s(OLA_KEEP, nw.y + old.y, nw.y);
        }
        *dst = nw;
      }
      wave_lds_sync();
    }
    inv4 = *reinterpret_cast<const float4*>(&A.invn[(tid & 63) * 4]);
  } else {
    fft512_inv(v, fb, tw512, c);
    // synthesis window, store the time-domain frame (natural order) into this frame's LDS slice
#pragma unroll
    for (int r = 0; r < 32; ++r) fb[c + 16 * r] = {v[r].x * wsyn[r].x, v[r].y * wsyn[r].y};
  }
  __syncthreads();

  // overlap-add: tile hop jj (ext hop tf_tile + jj) = sum over tile frames i = jj-3..jj of quarter jj-i
  const float* fr = reinterpret_cast<const float*>(regions);
  const int s4 = (tid & 63) * 4;
  const float4 n4 = inv4;
  const int jj_lo = seam ? 0 : 3, jj_hi = seam ? NF + 3 : NF;
  const bool inkernel = LEAN && seam && A.part2 != nullptr;
  // in-kernel seam: per wave the TRAILING hop first (published early), the interior hops, the LEADING hop last (the
  // previous tile -- one ticket earlier -- has usually published by then); publishing never waits
  const int n_it = (jj_hi - jj_lo - (tid >> 6) + WAVES - 1) / WAVES;
  for (int it = 0; it < n_it; ++it) {
    int jj = jj_lo + (tid >> 6) + it * WAVES;
    if (inkernel && (tid >> 6) < 3) jj = it == 0 ? NF + (tid >> 6) : (it == n_it - 1 ? (tid >> 6) : (tid >> 6) + WAVES * it);
    const int64_t h = tf_tile + jj;
    if (h >= A.h_end || h < A.h_begin) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool all_valid = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t ti = tf_tile + jj - q;
      if (ti < 0 || ti >= G.T) all_valid = false;
    }
    if constexpr (LEAN) {
      // hop jj lives in the accumulators of wave jj/4 (local hop jj%4 .. ) and, for jj%4 <= 2, of the
      // wave before it (local hop jj%4 + 4); fixed order: earlier wave first
      const int wh = jj >> 2, lh = jj & 3;
      if (wh >= 1 && wh - 1 < WAVES && lh <= 2) {
        float4 f4 = *reinterpret_cast<const float4*>(&fr[(wh - 1) * WAVE_CX_H * 2 + (lh + 4) * HPITCH + s4]);
        acc = f4;
      }
      if (wh < WAVES) {
        float4 f4 = *reinterpret_cast<const float4*>(&fr[wh * WAVE_CX_H * 2 + lh * HPITCH + s4]);
        acc.x += f4.x; acc.y += f4.y; acc.z += f4.z; acc.w += f4.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = jj - q;                     // tile frame contributing quarter q
        const int64_t ti = tf_tile + i;
        if (i >= 0 && i < NF && ti >= 0 && ti < G.T) {
          const int off = ((i >> 2) * WAVE_CX + frame_base(i & 3)) * 2 + 256 * q + s4;  // float index
          float4 f4 = *reinterpret_cast<const float4*>(&fr[off]);
          acc.x += f4.x; acc.y += f4.y; acc.z += f4.z; acc.w += f4.w;
        }
      }
    }
    if (inkernel && jj >= NF) {
      unsigned long long* dst = A.part2 + (((size_t)u * A.n_tiles + bx) * 3 + (jj - NF)) * 256 + s4;
      const op_v4u ga = {__float_as_uint(acc.x), A.epoch, __float_as_uint(acc.y), A.epoch};
      const op_v4u gb = {__float_as_uint(acc.z), A.epoch, __float_as_uint(acc.w), A.epoch};
      op_st16_sc1(dst, ga);
      op_st16_sc1(dst + 2, gb);
      continue;
    }
    if (inkernel && jj < 3) {
      // h >= h_begin implies bx >= 1: the previous tile exists, and its workgroup holds the previous ticket
      const unsigned long long* src = A.part2 + (((size_t)u * A.n_tiles + bx - 1) * 3 + jj) * 256 + s4;
      op_v4u ga, gb;
      for (int spin = 0;; ++spin) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(ga), "=&v"(gb) : "v"(src) : "memory");
        const unsigned e = A.epoch;
        if (!LOSE && ga[1] == e && ga[3] == e && gb[1] == e && gb[3] == e) break;
        if (LOSE || spin >= OP_SPIN_MAX) {
          atomicOr_system(A.err, 4u);
          // the previous tile's share never arrived: these hops become NaN (a device-tensor caller that does not
          // check the error word must not receive a plausible partial sum)
          ga[0] = ga[2] = gb[0] = gb[2] = 0x7fc00000u;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      // trailing partial of the previous tile + leading partial of this one (the order k_ola_seam adds them in)
      acc.x = __uint_as_float(ga[0]) + acc.x;
      acc.y = __uint_as_float(ga[2]) + acc.y;
      acc.z = __uint_as_float(gb[0]) + acc.z;
      acc.w = __uint_as_float(gb[2]) + acc.w;
    } else if (jj < 3 || jj >= NF) {
      // seam hop: partial sum only; slot 0..2 = leading hops, 3..5 = trailing hops of this tile
      const int slot = jj < 3 ? jj : 3 + (jj - NF);
      float* dst = A.part + (((u * A.n_tiles + bx) * 6 + slot) * 256 + s4);
      *reinterpret_cast<float4*>(dst) = acc;
      continue;
    }
    if (!A.normalize) {
      // adjoint: un-normalised overlap-add
    } else if (all_valid) {
      acc.x *= n4.x; acc.y *= n4.y; acc.z *= n4.z; acc.w *= n4.w;
    } else {
      float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ti = h - q;
        if (ti >= 0 && ti < G.T) {
          float4 w4 = *reinterpret_cast<const float4*>(&A.wsq[256 * q + s4]);
          nrm.x += w4.x; nrm.y += w4.y; nrm.z += w4.z; nrm.w += w4.w;
        }
      }
      acc.x /= (nrm.x > 1e-10f ? nrm.x : 1.f);
      acc.y /= (nrm.y > 1e-10f ? nrm.y : 1.f);
      acc.z /= (nrm.z > 1e-10f ? nrm.z : 1.f);
      acc.w /= (nrm.w > 1e-10f ? nrm.w : 1.f);
    }
    {
      // whole hop inside the kept range of a float32 output: one 16-byte store per lane
      const int64_t pb = h * 256 - G.padL;
      const int64_t gi0 = chunk * A.om.g_step + (pb - A.om.p0);
      if (A.om.dtype == 0 && pb >= A.om.p0 && pb + 256 <= A.om.p1 && pb + 256 <= G.Lout && gi0 >= A.om.g_lo &&
          gi0 + 256 <= A.om.g_hi && !(SG_ABLATE & 4)) {
        float* dst = (float*)A.om.out + (row * A.om.stride + gi0 - A.om.g0 + s4);
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = acc;
          continue;
        }
      }
    }
    const float vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t p = h * 256 + s4 + e - G.padL;  // unit-local output position
      if (p < A.om.p0 || p >= A.om.p1) continue;
      const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
      if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
      if ((SG_ABLATE & 4) && vals[e] != 12345.678f) continue;
      store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? vals[e] : 0.f);
    }
  }
}

// Seam hops of abutting tiles: hop h = tf0 + NF*(b+1) + k (k = 0..2) is the trailing partial k of
// tile b plus the leading partial k of tile b+1, normalised and stored here.
template <int NF>
__global__ __launch_bounds__(256) void k_ola_seam(ApplyArgs A) {
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t b = blockIdx.x;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  const int s = threadIdx.x;
  const float* pa = A.part + ((u * A.n_tiles + b) * 6 + 3) * 256;      // trailing hops of tile b
  const float* pb = A.part + ((u * A.n_tiles + b + 1) * 6 + 0) * 256;  // leading hops of tile b+1
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int64_t h = A.h_begin - 3 + (int64_t)NF * (b + 1) + k;
    if (h < A.h_begin || h >= A.h_end) continue;
    float val = pa[k * 256 + s] + pb[k * 256 + s];
    if (A.normalize) {
      if (h - 3 >= 0 && h < G.T) {
        val *= A.invn[s];  // same reciprocal table as the in-tile hops
      } else {
        float nrm = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t ti = h - q;
          if (ti >= 0 && ti < G.T) nrm += A.wsq[256 * q + s];
        }
        val /= (nrm > 1e-10f ? nrm : 1.f);
      }
    }
    const int64_t p = h * 256 + s - G.padL;
    if (p < A.om.p0 || p >= A.om.p1) continue;
    const int64_t gi = chunk * A.om.g_step + (p - A.om.p0);
    if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
    store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? val : 0.f);
  }
}

}  // namespace fast
}  // namespace sg

// =======================================================================================
// Fast decision kernel: float32 STFT + exact float64 re-evaluation of ambiguous cells.
//
// The stationary mask is a hard compare |X[k]|^2 > T2[k] (one flipped cell
// moves the output by ~5e-3 of peak), so the decision must agree with a float64 evaluation.
// Almost every cell is far from its threshold: the float32 transform decides those; a cell is
// "ambiguous" when | |X| - T | <= delta with delta = 2^-16 * ||x w||_2, ~60x the RMS rounding
// error of the float32 pipeline (window rounding + 10 butterfly levels).  Ambiguous cells
// (~1e-5 of all cells on noise-like input) are re-evaluated exactly: all 64 lanes cooperate on
// the 1024-term float64 DFT sum of that one bin.
// =======================================================================================
namespace sg {
namespace fast {

struct DecideArgs {
  View view;
  Geom g;
  const float* win;        // analysis window, float32 (1024)
  const double* win64;     // analysis window, float64 (1024)
  const cf* tw512;         // w_512^j  float32 (512)
  const cf* tw1024;        // w_1024^j float32 (512)
  const cx<double>* tw64;  // w_1024^j float64 (512)
  ThreshConsts tc;         // T2 (raw power compare constants), thresh, pmax, need_floor
  double mag_scale, top_db;
  unsigned long long* bits;  // [units][T][wpr]
  int wpr;
  int64_t t_begin, t_end;    // frames to decide
  int quads_per_wave;        // consecutive frame quads handled by one wave
};


// exact float64 |X[f]|^2 of frame t, computed by the whole wavefront (rare path: kept out of line
// so that its float64 temporaries do not inflate the register budget of the main loop)
__device__ __forceinline__ double exact_power(const DecideArgs& A, int64_t row, int64_t chunk, int64_t t, int f,
                                              int lane) {
  const int64_t s0 = t * A.g.H - A.g.padL;
  double re = 0.0, im = 0.0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int m = lane + 64 * i;
    const double xv = view_sample(A.view, row, chunk, s0 + m) * A.win64[m];
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

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 3) void k_decide_fast(DecideArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  const int need = A.tc.need_floor[u];
  const bool floor_live = need == 1;

  // effective compare constants (4x the raw-power constant: the split below works on 2X) as
  // float32 in LDS, permuted like the mask rows: entry c*32 + e = bin_of_entry(c, e), entry 512 = bin 512
  float* s_t2 = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  auto t2eff = [&](int f) -> double {
    double v = A.tc.T2[f];
    if (floor_live) {
      double fl = cell_db(A.tc.pmax[u * G.FS + f], A.mag_scale) - A.top_db;
      if (fl > A.tc.thresh[f]) v = -1.0;
    }
    if (need == 2) v = T2_NEVER;
    return v;
  };
  for (int i = tid; i <= 512; i += WAVES * 64) {
    double v = t2eff(perm_inv(i));
    // "every cell passes" as a huge negative constant: P - T > 0, and (P - T)^2 overflows to +inf while
    // d2 * (P + T) is negative, so the ambiguity test fails without an extra T >= 0 term
    s_t2[t2_pos(i)] = t2_to_f32(v, 4.0);
  }
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
  const cf wl0 = A.tw1024[c];  // w_1024^c (lane 0: 1)
  // window table in LDS (behind the compare constants); interior blocks of float32 input also stage
  // the contiguous sample span of their 4*WAVES frames in the (still idle) exchange slices -- see
  // k_apply_fast
  float* swin = s_t2 + T2_FLOATS;
  stage_tables<WAVES * 64, 256>(tw512, A.tw512, swin, A.win, tid);
  constexpr int NFB = 4 * WAVES, SPAN = (NFB - 1) * 256 + 1024, XPITCH = 288;
  static_assert((SPAN / 256) * XPITCH <= WAVES * WAVE_CX_H * 2, "span must fit the exchange slices");
  const int64_t tqb = A.t_begin + (int64_t)blockIdx.x * NFB;  // first frame of the workgroup
  // every block stages its span (see k_apply_fast): 16-byte loads in the interior, checked loads at the edges
  bool blk_vec;
  {
    const int64_t s0b = tqb * 256 - G.padL;
    const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
    const float* sp = (const float*)A.view.x + row * A.view.stride + gb;
    const bool interior = tqb + NFB <= A.t_end && tqb + NFB <= G.T && s0b >= 0 && s0b + SPAN <= A.view.Lp &&
                          gb >= A.view.lo && gb + SPAN <= A.view.hi;
    const int16_t* sp16 = (const int16_t*)A.view.x + row * A.view.stride + gb;
    const bool vec16 = A.view.dtype == 2 && interior && (reinterpret_cast<uintptr_t>(sp16) & 7) == 0;
    blk_vec = (A.view.dtype == 0 && interior && (reinterpret_cast<uintptr_t>(sp) & 15) == 0) || vec16;
    float* xs = reinterpret_cast<float*>(regions);
    if (vec16) {
      stage_span_vec_i16<WAVES * 64, SPAN, XPITCH>(xs, sp16, tid);
    } else if (blk_vec) {
      stage_span_vec<WAVES * 64, SPAN, XPITCH>(xs, sp, tid);
    } else {
      for (int i = tid; i < SPAN; i += WAVES * 64)
        xs[(i >> 8) * XPITCH + (i & 255)] = (float)view_sample(A.view, row, chunk, s0b + i);
    }
  }
  __syncthreads();

  {
    // one frame quad per wave (no loop: loop-invariant twiddle/window loads would be hoisted and
    // pin >100 VGPRs, costing the second wave per SIMD)
    const int64_t tq = tqb + wave * 4;  // first frame of the quad
    const int64_t t = tq + g;
    const bool fvalid = t < A.t_end && t < G.T;
    cf v[32];
    float nrm2 = 0.f;
    {
      const float* xs = reinterpret_cast<const float*>(regions) + (4 * wave + g) * XPITCH + 2 * c;
      const float2* wl2 = reinterpret_cast<const float2*>(swin + 2 * c);
      if (blk_vec) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          const float2 x2 = *reinterpret_cast<const float2*>(xs + (r >> 3) * XPITCH + 32 * (r & 7));
          const float2 w2 = wl2[16 * r];
          v[r] = {x2.x * w2.x, x2.y * w2.y};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          float2 x2 = *reinterpret_cast<const float2*>(xs + (r >> 3) * XPITCH + 32 * (r & 7));
          if (!fvalid) x2 = make_float2(0.f, 0.f);
          const float2 w2 = wl2[16 * r];
          v[r] = {x2.x * w2.x, x2.y * w2.y};
        }
      }
    }
    __syncthreads();  // the span may now be overwritten by the exchanges
    if (tq >= A.t_end) return;  // wave-uniform; no barrier below
    {
#pragma unroll
      for (int r = 0; r < 32; ++r) nrm2 += v[r].x * v[r].x + v[r].y * v[r].y;
      // sum over the 16 lanes of this frame
      nrm2 += __shfl_xor(nrm2, 1);
      nrm2 += __shfl_xor(nrm2, 2);
      nrm2 += __shfl_xor(nrm2, 4);
      nrm2 += __shfl_xor(nrm2, 8);
    }
    {
      // keep the loop-invariant twiddle reads inside the loop: hoisted, they would pin ~100 VGPRs
      // (an opaque zero OFFSET: an opaque pointer would lose the LDS address space -> FLAT loads)
      int z0 = 0;
      asm volatile("" : "+v"(z0));
      fft512_fwd_half(v, fb, tw512 + z0, c);
    }
    cf wl = wl0;
    asm volatile("" : "+v"(wl.x), "+v"(wl.y));
    float t2[32];
    float t2_512;
    {
      int zt = 0;
      asm volatile("" : "+v"(zt));
      const float* tp = s_t2 + zt;
      const float4* t4 = reinterpret_cast<const float4*>(tp + c * T2_PITCH);
#pragma unroll
      for (int q4 = 0; q4 < 8; ++q4) {
        float4 x = t4[q4];
        t2[4 * q4] = x.x; t2[4 * q4 + 1] = x.y; t2[4 * q4 + 2] = x.z; t2[4 * q4 + 3] = x.w;
      }
      t2_512 = tp[T2_POS512];
    }

    // powers of the lane's 32 bins (x4): P4[k] = |E2 + w O2|^2, P4[N-k] = |E2 - w O2|^2 with
    // E2 = a + conj(b), O2 = (a - conj(b)) / i; each is decided as soon as it exists.
    // ambiguous:  (P4 - T4)^2 <= 2 * (2 delta)^2 * (P4 + T4)  (implied by |2|X| - 2T| <= 2 delta),
    // delta^2 = 2^-32 * nrm2.
    // 2 * (2 delta)^2; a silent frame (nrm2 == 0) gets a negative factor: no ambiguous cells (the tests
    // below are then pure vector compares -- boolean terms would be combined on the scalar unit)
    const float d2 = nrm2 > 0.f ? 8.0f * 2.3283064e-10f * nrm2 : -1.0f;
    unsigned pred = 0, amb = 0;
    auto decide = [&](float P, float T, int q) {
      const float diff = P - T;
      pred |= (diff > 0.f ? 1u : 0u) << q;
      amb |= ((diff * diff <= d2 * (P + T)) ? 1u : 0u) << q;
    };
    auto pair_power = [&](cf a, cf b, cf w, float& Pk, float& Pn) {
      cf p, q;
      split_pair(a, b, w, p, q);
      Pk = p.x * p.x + p.y * p.y;
      Pn = q.x * q.x + q.y * q.y;
    };
    // One instruction stream for all lanes (see k_apply_fast): decisions are made per ENTRY
    // (bin_of_entry); lane 0 selects its operands differently and its bits are permuted back to
    // register order afterwards.
    const bool l0 = c == 0;
    const cf wlo = wl;
    cf whi = wl;
    {
      const cf w16 = A.tw1024[16];
      if (l0) whi = {-w16.y, w16.x};  // i * w_1024^16
    }
    auto sel = [&](cf a0, cf a1) -> cf { return {l0 ? a0.x : a1.x, l0 ? a0.y : a1.y}; };
    bool pred512 = false, amb512 = false;
    {
      float Pk, Pn;
      pair_power(v[0], v[31], wlo, Pk, Pn);
      const cf a = v[0];
      const float x0 = 2.f * (a.x + a.y), xN = 2.f * (a.x - a.y);
      const float P256 = 4.f * (v[8].x * v[8].x + v[8].y * v[8].y);
      decide(l0 ? x0 * x0 : Pk, t2[0], 0);
      decide(l0 ? P256 : Pn, t2[31], 31);
      const float P5 = xN * xN, d5 = P5 - t2_512;
      pred512 = l0 && d5 > 0.f;
      amb512 = l0 && d5 * d5 <= d2 * (P5 + t2_512);
    }
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      const cf a = sl < 8 ? v[sl] : sel(v[8 + sl], v[sl]);
      const cf b = sl < 8 ? sel(v[16 - sl], v[31 - sl]) : sel(v[39 - sl], v[31 - sl]);
      const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
      float Pk, Pn;
      pair_power(a, b, w, Pk, Pn);
      decide(Pk, t2[sl], sl);
      decide(Pn, t2[31 - sl], 31 - sl);
    }
    if (!fvalid) { amb = 0; amb512 = false; }
    // exact re-evaluation, one cell at a time, whole wave cooperating
    while (true) {
      const unsigned long long pending = __ballot(amb != 0 || amb512);
      if (pending == 0) break;
      const int src = __ffsll((long long)pending) - 1;
      const unsigned amb_s = (unsigned)__shfl((int)amb, src);
      const int amb512_s = __shfl((int)amb512, src);
      const int q = amb_s ? (__ffs((int)amb_s) - 1) : 32;
      (void)amb512_s;
      const int cs = src & 15, gs = src >> 4;
      const int f = q < 32 ? bin_of_entry(cs, q) : 512;
      const double P = exact_power(A, row, chunk, tq + gs, f, lane);
      const bool pass = P > t2eff(f);
      if (lane == src) {
        if (q < 32) {
          pred = (pred & ~(1u << q)) | ((pass ? 1u : 0u) << q);
          amb &= ~(1u << q);
        } else {
          pred512 = pass;
          amb512 = false;
        }
      }
    }
    if (l0)  // entry -> register: e 0..7 -> 0..7, 8..23 -> 16..31, 24..30 -> 9..15, 31 -> 8
      pred = (pred & 0xffu) | ((pred & 0x00ffff00u) << 8) | ((pred >> 15) & 0xfe00u) | ((pred >> 23) & 0x100u);
    // pack: ballots over the wave give 16 consecutive bins per frame and slot.  Lane (g, c = m) keeps
    // the four ballots of word m with vector selects and slices out its frame's 16-bit fields with
    // per-lane shifts afterwards (doing the slicing per frame on the scalar unit cost ~450 SALU
    // instructions per wave; the CU has one scalar unit for all its waves).
    // 16 x 16 bit transpose of both halves of `pred` across the frame's 16 lanes (see k_gate_onepass): lane k then
    // holds entry k (low half) and entry 16 + k (high half) of lanes 0..15
    unsigned tr = pred;
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
    const unsigned long long b8 = __ballot(pred512);
    unsigned long long myword;  // lane c < 9 of group g ends up with word c of frame tq + g
    {
      const int sh = 16 * g;
      const int srcl = (lane & 48) | ((2 * c) & 15);
      const unsigned wa = (unsigned)__shfl((int)tr, srcl), wb2 = (unsigned)__shfl((int)tr, srcl + 1);
      const unsigned f0 = wa & 0xffffu;    // bins 64m      + c
      const unsigned f1 = wa >> 16;        // bins 64m + 16 + twisted
      const unsigned f2 = wb2 & 0xffffu;   // bins 64m + 32 + c
      const unsigned f3 = wb2 >> 16;       // bins 64m + 48 + twisted
      // row-2 slots hold bins 16, 31, 30, ..., 17 (c = 0, 1, ..., 15): undo the order
      const unsigned r1 = (((__brev(f1) >> 16) << 1) | (f1 & 1u)) & 0xffffu;
      const unsigned r3 = (((__brev(f3) >> 16) << 1) | (f3 & 1u)) & 0xffffu;
      myword = (unsigned long long)(f0 | (r1 << 16)) | ((unsigned long long)(f2 | (r3 << 16)) << 32);
      if (c == 8) myword = (b8 >> sh) & 1ull;
    }
    if (fvalid && c < 9) A.bits[(u * G.T + t) * (int64_t)A.wpr + c] = myword;
  }
}

}  // namespace fast
}  // namespace sg

// =======================================================================================
// Magnitude STFT for the non-stationary masks (default geometry): float32 fast core, |X| stored
// in natural bin order [unit][frame][FS].
// =======================================================================================
namespace sg {
namespace fast {

// |X| = sqrt(|2X|^2) / 2 with the bare v_sqrt_f32 (1 ulp): sqrtf's IEEE sequence -- scaling for denormal arguments,
// a Newton step -- is 8 instructions per bin, a twelfth of k_mag_fast.  Arguments below 1.2e-38 (magnitudes below
// 1e-19: DESIGN.md section 1, dynamic range) flush to zero.
__device__ __forceinline__ float half_sqrt(float P4) { return 0.5f * __builtin_amdgcn_sqrtf(P4); }

struct MagArgs {
  View view;
  Geom g;
  const float* win;
  const cf* tw512;
  const cf* tw1024;
  float* mag;  // [units][T][FS]
  // non-stationary gate: the recurrence partials of nonstat.hpp for this block's 16 frames ride along (k_iir_part read the
  // whole |X| field once more for them: 57 us and 287 MB of a 10-minute call).  sub == nullptr: magnitudes only.
  double iir_b;
  double* sub;  // [units][blocks][2][FS]: e = sum_t b c^(end-1-t) A[t],  E0 = sum_t b c^(t-start) s0[t]  (zero-state forward response s0)
};
constexpr int MAG_TILE_PITCH = 520;   // floats per frame row of the block's |X| tile in LDS (4 rows in each wave's exchange slice)

// (round 6: a FOURTH workgroup per CU -- 128 registers: 10 spilled, and the window read from global memory / L1 instead of LDS to
// get under 40 KB per workgroup -- measured 131 -> 151 us per ten minutes: DESIGN 8)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 3) void k_mag_fast(MagArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const Geom& G = A.g;
  const int64_t u = blockIdx.y;
  const int64_t row = (A.view.unit0 + u) / A.view.n_chunks;
  const int64_t chunk = A.view.c0 + (A.view.unit0 + u) % A.view.n_chunks;
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);
  // window table and (interior blocks of float32 input) the block's contiguous sample span in LDS, as
  // in k_apply_fast / k_decide_fast
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  stage_tables<WAVES * 64, 256>(tw512, A.tw512, swin, A.win, tid);
  constexpr int NFB = 4 * WAVES, SPAN = (NFB - 1) * 256 + 1024, XPITCH = 288;
  const int64_t tqb = (int64_t)blockIdx.x * NFB;
  // Every block stages its span: interior blocks of float32 input with 16-byte loads, the others (row edges,
  // other sample types) sample by sample through view_sample -- zero outside the readable range, like the frames
  // they belong to.  No per-lane gather path: short rows (TorchGate: 4 blocks per row, 2 of them at an edge) run
  // the same code as long ones.
  bool blk_vec;
  {
    const int64_t s0b = tqb * 256 - G.padL;
    const int64_t gb = chunk * A.view.cs - A.view.pad + s0b;
    const float* sp = (const float*)A.view.x + row * A.view.stride + gb;
    blk_vec = A.view.dtype == 0 && tqb + NFB <= G.T && s0b >= 0 && s0b + SPAN <= A.view.Lp &&
              gb >= A.view.lo && gb + SPAN <= A.view.hi && (reinterpret_cast<uintptr_t>(sp) & 15) == 0;
    float* xs = reinterpret_cast<float*>(regions);
    if (blk_vec) {
      stage_span_vec<WAVES * 64, SPAN, XPITCH>(xs, sp, tid);
    } else {
      for (int i = tid; i < SPAN; i += WAVES * 64)
        xs[(i >> 8) * XPITCH + (i & 255)] = (float)view_sample(A.view, row, chunk, s0b + i);
    }
  }
  __syncthreads();
  const int64_t tq = tqb + wave * 4;
  const int64_t t = tq + g;
  const bool fvalid = t < G.T;
  cf v[32];
  {
    const float* xs = reinterpret_cast<const float*>(regions) + (4 * wave + g) * XPITCH + 2 * c;
    const float2* wl2 = reinterpret_cast<const float2*>(swin + 2 * c);
    if (blk_vec) {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const float2 x2 = *reinterpret_cast<const float2*>(xs + (r >> 3) * XPITCH + 32 * (r & 7));
        const float2 w2 = wl2[16 * r];
        v[r] = {x2.x * w2.x, x2.y * w2.y};
      }
    } else {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        float2 x2 = *reinterpret_cast<const float2*>(xs + (r >> 3) * XPITCH + 32 * (r & 7));
        if (!fvalid) x2 = make_float2(0.f, 0.f);   // frames past the end of the row: zeros
        const float2 w2 = wl2[16 * r];
        v[r] = {x2.x * w2.x, x2.y * w2.y};
      }
    }
  }
  __syncthreads();  // the span may now be overwritten by the exchanges
  static_assert(4 * MAG_TILE_PITCH * sizeof(float) <= WAVE_CX_H * sizeof(cf), "four tile rows per exchange slice");
  float* trow = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + g * MAG_TILE_PITCH;   // this frame's row of the |X| tile
  const bool with_sub = A.sub != nullptr;
  if (tq < G.T) {
  fft512_fwd_half(v, fb, tw512, c);
  const bool l0 = c == 0;
  const cf wlo = A.tw1024[c];
  cf whi = wlo;
  {
    const cf w16 = A.tw1024[16];
    if (l0) whi = {-w16.y, w16.x};
  }
  auto sel = [&](cf a0, cf a1) -> cf { return {l0 ? a0.x : a1.x, l0 ? a0.y : a1.y}; };
  float* mrow = A.mag + (u * G.T + (fvalid ? t : 0)) * (int64_t)G.FS;
  // (with_sub: the rows go to the block's LDS tile only and are written to the field from there, 16 bytes per lane and
  // store, after the barrier below -- 2 stores per lane and frame row instead of 33 four-byte ones scattered over 64-byte runs)
  auto put = [&](int e, float P4) {  // |X| = sqrt(P4) / 2 at the bin of entry e
    const float m = half_sqrt(P4);
    if (with_sub) trow[bin_of_entry(c, e)] = m;   // (the wave's own exchange slice: its transform is done)
    else if (fvalid) mrow[bin_of_entry(c, e)] = m;
  };
  auto pair_power = [&](cf a, cf b, cf w, float& Pk, float& Pn) {
    cf p, q;
    split_pair(a, b, w, p, q);
    Pk = p.x * p.x + p.y * p.y;
    Pn = q.x * q.x + q.y * q.y;
  };
  {
    float Pk, Pn;
    pair_power(v[0], v[31], wlo, Pk, Pn);
    const cf a = v[0];
    const float x0 = 2.f * (a.x + a.y), xN = 2.f * (a.x - a.y);
    const float P256 = 4.f * (v[8].x * v[8].x + v[8].y * v[8].y);
    put(0, l0 ? x0 * x0 : Pk);
    put(31, l0 ? P256 : Pn);
    if (l0 && with_sub) trow[512] = 0.5f * fabsf(xN);
    else if (l0 && fvalid) mrow[512] = 0.5f * fabsf(xN);
  }
#pragma unroll
  for (int sl = 1; sl < 16; ++sl) {
    const cf a = sl < 8 ? v[sl] : sel(v[8 + sl], v[sl]);
    const cf b = sl < 8 ? sel(v[16 - sl], v[31 - sl]) : sel(v[39 - sl], v[31 - sl]);
    const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
    float Pk, Pn;
    pair_power(a, b, w, Pk, Pn);
    put(sl, Pk);
    put(31 - sl, Pn);
  }
  }   // tq < G.T
  if (!with_sub) return;
  // ---- the block's frames as one sub-tile of the time recurrence: float64 like k_iir_part ----
  __syncthreads();
  {
    const int n = (int)min<int64_t>((int64_t)NFB, G.T - tqb);   // frames of this block inside the unit
    // the block's |X| rows: LDS tile -> field, coalesced (row r of the tile = frame tqb + r; 128 float4 + bin 512 per row)
    {
      float* mbase = A.mag + (u * G.T + tqb) * (int64_t)G.FS;
      for (int i = tid; i < n * 128; i += WAVES * 64) {
        const int r = i >> 7, q = i & 127;
        const float* rowp = reinterpret_cast<const float*>(regions + (r >> 2) * WAVE_CX_H) + (r & 3) * MAG_TILE_PITCH;
        *reinterpret_cast<float4*>(mbase + (int64_t)r * G.FS + 4 * q) = *reinterpret_cast<const float4*>(rowp + 4 * q);
      }
      if (tid < n) {
        const float* rowp = reinterpret_cast<const float*>(regions + (tid >> 2) * WAVE_CX_H) + (tid & 3) * MAG_TILE_PITCH;
        mbase[(int64_t)tid * G.FS + 512] = rowp[512];
      }
    }
    const double b = A.iir_b, cc = 1.0 - b;
    static_assert(WAVES * 64 == 256, "bins tid, tid + 256 and (thread 0) 512");
    // three independent chains per thread, interleaved (the recurrence is serial in t, float64 fma latency ~8 cycles)
    const bool third = tid == 0;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, E00 = 0.0, E01 = 0.0, E02 = 0.0, pw = b;
    auto step = [&](int r) {
      const float* rowp = reinterpret_cast<const float*>(regions + (r >> 2) * WAVE_CX_H) + (r & 3) * MAG_TILE_PITCH;
      const double a0 = (double)rowp[tid], a1 = (double)rowp[tid + 256], a2 = third ? (double)rowp[512] : 0.0;
      e0 = b * a0 + cc * e0;
      e1 = b * a1 + cc * e1;
      e2 = b * a2 + cc * e2;
      E00 += pw * e0;
      E01 += pw * e1;
      E02 += pw * e2;
      pw *= cc;
    };
    if (n == NFB) {   // every block but a unit's last: straight-line (all 48 LDS reads up front)
#pragma unroll
      for (int r = 0; r < NFB; ++r) step(r);
    } else {
      for (int r = 0; r < n; ++r) step(r);
    }
    double* o = A.sub + ((u * gridDim.x + blockIdx.x) * 2) * (int64_t)G.FS;
    o[tid] = e0;
    o[tid + 256] = e1;
    o[G.FS + tid] = E00;
    o[G.FS + tid + 256] = E01;
    if (third) { o[512] = e2; o[G.FS + 512] = E02; }
  }
}

}  // namespace fast
}  // namespace sg
