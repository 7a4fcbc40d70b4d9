// One-pass stationary gate for the default geometry (n_fft = win = 1024, hop = 256, float32 transforms):
// every frame's forward transform is computed ONCE.
//
//   k_decide_fast + k_smooth_bits2 + k_apply_fast          (3 transforms per frame, K round trip in HBM)
//     ->  k_gate_onepass                                    (2 transforms per frame, no mask field in HBM)
//
// A workgroup owns one TILE of 16 consecutive frames of one unit (4 wavefronts x 4 frames, the register
// FFT of fastpath.hpp).  After the forward transform it decides its 16 x 513 cells (float32 + exact
// float64 refinement, bit-identical to k_decide_fast), and keeps the spectra IN REGISTERS while
//   1. the mask bits of the tile (16 x 9 words = 1152 B) are PUBLISHED to the neighbouring tiles through
//      a tile-blocked exchange buffer in HBM (write-through stores, drained, then an epoch-valued flag);
//   2. it waits for the flags of tiles j-1 and j+1 and reads their nt adjacent rows (nt <= 16: the time
//      half-width of the smoothing filter, base.py:115) -- 2*nt*72 B;
//   3. the (16 + 2 nt) x 513 bit tile is smoothed in LDS with the exact integer separable triangle
//      filter (the arithmetic of k_smooth_bits2: table lookups along f, weighted sums along t) into the
//      uint16 weight sums K of its own 16 frames;
// then x mask -> inverse transform -> window -> overlap-add -> store exactly as k_apply_fast<LEAN>.
// The smoothing buffers live in the exchange slices, which are idle between the two transforms: the
// kernel needs no more LDS than k_apply_fast (3 workgroups per CU).
//
// Inter-workgroup protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
// visibility"): payload and flag are 8/4-byte agent-scope relaxed atomics on both sides (sc1 stores and
// loads: write-through, L1-bypassing -- per-XCD L2s are not coherent), the payload stores are drained
// (s_waitcnt vmcnt(0) + workgroup barrier) before the flag is stored; every tile's payload is 1152 B =
// 9 x 128 B, 128-byte aligned: no cache line is shared between producers.
// Deadlock freedom does not rely on dispatch order: a workgroup takes a TICKET (atomic counter) when it
// starts and works on tile number `ticket`; it publishes before it waits for anything, so the only
// workgroup that can wait for a tile nobody has started yet is the one with the highest ticket, and
// every other resident workgroup finishes and frees its slot.
#pragma once
#include "fastpath.hpp"

namespace sg {
namespace fast {

constexpr int OP_XW = 9;                 // 64-bit words per frame row (513 bins)
constexpr int OP_TILE_WORDS = 16 * OP_XW * 2;  // payload of one tile: 288 tagged granules = 2304 B (18 x 128 B)
constexpr int OP_MAX_NT = 16;            // neighbours hold 16 frames each
constexpr int OP_KP = 528;               // K row pitch (uint16 entries) of a wave's private K tile
// The kernel's nine constant tables live in ONE device buffer at fixed offsets (OnePassArgs::tab): one base pointer in the
// kernel arguments instead of nine -- 16 scalar registers fewer in a kernel that spilled 44 of them.
constexpr int OP_TAB_WIN = 0, OP_TAB_WSQ = 4096, OP_TAB_INVN = 8192, OP_TAB_TW512 = 9216, OP_TAB_TW1024 = 13312,
              OP_TAB_WIN64 = 17408, OP_TAB_TW64 = 25600, OP_TAB_MCONST = 33792, OP_TAB_EXP8 = 35328, OP_TAB_BYTES = 37376;

#ifndef OP_LATE_ARGS
#define OP_LATE_ARGS 1   // 1: the epilogue re-reads its arguments from the kernel-argument segment (see "LATE ARGUMENTS" in the kernel)
#endif
#ifndef OP_TRACE
#define OP_TRACE 0
#endif
#ifndef OP_ARGCHECK
#define OP_ARGCHECK 0
#endif
#ifndef OP_TILECOUNT
#define OP_TILECOUNT 0
#endif
#ifndef OP_NO_LOOPTOP_WAIT
#define OP_NO_LOOPTOP_WAIT 0   // (diagnosis) 1: without the explicit LDS wait before the persistent loop's top barrier (the state that lost hand-offs)
#endif
#ifndef OP_BACKOFF
#define OP_BACKOFF 0   // (diagnosis) 1: a poll that has missed 8 times sleeps ~3.5 us between tries instead of 64 cycles (is it the polls' own traffic?)
#endif
#if OP_BACKOFF
#define OP_POLL_SLEEP(spin) do { if ((spin) >= 8) __builtin_amdgcn_s_sleep(127); else __builtin_amdgcn_s_sleep(1); } while (0)
#else
#define OP_POLL_SLEEP(spin) __builtin_amdgcn_s_sleep(1)
#endif
#ifndef OP_WHO
#define OP_WHO 0   // (diagnosis) 1: a poll that gives up leaves its tile's ticket, what it waited for and the tag it saw in the error words 8..13
#endif
#ifndef OP_GLOBAL_DRAW
#define OP_GLOBAL_DRAW 0   // (diagnosis) 1: the mid-tile ticket draws as GLOBAL atomics (the pointer re-read from the kernel-argument segment is a generic one: flat_atomic_add)
#endif
#if OP_GLOBAL_DRAW
#define OP_DRAW_TICKET(p64) __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned*)(p64), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define OP_DRAW_TICKET(p64) atomicAdd((unsigned*)(uintptr_t)(p64), 1u)
#endif
#ifndef OP_DRAW_TOP
#define OP_DRAW_TOP 0   // (diagnosis, with -DOP_PF_AT=6) 1: the persistent loop draws its next ticket at the loop top -- no ticket is held while the previous tile is finished
#endif
#ifndef OP_RESTAGE
#define OP_RESTAGE 0   // (diagnosis) 1: the persistent loop stages its LDS tables again in every iteration (is a table overwritten between tiles?)
#endif
#ifndef OP_MAX_ITERS
#define OP_MAX_ITERS 0   // (diagnosis) 1: the persistent instantiation on a grid of one workgroup per tile, no second iteration -- is it the LOOP or the code?
#endif
#ifndef OP_LATE_START_US
#define OP_LATE_START_US 0
#endif
#ifndef OP_EXP_STALL_NS
#define OP_EXP_STALL_NS 0   // development experiment (see OP_STAMP(4))
#endif
#ifndef OP_INPLACE
#define OP_INPLACE 0   // 1: split / merge in place after a one-off permutation of lane 0's registers (k_row_gate's formulation)
                       // instead of operand selects per pair.  Round 4: same instruction count (96 + 8 v_cndmask either way),
                       // same 168 VGPRs and SGPR spills in this kernel -- no gain here, kept for A/B builds
#endif
struct OnePassArgs {
  ApplyArgs A;              // view, geometry, output map, tables, seam buffer (A.K / A.Mf unused)
  // the caller's samples in their own dtype (A.view may be a float32 copy): exact refinement.  Only these three fields
  // differ from A.view (96 bytes of kernel arguments fewer than a second View)
  const void* x_exact;
  int64_t stride_exact;
  int dtype_exact;
  const char* tab;          // the constant tables (OP_TAB_*): float32 window, window^2, 1/envelope, w_512^(k1 c), w_1024^j,
                            // float64 window and w_1024^j (exact refinement), MFMA operands, byte -> 8 bytes expansion
  ThreshConsts tc;
  double mag_scale, top_db;
  unsigned long long* xbits;  // [units][n_tiles + 2][16][OP_XW][2] published mask bits: granules {32 bits, epoch}
  unsigned* ticket;           // work counter: never reset, a launch takes exactly units * (n_tiles + 2) tickets
  unsigned ticket_base;       // its value before this launch
  unsigned epoch;
  unsigned* err;              // host-mapped word: bit 0 / 1 = a bit / partial-hop hand-off timed out
  int nf, nt;
  float prop, inv_ktot;       // PROP instantiation: prop_decrease and 1 / ktot (A.kscale = 1/512 then)
  unsigned long long* part2;  // [units][n_tiles][3][256] trailing partial hops of every tile: granules {float, epoch}
  // In-kernel floor test (see "floor test" in the kernel): alim = bit pattern of the largest max|x| for which no band's
  // -top_db floor can be live (alim[2 .. 2 + OP_ALIM_BLOCKS), see fastpath.hpp), null when the flags in tc.need_floor were computed up front
  // (k_unit_absmax + k_prep_thresh).  The REDO instantiation is the second launch of such a call: only the units whose
  // test fired run.
  // alim[1]: tc.need_tag of the last call in which some unit reported: the second launch returns at once -- before tables
  // and ticket -- when it is another call's (7152 workgroups that only took their ticket and left cost 83 us:
  // tools/ubench/ticket_atomic.hip)
  unsigned* alim;
  int scan_q;                 // in-kernel floor test: samples of the unit window's unstaged part that each tile scans
  unsigned total_tiles;       // units * (n_tiles + 2): PERSIST workgroups draw tickets until they get one >= this
#if OP_TRACE
  unsigned* trace;                   // [workgroups][4 waves][16] shader cycles per phase (slot 15 = 1: tile completed): development builds
#endif
#if OP_WHO
  unsigned* who;                     // [ticket < 4096][4]: drawn by (workgroup + 1 | iteration << 16), while on tile, started by, past its bits poll -- development builds
#endif
};

// exact float64 |X[f]|^2 of frame t (see k_decide_fast).  A rare path (about one wave in fifty): with OP_LATE_ARGS its
// arguments -- a dozen 64-bit values of the view -- are read from the kernel-argument segment HERE (scalar loads through an
// opaque pointer) instead of staying live in scalar registers from the entry block to the decision stage.
#if OP_LATE_ARGS
#define OP_XARG(type, member) (*(const __attribute__((address_space(4))) type*)(kp4 + __builtin_offsetof(OnePassArgs, member)))
#else
#define OP_XARG(type, member) (P.member)
#endif
__device__ __forceinline__ double op_exact_power(const OnePassArgs& P, int64_t row, int64_t chunk, int64_t t, int f,
                                                 int lane) {
#if OP_LATE_ARGS
  const __attribute__((address_space(4))) char* kp4 = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp4));
#endif
  const int64_t s0 = t * OP_XARG(int32_t, A.g.H) - OP_XARG(int32_t, A.g.padL);
  const int64_t v_cs = OP_XARG(int64_t, A.view.cs), v_pad = OP_XARG(int64_t, A.view.pad), v_Lp = OP_XARG(int64_t, A.view.Lp),
                v_lo = OP_XARG(int64_t, A.view.lo), v_hi = OP_XARG(int64_t, A.view.hi), x_stride = OP_XARG(int64_t, stride_exact);
  const void* const x_exact = (const void*)(uintptr_t)OP_XARG(unsigned long long, x_exact);
  const int x_dtype = OP_XARG(int, dtype_exact);
  const char* const tab = (const char*)(uintptr_t)OP_XARG(unsigned long long, tab);
  double re = 0.0, im = 0.0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int m = lane + 64 * i;
    // the ORIGINAL samples: view_sample on A.view's geometry with the caller's pointer / dtype / stride
    double xs = 0.0;
    {
      const int64_t sp = s0 + m;
      const int64_t gi = chunk * v_cs - v_pad + sp;
      if (sp >= 0 && sp < v_Lp && gi >= v_lo && gi < v_hi) xs = load_sample(x_exact, x_dtype, row * x_stride + gi);
    }
    const double xv = xs * reinterpret_cast<const double*>(tab + OP_TAB_WIN64)[m];
    const int j = (f * m) & 1023;
    cx<double> w = reinterpret_cast<const cx<double>*>(tab + OP_TAB_TW64)[j & 511];
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

// OP_ABLATE (development only, default 0; results are wrong): 1 no decision stage, 2 no wait for the
// neighbours' flags, 8 no exact refinement, 16 no wait for the previous tile's trailing hops
#ifndef OP_ABLATE
#define OP_ABLATE 0
#endif
// OP_TRACE (development only): per-phase shader-clock stamps of every non-halo wave, summed into P.trace (tools/
// trace_onepass.sh): where a tile's lifetime goes
#ifndef OP_TRACE
#define OP_TRACE 0
#endif
#if OP_TRACE
#define OP_STAMP(i)                                                                                   \
  do {                                                                                                \
    const long long t_now_ = clock64();                                                               \
    if (lane == 0 && t_slot_) t_slot_[i] = (unsigned)(t_now_ - t_prev_);                              \
    t_prev_ = t_now_;                                                                                 \
  } while (0)
#else
#define OP_STAMP(i) do { } while (0)
#endif

// PROP: prop_decrease < 1 (stationary.py:108-114 applies it BEFORE the smoothing: mask = p K / ktot + (1 - p) edge,
// edge = the smoothing filter's weight inside the spectrogram -- 1 except near its borders)
// LOSE (tests only, SG_OPT_INJECT_HANDOFF_FAULT bits 3..4): an instantiation whose polls give up at once -- the timeout
// branch of every hand-off (error word, NaN-poisoned hops) runs without a second of spinning and without a test
// argument in the product kernel, whose register allocation sits at the 168-VGPR edge.
// REDO: the second launch of a call with the in-kernel floor test (its own instantiation: its launches are their own row in
// a profile -- when no chunk reported they return at once, and would halve the gate's average duration -- and the first
// launch carries no test for it).
// PERSIST (round 6): a workgroup LOOPS over tickets instead of handling one tile.  What that buys -- a tile's first two phases
// (ticket atomic + constant tables: one memory round trip; span loads + compare constants: a second one) were 11.4 k of the
// 58.7 k cycles a wave lives (profiles/r03_onepass_phase_trace.txt), pure memory latency that the other two workgroups of the
// CU only partly cover (the kernel runs tiles x lifetime / 768 slots almost exactly):
//   * twiddles, window, byte-expansion table, compare constants (a lazy first launch has need == 0 in every unit: they do not
//     depend on the tile) and the floor test's bound are staged ONCE per workgroup;
//   * the NEXT ticket is drawn by thread 0 right after the neighbours' bits have arrived (never earlier: a workgroup that
//     held ticket j + 1 while waiting for tile j + 1's bits would wait for itself) and travels through LDS under the
//     smoothing stage's closing barrier;
//   * the next tile's span (five 16-byte loads per thread) and its slice of the floor test are issued after the inverse
//     transform, when the 64 spectrum registers are about to die, and land under the overlap-add and the epilogue;
//     the loop top only moves them from registers to LDS.
// Round 4's attempt at this died of loop-carried kernel arguments (160 spilled SGPRs).  Here NOTHING of the argument struct
// is loop-carried: every iteration reads `P` through late_args() -- an opaque pointer to the kernel-argument segment made
// inside the loop body, so no load from it can be hoisted -- and thread / lane indices come from an opaque copy of
// threadIdx.x.  Loop-carried: the parity of the ticket slot, the "prefetched" flag, one VGPR of floor-test bound and the 25
// VGPRs of prefetched samples.
// Deadlock freedom as before (tickets; publish before wait); a launch needs two resident workgroups, and draws
// total_tiles + gridDim.x tickets (every workgroup ends on one ticket >= total_tiles).
// Not for REDO / LOSE / a-priori floor flags (compare constants depend on the unit there) / SG_OPT_TILE_ORDER 1.
#ifndef OP_LATE_P
#define OP_LATE_P 1   // 1: every instantiation reads its arguments through late_args() (0 SGPR spills; 0: the by-value struct outside PERSIST)
#endif
#ifndef OP_OCC
#define OP_OCC 3   // workgroups per CU the register budget is set for (development builds: 2 shows the unconstrained pressure)
#endif
#define OP_DONE { if (PERSIST) continue; return; }
template <int WAVES, bool PROP, bool LOSE = false, bool REDO = false, bool PERSIST = false>
__global__ __launch_bounds__(WAVES * 64, OP_OCC) void k_gate_onepass(OnePassArgs Pk) {
  static_assert(WAVES == 4, "tile = 16 frames");
  static_assert(!PERSIST || (!LOSE && !REDO), "the persistent loop is the lazy first launch only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  cf* regions = tw512 + FN;
  float* swin = reinterpret_cast<float*>(regions + WAVES * WAVE_CX_H);
  float* s_t2 = swin + 1024;
  unsigned long long* s_exp = reinterpret_cast<unsigned long long*>(s_t2 + T2_FLOATS);
  double* s_t2d = reinterpret_cast<double*>(s_exp + 256);   // exact compare constants (refinement), same order
  unsigned* s_misc = reinterpret_cast<unsigned*>(s_t2d + 514);   // [0] ticket, [1] lost hand-off, [4], [5]: PERSIST ticket slots
  unsigned char* s_ef = reinterpret_cast<unsigned char*>(s_misc + 8);  // PROP: integer weight (<= 81) of the valid taps along f, per bin
  constexpr int NF = 4 * WAVES;
  constexpr int SPAN = (NF - 1) * 256 + 1024, XPITCH = 288;
  static_assert((SPAN / 256) * XPITCH <= WAVES * WAVE_CX_H * 2, "span must fit the exchange slices");
  // span loads: one dword per thread and 256-sample row (lane-contiguous, 19 per thread).  16-byte loads (five per thread)
  // need aligned register QUADS; held across the second half of a tile (PERSIST) the allocator could not place them next to
  // the 64 spectrum registers and spilled all of them -- a scratch store that waits for the load it was meant to hide
  constexpr int NQ = SPAN / 256;
  static_assert(SPAN % 256 == 0 && WAVES * 64 == 256, "one sample per thread and row");
  constexpr int SCAN_REG = 5;                     // floor-test slices up to 5 x 256 samples ride in registers

  if (REDO && Pk.alim[1] != Pk.tc.need_tag) return;   // second launch of a call none of whose units reported (the common case)
#if OP_LATE_START_US
  // (diagnosis) some workgroups of the persistent grid start late, as they do when another kernel holds their compute unit
  if (PERSIST && (blockIdx.x % 5u) == 1u) {
    const unsigned long long t0_ = wall_clock64();
    while ((long long)(wall_clock64() - t0_) < (long long)OP_LATE_START_US * 100 * (1 + (blockIdx.x % 7u))) __builtin_amdgcn_s_sleep(8);
  }
#endif
  double t2pre[3];   // compare constants of entries tid, tid + 256 and 512 (they do not depend on the ticket)
  unsigned alim_v = 0u;
  {
    const int tid = threadIdx.x, lane = tid & 63;
    // table loads first, the ticket's atomic behind them in the same queue: one memory round trip, not two
    static_assert(FN == 2 * WAVES * 64 && WAVES * 64 == 256, "one pass of the prologue loads per thread");
    const cf tw_a = reinterpret_cast<const cf*>(Pk.tab + OP_TAB_TW512)[(tid >> 4) * (tid & 15)];
    const cf tw_b = reinterpret_cast<const cf*>(Pk.tab + OP_TAB_TW512)[((tid + 256) >> 4) * (tid & 15)];
    const float4 w4 = reinterpret_cast<const float4*>(Pk.tab + OP_TAB_WIN)[tid];
    const unsigned long long e8 = reinterpret_cast<const unsigned long long*>(Pk.tab + OP_TAB_EXP8)[tid];
    t2pre[0] = Pk.tc.T2[perm_inv(tid)];
    t2pre[1] = Pk.tc.T2[perm_inv(tid + 256)];
    t2pre[2] = Pk.tc.T2[perm_inv(512)];
    if (!REDO && Pk.alim != nullptr) {
      // the floor test's compare constant: a VECTOR load behind the table loads (as a scalar load the compiler places it
      // at its use, after the span has landed: one more exposed round trip per tile, 5 us of the kernel)
      int z = 2 + min(lane & 15, OP_ALIM_BLOCKS - 1);   // one bound per band block of the noise statistics: minimum below
      asm volatile("" : "+v"(z));
      alim_v = Pk.alim[z];
    }
    if (tid == 0) {
      // (ticket_base == 0xffffffff: SG_OPT_TILE_ORDER 1, the block index instead of a ticket)
      s_misc[PERSIST ? 4 : 0] = Pk.ticket_base == 0xffffffffu ? blockIdx.x : atomicAdd(Pk.ticket, 1u) - Pk.ticket_base;
      s_misc[1] = 0u;   // set when a hand-off of this tile is lost: its output hops are POISONED (NaN), never plausible garbage
    }
    tw512[tid] = tw_a;
    tw512[tid + 256] = tw_b;
    reinterpret_cast<float4*>(swin)[tid] = w4;
    s_exp[tid] = e8;
    if constexpr (PROP) {
      for (int f = tid; f <= 512; f += WAVES * 64) {
        const int lo = max(-Pk.nf, -f), hi = min(Pk.nf, 512 - f);
        int sum = 0;
        for (int a = lo; a <= hi; ++a) sum += Pk.nf + 1 - (a < 0 ? -a : a);
        s_ef[f] = (unsigned char)sum;
      }
    }
    if constexpr (PERSIST) {
      // need == 0 in every unit of a lazy first launch: one set of compare constants for all of the workgroup's tiles
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int i = tid + k * WAVES * 64;
        if (i > 512) break;
        s_t2[t2_pos(i)] = t2_to_f32(t2pre[k], 4.0);
        s_t2d[i] = t2pre[k];
      }
      for (int off = 1; off < 16; off <<= 1) alim_v = min(alim_v, (unsigned)__shfl_xor((int)alim_v, off));
      if (tid == 0) s_misc[2] = alim_v;   // (not a loop-carried register: it would live in scratch)
#if OP_TILECOUNT == 2
      if (tid == 0) { s_misc[3] = Pk.total_tiles; s_misc[6] = (unsigned)(uintptr_t)Pk.err; s_misc[7] = (unsigned)((uintptr_t)Pk.err >> 32); }
#endif
    }
  }
  __syncthreads();

  // what one tile reads from the recording: its span, and its slice of the floor test's scan (see "floor test" below).
  // A function of the ticket alone: the loop top evaluates it for the tile at hand, the prefetch for the next one.
  struct TileSrc {
    const float* sp;        // first sample of the span
    bool blk_vec;           // interior tile of float32 input: five aligned 16-byte loads per thread
    const float* sc_base;   // floor-test slice that rides in registers (null: none, or the slow loop)
    int sc_n1;              // its length - 1
    int64_t s0b, sc_lo, sc_last, sc_lenA, sc_c0, sc_c1;
  };
  auto tile_src = [](const OnePassArgs& P, int64_t row, int64_t chunk, int jt, bool test) -> TileSrc {
    const ApplyArgs& A = P.A;
    TileSrc S;
    const int64_t tf_tile = A.h_begin - 3 + (int64_t)jt * NF;
    S.s0b = tf_tile * 256 - A.g.padL;
    const int64_t gb = chunk * A.view.cs - A.view.pad + S.s0b;
    S.sp = (const float*)A.view.x + row * A.view.stride + gb;
    S.blk_vec = A.view.dtype == 0 && tf_tile >= 0 && tf_tile + NF <= A.g.T && S.s0b >= 0 && S.s0b + SPAN <= A.view.Lp &&
                gb >= A.view.lo && gb + SPAN <= A.view.hi && (reinterpret_cast<uintptr_t>(S.sp) & 15) == 0;
    S.sc_base = nullptr;
    S.sc_n1 = 0;
    S.sc_lo = S.sc_last = S.sc_lenA = S.sc_c0 = S.sc_c1 = 0;
    if (test) {
      const int64_t g0 = chunk * A.view.cs - A.view.pad;
      const int64_t s_lo = max<int64_t>(0, A.view.lo - g0), s_hi = min<int64_t>(A.view.Lp, A.view.hi - g0);
      const int64_t sp0 = (A.h_begin - 3 - NF) * 256 - A.g.padL;                                   // tile -1's span begins
      const int64_t sp1 = (A.h_begin - 3 + (int64_t)A.n_tiles * NF) * 256 - A.g.padL + SPAN;       // tile n_tiles' span ends
      const int64_t first = min(s_hi, max(s_lo, sp0));
      S.sc_lo = s_lo;
      S.sc_last = max(s_lo, min(s_hi, sp1));
      S.sc_lenA = first - s_lo;
      S.sc_c0 = (int64_t)(jt + 1) * P.scan_q;
      S.sc_c1 = min(S.sc_c0 + P.scan_q, S.sc_lenA + (s_hi - S.sc_last));
      if (A.view.dtype == 0 && S.sc_c1 > S.sc_c0 && S.sc_c1 - S.sc_c0 <= SCAN_REG * WAVES * 64 &&
          (S.sc_c1 <= S.sc_lenA || S.sc_c0 >= S.sc_lenA)) {
        S.sc_base = (const float*)A.view.x + row * A.view.stride + g0 +
                    (S.sc_c1 <= S.sc_lenA ? s_lo + S.sc_c0 : S.sc_last + (S.sc_c0 - S.sc_lenA));
        S.sc_n1 = (int)(S.sc_c1 - S.sc_c0) - 1;
      }
    }
    return S;
  };
  // loop-carried (PERSIST): the samples of the tile at hand, issued by the previous iteration
  float q[NQ];
  float sc[SCAN_REG];
  bool pf = false;

#if OP_EXP_STALL_NS
  const unsigned long long op_exp_t0_ = wall_clock64();
#endif
  for (unsigned iter = 0; iter == 0u || PERSIST; ++iter) {
  // ---- PERSIST: nothing of the arguments or the thread's indices survives an iteration (see above) ----
  const OnePassArgs& P = (PERSIST || OP_LATE_P) ? *late_args<OnePassArgs>() : Pk;
  int tid = threadIdx.x;
#if OP_ARGCHECK
  // (diagnosis) do the arguments re-read from the kernel-argument segment still equal the ones the kernel started with?
#if OP_ARGCHECK == 2
  if (PERSIST && threadIdx.x == 1) {   // every word of the argument struct, by a thread that draws no ticket
    const unsigned* a_ = reinterpret_cast<const unsigned*>(&P);
    const unsigned* b_ = reinterpret_cast<const unsigned*>(&Pk);
    bool same_ = true;
    for (unsigned w_ = 0; w_ < sizeof(OnePassArgs) / 4; ++w_) same_ = same_ && a_[w_] == b_[w_];
    if (!same_) atomicOr_system(Pk.err, 0x80u);
  }
#else
  if (PERSIST && threadIdx.x == 0 && (P.total_tiles != Pk.total_tiles || P.epoch != Pk.epoch || P.xbits != Pk.xbits || P.A.view.x != Pk.A.view.x))
    atomicOr_system(Pk.err, 0x80u);
#endif
#endif
#if OP_TRACE
  long long t_prev_ = clock64();   // (PERSIST: phase 0 = the wait at the loop-top barrier)
  unsigned* t_slot_ = nullptr;   // known once the ticket is
#endif
  if constexpr (PERSIST) {
    asm volatile("" : "+v"(tid));
    // (round 6, found with the builds listed in DESIGN 3: on the halo tiles' path to this barrier -- s_misc[next slot] = ticket;
    // continue -- the compiler emits NO s_waitcnt lgkmcnt(0) between the ds_write and the s_barrier (gfx950 has no automatic
    // wait before a barrier; the normal path's barrier has one).  Alone on the GPU the LDS write always landed before the other
    // waves' read of the slot; next to a kernel that keeps the LDS queues busy a wave could read the slot's PREVIOUS content --
    // the ticket of two tiles ago -- and the workgroup ran a tile with two different tickets: mis-gated tiles, lost hand-offs,
    // wild addresses.  The wait is explicit now.)
    if (iter != 0u) {
#if !OP_NO_LOOPTOP_WAIT
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      __syncthreads();   // the previous tile's epilogue has read every hop accumulator: the slices are free
    }
#if OP_DRAW_TOP
    if (iter != 0u) {
      if (threadIdx.x == 0) s_misc[4u + (iter & 1u)] = atomicAdd(P.ticket, 1u) - P.ticket_base;
      __syncthreads();
    }
#endif
  }
  const ApplyArgs& A = P.A;
  const Geom& G = A.g;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const unsigned tk_slot = PERSIST ? 4u + (iter & 1u) : 0u;
  const int ntt = A.n_tiles + 2;                 // tiles per unit incl. one decide-only halo tile per side
  const unsigned ticket = PERSIST ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_misc[tk_slot]) : s_misc[0];
#if OP_TILECOUNT == 2
  // (diagnosis) does the late read of total_tiles still equal what the workgroup saw when it started?  Reported through the err pointer
  // stashed in LDS at the start (not a late read, and no by-value argument inside the loop)
  if (PERSIST && threadIdx.x == 0 && P.total_tiles != s_misc[3]) {
    unsigned* e_ = (unsigned*)(((uintptr_t)s_misc[7] << 32) | (uintptr_t)s_misc[6]);
    atomicOr_system(e_, 0x40u);
    atomicMax_system(e_ + 6, P.total_tiles);
    atomicMax_system(e_ + 7, iter);
  }
  if (PERSIST && ticket >= s_misc[3]) return;
#endif
#if OP_WHO
  if (PERSIST && threadIdx.x == 0 && ticket < 4096u) P.who[ticket * 4 + 2] = (blockIdx.x + 1u) | (iter << 16);   // who started the tile
#endif
  if (PERSIST && ticket >= P.total_tiles) return;   // (workgroup-uniform)
#if OP_TILECOUNT == 1
  if (PERSIST && threadIdx.x == 0) { atomicAdd_system(P.err + 4, 1u); atomicAdd_system(P.err + 5, ticket); }   // (diagnosis) every ticket taken up
#endif
#if OP_RESTAGE
  if (PERSIST && iter != 0u) {
    const int t_ = threadIdx.x;
    tw512[t_] = reinterpret_cast<const cf*>(P.tab + OP_TAB_TW512)[(t_ >> 4) * (t_ & 15)];
    tw512[t_ + 256] = reinterpret_cast<const cf*>(P.tab + OP_TAB_TW512)[((t_ + 256) >> 4) * (t_ & 15)];
    reinterpret_cast<float4*>(swin)[t_] = reinterpret_cast<const float4*>(P.tab + OP_TAB_WIN)[t_];
    s_exp[t_] = reinterpret_cast<const unsigned long long*>(P.tab + OP_TAB_EXP8)[t_];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = t_ + k * WAVES * 64;
      if (i > 512) break;
      const double v_ = P.tc.T2[perm_inv(i)];
      s_t2[t2_pos(i)] = t2_to_f32(v_, 4.0);
      s_t2d[i] = v_;
    }
  }
#endif
  if (PERSIST && tid == 0) {
    s_misc[1] = 0u;
#if OP_TK_LATE
    s_misc[tk_slot ^ 1u] = 0xffffffffu;   // "the next ticket has not arrived yet"
#endif
  }
#if OP_TRACE
  t_slot_ = P.trace + ((size_t)ticket * 4 + wave) * 16;
  if (lane == 0) t_slot_[14] = (blockIdx.x + 1u) | (iter << 16);   // who took the ticket up (sg_debug_counter 8 reads it after a lost hand-off)
#endif
  OP_STAMP(0);   // tables + ticket
  const int64_t u = ticket / (unsigned)ntt;
  const int jt = (int)(ticket % (unsigned)ntt) - 1;   // -1 and n_tiles: halo tiles
  const bool halo_tile = jt < 0 || jt >= A.n_tiles;
  // (32-bit division: unit indices fit, and a 64-bit divide is ~150 instructions of every workgroup's prologue)
  const unsigned gu = (unsigned)(A.view.unit0 + u), nch = (unsigned)A.view.n_chunks;
  const int64_t row = gu / nch;
  const int64_t chunk = A.view.c0 + gu % nch;
  // Floor flags of the unit.  lazy (P.alim set): the first launch assumes "not live" and runs the floor test on the
  // samples it stages (below); the second launch (REDO) serves exactly the units whose test fired.
  const bool lazy = PERSIST || P.alim != nullptr;
  const int need = (lazy && !REDO) ? 0 : need_of(P.tc, u);
  if (REDO && need == 0) return;   // whole workgroup
  // (first launch: the second launch's work counter -- it only counts when a unit reported -- starts from zero)
  if (!REDO && lazy && ticket == 0u && tid == 0) P.ticket[8] = 0u;
  const bool floor_live = need == 1;

  // compare constants (x4: the split works on 2X), permuted like the lanes' entries -- see k_decide_fast
  auto t2eff = [&](int f, double v) -> double {
    if (floor_live) {
      double fl = cell_db(P.tc.pmax[u * G.FS + f], P.mag_scale) - P.top_db;
      if (fl > P.tc.thresh[f]) v = -1.0;
    }
    if (need & 2) v = T2_NEVER;   // non-finite sample in the unit (wins over 1: lazy tiles OR their verdicts together)
    return v;
  };
  const int64_t tf_tile = A.h_begin - 3 + (int64_t)jt * NF;  // first frame of the tile (abutting tiles)
  const int64_t tq = tf_tile + 4 * wave;
  const int64_t t = tq + g;                                  // this lane group's frame
  const bool fvalid = t >= 0 && t < G.T;
  cf* fb = regions + wave * WAVE_CX_H + frame_base_h(g);

  // ---- stage the tile's contiguous sample span: interior tiles of float32 input with 16-byte loads, the others
  // (unit edges, halo tiles) sample by sample through view_sample -- zero outside the readable range ------------
  bool blk_vec;
  {
    // ---- floor test (lazy, first launch).  k_unit_absmax + k_prep_thresh read the whole recording before the gate to
    // decide, per unit, whether _amp_to_db's floor max(dB, band max - top_db) (spectralgate/utils.py:16) can lift a band
    // over its threshold: max|x| sum|w| bounds every |X|.  The tiles of a unit stage nearly all of its window anyway:
    // each compares the largest sample it staged with the same bound (alim).  The rest of the window -- the chunk's
    // padding, which no tile of THIS unit transforms: A = [s_lo, first span) and B = [last span's end, s_hi) -- is dealt
    // to the unit's tiles in slices of P.scan_q samples of A ++ B (~1200 at the default chunking: five loads per thread,
    // in flight with the span's).  A tile whose test fires reports its unit: need_floor[u] |= 1 (2: a non-finite
    // sample), the unit's band maxima cleared for the float64 pre-pass that follows.  This launch's result for a reported
    // unit is overwritten by the second (REDO).
    const bool test = !REDO && lazy;
    const TileSrc S = tile_src(P, row, chunk, jt, test);
    blk_vec = S.blk_vec;
    const int64_t s0b = S.s0b;
    auto fill_t2 = [&]() {
      if constexpr (!PERSIST) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int i = tid + k * WAVES * 64;
          if (i > 512) break;
          const double v = t2eff(perm_inv(i), t2pre[k]);
          s_t2[t2_pos(i)] = t2_to_f32(v, 4.0);
          s_t2d[i] = v;   // the rare exact re-evaluation compares against this (no log10 in the hot kernel body)
        }
      }
    };
    float* xs = reinterpret_cast<float*>(regions);
    unsigned mi = 0u;   // max |x| seen by this thread, as a bit pattern (sign cleared: NaN / Inf order above every finite value)
    auto ab = [](float x) -> unsigned { return __float_as_uint(x) & 0x7fffffffu; };
    // issued BEHIND the span's loads (every later phase waits for the span), consumed after them.  One scalar base + a
    // clamped lane offset: the slice's tail re-reads its last sample (a predicated load is a branch and a copy)
    if (!pf) {
      if (S.blk_vec) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = S.sp[tid + k * 256];
      }
      if (S.sc_base != nullptr) {
#pragma unroll
        for (int k = 0; k < SCAN_REG; ++k) sc[k] = S.sc_base[min(tid + k * WAVES * 64, S.sc_n1)];
      }
    }
    fill_t2();   // (span loads in flight while the compare constants are built)
    if (blk_vec) {
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        xs[k * XPITCH + tid] = q[k];
        mi = max(mi, ab(q[k]));
      }
    } else {
      for (int i = tid; i < SPAN; i += WAVES * 64) {
        const float xv = (float)view_sample(A.view, row, chunk, s0b + i);
        xs[(i >> 8) * XPITCH + (i & 255)] = xv;
        mi = max(mi, ab(xv));
      }
    }
    if (test) {
      if (S.sc_base != nullptr) {
#pragma unroll
        for (int k = 0; k < SCAN_REG; ++k) mi = max(mi, ab(sc[k]));
      } else {
        // the slice that straddles A | B, long slices (padding much longer than the kept part), other sample types: a
        // loop after the span
        for (int64_t i = S.sc_c0 + tid; i < S.sc_c1; i += WAVES * 64)
          mi = max(mi, ab((float)view_sample(A.view, row, chunk, i < S.sc_lenA ? S.sc_lo + i : S.sc_last + (i - S.sc_lenA))));
      }
      unsigned alim_t = PERSIST ? s_misc[2] : alim_v;
      if constexpr (!PERSIST)
        for (int off = 1; off < 16; off <<= 1) alim_t = min(alim_t, (unsigned)__shfl_xor((int)alim_t, off));
      const bool hit = mi >= alim_t;
      if (__any(hit)) {   // wave-uniform, rare
        const bool nonfinite = __any(mi >= 0x7f800000u);
        double* pm = const_cast<double*>(P.tc.pmax) + u * G.FS;
        for (int f = lane; f < G.FS; f += 64) pm[f] = 0.0;
        if (lane == 0) {
          atomicMax(reinterpret_cast<unsigned*>(const_cast<int*>(P.tc.need_floor)) + u, (P.tc.need_tag << 2) | (nonfinite ? 2u : 1u));
          P.alim[1] = P.tc.need_tag;
          P.err[1] = P.epoch;   // host-mapped: "a unit of launch `epoch` reported" (the host picks the a-priori test next time)
        }
      }
    }
  }
  __syncthreads();  // tables, compare constants and span staged
  OP_STAMP(1);   // span + compare constants

  // ---- gather: v[r] = (x[2c + 32r], x[2c + 32r + 1]) * window ---------------------------------------
  cf v[32];
  float nrm2 = 0.f;
  {
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
        if (!fvalid) x2 = make_float2(0.f, 0.f);   // frames before / past the unit: zeros
        const float2 w2 = wl[16 * r];
        v[r] = {x2.x * w2.x, x2.y * w2.y};
      }
    }
  }
  __syncthreads();  // every lane has its samples: the span may be overwritten by the exchanges
  OP_STAMP(2);   // gather
  {
#pragma unroll
    for (int r = 0; r < 32; ++r) nrm2 += v[r].x * v[r].x + v[r].y * v[r].y;
    nrm2 += __shfl_xor(nrm2, 1);
    nrm2 += __shfl_xor(nrm2, 2);
    nrm2 += __shfl_xor(nrm2, 4);
    nrm2 += __shfl_xor(nrm2, 8);
  }
  {
    // an opaque ZERO OFFSET (not an opaque pointer: that would lose the LDS address space and turn every
    // access into a FLAT instruction) keeps the loop-invariant twiddle reads from being hoisted and pinned
    int z0 = 0;
    asm volatile("" : "+v"(z0));
    fft512_fwd_half(v, fb, tw512 + z0, c);
  }
  OP_STAMP(3);   // forward transform

  // ---- decide (k_decide_fast): mask bits of this lane's 32 entries -----------------------------------
  // ---- real-FFT split, ONCE: conjugate pair (a, b) = (Zc[k], Zc[N-k]) -> X2[k] = E + w O, conj-pair value
  // X2n = E - w O (both 2x the spectrum; E = a + conj b, O = (a - conj b)/i).  Slot order (see k_apply_fast):
  // lanes >= 1 pair registers (sl, 31 - sl); lane 0 pairs its self-conjugate rows differently, handled by
  // selects on the way in HERE only -- the decision stage and the mask stage both read pa/pb in slot order.
#if OP_INPLACE
  // In place (rowgate.hpp's formulation): lane 0 permutes its registers ONCE so that register index == entry index in
  // every lane, then split / merge work on (v[s], v[31 - s]) without operand selects -- ~100 v_cndmask fewer per wave
  // and the 64 registers of the transform are the 64 split values (no second array alive beside them).
  // pa[sl] / pb[sl] below are v[sl] / v[31 - sl]; lane 0 keeps its unpaired registers raw in v[0] (bins 0 / 512) and
  // v[31] (bin 256).  Same split_pair / merge_pair arithmetic in the same order: bit-identical results.
  const bool l0 = c == 0;
  cf wlo = reinterpret_cast<const cf*>(P.tab + OP_TAB_TW1024)[c];
  asm volatile("" : "+v"(wlo.x), "+v"(wlo.y));
  cf whi = wlo;
  {
    const cf w16 = reinterpret_cast<const cf*>(P.tab + OP_TAB_TW1024)[16];
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
#define OP_PA(sl) v[sl]
#define OP_PB(sl) v[31 - (sl)]
  const cf raw0 = v[0], raw8 = v[31];   // (lane 0 only: the same registers)
#else
  cf pa[16], pb[16];
  const cf raw0 = v[0], raw8 = v[8];   // lane 0: bins 0 / 512 and bin 256 are not part of a pair
  const bool l0 = c == 0;
  cf wlo = reinterpret_cast<const cf*>(P.tab + OP_TAB_TW1024)[c];
  asm volatile("" : "+v"(wlo.x), "+v"(wlo.y));
  cf whi = wlo;
  {
    const cf w16 = reinterpret_cast<const cf*>(P.tab + OP_TAB_TW1024)[16];
    if (l0) whi = {-w16.y, w16.x};  // i * w_1024^16
  }
  {
    auto sel = [&](cf a0, cf a1) -> cf { return {l0 ? a0.x : a1.x, l0 ? a0.y : a1.y}; };
    auto split = [&](cf a, cf b2, cf w, cf& xa, cf& xb) { split_pair(a, b2, w, xa, xb); };
    split(v[0], v[31], wlo, pa[0], pb[0]);
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      const cf a = sl < 8 ? v[sl] : sel(v[8 + sl], v[sl]);
      const cf b2 = sl < 8 ? sel(v[16 - sl], v[31 - sl]) : sel(v[39 - sl], v[31 - sl]);
      const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
      split(a, b2, w, pa[sl], pb[sl]);
    }
  }

#define OP_PA(sl) pa[sl]
#define OP_PB(sl) pb[sl]
#endif
  OP_STAMP(4);   // split
#if OP_EXP_STALL_NS
  // (experiment, tools/experiments/stats_in_launch.sh: what a first-round tile would lose waiting at its decision stage for
  // thresholds that an in-launch statistics chain publishes OP_EXP_STALL_NS after the launch starts; results unchanged)
  if (PERSIST && iter == 0u)
    while ((long long)(wall_clock64() - op_exp_t0_) * 10 < (long long)OP_EXP_STALL_NS) __builtin_amdgcn_s_sleep(2);
#endif
  // ---- decide (k_decide_fast): mask bits of this lane's 32 entries -----------------------------------
  unsigned long long myword;  // lane c < 9 of group g: word c of frame tq + g
  {
    // compare constants are read from LDS where they are used (the spectra stay live through this phase:
    // 32 more registers for a preloaded table would spill)
    int zt = 0;
    asm volatile("" : "+v"(zt));
    const float* t2 = s_t2 + c * T2_PITCH + zt;
    const float t2_512 = s_t2[T2_POS512];
    const float d2 = nrm2 > 0.f ? 8.0f * 2.3283064e-10f * nrm2 : -1.0f;
    // Decisions as SIGN BITS shifted into accumulators (v_alignbit: acc = acc << 1 | sign), no compares or selects:
    //   passes    <=>  T - P < 0                      (sign of nd)
    //   ambiguous <=>  d2 (P + T) - (T - P)^2 >= 0    (sign of e CLEAR)
    // Slot sl decides entry sl (accumulators A, in slot order: bit 15 - sl) and entry 31 - sl (accumulators B: bit
    // 15 - sl = entry 16 + that bit, already in place); A is bit-reversed at the end.
    unsigned pA = 0, pB = 0, nA = 0, nB = 0;
    auto decideA = [&](float Pw, float T) {
      const float nd = T - Pw;
      const float e = fmaf(-nd, nd, d2 * (Pw + T));
      pA = __builtin_amdgcn_alignbit(pA, __float_as_uint(nd), 31);
      nA = __builtin_amdgcn_alignbit(nA, __float_as_uint(e), 31);
    };
    auto decideB = [&](float Pw, float T) {
      const float nd = T - Pw;
      const float e = fmaf(-nd, nd, d2 * (Pw + T));
      pB = __builtin_amdgcn_alignbit(pB, __float_as_uint(nd), 31);
      nB = __builtin_amdgcn_alignbit(nB, __float_as_uint(e), 31);
    };
    bool pred512 = false, amb512 = false;
    {
      const float Pk = OP_PA(0).x * OP_PA(0).x + OP_PA(0).y * OP_PA(0).y;
      const float Pn = OP_PB(0).x * OP_PB(0).x + OP_PB(0).y * OP_PB(0).y;
      const float x0 = 2.f * (raw0.x + raw0.y), xN = 2.f * (raw0.x - raw0.y);
      const float P256 = 4.f * (raw8.x * raw8.x + raw8.y * raw8.y);
      decideA(l0 ? x0 * x0 : Pk, t2[0]);
      decideB(l0 ? P256 : Pn, t2[31]);
      const float P5 = xN * xN, d5 = P5 - t2_512;
      pred512 = l0 && d5 > 0.f;
      amb512 = l0 && d5 * d5 <= d2 * (P5 + t2_512);
    }
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      if ((sl & 3) == 0) __builtin_amdgcn_sched_barrier(0);  // keep the constant loads next to their use
      decideA(OP_PA(sl).x * OP_PA(sl).x + OP_PA(sl).y * OP_PA(sl).y, t2[sl]);
      decideB(OP_PB(sl).x * OP_PB(sl).x + OP_PB(sl).y * OP_PB(sl).y, t2[31 - sl]);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned pred = (__brev(pA) >> 16) | (pB << 16);
    unsigned amb = ~((__brev(nA) >> 16) | (nB << 16));
    // a unit with non-finite samples (compare constants = T2_NEVER): NaN powers have no meaningful sign
    if (need == 2) { pred = 0; amb = 0; pred512 = false; amb512 = false; }
    if (!fvalid) { amb = 0; amb512 = false; pred = 0; pred512 = false; }
    // exact re-evaluation, one cell at a time, whole wave cooperating.  Rare (about one wave in fifty has an
    // ambiguous cell), but its float64 temporaries do not fit next to the 64 registers of the split spectra:
    // the wave parks half of them (pb: 8 KB) in its idle exchange slice for the duration.
    if ((OP_ABLATE & 8) == 0 && __ballot(amb != 0 || amb512) != 0ull) {
      float* park = reinterpret_cast<float*>(regions + wave * WAVE_CX_H) + lane;
      static_assert(64 * 32 * 4 <= WAVE_CX_H * 8, "parked registers must fit the wave's slice");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        park[(2 * i) * 64] = OP_PB(i).x;
        park[(2 * i + 1) * 64] = OP_PB(i).y;
      }
      while (true) {
        const unsigned long long pending = __ballot(amb != 0 || amb512);
        if (pending == 0) break;
        const int src = __ffsll((long long)pending) - 1;
        const unsigned amb_s = (unsigned)__shfl((int)amb, src);
        const int q = amb_s ? (__ffs((int)amb_s) - 1) : 32;
        const int cs = src & 15, gs = src >> 4;
        const int f = q < 32 ? bin_of_entry(cs, q) : 512;
        const double Pe = op_exact_power(P, row, chunk, tq + gs, f, lane);
        const bool pass = Pe > s_t2d[q < 32 ? cs * 32 + q : 512];
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
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        OP_PB(i).x = park[(2 * i) * 64];
        OP_PB(i).y = park[(2 * i + 1) * 64];
      }
      wave_lds_sync();
    }
    if (l0)  // entry -> register: e 0..7 -> 0..7, 8..23 -> 16..31, 24..30 -> 9..15, 31 -> 8
      pred = (pred & 0xffu) | ((pred & 0x00ffff00u) << 8) | ((pred >> 15) & 0xfe00u) | ((pred >> 23) & 0x100u);
    // Lane (g, c) holds the 32 decisions of frame g in ENTRY order; word m of the frame's bit row needs entry 2m (and
    // 16 + 2m, 2m + 1, 17 + 2m) of all 16 lanes.  A 16 x 16 bit transpose of both 16-bit halves at once -- four
    // xor-shuffle steps, each swapping the off-diagonal s x s blocks -- leaves in lane k: low half = entry k of lanes
    // 0..15, high half = entry 16 + k.  (32 wave ballots + per-lane selects did the same in ~200 instructions.)
    unsigned tr = pred;
    // (`up` = lanes whose column index has bit `sft` set: a compile-time lane mask -> SGPR-pair selects, sel_s of fastpath.hpp)
    auto tstep = [&](int sft, unsigned msk, unsigned long long up) {
      const unsigned y = (unsigned)__shfl_xor((int)tr, sft);
      const unsigned ysh = sel_s(up, y >> sft, y << sft);
      const unsigned mk = sel_s(up, msk, ~msk);
      tr = (tr & ~mk) | (ysh & mk);
    };
    tstep(8, 0x00ff00ffu, 0xff00ff00ff00ff00ull);
    tstep(4, 0x0f0f0f0fu, 0xf0f0f0f0f0f0f0f0ull);
    tstep(2, 0x33333333u, 0xccccccccccccccccull);
    tstep(1, 0x55555555u, 0xaaaaaaaaaaaaaaaaull);
    const unsigned long long b8 = __ballot(pred512);
    {
      const int sh = 16 * g;
      const int srcl = (lane & 48) | ((2 * c) & 15);
      const unsigned wa = (unsigned)__shfl((int)tr, srcl), wb2 = (unsigned)__shfl((int)tr, srcl + 1);
      const unsigned f0 = wa & 0xffffu, f1 = wa >> 16, f2 = wb2 & 0xffffu, f3 = wb2 >> 16;
      const unsigned r1 = (((__brev(f1) >> 16) << 1) | (f1 & 1u)) & 0xffffu;
      const unsigned r3 = (((__brev(f3) >> 16) << 1) | (f3 & 1u)) & 0xffffu;
      myword = (unsigned long long)(f0 | (r1 << 16)) | ((unsigned long long)(f2 | (r3 << 16)) << 32);
      if (c == 8) myword = (b8 >> sh) & 1ull;
    }
  }

  OP_STAMP(5);   // decide (+ refinement) + transpose
  // ---- publish this tile's bits; the spectra stay in v[] ---------------------------------------------
  // (LATE ARGUMENTS, see the epilogue: what the middle of the kernel needs -- exchange buffer, epoch, error word, tables,
  // mask scale, smoothing width -- is read from the kernel-argument segment here, after the prologue's peak of live scalars)
#if OP_LATE_ARGS
  typedef const __attribute__((address_space(4))) char* op_kpm_t;
  op_kpm_t kpm = (op_kpm_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kpm));
#define OP_MARG(type, member) (*(const __attribute__((address_space(4))) type*)(kpm + __builtin_offsetof(OnePassArgs, member)))
  const unsigned m_ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)s_misc[tk_slot]);
  const unsigned m_ntt = (unsigned)(OP_MARG(int, A.n_tiles) + 2);
  const int64_t m_u = m_ticket / m_ntt;
  const int m_jt = (int)(m_ticket % m_ntt) - 1;
#else
#define OP_MARG(type, member) (P.member)
  const unsigned m_ntt = (unsigned)ntt;
  const int64_t m_u = u;
  const int m_jt = jt;
#endif
  unsigned long long* const m_xbits = (unsigned long long*)(uintptr_t)OP_MARG(unsigned long long, xbits);
  const unsigned m_epoch = OP_MARG(unsigned, epoch);
  unsigned* const m_err = (unsigned*)(uintptr_t)OP_MARG(unsigned long long, err);
  const char* const m_tab = (const char*)(uintptr_t)OP_MARG(unsigned long long, tab);
  const float m_kscale = OP_MARG(float, A.kscale);
  const int m_nt = OP_MARG(int, nt);
  [[maybe_unused]] const float m_inv_ktot = OP_MARG(float, inv_ktot), m_prop = OP_MARG(float, prop);
  unsigned long long* xb_mine = m_xbits + ((size_t)m_u * m_ntt + (m_jt + 1)) * OP_TILE_WORDS;
  // data-tagged granules: every 8-byte store carries 32 mask bits and the launch epoch.  A consumer polls the
  // granules it needs until their tags are current: no separate flag, no drain of the stores, one write-through
  // and one read on the critical path instead of two of each.
  if (c < OP_XW) {
    const op_v4u gr = {(unsigned)myword, m_epoch, (unsigned)(myword >> 32), m_epoch};
    op_st16_sc1(&xb_mine[((4 * wave + g) * OP_XW + c) * 2], gr);
  }
  __syncthreads();   // every wave is past its forward exchange: the slices are idle from here
  if (halo_tile) {
    if constexpr (PERSIST) {
      // a halo tile ends here: its next ticket is drawn on the spot (two tiles in 149 at the default chunking)
#if OP_MAX_ITERS == 1 || OP_DRAW_TOP
      if (tid == 0) s_misc[tk_slot ^ 1u] = 0xffffffffu;
#else
      if (tid == 0) s_misc[tk_slot ^ 1u] = OP_DRAW_TICKET(OP_MARG(unsigned long long, ticket)) - OP_MARG(unsigned, ticket_base);
#endif
      pf = false;
      // (defined on this path too: left alone, the registers of the samples staged at the loop top would stay live up to here)
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k] = 0.f;
#pragma unroll
      for (int k = 0; k < SCAN_REG; ++k) sc[k] = 0.f;
    }
    OP_DONE;
  }
  OP_STAMP(6);   // publish + barrier

  // ---- smoothing on the matrix cores (exact integer arithmetic, v_mfma_i32_16x16x32_i8) ----------------
  // The separable triangle filter is two small dense contractions per 16-bin block:
  //   H[row][bin]  = sum_k  bit[row][16 b - 8 + k] * vf[k - 8 - j]        (A = 16 rows x 32 bins of 0/1 bytes,
  //                                                                          B = 32 x 16 band matrix, constant)
  //   K[frame][bin] = sum_row vt[row - frame - m_nt] * H[row][bin]           (A = 16 frames x 32 row slots, constant,
  //                                                                          B = H as bytes: H <= (nf+1)^2 <= 81)
  // The first product's result layout (lane = bin column, 4 consecutive rows per lane group) IS the second
  // product's B layout once its k slots are numbered accordingly, so H never leaves the registers.  Row block 0
  // = the tile's OWN 16 rows: its H is computed before the neighbours' flags are polled (the hand-off latency
  // hides behind it); row blocks 1, 2 = the 2 m_nt neighbour rows.
  constexpr int WP = OP_XW + 2;        // one zero word on each side of every bit row
  constexpr int WPB = WP * 8;          // bytes per row
  constexpr int SLICE_B = WAVE_CX_H * 8;
  char* rbytes = reinterpret_cast<char*>(regions);
  unsigned long long* wb = reinterpret_cast<unsigned long long*>(rbytes + 4 * OP_KP * 2);  // tail of slice 0: 48 rows
  static_assert(4 * OP_KP * 2 + 48 * WPB <= SLICE_B, "bit rows must fit behind wave 0's K rows");
  const unsigned char* wbb = reinterpret_cast<const unsigned char*>(wb);
  if (c < OP_XW) wb[(m_nt + 4 * wave + g) * WP + 1 + c] = myword;
  for (int r = tid; r < 48; r += WAVES * 64) {
    wb[r * WP] = 0ull;
    wb[r * WP + WP - 1] = 0ull;
  }
  const int q4 = lane >> 4, j16 = lane & 15;
  const long Bf = (long)reinterpret_cast<const unsigned long long*>(m_tab + OP_TAB_MCONST)[lane], At1 = (long)reinterpret_cast<const unsigned long long*>(m_tab + OP_TAB_MCONST)[64 + lane], At2 = (long)reinterpret_cast<const unsigned long long*>(m_tab + OP_TAB_MCONST)[128 + lane];
  const bool three = 2 * m_nt > 16;      // a third row block (wave-uniform)
  // neighbour-list index m -> tile row: m < m_nt: row m (previous tile), else row m_nt + 16 + (m - m_nt) (next tile)
  const int m1 = j16, m2 = 16 + j16;
  const int r1 = m1 < 2 * m_nt ? (m1 < m_nt ? m1 : 16 + m1) : 0;
  const int r2 = m2 < 2 * m_nt ? (m2 < m_nt ? m2 : 16 + m2) : 0;
  __syncthreads();
  typedef int op_v4i __attribute__((ext_vector_type(4)));
  const op_v4i zero4 = {0, 0, 0, 0};
  unsigned hown[9];   // H of the own rows, 4 bytes per 16-bin block
  {
    const unsigned char* rp = wbb + (m_nt + j16) * WPB + 7 + q4;
#pragma unroll
    for (int nb = 0; nb < 9; ++nb) {
      const int b = wave + 4 * nb;
      hown[nb] = 0u;
      if (b < 33) {
        const long a = (long)s_exp[rp[2 * b]];
        const op_v4i h = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, Bf, zero4, 0, 0, 0);
        hown[nb] = (unsigned)h[0] | ((unsigned)h[1] << 8) | ((unsigned)h[2] << 16) | ((unsigned)h[3] << 24);
      }
    }
  }
  OP_STAMP(7);   // zero fill + barrier + own rows on the matrix cores
  // neighbour rows: one 16-byte load per 64-bit word (2 m_nt x 9 words <= 288: at most two per thread), polled
  // until both tags are current
  for (int i = tid; i < 2 * m_nt * OP_XW; i += WAVES * 64) {
    const int side = i >= m_nt * OP_XW;
    const int rem = i - side * m_nt * OP_XW;
    const int rr = rem / OP_XW, w = rem - rr * OP_XW;
    const unsigned long long* src = side ? xb_mine + OP_TILE_WORDS + (rr * OP_XW + w) * 2
                                         : xb_mine - OP_TILE_WORDS + ((NF - m_nt + rr) * OP_XW + w) * 2;
    op_v4u gr = op_ld16_sc1(src);
    for (int spin = 0; !(OP_ABLATE & 2) && (LOSE || gr[1] != m_epoch || gr[3] != m_epoch); ++spin) {
      if (LOSE || spin >= OP_SPIN_MAX) {   // every spin is bounded: report instead of hanging the device
        atomicOr_system(m_err, 1u);
#if OP_WHO
        if (atomicAdd_system(m_err + 14, 1u) == 0u) { m_err[8] = m_ticket; m_err[9] = (unsigned)i; m_err[10] = gr[1]; m_err[11] = m_epoch; }
#endif
        s_misc[1] = 1u;            // the tile's mask is unknown: every hop it finalises or hands on becomes NaN
        break;
      }
      OP_POLL_SLEEP(spin);
      gr = op_ld16_sc1(src);
    }
    wb[(side ? m_nt + NF + rr : rr) * WP + 1 + w] = (unsigned long long)gr[0] | ((unsigned long long)gr[2] << 32);
  }
  __syncthreads();
  OP_STAMP(8);   // neighbours' bits (poll) + barrier
  // PERSIST: the next ticket.  Drawn only now -- this tile no longer waits for a tile with a HIGHER ticket -- and consumed at
  // the end of the smoothing stage: the atomic's round trip (~1.5 us) runs under the matrix-core work
  [[maybe_unused]] unsigned nx_raw = 0u;
  if constexpr (PERSIST) {
#if OP_MAX_ITERS == 1 || OP_DRAW_TOP
    if (tid == 0) nx_raw = 0xffffffffu + OP_MARG(unsigned, ticket_base);   // "past the last tile": the workgroup leaves at the loop top
#else
    if (tid == 0) nx_raw = OP_DRAW_TICKET(OP_MARG(unsigned long long, ticket));
#endif
  }
  {
    const unsigned char* rp1 = wbb + r1 * WPB + 7 + q4;
    const unsigned char* rp2 = wbb + r2 * WPB + 7 + q4;
#pragma unroll
    for (int nb = 0; nb < 9; ++nb) {
      const int b = wave + 4 * nb;
      if (b < 33) {
        const long a1 = (long)s_exp[rp1[2 * b]];
        const op_v4i h1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, Bf, zero4, 0, 0, 0);
        const unsigned p1 = (unsigned)h1[0] | ((unsigned)h1[1] << 8) | ((unsigned)h1[2] << 16) | ((unsigned)h1[3] << 24);
        const long bt1 = (long)(((unsigned long long)p1 << 32) | (unsigned long long)hown[nb]);
        op_v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8(At1, bt1, zero4, 0, 0, 0);
        if (three) {
          const long a2 = (long)s_exp[rp2[2 * b]];
          const op_v4i h2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a2, Bf, zero4, 0, 0, 0);
          const unsigned p2 = (unsigned)h2[0] | ((unsigned)h2[1] << 8) | ((unsigned)h2[2] << 16) | ((unsigned)h2[3] << 24);
          d = __builtin_amdgcn_mfma_i32_16x16x32_i8(At2, (long)(unsigned long long)p2, d, 0, 0, 0);
        }
        // lane group q4 holds output frames 4 q4 .. 4 q4 + 3 = wave q4's frames: K rows live in THAT wave's slice
        unsigned short* kd = reinterpret_cast<unsigned short*>(rbytes + q4 * (SLICE_B + 64)) + 16 * b + j16;  // +64 B per slice: the four lane groups hit disjoint banks
        kd[0] = (unsigned short)d[0];
        kd[OP_KP] = (unsigned short)d[1];
        kd[2 * OP_KP] = (unsigned short)d[2];
        kd[3 * OP_KP] = (unsigned short)d[3];
      }
    }
  }
#if !OP_TK_LATE
  if constexpr (PERSIST) {
    if (tid == 0) s_misc[tk_slot ^ 1u] = nx_raw - OP_MARG(unsigned, ticket_base);
#if OP_WHO
    if (tid == 0) {   // who drew the ticket, and while working on which tile
      unsigned* w_ = (unsigned*)(uintptr_t)OP_MARG(unsigned long long, who);
      const unsigned nx_ = nx_raw - OP_MARG(unsigned, ticket_base);
      if (nx_ < 4096u) { w_[nx_ * 4] = (blockIdx.x + 1u) | (iter << 16); w_[nx_ * 4 + 1] = m_ticket; }
      if (m_ticket < 4096u) w_[m_ticket * 4 + 3] = 1u;   // this tile is past its neighbours' bits
    }
#endif
  }
#endif
  __syncthreads();  // K of all 16 frames complete; from here every wave touches only its own slice
  OP_STAMP(9);   // smoothing on the matrix cores + barrier

  const unsigned short* kt = reinterpret_cast<const unsigned short*>(rbytes + wave * (SLICE_B + 64));
  // entry e of this lane = bin c + 32 e (e < 16) or (32 - c) + 32 (e - 16); lane 0 pairs its bins
  // differently (bin_of_entry): read where used, two 16-bit LDS loads per conjugate pair
  const unsigned short* krow = kt + g * OP_KP;
  const unsigned short* k_lo = krow + (c == 0 ? 0 : c);        // entries e < 16 of lanes c >= 1
  const unsigned short* k_hi = krow + (c == 0 ? 0 : 32 - c);   // entries e >= 16
  // PROP: weight of the valid taps along t for this lane group's frame (closed form of the triangle's tails)
  float tt = 0.f;
  if constexpr (PROP) {
    const int64_t tl_ = t < m_nt ? m_nt - t : 0, tr_ = (G.T - 1 - t) < m_nt ? m_nt - (G.T - 1 - t) : 0;
    tt = (float)((int64_t)(m_nt + 1) * (m_nt + 1) - tl_ * (tl_ + 1) / 2 - tr_ * (tr_ + 1) / 2);
  }
  const unsigned char* e_lo = s_ef + (c == 0 ? 0 : c);
  const unsigned char* e_hi = s_ef + (c == 0 ? 0 : 32 - c);
  // the float mask exactly as k_k16_to_mask writes it: p * (K / ktot) + (1 - p) * edge, edge = tf * tt / ktot
  auto mfull = [&](unsigned short kv, float tf) -> float {
    const float edge = tf * tt * m_inv_ktot;
    return m_prop * ((float)kv * m_inv_ktot) + (1.0f - m_prop) * edge;
  };
  const float k512 = PROP ? mfull(krow[512], (float)s_ef[512]) * m_kscale : (float)krow[512] * m_kscale;
  auto mval = [&](int q, float scale) -> float {
    const int b0 = bin_of_entry(0, q);                         // lane 0 (compile-time)
    const unsigned short* pl = q < 16 ? k_lo : k_hi;
    const int off = q < 16 ? 32 * q : 32 * (q - 16);
    const unsigned short kv = c == 0 ? krow[b0] : pl[off];
    if constexpr (PROP) {
      const float tf = (float)(c == 0 ? s_ef[b0] : (q < 16 ? e_lo : e_hi)[off]);
      return mfull(kv, tf) * scale;
    }
    return (float)kv * scale;
  };
  {
    // mask -> merge (the second half of pair_mask): Yk = X2[k] mk, Yn = conj-pair value x mn, then back to the
    // half-size complex spectrum.  The four 1/2 factors of split and merge ride in the mask scale.
    const float ks = m_kscale * 0.25f;
#if OP_INPLACE
    {
      // slot 0: lanes >= 1 merge the pair (v[0], v[31]); lane 0: bins 0 / 512 from v[0], bin 256 = v[31] scaled
      const cf r0 = v[0], r31 = v[31];
      cf xa = r0, xb = r31;
      merge_pair(xa, xb, wlo, mval(0, ks), mval(31, ks));
      const float y0 = (r0.x + r0.y) * mval(0, m_kscale);
      const float yN = (r0.x - r0.y) * k512;
      const cf z0 = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
      const float m8 = mval(31, m_kscale);  // entry 31 of lane 0 = bin 256
      const cf z8 = {r31.x * m8, r31.y * m8};
      v[0] = {l0 ? z0.x : xa.x, l0 ? z0.y : xa.y};
      v[31] = {l0 ? z8.x : xb.x, l0 ? z8.y : xb.y};
    }
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
      merge_pair(v[sl], v[31 - sl], w, mval(sl, ks), mval(31 - sl, ks));
    }
    rg_lane0_from_entries(v, l0);   // back to the transform's register order
  }
#else
    auto sel = [&](cf a0, cf a1) -> cf { return {l0 ? a0.x : a1.x, l0 ? a0.y : a1.y}; };
    auto merge = [&](cf& xa, cf& xb, cf w, float mk, float mn) { merge_pair(xa, xb, w, mk, mn); };
    merge(pa[0], pb[0], wlo, mval(0, ks), mval(31, ks));
    cf z0, z8;
    {
      const float y0 = (raw0.x + raw0.y) * mval(0, m_kscale);
      const float yN = (raw0.x - raw0.y) * k512;
      z0 = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
      const float m8 = mval(31, m_kscale);  // entry 31 of lane 0 = bin 256
      z8 = {raw8.x * m8, raw8.y * m8};
    }
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
      merge(pa[sl], pb[sl], w, mval(sl, ks), mval(31 - sl, ks));
    }
    // scatter back: register i receives, for lanes >= 1, entry i; for lane 0, the entry that lives in register i
    v[0] = sel(z0, pa[0]);
#pragma unroll
    for (int i = 1; i < 8; ++i) v[i] = pa[i];
    v[8] = sel(z8, pa[8]);
#pragma unroll
    for (int i = 9; i < 16; ++i) v[i] = sel(pb[16 - i], pa[i]);
#pragma unroll
    for (int i = 16; i < 24; ++i) v[i] = sel(pa[i - 8], pb[31 - i]);
#pragma unroll
    for (int i = 24; i < 31; ++i) v[i] = sel(pb[39 - i], pb[31 - i]);
    v[31] = sel(pb[8], pb[0]);
  }
#endif
  wave_lds_sync();  // K tile consumed: the slice is reused by the inverse transform
  OP_STAMP(10);  // mask + merge
#if OP_TK_LATE
  if constexpr (PERSIST) {
    // the next ticket, from thread 0 to the other waves through a polled LDS word (no barrier between here and the prefetch)
    if (tid == 0) *(volatile unsigned*)&s_misc[tk_slot ^ 1u] = nx_raw - OP_MARG(unsigned, ticket_base);
  }
#endif

  // ---- inverse transform, synthesis window, wave-private overlap-add (k_apply_fast<LEAN>) -------------
  {
    // fresh address arithmetic for the inverse exchange: shared with the forward transform (CSE) the 16 row
    // addresses would stay live -- and be spilled -- across the whole smoothing phase
    int zi = 0, ci = c;
    asm volatile("" : "+v"(zi), "+v"(ci));
    fft512_inv_half(v, fb + zi, tw512 + zi, ci);
  }
  OP_STAMP(11);  // inverse transform
  float4 n4;   // 1 / window envelope of the lane's four samples of a hop
  auto load_n4 = [&]() { n4 = *reinterpret_cast<const float4*>(&reinterpret_cast<const float*>(m_tab + OP_TAB_INVN)[(tid & 63) * 4]); };
  if constexpr (PERSIST) load_n4();   // ahead of the prefetch: the epilogue's wait for it must not include the next tile's samples
  auto prefetch_next = [&]() {
    // ---- the NEXT tile's samples: loads issued here, consumed at the loop top (see PERSIST above) ----
    const OnePassArgs& Pn = *late_args<OnePassArgs>();
#if OP_TK_LATE
    unsigned nx_v;
    while ((nx_v = *(volatile unsigned*)&s_misc[tk_slot ^ 1u]) == 0xffffffffu) __builtin_amdgcn_s_sleep(1);
    const unsigned nx = (unsigned)__builtin_amdgcn_readfirstlane((int)nx_v);
#else
    const unsigned nx = (unsigned)__builtin_amdgcn_readfirstlane((int)s_misc[tk_slot ^ 1u]);
#endif
    pf = true;
    // (every path DEFINES q / sc: a conditional assignment would keep the previous tile's values alive -- 24 registers --
    // through a whole iteration)
    // q / sc are DEFINED on every path, each by one if / else (a conditional assignment -- or a flag the optimiser cannot
    // thread -- keeps the previous tile's values alive through a whole iteration, and the allocator then parks all 24
    // in scratch: a store that waits for the very load it was meant to hide)
    const bool n_live = nx < Pn.total_tiles;
    const unsigned n_ntt = (unsigned)(Pn.A.n_tiles + 2);
    const unsigned n_tk = n_live ? nx : 0u;
    const unsigned n_u = n_tk / n_ntt;
    const int n_jt = (int)(n_tk % n_ntt) - 1;
    const unsigned n_gu = (unsigned)(Pn.A.view.unit0 + n_u), n_nch = (unsigned)Pn.A.view.n_chunks;
    const TileSrc N = tile_src(Pn, (int64_t)(n_gu / n_nch), Pn.A.view.c0 + n_gu % n_nch, n_jt, true);
    if (n_live && N.blk_vec) {
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k] = N.sp[tid + k * 256];
    } else {
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k] = 0.f;
    }
    if (n_live && N.sc_base != nullptr) {
#pragma unroll
      for (int k = 0; k < SCAN_REG; ++k) sc[k] = N.sc_base[min(tid + k * WAVES * 64, N.sc_n1)];
    } else {
#pragma unroll
      for (int k = 0; k < SCAN_REG; ++k) sc[k] = 0.f;
    }
  };
#ifndef OP_PF_AT
#define OP_PF_AT 2   // where the prefetch is issued: 0 before the overlap-add, 1 / 2 / 3 after that many quarters of it, 4 after it, 5 at the end of the epilogue
#endif
#ifndef OP_TK_LATE
#define OP_TK_LATE 0  // 1: the next ticket reaches the other waves through a polled LDS word after the mask stage, not under the smoothing stage's barrier
#endif
  if constexpr (PERSIST && OP_PF_AT == 0) prefetch_next();
  float* acc = reinterpret_cast<float*>(regions + wave * WAVE_CX_H);
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
        float2 nw = {v[r].x * ws.x, v[r].y * ws.y};
        if (j != 0) {
          const float2 old = *dst;
          nw.x = sel_s(OLA_KEEP, nw.x + old.x, nw.x);
          nw.y = sel_s(OLA_KEEP, nw.y + old.y, nw.y);
        }
        *dst = nw;
      }
      wave_lds_sync();
      if constexpr (PERSIST) { if (j + 1 == OP_PF_AT) prefetch_next(); }
    }
  }
  if constexpr (!PERSIST) load_n4();
  __syncthreads();
  OP_STAMP(12);  // window + wave-private overlap-add + barrier
#if OP_TRACE
  if (lane == 0) t_slot_[15] = 1u;
#endif

  // ---- cross-wave combine, normalise, store (seam mode: abutting tiles) -------------------------------
  // LATE ARGUMENTS (round 5).  The output map, hop range, seam buffer and the tile's coordinates are only needed from here
  // on, but as ordinary kernel arguments they are loaded in the entry block and stay live through every phase: the
  // kernel ran out of scalar registers and parked them in VGPR lanes (v_writelane / 329 v_readlane -- each a 4-cycle slot
  // of the vector pipe, the kernel's binding resource: profiles/r05_valu_classes.txt).  Here they are read again from the
  // kernel-argument segment through an OPAQUE pointer (scalar loads, off the vector pipe; not mergeable with the entry
  // block's loads), and the tile's coordinates are recomputed from the ticket, which is still in LDS.
#if OP_LATE_ARGS
  typedef const __attribute__((address_space(4))) char* op_kp_t;
  op_kp_t kp4 = (op_kp_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp4));
#define OP_KARG(type, member) (*(const __attribute__((address_space(4))) type*)(kp4 + __builtin_offsetof(OnePassArgs, member)))
#define OP_KPTR(type, member) ((type)(uintptr_t)OP_KARG(unsigned long long, member))
#else
#define OP_KARG(type, member) (P.member)
#define OP_KPTR(type, member) ((type)P.member)
#endif
  const int64_t e_gstep = OP_KARG(int64_t, A.om.g_step), e_p0 = OP_KARG(int64_t, A.om.p0), e_p1 = OP_KARG(int64_t, A.om.p1);
  void* const e_out = OP_KPTR(void*, A.om.out);
  const int64_t e_ostride = OP_KARG(int64_t, A.om.stride), e_g0 = OP_KARG(int64_t, A.om.g0), e_glo = OP_KARG(int64_t, A.om.g_lo),
                e_ghi = OP_KARG(int64_t, A.om.g_hi);
  const int e_odtype = OP_KARG(int, A.om.dtype), e_normalize = OP_KARG(int, A.normalize), e_ntiles = OP_KARG(int, A.n_tiles);
  const int64_t e_hbegin = OP_KARG(int64_t, A.h_begin), e_hend = OP_KARG(int64_t, A.h_end);
  unsigned long long* const e_part2 = OP_KPTR(unsigned long long*, part2);
  const unsigned e_epoch = OP_KARG(unsigned, epoch);
  unsigned* const e_err = OP_KPTR(unsigned*, err);
  const char* const e_tab = OP_KPTR(const char*, tab);
  const int e_padL = OP_KARG(int32_t, A.g.padL);
  const int64_t e_T = OP_KARG(int64_t, A.g.T), e_Lout = OP_KARG(int64_t, A.g.Lout);
#if OP_LATE_ARGS
  const unsigned e_ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)s_misc[tk_slot]);
  const unsigned e_ntt = (unsigned)(e_ntiles + 2);
  const int64_t e_u = e_ticket / e_ntt;
  const int e_jt = (int)(e_ticket % e_ntt) - 1;
  const unsigned e_gu = (unsigned)(OP_KARG(int64_t, A.view.unit0) + e_u), e_nch = (unsigned)OP_KARG(int32_t, A.view.n_chunks);
  const int64_t e_row = e_gu / e_nch;
  const int64_t e_chunk = OP_KARG(int64_t, A.view.c0) + e_gu % e_nch;
  const int64_t e_tf = e_hbegin - 3 + (int64_t)e_jt * NF;
#else
  const int64_t e_u = u, e_row = row, e_chunk = chunk, e_tf = tf_tile;
  const int e_jt = jt;
#endif
  const float* fr = reinterpret_cast<const float*>(regions);
  const int s4 = (tid & 63) * 4;
  // A lost hand-off must not look like audio: a tile whose neighbour bits never arrived writes NaN to every hop it
  // finalises and publishes NaN partials (the next tile's straddling hops inherit them); a tile whose predecessor's
  // partial hops never arrived writes NaN to those three hops.  The error word still reports it (sg_check_errors).
  const float poison = s_misc[1] != 0u ? __uint_as_float(0x7fc00000u) : 0.f;
  // Hops that straddle two tiles: tile j publishes its three TRAILING partial hops (un-normalised sums) the
  // same way as the mask bits (write-through stores, drained, epoch flag per hop); tile j+1 adds its LEADING
  // partials and finalises them.  Per wave: trailing hop first (published early), interior hops, leading hop
  // last (tile j, one ticket earlier, has usually published by then).  Publishing never waits: no cycles.
  // Interior tiles (every hop inside the output range, all four frames of each hop present, float32 output,
  // aligned rows): straight-line version of the loop below, same sums in the same order.
  {
    const int64_t pb0 = e_tf * 256 - e_padL;
    const int64_t gi00 = e_chunk * e_gstep + (pb0 - e_p0);
    float* dbase = (float*)e_out + (e_row * e_ostride + gi00 - e_g0);
    float* dst0 = dbase + s4;
    const bool tile_fast = e_odtype == 0 && e_normalize && e_tf >= 3 && e_tf + NF + 2 < e_T &&
                           e_tf >= e_hbegin && e_tf + NF + 2 < e_hend && pb0 >= e_p0 &&
                           pb0 + NF * 256 <= e_p1 && pb0 + NF * 256 <= e_Lout && gi00 >= e_glo &&
                           gi00 + NF * 256 <= e_ghi && (reinterpret_cast<uintptr_t>(dbase) & 15) == 0 && WAVES == 4;
    if (tile_fast) {
      constexpr int R = WAVE_CX_H * 2;
      auto ld4 = [&](int off) { return *reinterpret_cast<const float4*>(&fr[off + s4]); };
      auto fin = [&](float4 a, int jj) {
        a.x = (a.x + poison) * n4.x; a.y = (a.y + poison) * n4.y; a.z = (a.z + poison) * n4.z; a.w = (a.w + poison) * n4.w;
        *reinterpret_cast<float4*>(dst0 + jj * 256) = a;
      };
      if (wave == 3) {
#pragma unroll
        for (int it = 0; it < 4; ++it) fin(ld4(it * R + 3 * HPITCH), 3 + 4 * it);
        OP_STAMP(13);
        if constexpr (PERSIST && OP_PF_AT == 5) prefetch_next();
        OP_DONE;
      }
      {
        const float4 a4 = ld4((WAVES - 1) * R + (wave + 4) * HPITCH);
        unsigned long long* dst = e_part2 + (((size_t)e_u * e_ntiles + e_jt) * 3 + wave) * 256 + s4;
        const op_v4u ga = {__float_as_uint(a4.x + poison), e_epoch, __float_as_uint(a4.y + poison), e_epoch};
        const op_v4u gb = {__float_as_uint(a4.z + poison), e_epoch, __float_as_uint(a4.w + poison), e_epoch};
        op_st16_sc1(dst, ga);
        op_st16_sc1(dst + 2, gb);
      }
#pragma unroll
      for (int it = 1; it < 4; ++it) {
        float4 a4 = ld4((it - 1) * R + (wave + 4) * HPITCH);
        const float4 f4 = ld4(it * R + wave * HPITCH);
        a4.x += f4.x; a4.y += f4.y; a4.z += f4.z; a4.w += f4.w;
        fin(a4, wave + 4 * it);
      }
      {
        float4 a4 = ld4(wave * HPITCH);
        const unsigned long long* src = e_part2 + (((size_t)e_u * e_ntiles + e_jt - 1) * 3 + wave) * 256 + s4;
        op_v4u ga, gb;
        for (int spin = 0;; ++spin) {
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(ga), "=&v"(gb) : "v"(src) : "memory");
          const unsigned e = e_epoch;
          if ((OP_ABLATE & 16) || (!LOSE && ga[1] == e && ga[3] == e && gb[1] == e && gb[3] == e)) break;
          if (LOSE || spin >= OP_SPIN_MAX) {
            atomicOr_system(e_err, 2u);
#if OP_WHO
            if (atomicAdd_system(e_err + 15, 1u) == 0u) { e_err[12] = e_ticket; e_err[13] = ga[1]; }
#endif
            ga[0] = ga[2] = gb[0] = gb[2] = 0x7fc00000u;   // the previous tile's share is unknown: NaN, not a partial sum
            break;
          }
          OP_POLL_SLEEP(spin);
        }
        a4.x = __uint_as_float(ga[0]) + a4.x;
        a4.y = __uint_as_float(ga[2]) + a4.y;
        a4.z = __uint_as_float(gb[0]) + a4.z;
        a4.w = __uint_as_float(gb[2]) + a4.w;
        fin(a4, wave);
      }
      OP_STAMP(13);  // cross-wave combine, hand-off of the straddling hops, stores
      if constexpr (PERSIST && OP_PF_AT == 5) prefetch_next();
      OP_DONE;
    }
  }
  for (int it = 0; it < 5; ++it) {
    const int jj = wave < 3 ? (it == 0 ? NF + wave : (it == 4 ? wave : wave + 4 * it)) : (it < 4 ? 3 + 4 * it : -1);
    if (jj < 0) break;
    const int64_t h = e_tf + jj;
    if (h >= e_hend || h < e_hbegin) continue;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool all_valid = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t ti = e_tf + jj - q;
      if (ti < 0 || ti >= e_T) all_valid = false;
    }
    {
      const int wh = jj >> 2, lh = jj & 3;
      if (wh >= 1 && wh - 1 < WAVES && lh <= 2) {
        a4 = *reinterpret_cast<const float4*>(&fr[(wh - 1) * WAVE_CX_H * 2 + (lh + 4) * HPITCH + s4]);
      }
      if (wh < WAVES) {
        const float4 f4 = *reinterpret_cast<const float4*>(&fr[wh * WAVE_CX_H * 2 + lh * HPITCH + s4]);
        a4.x += f4.x; a4.y += f4.y; a4.z += f4.z; a4.w += f4.w;
      }
    }
    if (jj >= NF) {
      // trailing partial hop -> tagged granules {float, epoch}
      const int k = jj - NF;
      unsigned long long* dst = e_part2 + (((size_t)e_u * e_ntiles + e_jt) * 3 + k) * 256 + s4;
      const op_v4u ga = {__float_as_uint(a4.x + poison), e_epoch, __float_as_uint(a4.y + poison), e_epoch};
      const op_v4u gb = {__float_as_uint(a4.z + poison), e_epoch, __float_as_uint(a4.w + poison), e_epoch};
      op_st16_sc1(dst, ga);
      op_st16_sc1(dst + 2, gb);
      continue;
    }
    if (jj < 3) {
      // h >= h_begin implies e_jt >= 1: the previous tile exists
      const unsigned long long* src = e_part2 + (((size_t)e_u * e_ntiles + e_jt - 1) * 3 + jj) * 256 + s4;
      op_v4u ga, gb;
      for (int spin = 0;; ++spin) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(ga), "=&v"(gb) : "v"(src) : "memory");
        const unsigned e = e_epoch;
        if ((OP_ABLATE & 16) || (!LOSE && ga[1] == e && ga[3] == e && gb[1] == e && gb[3] == e)) break;
        if (LOSE || spin >= OP_SPIN_MAX) {
          atomicOr_system(e_err, 2u);
          ga[0] = ga[2] = gb[0] = gb[2] = 0x7fc00000u;
          break;
        }
        OP_POLL_SLEEP(spin);
      }
      // trailing partial of the previous tile + leading partial of this one (the order k_ola_seam adds them in)
      a4.x = __uint_as_float(ga[0]) + a4.x;
      a4.y = __uint_as_float(ga[2]) + a4.y;
      a4.z = __uint_as_float(gb[0]) + a4.z;
      a4.w = __uint_as_float(gb[2]) + a4.w;
    }
    a4.x += poison; a4.y += poison; a4.z += poison; a4.w += poison;
    if (!e_normalize) {
    } else if (all_valid) {
      a4.x *= n4.x; a4.y *= n4.y; a4.z *= n4.z; a4.w *= n4.w;
    } else {
      float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ti = h - q;
        if (ti >= 0 && ti < e_T) {
          const float4 w4 = *reinterpret_cast<const float4*>(&reinterpret_cast<const float*>(e_tab + OP_TAB_WSQ)[256 * q + s4]);
          nrm.x += w4.x; nrm.y += w4.y; nrm.z += w4.z; nrm.w += w4.w;
        }
      }
      a4.x /= (nrm.x > 1e-10f ? nrm.x : 1.f);
      a4.y /= (nrm.y > 1e-10f ? nrm.y : 1.f);
      a4.z /= (nrm.z > 1e-10f ? nrm.z : 1.f);
      a4.w /= (nrm.w > 1e-10f ? nrm.w : 1.f);
    }
    {
      const int64_t pb = h * 256 - e_padL;
      const int64_t gi0 = e_chunk * e_gstep + (pb - e_p0);
      if (e_odtype == 0 && pb >= e_p0 && pb + 256 <= e_p1 && pb + 256 <= e_Lout && gi0 >= e_glo &&
          gi0 + 256 <= e_ghi) {
        float* dst = (float*)e_out + (e_row * e_ostride + gi0 - e_g0 + s4);
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = a4;
          continue;
        }
      }
    }
    const float vals[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t p = h * 256 + s4 + e - e_padL;
      if (p < e_p0 || p >= e_p1) continue;
      const int64_t gi = e_chunk * e_gstep + (p - e_p0);
      if (gi < e_glo || gi >= e_ghi) continue;
      store_sample(e_out, e_odtype, e_row * e_ostride + gi - e_g0, p < e_Lout ? vals[e] : 0.f);
    }
  }
  if constexpr (PERSIST && OP_PF_AT == 5) prefetch_next();
  }   // tile loop (one iteration unless PERSIST)
}

}  // namespace fast
}  // namespace sg
