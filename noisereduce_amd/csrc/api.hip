// libmi355gate.so -- host side of the C ABI declared in include/mi355gate.h.
// Builds tables, owns the workspace, batches (channel, chunk) units and enqueues the
// kernels of kernels.hpp on the caller's HIP stream.  gfx950 only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mi355gate_debug.h"
// Floating-point contraction only INSIDE a source expression (the language rule), never across statements: with the
// compiler's default (fast) the back end fuses a product into whichever neighbouring add it meets first, and that
// choice changes with unrelated edits -- two kernels that share a formula then round differently in the cells where
// both products of a sum are inexact.  The one-pass gate and the three-kernel path are bit-identical by
// construction only under this rule; where a fused multiply-add is wanted across statements the code says fmaf().
#pragma clang fp contract(on)
#include "kernels.hpp"
#include "fused.hpp"
#include "czt.hpp"
#include "mixed.hpp"
#include "onepass.hpp"
#include "rowgate.hpp"
#include "rowbwd.hpp"
#include "fast512.hpp"
#include "onepass512.hpp"
#include "fast256.hpp"
#include "onepass256.hpp"
#include "fast2048.hpp"
#include "onepass2048.hpp"
#include "nonstat.hpp"
#include "fast64.hpp"
#include "big.hpp"
#include "exact.hpp"
#include "apply64.hpp"

// complex transform length from which a whole 256-thread workgroup (instead of one wavefront) works on ONE frame in
// the general LDS kernels: at N = 2048 (n_fft = 4096) a wavefront holds 4 radix-8 butterflies = 64 complex values per
// lane -- 256 VGPRs and scratch, one wave per SIMD
#ifndef SG_TEAM_N
#define SG_TEAM_N 2048
#endif
// (round 5) threads that share one frame in the LDS-Stockham kernels (fft_wave.hpp): the whole workgroup from N = SG_TEAM_N
// on, one wavefront for 512 <= N < SG_TEAM_N, and SUB-wavefront teams for short frames -- a radix-8 pass over N complex
// points has N / 8 butterflies, so a 64-lane team on N = 128 (n_fft = 256) idles 48 lanes in two of its three passes.
// SG_SHORT_TEAMS=0 at build time restores one wavefront per frame (A/B).
#ifndef SG_SHORT_TEAMS
#define SG_SHORT_TEAMS 1
#endif
#ifndef SG_CHAIN_PAR
#define SG_CHAIN_PAR 1    // the non-stationary gate's tile chain in 16 parallel runs per band (nonstat.hpp: k_iir_chain_par); 0: A/B
#if OP_WHO
static unsigned* g_who_dev = nullptr;     // (development builds) per workgroup of the persistent gate: ticket started, iteration, ticket drawn next, stage
#endif
#if OP_TRACE
static unsigned* g_trace_dev = nullptr;   // (development builds) the one-pass gate's phase trace of the last first launch: sg_debug_counter 8
static size_t g_trace_tiles = 0;
static int g_trace_ntt = 1;
#endif
#endif
#ifndef SG_STATS_TEAM
#define SG_STATS_TEAM 1   // float64 noise-clip transform of 1024 points (n_fft = 2048) by a whole workgroup (launch_stft_n); 0: A/B
#endif
template <int N>
constexpr int team_threads() {
  return N >= SG_TEAM_N ? 256 : (!SG_SHORT_TEAMS ? 64 : (N <= 128 ? 16 : (N == 256 ? 32 : 64)));
}
// teams per workgroup: 256 threads unless the transform buffers (`elem` bytes per complex point) would not fit
template <int N>
constexpr int team_count(size_t elem) {
  return N >= SG_TEAM_N ? 1 : (team_threads<N>() < 64 ? 256 / team_threads<N>() : ((N * elem > 16384) ? 2 : 4));
}
#ifndef SG_APPLY_WAVES
#define SG_APPLY_WAVES 4  // wavefronts per workgroup of k_apply_fast: tile = 4*W frames -> 4*W-3 hops
#endif

using namespace sg;

namespace {
thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

std::string fmt(const char* f, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}
}  // namespace

struct sg_handle {
  sg_params p{};
  int n = 0, N = 0, W = 0, H = 0, F = 0, FS = 0, padL = 0;
  double sum_w = 0.0;   // sum of the analysis window (scipy 'spectrum' scaling)
  double mag_scale = 1; // |Z| = sqrt(P) * mag_scale  (S: 1/sum_w, T: 1)
  // device tables
  DevBuf tw64, tw32, wfull64, wa32, ws32, wsq32, kf, kt;
  // per-band threshold (S stationary): double[FS]
  DevBuf thresh;
  bool has_thresh = false;
  // workspace
  DevBuf P, pmax, thr_rows, raw, M, seg, yn;
  DevBuf bits, K16, umax, need, T2;  // fused stationary path
  DevBuf nss;                        // non-stationary gate: recurrence partials of k_mag_fast's 16-frame blocks
  DevBuf alim;                       // one-pass gate, in-kernel floor test: [1] tag of the last call that reported, [2..11) bounds on max|x| (k_colstats1_final / k_prep_thresh_lazy)
  bool t2_ready = false;             // T2 / alim hold the compare constants of the CURRENT threshold (any writer of thresh clears it)
  DevBuf logtab;                     // db_fast (kernels.hpp): {rd(1 / c_i), -log2 of it} for 128 mantissa centres
  DevBuf part;                       // partial reductions of the column statistics
  DevBuf tw512, invn;                // fast path tables (n_fft = 1024, hop = 256)
  DevBuf seam;                       // partial seam hops of abutting apply tiles
  DevBuf ftab;                       // k_smooth_bits2 phase-1 lookup tables (nf <= 5)
  int sm2_tt = 64;                   // k_smooth_bits2 tile height (frames)
  // one-pass gate (onepass.hpp): published mask bits per tile, publication flags, work counter, tables
  DevBuf xbits, xpart, xticket, xtick2, ftab3, xexp, optab;
  DevBuf xin;                        // float32 copy of a recording held in another sample dtype
  DevBuf nsp, nsc;                   // non-stationary mask: per-sub-tile partials / carries (nonstat.hpp)  // ftab3: per-lane MFMA operands, xexp: bit -> byte table
  unsigned* err_host = nullptr;      // host-mapped error word written by k_gate_onepass when a hand-off times out
  unsigned* err_dev = nullptr;
  unsigned inject_fault = 0;         // SG_OPT_INJECT_HANDOFF_FAULT (tests): error bits the next hand-off launch reports
  unsigned lose_now = 0;             // bits 3..5 of the option, consumed by the next hand-off launch: the kernel itself treats
                                     // the hand-off as lost (bounded-poll timeout path: error word + NaN-poisoned hops)
  unsigned ticket_base = 0;          // tickets handed out by all previous launches
  unsigned epoch = 0;                // launch counter: the value a tile's flag must carry to be current
  bool fast_integer = false;         // SG_OPT_FAST_INTEGER: integer outputs from the float32 kernels (<= 1 LSB off)
  bool force_exact = false;          // SG_OPT_FORCE_EXACT: float64 pipeline (exact.hpp) whatever the output dtype
  bool exact_materialised = false;   // SG_OPT_EXACT_MATERIALISED: the float64 pipeline always through exact.hpp's materialised fields
  DevBuf norm64;                     // sum_q win^2[256 q + s] as doubles (default geometry: k_apply_fast64's window envelope)
  unsigned need_era = 0;             // epoch >> 30 of the last one-pass gate call (tagged floor-test flags, stage_onepass)
  DevBuf xP, xraw, xM, xtmp, xseg;   // fields of the exact path
  bool force_split = false;          // SG_OPT_FORCE_SPLIT: decide / smooth / apply as three kernels
  int rowgate_mode = 0;              // SG_OPT_FORCE_NOROWGATE: 0 = by batch size, 1 = never, 2 = whenever the shape is eligible
  int64_t n_floor_lazy = 0, n_floor_apriori = 0;   // sg_debug_counter 1 / 2
  int tile_order = 0;                // SG_OPT_TILE_ORDER: 0 (default, round 6) = persistent workgroups looping over tickets (onepass.hpp PERSIST), 2 = one ticket-drawn tile per workgroup (the kernel of rounds 2-5), 1 = tile = block index
                                     // block index (no ticket), 0 = persistent workgroups looping over tickets (round 6: 1 % faster alone on the
                                     // GPU, NOT the default -- next to TorchGate's row gate on a second stream it mis-gates a tile now and then:
                                     // tests/tools/soak_handoff.py)
  int n_cu = 0;                      // compute units of the handle's device (persistent grids)
  int floor_test = 0;                // SG_OPT_FLOOR_TEST: one-pass gate's floor test 0 = predicted, 1 = a priori, 2 = in the gate kernel
  int rg_shape = 16;                 // SG_OPT_ROWGATE_SHAPE: waves per workgroup of the row gate (16 x 1 quad, or 8 x 2 quads)
  bool rg_tap = false;               // SG_OPT_ROWGATE_TAP: keep the row gate's float32 power tile (stage tap 4)
  bool dbg_rg = false;               // the last batch ran on the row gate
  DevBuf rg_count;                   // k_row_gate: number of (row, band) pairs that took the exact path (one counter, never reset)
  bool dbg_xbits = false;            // the last batch's mask bits live in xbits (tile-blocked)
  int dbg_trows = 16, dbg_tstep = 16, dbg_twords = 0, dbg_txw = 9;   // ... in tiles of trows frames every tstep frames, twords granule halves, txw words per row
  int64_t dbg_tf0 = 0;               // first frame of tile 0 of that batch
  int dbg_ntt = 0;                   // tiles per unit incl. the two halo tiles
  int big_M = 0;                     // > 0: long frames (n_fft > 8192, or > 4096 and not a power of two): four-step
                                     // transform of size M through HBM (big.hpp); big_czt: chirp-z on top of it
  int big_czt = 0;
  DevBuf big_twM, big_tw2, big_ch, big_bh, big_W, big_W2;
  int czt_M = 0;                     // > 0: n_fft is not a power of two -> chirp-z kernels (czt.hpp) of size M
  bool mr_ok = false;                // n_fft even, n_fft / 2 <= 2048 with prime factors <= 13, not a power of two: the float32 and
  MrPlan mr{};                       // float64 STFT / decision / apply kernels of mixed.hpp (run-time radix schedule) instead of chirp-z
  DevBuf mr_pt32, mr_pt64;           // the plan's per-pass twiddle tables (mr_pass_tables)
  DevBuf o5tab, o25tab, o20tab;      // k_gate_onepass512 / 256 / 2048: MFMA operands + byte expansion (onepass512.hpp, onepass256.hpp, onepass2048.hpp)
  DevBuf czt_tw64, czt_ch64, czt_bh64, czt_tw32, czt_ch32, czt_bh32;
  bool force_noseam = false;
  bool force_nolean = false;         // SG_OPT_FORCE_NOLEAN: full-size slices + stored frames (2 waves/SIMD)
  bool fast_ok = false;              // default geometry: fused apply kernel available
  bool fast5_ok = false;             // n_fft = win = 512, hop = 128: register-transform kernels of fast512.hpp
  DevBuf invn5;                      // 1 / window envelope per hop phase (128) of that geometry
  bool fast25_ok = false;            // n_fft = win = 256, hop = 64: register-transform kernels of fast256.hpp (round 5)
  DevBuf invn25;                     // 1 / window envelope per hop phase (64)
  bool fast20_ok = false;            // n_fft = win = 2048, hop = 512: register-transform kernels of fast2048.hpp
  DevBuf invn20;                     // 1 / window envelope per hop phase (512)
  bool force_nofast = false;
  bool force_f64_decide = false;     // SG_OPT_FORCE_F64_DECIDE: float64 STFT for every mask decision
  int64_t ktot = 1;                  // (nf+1)^2 (nt+1)^2: integer weight total of the smoothing filter
  bool fused_ok = false;
  double sum_abs_w = 0.0;
  int64_t dbg_units = 0, dbg_T = 0;
  bool dbg_has_P = false;
  bool dbg_has_raw = true;           // the unsmoothed mask field exists (not when k_iir_mask / k_box_mask produced the mask in one kernel)
  bool dbg_fused = false;
  bool dbg_fast = false;
  bool dbg_k16_only = false;         // the last fused general-geometry batch left K counts only (no float mask field): dbg_g
  Geom dbg_g{};
  int64_t dbg_db = 0, dbg_de = 0;  // frames whose mask bits were decided in the last batch
  bool force_unfused = false;  // sg_set_option(SG_OPT_FORCE_UNFUSED): materialised v1 path
  // per-kernel timing with HIP events on the launch stream (sg_profile_*)
  bool roctx_on = false;       // SG_ROCTX=1: a roctx range around every stage's enqueue
  bool prof_on = false;
  uint64_t prof_mask = ~0ull;  // stages that get an event pair (sg_profile_select)
  int prof_override = -1;  // >= 0: book every launch under this stage (noise statistics)
  struct ProfRec { int stage; hipEvent_t a, b; };
  std::vector<ProfRec> prof_live;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[SG_N_STAGES] = {0};
  int64_t prof_cnt[SG_N_STAGES] = {0};
  std::string err;
};

namespace {
// Optional roctx ranges around every stage's enqueue (SG_ROCTX=1 in the environment at sg_create): the stages show
// up by name in `rocprofv3 --marker-trace` next to the kernel trace.  libroctx64 is looked up at run time: the library
// does not depend on it.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    // rocprofv3 listens to the rocprofiler-sdk flavour; libroctx64 is the roctracer one (older tools)
    void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
const Roctx& roctx() {
  static const Roctx r;
  return r;
}

// RAII: records an event pair around one kernel launch when profiling is enabled.
struct ProfScope {
  sg_handle* h;
  hipStream_t st;
  int idx = -1;
  static hipEvent_t get(sg_handle* h) {
    if (!h->prof_pool.empty()) {
      hipEvent_t e = h->prof_pool.back();
      h->prof_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  bool ranged = false;
  ProfScope(sg_handle* h_, int stage, hipStream_t st_) : h(h_), st(st_) {
    if (h->roctx_on && roctx().push) {
      roctx().push(sg_stage_name(h->prof_override >= 0 ? h->prof_override : stage));
      ranged = true;
    }
    if (!h->prof_on) return;
    const int booked = h->prof_override >= 0 ? h->prof_override : stage;
    if (!((h->prof_mask >> booked) & 1ull)) return;
    sg_handle::ProfRec r{h->prof_override >= 0 ? h->prof_override : stage, get(h), get(h)};
    (void)hipEventRecord(r.a, st);
    h->prof_live.push_back(r);
    idx = (int)h->prof_live.size() - 1;
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(h->prof_live[idx].b, st);
    if (ranged) roctx().pop();
  }
};
}  // namespace

#define HIPCHK(h, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (h)->err = fmt("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SG_E_HIP;                                                                       \
    }                                                                                        \
  } while (0)

#define FAIL(h, code, ...)      \
  do {                          \
    (h)->err = fmt(__VA_ARGS__); \
    return (code);              \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size): the call costs microseconds of host
// time, and a short call enqueues ten launches
static hipError_t set_lds(const void* kern, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  size_t& have = done[{dev, kern}];
  if (have >= bytes) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

static int ensure(sg_handle* h, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return SG_OK;
  if (b.p) {
    HIPCHK(h, hipDeviceSynchronize());  // buffer may still be in use by enqueued work
    HIPCHK(h, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    b.p = nullptr;
    FAIL(h, SG_E_NOMEM, "workspace allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
  }
  b.bytes = bytes;
  return SG_OK;
}

// ensure() + zero fill when the buffer was (re)allocated: for state that kernels keep clean themselves
static int ensure_zeroed(sg_handle* h, DevBuf& b, size_t bytes, hipStream_t st, bool* fresh = nullptr) {
  // freshness from the SIZE, not the pointer: the allocator may hand the freed address straight back
  const bool grow = b.bytes < bytes;
  int rc = ensure(h, b, bytes);
  if (rc) return rc;
  if (fresh) *fresh = grow;
  if (grow) HIPCHK(h, hipMemsetAsync(b.p, 0, b.bytes, st));
  return SG_OK;
}

static int upload(sg_handle* h, DevBuf& b, const void* src, size_t bytes) {
  int rc = ensure(h, b, bytes);
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  return SG_OK;
}

static void free_buf(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

static int64_t frames_for(const sg_handle* h, int64_t L) {
  // S: zero extension W//2 per side, padded=False (scipy/_spectral_py.py:2052,2185-2189).
  // T: torch.stft(center=True) pads n_fft//2 per side: 1 + (L + 2 (n//2) - n) // hop
  //    (= 1 + L // hop for even n_fft).
  if (h->p.variant == SG_VARIANT_S) return (L + 2 * (int64_t)(h->W / 2) - h->W) / h->H + 1;
  return 1 + (L + 2 * (int64_t)(h->n / 2) - h->n) / h->H;
}
static int64_t outlen_for(const sg_handle* h, int64_t L) {
  int64_t T = frames_for(h, L);
  if (h->p.variant == SG_VARIANT_S) return (T - 1) * h->H + h->W - 2 * (int64_t)(h->W / 2);
  // torch.istft(center=True) trims n_fft//2 from both ends of the n_fft + (T-1) hop buffer
  return (T - 1) * (int64_t)h->H + (h->n - 2 * (int64_t)(h->n / 2));
}

static Geom make_geom(const sg_handle* h, int64_t Lp) {
  Geom g;
  g.n = h->n; g.W = h->W; g.H = h->H; g.F = h->F; g.FS = h->FS; g.padL = h->padL;
  g.T = frames_for(h, Lp);
  g.Lout = outlen_for(h, Lp);
  return g;
}

// ------------------------------------------------------------------------------------------
// kernel dispatch on the FFT size
// ------------------------------------------------------------------------------------------
template <typename TC, int N>
static hipError_t launch_stft_n(const View& v, const Geom& g, int64_t units, const void* tw, const void* wfull,
                                double* P, float* mag, double* z, double zscale, hipStream_t st,
                                unsigned long long* pmax_bits) {
  // n_fft = 8192 (N = 4096): the whole workgroup cooperates on one frame
  constexpr int NT = team_threads<N>();
  constexpr int WAVES = team_count<N>(sizeof(cx<TC>));
  size_t lds = (size_t)(N + WAVES * lpn<TC>(N)) * sizeof(cx<TC>);
  // few units (the noise clip): one frame per wave so that the grid still covers the chip
  const bool small = units * ((g.T + WAVES * 4 - 1) / (WAVES * 4)) < 1024;
  auto launch = [&](auto kern, int fpw) -> hipError_t {
    if (lds > 65536) {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)((g.T + WAVES * fpw - 1) / (WAVES * fpw)), (unsigned)units);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * NT), lds, st, v, g, (const cx<TC>*)tw, (const TC*)wfull, P, mag,
                       z, zscale, pmax_bits);
    return hipGetLastError();
  };
  if constexpr (sizeof(TC) == 8 && N >= 1024 && N < SG_TEAM_N && SG_STATS_TEAM) {
    // (round 5) the noise clip in float64 at n_fft = 2048: 293 workgroups of four one-frame wavefronts are one
    // latency-bound round of 32 us.  The whole workgroup on one frame (the N >= SG_TEAM_N shape): four times the
    // workgroups, each transform split over 256 threads: 16.5 us.  (n_fft = 1024 measured too: 14.0 -> 15.9 us, so not there.)
    if (small && units * g.T <= 16384) {
      const size_t lds1 = (size_t)(N + lpn<TC>(N)) * sizeof(cx<TC>);
      auto kern = k_stft<TC, N, 1, 1, 256>;
      if (lds1 > 65536) {
        hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds1);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)g.T, (unsigned)units), dim3(256), lds1, st, v, g, (const cx<TC>*)tw,
                         (const TC*)wfull, P, mag, z, zscale, pmax_bits);
      return hipGetLastError();
    }
  }
  if (small) return launch(k_stft<TC, N, WAVES, 1, NT>, 1);
  return launch(k_stft<TC, N, WAVES, 4, NT>, 4);
}

template <typename TC>
static hipError_t launch_stft(int N, const View& v, const Geom& g, int64_t units, const void* tw,
                              const void* wfull, double* P, float* mag, double* z, double zscale,
                              hipStream_t st, unsigned long long* pmax_bits = nullptr) {
  switch (N) {
    case 32: return launch_stft_n<TC, 32>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
    case 64: return launch_stft_n<TC, 64>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
    case 128: return launch_stft_n<TC, 128>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
    case 256: return launch_stft_n<TC, 256>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
    case 512: return launch_stft_n<TC, 512>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
    case 1024: return launch_stft_n<TC, 1024>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
    case 2048: return launch_stft_n<TC, 2048>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
    case 4096: return launch_stft_n<TC, 4096>(v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
  }
  return hipErrorInvalidValue;
}

template <int N, int MODE>
static hipError_t launch_bits_n(const View& v, const Geom& g, int64_t units, const void* tw, const void* wfull,
                                const ThreshConsts& tc, double mag_scale, double top_db,
                                unsigned long long* pmax_bits, unsigned long long* bits, int wpr,
                                hipStream_t st) {
  // from N = 2048 (n_fft = 4096) on the 256 threads of a workgroup share one frame (k_stft does the same): a single wave's
  // 2048-point float64 pass held 512 registers and 476 B of scratch
  constexpr int NT = N >= SG_TEAM_N ? 256 : 64;
  constexpr int WAVES = N >= SG_TEAM_N ? 1 : ((N * sizeof(cx<double>) > 16384) ? 2 : 4);
  constexpr int FPW = 4;
  size_t lds = (size_t)(N + WAVES * lpn<double>(N)) * sizeof(cx<double>) + (size_t)(N + 1) * sizeof(double);
  auto kern = k_stft_bits<N, WAVES, FPW, MODE, NT>;
  if (lds > 65536) {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((g.T + WAVES * FPW - 1) / (WAVES * FPW)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * NT), lds, st, v, g, (const cx<double>*)tw, (const double*)wfull, tc,
                     mag_scale, top_db, pmax_bits, bits, wpr);
  return hipGetLastError();
}

template <int MODE>
static hipError_t launch_bits(int N, const View& v, const Geom& g, int64_t units, const void* tw,
                              const void* wfull, const ThreshConsts& tc, double mag_scale, double top_db,
                              unsigned long long* pmax_bits, unsigned long long* bits, int wpr, hipStream_t st) {
  switch (N) {
    case 32: return launch_bits_n<32, MODE>(v, g, units, tw, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr, st);
    case 64: return launch_bits_n<64, MODE>(v, g, units, tw, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr, st);
    case 128: return launch_bits_n<128, MODE>(v, g, units, tw, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr, st);
    case 256: return launch_bits_n<256, MODE>(v, g, units, tw, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr, st);
    case 512: return launch_bits_n<512, MODE>(v, g, units, tw, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr, st);
    case 1024: return launch_bits_n<1024, MODE>(v, g, units, tw, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr, st);
    case 2048: return launch_bits_n<2048, MODE>(v, g, units, tw, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr, st);
  }
  return hipErrorInvalidValue;
}

template <int N>
static hipError_t launch_decide_lds_n(const sg_handle* h, const View& v, const Geom& g, int64_t units,
                                      const ThreshConsts& tc, unsigned long long* bits, int wpr, hipStream_t st) {
  constexpr int NT = team_threads<N>() > 64 ? 64 : team_threads<N>();   // (a frame never spans wavefronts here)
  constexpr int WAVES = NT < 64 ? 256 / NT : ((N * sizeof(cx<float>) > 8192) ? 2 : 4);
  constexpr int FPW = 4;
  const size_t lds = (size_t)(N + WAVES * N) * sizeof(cx<float>) + (size_t)(N + 1) * sizeof(float);
  auto kern = k_decide_lds<N, WAVES, FPW, NT>;
  if (lds > 65536) {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((g.T + WAVES * FPW - 1) / (WAVES * FPW)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * NT), lds, st, v, g, (const cx<float>*)h->tw32.p,
                     (const float*)h->wa32.p, (const cx<double>*)h->tw64.p, (const double*)h->wfull64.p, tc,
                     h->mag_scale, h->p.top_db, bits, wpr);
  return hipGetLastError();
}

static hipError_t launch_decide_mr(const sg_handle* h, const View& v, const Geom& g, int64_t units, const ThreshConsts& tc,
                                   unsigned long long* bits, int wpr, hipStream_t st);
static hipError_t launch_decide_lds(const sg_handle* h, const View& v, const Geom& g, int64_t units,
                                    const ThreshConsts& tc, unsigned long long* bits, int wpr, hipStream_t st) {
  if (h->mr_ok) return launch_decide_mr(h, v, g, units, tc, bits, wpr, st);
  switch (h->N) {
    case 32: return launch_decide_lds_n<32>(h, v, g, units, tc, bits, wpr, st);
    case 64: return launch_decide_lds_n<64>(h, v, g, units, tc, bits, wpr, st);
    case 128: return launch_decide_lds_n<128>(h, v, g, units, tc, bits, wpr, st);
    case 256: return launch_decide_lds_n<256>(h, v, g, units, tc, bits, wpr, st);
    case 512: return launch_decide_lds_n<512>(h, v, g, units, tc, bits, wpr, st);
    case 1024: return launch_decide_lds_n<1024>(h, v, g, units, tc, bits, wpr, st);
    case 2048: return launch_decide_lds_n<2048>(h, v, g, units, tc, bits, wpr, st);
  }
  return hipErrorInvalidValue;
}

template <int N>
static hipError_t launch_apply_n(const View& v, const Geom& g, int64_t units, const void* tw, const float* wa,
                                 const float* ws, const float* M, float* seg, hipStream_t st,
                                 const unsigned short* K16, float kscale) {
  constexpr int NT = team_threads<N>();
  constexpr int WAVES = team_count<N>(sizeof(cx<float>));
  constexpr int FPW = 4;
  size_t lds = (size_t)(N + WAVES * lpn<float>(N)) * sizeof(cx<float>);
  auto kern = k_apply_istft<N, WAVES, FPW, NT>;
  if (lds > 65536) {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((g.T + WAVES * FPW - 1) / (WAVES * FPW)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * NT), lds, st, v, g, (const cx<float>*)tw, wa, ws, M, seg, K16, kscale);
  return hipGetLastError();
}

static hipError_t launch_apply(int N, const View& v, const Geom& g, int64_t units, const void* tw,
                               const float* wa, const float* ws, const float* M, float* seg, hipStream_t st,
                               const unsigned short* K16 = nullptr, float kscale = 0.f) {
  switch (N) {
    case 32: return launch_apply_n<32>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
    case 64: return launch_apply_n<64>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
    case 128: return launch_apply_n<128>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
    case 256: return launch_apply_n<256>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
    case 512: return launch_apply_n<512>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
    case 1024: return launch_apply_n<1024>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
    case 2048: return launch_apply_n<2048>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
    case 4096: return launch_apply_n<4096>(v, g, units, tw, wa, ws, M, seg, st, K16, kscale);
  }
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// mixed-radix kernels (mixed.hpp): frame lengths 2 N with N = 2^a 3^b 5^c 7^d 11^e 13^f <= 2048 that are not powers of two
// ------------------------------------------------------------------------------------------
template <typename TC>
static hipError_t launch_stft_mr(const sg_handle* h, const View& v, const Geom& g, int64_t units, const void* wfull, double* P,
                                 float* mag, double* z, double zscale, hipStream_t st, unsigned long long* pmax_bits) {
  if constexpr (sizeof(TC) == 8)
    return mr_launch_stft64(h->mr, v, g, units, (const cx<double>*)h->tw64.p, (const cx<double>*)h->mr_pt64.p, (const double*)wfull, P, mag, z, zscale,
                            pmax_bits, st);
  else
    return mr_launch_stft32(h->mr, v, g, units, (const cx<float>*)h->tw32.p, (const cx<float>*)h->mr_pt32.p, (const float*)wfull, P, mag, z, zscale,
                            pmax_bits, st);
}
template <int MODE>
static hipError_t launch_bits_mr(const sg_handle* h, const View& v, const Geom& g, int64_t units, const ThreshConsts& tc,
                                 unsigned long long* pmax_bits, unsigned long long* bits, int wpr, hipStream_t st) {
  return mr_launch_bits(MODE, h->mr, v, g, units, (const cx<double>*)h->tw64.p, (const cx<double>*)h->mr_pt64.p, (const double*)h->wfull64.p, tc, h->mag_scale,
                        h->p.top_db, pmax_bits, bits, wpr, st);
}
static hipError_t launch_decide_mr(const sg_handle* h, const View& v, const Geom& g, int64_t units, const ThreshConsts& tc,
                                   unsigned long long* bits, int wpr, hipStream_t st) {
  return mr_launch_decide(h->mr, v, g, units, (const cx<float>*)h->tw32.p, (const cx<float>*)h->mr_pt32.p, (const float*)h->wa32.p, (const cx<double>*)h->tw64.p,
                          (const double*)h->wfull64.p, tc, h->mag_scale, h->p.top_db, bits, wpr, st);
}
static hipError_t launch_apply_mr(const sg_handle* h, const View& v, const Geom& g, int64_t units, const float* wa, const float* ws,
                                  const float* M, float* seg, hipStream_t st, const unsigned short* K16, float kscale) {
  return mr_launch_apply(h->mr, v, g, units, (const cx<float>*)h->tw32.p, (const cx<float>*)h->mr_pt32.p, wa, ws, M, seg, K16, kscale, st);
}
template <int MODE>
static hipError_t launch_bits(int N, const View& v, const Geom& g, int64_t units, const void* tw, const void* wfull,
                              const ThreshConsts& tc, double mag_scale, double top_db, unsigned long long* pmax_bits,
                              unsigned long long* bits, int wpr, hipStream_t st);
// (the decision kernels of the fused path, whatever transform the frame length runs on)
template <int MODE>
static hipError_t launch_bits_any(const sg_handle* h, const View& v, const Geom& g, int64_t units, const ThreshConsts& tc,
                                  unsigned long long* pmax_bits, unsigned long long* bits, int wpr, hipStream_t st) {
  if (h->mr_ok) return launch_bits_mr<MODE>(h, v, g, units, tc, pmax_bits, bits, wpr, st);
  return launch_bits<MODE>(h->N, v, g, units, h->tw64.p, h->wfull64.p, tc, h->mag_scale, h->p.top_db, pmax_bits, bits, wpr, st);
}

// ------------------------------------------------------------------------------------------
// chirp-z kernels (n_fft not a power of two): dispatch on the convolution size M
// ------------------------------------------------------------------------------------------
template <int M>
struct CztShape {
  static constexpr int NT = M >= 2048 ? 256 : 64;  // threads per frame
  static constexpr int FR = M >= 2048 ? 1 : 4;     // frames in flight per workgroup
};

template <typename TC, int M>
static hipError_t launch_stft_czt_m(const View& v, const Geom& g, int64_t units, const CztTabs<TC>& tb,
                                    const void* wfull, double* P, float* mag, double* z, double zscale,
                                    hipStream_t st, unsigned long long* pmax_bits) {
  constexpr int NT = CztShape<M>::NT, FR = CztShape<M>::FR;
  const size_t lds = (size_t)FR * lpn<TC>(M) * sizeof(cx<TC>);
  auto kern = k_stft_czt<TC, M, NT, FR>;
  if (lds > 65536) {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  const int fpb = units * ((g.T + FR * 4 - 1) / (FR * 4)) < 1024 ? 1 : 4;
  dim3 grid((unsigned)((g.T + FR * fpb - 1) / (FR * fpb)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(NT * FR), lds, st, v, g, tb, (const TC*)wfull, P, mag, z, zscale, pmax_bits,
                     fpb);
  return hipGetLastError();
}

template <int M>
static hipError_t launch_apply_czt_m(const View& v, const Geom& g, int64_t units, const CztTabs<float>& tb,
                                     const float* wa, const float* ws, const float* Mk, float* seg,
                                     hipStream_t st) {
  constexpr int NT = CztShape<M>::NT, FR = CztShape<M>::FR;
  const size_t lds = (size_t)FR * lpn<float>(M) * sizeof(cx<float>);
  auto kern = k_apply_istft_czt<M, NT, FR>;
  if (lds > 65536) {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  const int fpb = 4;
  dim3 grid((unsigned)((g.T + FR * fpb - 1) / (FR * fpb)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(NT * FR), lds, st, v, g, tb, wa, ws, Mk, seg, fpb);
  return hipGetLastError();
}

#define SG_CZT_SWITCH(M_, CALL)            \
  switch (M_) {                            \
    case 64: return CALL(64);              \
    case 128: return CALL(128);            \
    case 256: return CALL(256);            \
    case 512: return CALL(512);            \
    case 1024: return CALL(1024);          \
    case 2048: return CALL(2048);          \
    case 4096: return CALL(4096);          \
    case 8192: return CALL(8192);          \
  }                                        \
  return hipErrorInvalidValue

template <typename TC>
static CztTabs<TC> czt_tabs(const sg_handle* h) {
  if (sizeof(TC) == 8)
    return {(const cx<TC>*)h->czt_tw64.p, (const cx<TC>*)h->czt_ch64.p, (const cx<TC>*)h->czt_bh64.p};
  return {(const cx<TC>*)h->czt_tw32.p, (const cx<TC>*)h->czt_ch32.p, (const cx<TC>*)h->czt_bh32.p};
}

// ------------------------------------------------------------------------------------------
// long frames (big.hpp): four-step transform through HBM, frames in batches of <= 256 MB of work buffer
// ------------------------------------------------------------------------------------------
static big::BigTabs big_tabs(const sg_handle* h) {
  return big::BigTabs{(const big::cd*)h->big_twM.p, (const big::cd*)h->big_tw2.p, (const big::cd*)h->big_ch.p,
                      (const big::cd*)h->big_bh.p, h->big_M, h->big_M / 16, h->big_czt};
}

template <int M2>
static hipError_t big_rows_m2(big::cd* W, const big::BigTabs& tb, int64_t nf, int mode, hipStream_t st) {
  const size_t lds = (size_t)lpn<double>(M2) * sizeof(big::cd);
  auto go = [&](auto kern) -> hipError_t {
    if (lds > 65536) {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nf * 16)), dim3(256), lds, st, W, tb, nf);
    return hipGetLastError();
  };
  if (mode == 0) return go(big::k_big_rows<M2, false, false>);
  if (mode == 1) return go(big::k_big_rows<M2, true, false>);
  return go(big::k_big_rows<M2, false, true>);
}

// mode 0: forward rows, 1: inverse rows, 2: forward rows x B then inverse rows (chirp-z)
static hipError_t big_rows(big::cd* W, const big::BigTabs& tb, int64_t nf, int mode, hipStream_t st) {
  switch (tb.M2) {
    case 1024: return big_rows_m2<1024>(W, tb, nf, mode, st);
    case 2048: return big_rows_m2<2048>(W, tb, nf, mode, st);
    case 4096: return big_rows_m2<4096>(W, tb, nf, mode, st);
  }
  return hipErrorInvalidValue;
}

// the length-n DFT of the staged frames: natural order in; out = permuted bins (power of two) or the chirp-z
// convolution in natural order
static hipError_t big_dft(big::cd* W, const big::BigTabs& tb, int64_t nf, hipStream_t st) {
  const dim3 gc((unsigned)((tb.M2 + 255) / 256), (unsigned)nf);
  hipLaunchKernelGGL(big::k_big_cols<false>, gc, dim3(256), 0, st, W, tb, nf);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if ((e = big_rows(W, tb, nf, tb.czt ? 2 : 0, st)) != hipSuccess) return e;
  if (tb.czt) {
    hipLaunchKernelGGL(big::k_big_cols<true>, gc, dim3(256), 0, st, W, tb, nf);
    e = hipGetLastError();
  }
  return e;
}

static int64_t big_batch(const sg_handle* h, int64_t total) {
  const int64_t per = (int64_t)h->big_M * (int64_t)sizeof(big::cd);
  return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(total, 32768), ((int64_t)256 << 20) / per));
}

static int big_stft(sg_handle* h, const View& v, const Geom& g, int64_t units, double* P, float* mag, double* z,
                    double zscale, hipStream_t st, unsigned long long* pmax_bits) {
  const big::BigTabs tb = big_tabs(h);
  const int64_t total = units * g.T, nb = big_batch(h, total);
  int rc = ensure(h, h->big_W, (size_t)nb * tb.M * sizeof(big::cd));
  if (rc) return rc;
  big::cd* W = (big::cd*)h->big_W.p;
  for (int64_t f0 = 0; f0 < total; f0 += nb) {
    const int64_t nf = std::min(nb, total - f0);
    hipLaunchKernelGGL(big::k_big_frames, dim3(64, (unsigned)nf), dim3(256), 0, st, v, g, tb, (const double*)h->wfull64.p,
                       W, f0, nf);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, big_dft(W, tb, nf, st));
    hipLaunchKernelGGL(big::k_big_out, dim3((unsigned)std::min<int64_t>(32, (g.F + 255) / 256), (unsigned)nf), dim3(256), 0,
                       st, (const big::cd*)W, g, tb, f0, nf, P, mag, z, zscale, pmax_bits);
    HIPCHK(h, hipGetLastError());
  }
  return SG_OK;
}

template <typename TM, typename TS>
static int big_apply(sg_handle* h, const View& v, const Geom& g, int64_t units, const TM* Mk, TS* seg,
                     hipStream_t st) {
  const big::BigTabs tb = big_tabs(h);
  const int64_t total = units * g.T, nb = big_batch(h, total);
  int rc = ensure(h, h->big_W, (size_t)nb * tb.M * sizeof(big::cd));
  if (!rc) rc = ensure(h, h->big_W2, (size_t)nb * tb.M * sizeof(big::cd));
  if (rc) return rc;
  big::cd *W = (big::cd*)h->big_W.p, *W2 = (big::cd*)h->big_W2.p;
  for (int64_t f0 = 0; f0 < total; f0 += nb) {
    const int64_t nf = std::min(nb, total - f0);
    hipLaunchKernelGGL(big::k_big_frames, dim3(64, (unsigned)nf), dim3(256), 0, st, v, g, tb, (const double*)h->wfull64.p,
                       W, f0, nf);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, big_dft(W, tb, nf, st));
    hipLaunchKernelGGL(big::k_big_mask<TM>, dim3(64, (unsigned)nf), dim3(256), 0, st, (const big::cd*)W, W2, g, tb, f0, nf, Mk);
    HIPCHK(h, hipGetLastError());
    if (tb.czt) {
      HIPCHK(h, big_dft(W2, tb, nf, st));   // inverse DFT = forward chirp-z of conj(Y), conjugated
    } else {
      HIPCHK(h, big_rows(W2, tb, nf, 1, st));
      hipLaunchKernelGGL(big::k_big_cols<true>, dim3((unsigned)((tb.M2 + 255) / 256), (unsigned)nf), dim3(256), 0, st, W2,
                         tb, nf);
      HIPCHK(h, hipGetLastError());
    }
    hipLaunchKernelGGL(big::k_big_seg<TS>, dim3(64, (unsigned)nf), dim3(256), 0, st, (const big::cd*)W2, g, tb, f0, nf,
                       (const double*)h->wfull64.p, seg);
    HIPCHK(h, hipGetLastError());
  }
  return SG_OK;
}

// forward STFT of `units` units on the kernels that fit the handle's frame length
template <typename TC>
static hipError_t stft_any(const sg_handle* h, const View& v, const Geom& g, int64_t units, double* P, float* mag,
                           double* z, double zscale, hipStream_t st, unsigned long long* pmax_bits = nullptr) {
  if (h->big_M) {
    const int rc = big_stft(const_cast<sg_handle*>(h), v, g, units, P, mag, z, zscale, st, pmax_bits);
    return rc == SG_OK ? hipSuccess : (rc == SG_E_NOMEM ? hipErrorOutOfMemory : hipErrorUnknown);
  }
  const void* wfull = sizeof(TC) == 8 ? h->wfull64.p : h->wa32.p;
  if (sizeof(TC) == 8 && h->fast_ok && !h->force_nofast && P && !mag && !z && units * ((g.T + 15) / 16) >= 512) {
    // default geometry, float64 powers only, enough 16-frame blocks for two per CU: register FFT
    // core (fast64.hpp).  (A single noise clip is faster on k_stft's one-frame-per-wave grid: 13.7 vs 18 us.)
    // Samples staged as float32 where that is exact (float32 / int16 recordings), as float64 otherwise.
    constexpr int WAVES = 4;
    fast::Pow64Args A{v, g, (const double*)h->wfull64.p, (const fast::cd*)h->tw64.p, P, pmax_bits};
    const size_t lds = (size_t)(fast::FN + WAVES * 4 * fast::FSLOTS_D + 32) * sizeof(fast::cd);
    dim3 grid((unsigned)((g.T + WAVES * 4 - 1) / (WAVES * 4)), (unsigned)units);
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, st, A);
      return hipGetLastError();
    };
    if (v.dtype == SG_F32 || v.dtype == SG_I16)
      return pmax_bits ? go(fast::k_power_fast64<WAVES, true>) : go(fast::k_power_fast64<WAVES, false>);
    return pmax_bits ? go(fast::k_power_fast64<WAVES, true, double>) : go(fast::k_power_fast64<WAVES, false, double>);
  }
  if (!h->czt_M) {
    const void* tw = sizeof(TC) == 8 ? h->tw64.p : h->tw32.p;
    return launch_stft<TC>(h->N, v, g, units, tw, wfull, P, mag, z, zscale, st, pmax_bits);
  }
  if (h->mr_ok) return launch_stft_mr<TC>(h, v, g, units, wfull, P, mag, z, zscale, st, pmax_bits);
  const CztTabs<TC> tb = czt_tabs<TC>(h);
#define SG_CALL(M) launch_stft_czt_m<TC, M>(v, g, units, tb, wfull, P, mag, z, zscale, st, pmax_bits)
  SG_CZT_SWITCH(h->czt_M, SG_CALL);
#undef SG_CALL
}

// the general apply kernel that can read the K counts of the fused bit-mask stages directly (k16_apply_ok)
static bool k16_apply_geom(const sg_handle* h) { return !h->big_M && (!h->czt_M || h->mr_ok); }

static hipError_t apply_any(const sg_handle* h, const View& v, const Geom& g, int64_t units, const float* Mk,
                            float* seg, hipStream_t st, const unsigned short* K16 = nullptr, float kscale = 0.f) {
  if (h->big_M) {
    const int rc = big_apply(const_cast<sg_handle*>(h), v, g, units, Mk, seg, st);
    return rc == SG_OK ? hipSuccess : (rc == SG_E_NOMEM ? hipErrorOutOfMemory : hipErrorUnknown);
  }
  const float* wa = (const float*)h->wa32.p;
  const float* ws = (const float*)h->ws32.p;
  if (!h->czt_M) return launch_apply(h->N, v, g, units, h->tw32.p, wa, ws, Mk, seg, st, K16, kscale);
  if (h->mr_ok) return launch_apply_mr(h, v, g, units, wa, ws, Mk, seg, st, K16, kscale);
  const CztTabs<float> tb = czt_tabs<float>(h);
#define SG_CALL(M) launch_apply_czt_m<M>(v, g, units, tb, wa, ws, Mk, seg, st)
  SG_CZT_SWITCH(h->czt_M, SG_CALL);
#undef SG_CALL
}

static DbFast db_fast_consts(const sg_handle* h) {
  const double eps = 2.220446049250313e-16;
  const double ymin = eps * 134217728.0 / h->mag_scale;   // eps 2^27 / mag_scale
  return DbFast{(const double*)h->logtab.p, 20.0 * std::log10(h->mag_scale), ymin * ymin, eps / h->mag_scale};
}

static unsigned grid_1d(int64_t work, int block) {
  int64_t b = (work + block - 1) / block;
  return (unsigned)std::min<int64_t>(std::max<int64_t>(b, 1), 256 * 32);
}

// ------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------
extern "C" int sg_version(void) { return SG_VERSION; }

extern "C" const char* sg_last_error(const sg_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static std::vector<double> triangle(int m) {
  // base.py:7-29 one axis: [1..m, m+1, m..1] / (m+1)
  std::vector<double> v(2 * m + 1);
  for (int i = 0; i <= 2 * m; ++i) v[i] = (double)(i <= m ? i + 1 : 2 * m + 1 - i) / (double)(m + 1);
  return v;
}

// Tables of the chirp-z kernels (czt.hpp) for a frame length n that is not a power of two:
// chirp conj(b[j]) = exp(-i pi j^2 / n) (j^2 reduced mod 2n in integers), B = FFT_M(b wrapped) / M
// (host radix-2 transform in long double), master twiddles w_{2M}^k of the M-point device core.
static int build_czt(sg_handle* h) {
  typedef long double ld;
  const ld PI = 3.14159265358979323846264338327950288L;
  const int n = h->n;
  int M = 64;
  while (M < 2 * n - 1) M *= 2;
  h->czt_M = M;
  std::vector<ld> br(M, 0.0L), bi(M, 0.0L), cr(n), ci(n);
  for (int j = 0; j < n; ++j) {
    const int64_t q = ((int64_t)j * j) % (2 * (int64_t)n);
    const ld a = PI * (ld)q / (ld)n;
    cr[j] = cosl(a);
    ci[j] = -sinl(a);
    br[j] = cr[j];
    bi[j] = -ci[j];
    if (j) {
      br[M - j] = br[j];
      bi[M - j] = bi[j];
    }
  }
  // in-place iterative radix-2 DIT
  for (int i = 1, j = 0; i < M; ++i) {
    int bit = M >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(br[i], br[j]);
      std::swap(bi[i], bi[j]);
    }
  }
  for (int len = 2; len <= M; len <<= 1) {
    for (int k = 0; k < len / 2; ++k) {
      const ld a = -2.0L * PI * (ld)k / (ld)len;
      const ld wr = cosl(a), wi = sinl(a);
      for (int i = k; i < M; i += len) {
        const int j = i + len / 2;
        const ld xr = br[j] * wr - bi[j] * wi, xi = br[j] * wi + bi[j] * wr;
        br[j] = br[i] - xr;
        bi[j] = bi[i] - xi;
        br[i] += xr;
        bi[i] += xi;
      }
    }
  }
  std::vector<cx<double>> tw64(M), ch64(n), bh64(M);
  std::vector<cx<float>> tw32(M), ch32(n), bh32(M);
  for (int k = 0; k < M; ++k) {
    const ld a = -PI * (ld)k / (ld)M;
    tw64[k] = {(double)cosl(a), (double)sinl(a)};
    tw32[k] = {(float)cosl(a), (float)sinl(a)};
    bh64[k] = {(double)(br[k] / (ld)M), (double)(bi[k] / (ld)M)};
    bh32[k] = {(float)(br[k] / (ld)M), (float)(bi[k] / (ld)M)};
  }
  for (int j = 0; j < n; ++j) {
    ch64[j] = {(double)cr[j], (double)ci[j]};
    ch32[j] = {(float)cr[j], (float)ci[j]};
  }
  int rc = upload(h, h->czt_tw64, tw64.data(), tw64.size() * sizeof(cx<double>));
  if (!rc) rc = upload(h, h->czt_ch64, ch64.data(), ch64.size() * sizeof(cx<double>));
  if (!rc) rc = upload(h, h->czt_bh64, bh64.data(), bh64.size() * sizeof(cx<double>));
  if (!rc) rc = upload(h, h->czt_tw32, tw32.data(), tw32.size() * sizeof(cx<float>));
  if (!rc) rc = upload(h, h->czt_ch32, ch32.data(), ch32.size() * sizeof(cx<float>));
  if (!rc) rc = upload(h, h->czt_bh32, bh32.data(), bh32.size() * sizeof(cx<float>));
  return rc;
}

// Host radix-2 transform in long double (tables only)
static void host_fft(std::vector<long double>& br, std::vector<long double>& bi) {
  typedef long double ld;
  const ld PI = 3.14159265358979323846264338327950288L;
  const int M = (int)br.size();
  for (int i = 1, j = 0; i < M; ++i) {
    int bit = M >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(br[i], br[j]);
      std::swap(bi[i], bi[j]);
    }
  }
  for (int len = 2; len <= M; len <<= 1) {
    for (int k = 0; k < len / 2; ++k) {
      const ld a = -2.0L * PI * (ld)k / (ld)len;
      const ld wr = cosl(a), wi = sinl(a);
      for (int i = k; i < M; i += len) {
        const int j = i + len / 2;
        const ld xr = br[j] * wr - bi[j] * wi, xi = br[j] * wi + bi[j] * wr;
        br[j] = br[i] - xr;
        bi[j] = bi[i] - xi;
        br[i] += xr;
        bi[i] += xi;
      }
    }
  }
}

// Tables of the long-frame kernels (big.hpp): w_M^j, the master twiddles of the M2-point row transform and, for a
// frame length that is not a power of two, the chirp exp(-i pi j^2 / n) and B = FFT_M(b wrapped) / M in the PERMUTED
// order pos(k) = (k & 15) * M2 + (k >> 4) the four-step transform leaves its output in.
static int build_big(sg_handle* h, bool pow2) {
  typedef long double ld;
  const ld PI = 3.14159265358979323846264338327950288L;
  const int n = h->n;
  int M = n;
  if (!pow2) {
    M = 16384;
    while (M < 2 * n - 1) M *= 2;
  }
  const int M2 = M / 16;
  h->big_M = M;
  h->big_czt = pow2 ? 0 : 1;
  std::vector<cx<double>> twM(M), tw2(M2);
  for (int j = 0; j < M; ++j) {
    const ld a = -2.0L * PI * (ld)j / (ld)M;
    twM[j] = {(double)cosl(a), (double)sinl(a)};
  }
  for (int k = 0; k < M2; ++k) {
    const ld a = -PI * (ld)k / (ld)M2;
    tw2[k] = {(double)cosl(a), (double)sinl(a)};
  }
  int rc = upload(h, h->big_twM, twM.data(), twM.size() * sizeof(cx<double>));
  if (!rc) rc = upload(h, h->big_tw2, tw2.data(), tw2.size() * sizeof(cx<double>));
  if (rc || pow2) return rc;
  std::vector<ld> br(M, 0.0L), bi(M, 0.0L);
  std::vector<cx<double>> ch(n), bh(M);
  for (int j = 0; j < n; ++j) {
    const int64_t q = ((int64_t)j * j) % (2 * (int64_t)n);
    const ld a = PI * (ld)q / (ld)n;
    ch[j] = {(double)cosl(a), (double)-sinl(a)};
    br[j] = cosl(a);
    bi[j] = sinl(a);
    if (j) {
      br[M - j] = br[j];
      bi[M - j] = bi[j];
    }
  }
  host_fft(br, bi);
  for (int k = 0; k < M; ++k)
    bh[(size_t)(k & 15) * M2 + (k >> 4)] = {(double)(br[k] / (ld)M), (double)(bi[k] / (ld)M)};
  rc = upload(h, h->big_ch, ch.data(), ch.size() * sizeof(cx<double>));
  if (!rc) rc = upload(h, h->big_bh, bh.data(), bh.size() * sizeof(cx<double>));
  return rc;
}

extern "C" int sg_create(const sg_params* p, const double* window_host, sg_handle** out) {
  if (!p || !out) {
    g_create_error = "sg_create: null argument";
    return SG_E_INVALID;
  }
  auto bad = [&](int code, const std::string& m) {
    g_create_error = m;
    return code;
  };
  int n = p->n_fft;
  // powers of two 64..8192 run on the Stockham kernels; every other length 4..4096 on the chirp-z
  // kernels (czt.hpp); longer frames -- powers of two up to 65536, any other length up to 32768 -- on the
  // four-step transform through HBM (big.hpp)
  const bool pow2 = n >= 64 && (n & (n - 1)) == 0;
  const bool bigf = (pow2 && n > 8192) || (!pow2 && n > 4096);
  if (n < 4 || (pow2 && n > 65536) || (!pow2 && n > 32768))
    return bad(SG_E_UNSUPPORTED,
               fmt("n_fft=%d unsupported: must be in [4, 32768], or a power of two up to 65536", n));
  if (p->win_length < 2 || p->win_length > n)
    return bad(SG_E_INVALID, fmt("win_length=%d must be in [2, n_fft=%d]", p->win_length, n));
  if (p->hop_length < 1 || p->hop_length > p->win_length)
    return bad(SG_E_INVALID, fmt("hop_length=%d must be in [1, win_length]", p->hop_length));
  if (p->variant != SG_VARIANT_S && p->variant != SG_VARIANT_T) return bad(SG_E_INVALID, "bad variant");
  if (p->smooth_mask && (p->n_grad_freq < 1 || p->n_grad_time < 1))
    return bad(SG_E_INVALID, "n_grad_freq / n_grad_time must be >= 1");
  if (p->variant == SG_VARIANT_S && (p->padding < 0 || p->chunk_size < 1))
    return bad(SG_E_INVALID, "chunk_size must be >= 1 and padding >= 0");
  if (!p->stationary && p->variant == SG_VARIANT_T && p->n_movemean < 1)
    return bad(SG_E_INVALID, "n_movemean must be >= 1");

  sg_handle* h = new sg_handle();
  h->p = *p;
  {
    const char* e = getenv("SG_ROCTX");
    h->roctx_on = e && e[0] == '1';
  }
  h->n = n;
  h->N = n / 2;
  h->W = p->win_length;
  h->H = p->hop_length;
  h->F = n / 2 + 1;
  h->FS = (h->F + 15) / 16 * 16;
  const int W = h->W;
  std::vector<double> w(W);
  if (window_host) {
    for (int k = 0; k < W; ++k) w[k] = window_host[k];
  } else {
    for (int k = 0; k < W; ++k) w[k] = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)k / (double)W);
  }
  h->sum_w = 0.0;
  for (int k = 0; k < W; ++k) h->sum_w += w[k];
  for (int k = 0; k < W; ++k) h->sum_abs_w += std::fabs(w[k]);
  {
    int64_t a = p->smooth_mask ? (int64_t)(p->n_grad_freq + 1) * (p->n_grad_freq + 1) : 1;
    int64_t b = p->smooth_mask ? (int64_t)(p->n_grad_time + 1) * (p->n_grad_time + 1) : 1;
    h->ktot = a * b;
    // fused (bit-mask) path: variant-S stationary gate whose integer smoothing sums fit uint16
    // frame lengths 2 N, N a product of 2, 3, 5, 7, 11, 13: the mixed-radix kernels (mixed.hpp) -- and with them the fused
    // (bit-mask) path -- instead of chirp-z
    h->mr_ok = !pow2 && !bigf && n % 2 == 0 && n >= 8 && mr_make_plan(n / 2, &h->mr) && getenv("SG_NO_MIXED_RADIX") == nullptr;
    h->fused_ok = p->variant == SG_VARIANT_S && p->stationary && h->ktot <= 65535 && (pow2 || h->mr_ok) && n <= 4096 &&
                  (!p->smooth_mask || p->n_grad_time <= 96);
    if (h->fused_ok && p->smooth_mask) {
      // the integer smoothing kernel holds (tt + 2 nt) rows of all F bins in LDS: 64-frame tiles, lower ones for long rows
      const int wpr = (h->F + 63) / 64;
      const bool small = (p->n_grad_freq + 1) * (p->n_grad_freq + 1) <= 255;
      size_t lds = 0;
      for (h->sm2_tt = smooth2_tt_max(h->F); h->sm2_tt >= 16; h->sm2_tt >>= 1) {
        const int rows = h->sm2_tt + 2 * p->n_grad_time;
        lds = smooth2_cf_bytes(rows + 2, h->F, small ? 1 : 2) + (size_t)rows * (wpr + 2) * 8 + 8192;
        if (lds <= 150 * 1024) break;
      }
      if (lds > 150 * 1024 || p->n_grad_freq > 30) h->fused_ok = false;
    }
  }
  // window embedded in an n_fft frame: scipy zero-pads the windowed frame at the END
  // (scipy/_spectral_py.py:2202) and extends the signal by W//2; torch centres the window
  // inside n_fft and pads the signal by n_fft//2.
  std::vector<double> wfull(n, 0.0);
  int left = (p->variant == SG_VARIANT_S) ? 0 : (n - W) / 2;
  for (int k = 0; k < W; ++k) wfull[left + k] = w[k];
  h->padL = (p->variant == SG_VARIANT_S) ? W / 2 : n / 2;
  h->mag_scale = (p->variant == SG_VARIANT_S) ? 1.0 / h->sum_w : 1.0;

  std::vector<cx<double>> tw64(h->N);
  std::vector<cx<float>> tw32(h->N);
  for (int k = 0; k < h->N; ++k) {
    long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
    tw64[k] = {(double)cosl(a), (double)sinl(a)};
    tw32[k] = {(float)cosl(a), (float)sinl(a)};
  }
  std::vector<float> wa32(n), ws32(n), wsq32(n);
  for (int k = 0; k < n; ++k) {
    wa32[k] = (float)wfull[k];
    // synthesis window incl. the inverse-transform normalisation: the half-size complex core leaves
    // a factor n/2, the chirp-z inverse a factor n
    ws32[k] = (float)(wfull[k] / ((pow2 || h->mr_ok) ? (double)h->N : (double)n));
    wsq32[k] = (float)(wfull[k] * wfull[k]);
  }
  int rc = SG_OK;
  if (!rc) rc = upload(h, h->tw64, tw64.data(), tw64.size() * sizeof(cx<double>));
  if (!rc && h->mr_ok) {
    std::vector<double> pt((size_t)std::max(1, h->mr.ptotal) * 2, 0.0);
    mr_pass_tables(h->mr, pt.data());
    std::vector<float> pt32(pt.begin(), pt.end());
    rc = upload(h, h->mr_pt64, pt.data(), pt.size() * sizeof(double));
    if (!rc) rc = upload(h, h->mr_pt32, pt32.data(), pt32.size() * sizeof(float));
  }
  {
    // db_fast: centre c_i = 1 + (i + 1/2) / 128 of the i-th mantissa slice; the logarithm is that of the ROUNDED
    // reciprocal, so that log2(m) = -log2(t_i) + log2(1 + (m t_i - 1)) holds exactly
    std::vector<double> lt(256);
    for (int i = 0; i < 128; ++i) {
      const double ti = (double)(1.0L / (1.0L + ((long double)i + 0.5L) / 128.0L));
      lt[2 * i] = ti;
      lt[2 * i + 1] = (double)(-log2l((long double)ti));
    }
    if (!rc) rc = upload(h, h->logtab, lt.data(), lt.size() * sizeof(double));
  }
  if (!rc) rc = upload(h, h->tw32, tw32.data(), tw32.size() * sizeof(cx<float>));
  if (!rc) rc = upload(h, h->wfull64, wfull.data(), wfull.size() * sizeof(double));
  if (!rc) rc = upload(h, h->wa32, wa32.data(), wa32.size() * sizeof(float));
  if (!rc) rc = upload(h, h->ws32, ws32.data(), ws32.size() * sizeof(float));
  if (!rc) rc = upload(h, h->wsq32, wsq32.data(), wsq32.size() * sizeof(float));
  if (!rc && bigf) rc = build_big(h, pow2);
  else if (!rc && !pow2) rc = build_czt(h);
  if (!rc && p->smooth_mask) {
    auto vf = triangle(p->n_grad_freq), vt = triangle(p->n_grad_time);
    double sf = 0, stt = 0;
    for (double x : vf) sf += x;
    for (double x : vt) stt += x;
    std::vector<float> kf(vf.size()), kt(vt.size());
    for (size_t i = 0; i < vf.size(); ++i) kf[i] = (float)(vf[i] / sf);
    for (size_t i = 0; i < vt.size(); ++i) kt[i] = (float)(vt[i] / stt);
    if (!rc) rc = upload(h, h->kf, kf.data(), kf.size() * sizeof(float));
    if (!rc) rc = upload(h, h->kt, kt.data(), kt.size() * sizeof(float));
  }
  if (!rc && n == 1024 && W == 1024 && h->H == 256) {
    std::vector<cx<float>> t512(512);
    for (int j = 0; j < 512; ++j) {
      long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / 512.0L;
      t512[j] = {(float)cosl(a), (float)sinl(a)};
    }
    std::vector<float> invn(256);
    for (int s2 = 0; s2 < 256; ++s2) {
      double acc = 0.0;
      for (int q = 0; q < 4; ++q) acc += wfull[256 * q + s2] * wfull[256 * q + s2];
      invn[s2] = (float)(acc > 1e-10 ? 1.0 / acc : 1.0);
    }
    std::vector<double> norm64(256);
    for (int s2 = 0; s2 < 256; ++s2) {
      double acc = 0.0;
      for (int q = 0; q < 4; ++q) acc += wfull[256 * q + s2] * wfull[256 * q + s2];   // frames t-3 .. t in scipy's order
      norm64[s2] = acc;
    }
    rc = upload(h, h->tw512, t512.data(), t512.size() * sizeof(cx<float>));
    if (!rc) rc = upload(h, h->invn, invn.data(), invn.size() * sizeof(float));
    if (!rc) rc = upload(h, h->norm64, norm64.data(), norm64.size() * sizeof(double));
    h->fast_ok = true;
  }
  if (!rc && n == 512 && W == 512 && h->H == 128) {
    // fast512.hpp: the 512-point complex transform of the default geometry carries two real frames of 512 samples
    std::vector<cx<float>> t512(512);
    for (int j = 0; j < 512; ++j) {
      long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / 512.0L;
      t512[j] = {(float)cosl(a), (float)sinl(a)};
    }
    std::vector<float> invn(128);
    for (int s2 = 0; s2 < 128; ++s2) {
      double acc = 0.0;
      for (int q = 0; q < 4; ++q) acc += wfull[128 * q + s2] * wfull[128 * q + s2];
      invn[s2] = (float)(acc > 1e-10 ? 1.0 / acc : 1.0);
    }
    rc = upload(h, h->tw512, t512.data(), t512.size() * sizeof(cx<float>));
    if (!rc) rc = upload(h, h->invn5, invn.data(), invn.size() * sizeof(float));
    h->fast5_ok = true;
  }
  if (!rc && n == 256 && W == 256 && h->H == 64) {
    // fast256.hpp: the same 512-point register transform carries FOUR real frames of 256 samples
    std::vector<cx<float>> t512(512);
    for (int j = 0; j < 512; ++j) {
      long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / 512.0L;
      t512[j] = {(float)cosl(a), (float)sinl(a)};
    }
    std::vector<float> invn(64);
    for (int s2 = 0; s2 < 64; ++s2) {
      double acc = 0.0;
      for (int q = 0; q < 4; ++q) acc += wfull[64 * q + s2] * wfull[64 * q + s2];
      invn[s2] = (float)(acc > 1e-10 ? 1.0 / acc : 1.0);
    }
    rc = upload(h, h->tw512, t512.data(), t512.size() * sizeof(cx<float>));
    if (!rc) rc = upload(h, h->invn25, invn.data(), invn.size() * sizeof(float));
    h->fast25_ok = true;
  }
  if (!rc && n == 2048 && W == 2048 && h->H == 512) {
    std::vector<float> invn(512);
    for (int s2 = 0; s2 < 512; ++s2) {
      double acc = 0.0;
      for (int q = 0; q < 4; ++q) acc += wfull[512 * q + s2] * wfull[512 * q + s2];
      invn[s2] = (float)(acc > 1e-10 ? 1.0 / acc : 1.0);
    }
    rc = upload(h, h->invn20, invn.data(), invn.size() * sizeof(float));
    h->fast20_ok = true;
  }
  if (!rc && p->smooth_mask && 8 + 2 * p->n_grad_freq <= 18) {
    // counts of 8 adjacent bins f..f+7 from the 18-bit window b[f-nf .. f-nf+17]: two 9-bit tables
    const int nf = p->n_grad_freq;
    std::vector<unsigned long long> tab(1024, 0ull);
    for (int half = 0; half < 2; ++half)
      for (int v = 0; v < 512; ++v) {
        unsigned long long packed = 0;
        for (int e = 0; e < 8; ++e) {
          int cnt = 0;
          for (int b = 0; b < 9; ++b)
            if ((v >> b) & 1) {
              int a = (half * 9 + b) - nf - e;  // tap offset of window bit relative to bin f+e
              if (a >= -nf && a <= nf) cnt += nf + 1 - (a < 0 ? -a : a);
            }
          packed |= (unsigned long long)cnt << (8 * e);
        }
        tab[half * 512 + v] = packed;
      }
    rc = upload(h, h->ftab, tab.data(), tab.size() * sizeof(unsigned long long));
  }
  if (!rc && h->fast_ok && p->smooth_mask && p->n_grad_freq <= 8 && p->n_grad_time <= fast::OP_MAX_NT) {
    // per-lane operands of the one-pass kernel's MFMA smoothing (onepass.hpp), v_mfma_i32_16x16x32_i8 layout:
    // lane l = (q = l / 16, j = l % 16) supplies bytes e = 0..7 = k slots 8 q + e of row/column j
    const int nf = p->n_grad_freq, nt = p->n_grad_time;
    std::vector<unsigned long long> mc(192, 0ull), ex(256, 0ull);
    auto wt = [&](int r, int i) {  // time weight of tile row r (frame r - nt) for output frame i
      const int d = r - i - nt, ad = d < 0 ? -d : d;
      return ad <= nt ? nt + 1 - ad : 0;
    };
    auto trow = [&](int m) { return m < nt ? m : 16 + m; };  // neighbour-list index -> tile row
    for (int l = 0; l < 64; ++l) {
      const int q = l / 16, j = l % 16;
      for (int e = 0; e < 8; ++e) {
        const int a = 8 * q + e - 8 - j, aa = a < 0 ? -a : a;
        mc[l] |= (unsigned long long)(aa <= nf ? nf + 1 - aa : 0) << (8 * e);
        int w1;
        if (e < 4) w1 = wt(nt + 4 * q + e, j);
        else { const int m = 4 * q + e - 4; w1 = m < 2 * nt ? wt(trow(m), j) : 0; }
        mc[64 + l] |= (unsigned long long)w1 << (8 * e);
        if (e < 4) { const int m = 16 + 4 * q + e; mc[128 + l] |= (unsigned long long)(m < 2 * nt ? wt(trow(m), j) : 0) << (8 * e); }
      }
    }
    for (int v = 0; v < 256; ++v)
      for (int e = 0; e < 8; ++e) ex[v] |= (unsigned long long)((v >> e) & 1) << (8 * e);
    rc = upload(h, h->ftab3, mc.data(), mc.size() * 8);
    if (!rc) rc = upload(h, h->xexp, ex.data(), ex.size() * 8);
    // the one-pass kernel reads all of its constant tables through ONE base pointer (onepass.hpp: OP_TAB_*)
    if (!rc) rc = ensure(h, h->optab, fast::OP_TAB_BYTES);
    if (!rc) {
      struct { const DevBuf* b; int off; size_t bytes; } parts[] = {
          {&h->wa32, fast::OP_TAB_WIN, 4096}, {&h->wsq32, fast::OP_TAB_WSQ, 4096}, {&h->invn, fast::OP_TAB_INVN, 1024},
          {&h->tw512, fast::OP_TAB_TW512, 4096}, {&h->tw32, fast::OP_TAB_TW1024, 4096}, {&h->wfull64, fast::OP_TAB_WIN64, 8192},
          {&h->tw64, fast::OP_TAB_TW64, 8192}, {&h->ftab3, fast::OP_TAB_MCONST, 1536}, {&h->xexp, fast::OP_TAB_EXP8, 2048}};
      for (const auto& pt : parts) {
        if (!pt.b->p || pt.b->bytes < pt.bytes) { h->err = "one-pass table arena: a source table is missing"; rc = SG_E_STATE; break; }
        if (hipMemcpy((char*)h->optab.p + pt.off, pt.b->p, pt.bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
          h->err = "one-pass table arena: hipMemcpy failed"; rc = SG_E_HIP; break;
        }
      }
    }
  }
  if (!rc && h->fast5_ok && p->smooth_mask && p->n_grad_freq <= fast::O5_MAX_NF && p->n_grad_time <= fast::O5_MAX_NT) {
    // operands of k_gate_onepass512's MFMA smoothing (onepass512.hpp), v_mfma_i32_16x16x32_i8 layout: lane l = (q = l / 16,
    // j = l % 16) supplies bytes e = 0..7 = k slots 8 q + e of row / column j
    const int nf = p->n_grad_freq, nt = p->n_grad_time;
    std::vector<unsigned long long> tb(448, 0ull);
    auto wt = [&](int d) { const int ad = d < 0 ? -d : d; return ad <= nt ? nt + 1 - ad : 0; };
    for (int l = 0; l < 64; ++l) {
      const int q = l / 16, j = l % 16;
      for (int e = 0; e < 8; ++e) {
        const int a = 8 * q + e - 8 - j, aa = a < 0 ? -a : a;
        tb[l] |= (unsigned long long)(aa <= nf ? nf + 1 - aa : 0) << (8 * e);
        // k slot 8 q + e: row 4 q + e of the first block (e < 4) / 16 + 4 q + (e - 4) of the second; output frame j at row nt + j
        const int r1 = e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4);
        tb[64 + l] |= (unsigned long long)wt(r1 - nt - j) << (8 * e);
        tb[128 + l] |= (unsigned long long)wt(32 + r1 - nt - j) << (8 * e);
      }
    }
    for (int v = 0; v < 256; ++v)
      for (int e = 0; e < 8; ++e) tb[192 + v] |= (unsigned long long)((v >> e) & 1) << (8 * e);
    rc = upload(h, h->o5tab, tb.data(), tb.size() * 8);
  }
  if (!rc && h->fast25_ok && p->smooth_mask && p->n_grad_freq <= fast::O25_MAX_NF && p->n_grad_time <= fast::O25_MAX_NT) {
    // k_gate_onepass256 (onepass256.hpp): the same operands with THREE k-blocks of time weights
    const int nf = p->n_grad_freq, nt = p->n_grad_time;
    std::vector<unsigned long long> tb(512, 0ull);
    auto wt = [&](int d) { const int ad = d < 0 ? -d : d; return ad <= nt ? nt + 1 - ad : 0; };
    for (int l = 0; l < 64; ++l) {
      const int q = l / 16, j = l % 16;
      for (int e = 0; e < 8; ++e) {
        const int a = 8 * q + e - 8 - j, aa = a < 0 ? -a : a;
        tb[l] |= (unsigned long long)(aa <= nf ? nf + 1 - aa : 0) << (8 * e);
        const int r1 = e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4);
        tb[64 + l] |= (unsigned long long)wt(r1 - nt - j) << (8 * e);
        tb[128 + l] |= (unsigned long long)wt(32 + r1 - nt - j) << (8 * e);
        tb[192 + l] |= (unsigned long long)wt(64 + r1 - nt - j) << (8 * e);
      }
    }
    for (int v = 0; v < 256; ++v)
      for (int e = 0; e < 8; ++e) tb[256 + v] |= (unsigned long long)((v >> e) & 1) << (8 * e);
    rc = upload(h, h->o25tab, tb.data(), tb.size() * 8);
  }
  if (!rc && h->fast20_ok && p->smooth_mask && p->n_grad_freq <= fast::O20_MAX_NF && p->n_grad_time <= fast::O20_MAX_NT) {
    // k_gate_onepass2048 (onepass2048.hpp): TWO band matrices (bins 16 b - 24 + k and 16 b + 8 + k against output bin 16 b + j), one
    // k-block of time weights
    const int nf = p->n_grad_freq, nt = p->n_grad_time;
    std::vector<unsigned long long> tb(448, 0ull);
    auto wt = [&](int d) { const int ad = d < 0 ? -d : d; return ad <= nt ? nt + 1 - ad : 0; };
    auto wf = [&](int d) { const int ad = d < 0 ? -d : d; return ad <= nf ? nf + 1 - ad : 0; };
    for (int l = 0; l < 64; ++l) {
      const int q = l / 16, j = l % 16;
      for (int e = 0; e < 8; ++e) {
        const int k = 8 * q + e;
        tb[l] |= (unsigned long long)wf(k - 24 - j) << (8 * e);
        tb[64 + l] |= (unsigned long long)wf(k + 8 - j) << (8 * e);
        const int r1 = e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4);
        tb[128 + l] |= (unsigned long long)wt(r1 - nt - j) << (8 * e);
      }
    }
    for (int v = 0; v < 256; ++v)
      for (int e = 0; e < 8; ++e) tb[192 + v] |= (unsigned long long)((v >> e) & 1) << (8 * e);
    rc = upload(h, h->o20tab, tb.data(), tb.size() * 8);
  }
  if (!rc) rc = ensure(h, h->thresh, (size_t)h->FS * sizeof(double));
  if (rc) {
    g_create_error = h->err;
    sg_destroy(h);
    return rc;
  }
  *out = h;
  return SG_OK;
}

extern "C" int sg_destroy(sg_handle* h) {
  if (!h) return SG_OK;
  (void)hipDeviceSynchronize();
  for (auto& r : h->prof_live) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : h->prof_pool) (void)hipEventDestroy(e);
  if (h->err_host) (void)hipHostFree(h->err_host);
  for (DevBuf* b : {&h->tw64, &h->tw32, &h->wfull64, &h->wa32, &h->ws32, &h->wsq32, &h->kf, &h->kt, &h->thresh,
                    &h->P, &h->pmax, &h->thr_rows, &h->raw, &h->M, &h->seg, &h->yn, &h->bits, &h->K16, &h->umax,
                    &h->need, &h->T2, &h->part, &h->tw512, &h->invn, &h->seam, &h->ftab, &h->xbits, &h->xpart,
                    &h->xticket, &h->xtick2, &h->ftab3, &h->xexp, &h->optab, &h->nsp, &h->nsc, &h->xin, &h->czt_tw64, &h->czt_ch64,
                    &h->czt_bh64, &h->czt_tw32, &h->czt_ch32, &h->czt_bh32, &h->logtab, &h->big_twM, &h->big_tw2,
                    &h->big_ch, &h->big_bh, &h->big_W, &h->big_W2, &h->xP, &h->xraw, &h->xM, &h->xtmp, &h->xseg, &h->invn5, &h->invn25, &h->invn20, &h->rg_count, &h->alim, &h->nss, &h->mr_pt32, &h->mr_pt64, &h->o5tab, &h->o25tab, &h->o20tab})
    free_buf(*b);
  delete h;
  return SG_OK;
}

extern "C" int sg_n_frames(const sg_handle* h, int64_t L, int64_t* n_frames) {
  if (!h || !n_frames) return SG_E_INVALID;
  *n_frames = frames_for(h, L);
  return SG_OK;
}
extern "C" int sg_output_length(const sg_handle* h, int64_t L, int64_t* out_len) {
  if (!h || !out_len) return SG_E_INVALID;
  *out_len = outlen_for(h, L);
  return SG_OK;
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
static int64_t ws_budget(const sg_handle* h) {
  return h->p.max_workspace_bytes > 0 ? h->p.max_workspace_bytes : (int64_t)8 << 30;
}

// Workspace per unit.  lean: the fused stationary path with the fused apply kernel only keeps the
// bit mask (T*wpr*8 B) and the uint16 weight sums (T*FS*2 B).
static size_t unit_bytes(const sg_handle* h, const Geom& g, bool lean) {
  size_t cells = (size_t)g.T * g.FS;
  if (lean) return (size_t)g.T * ((g.F + 63) / 64) * 8 + cells * 2 + (size_t)g.FS * 16 + 64 +
           (size_t)(g.T / 16 + 2) * 6 * 256 * 4;
  return cells * (8 + 4 + 4 + 2) + (size_t)g.T * g.n * 4 + (size_t)g.FS * 16 +
         (size_t)(g.T / NS_TT + 1) * 2 * g.FS * 8 * 2 +  // + partials and carries of the two-pass non-stationary mask
         (size_t)(g.T / 8 + 1) * 2 * g.FS * 8;           // + the magnitude kernels' per-tile partials (tiles of 8 frames at n_fft = 2048)
}

static int64_t units_per_batch(const sg_handle* h, const Geom& g, int64_t total, bool lean = false) {
  int64_t ub = ws_budget(h) / (int64_t)unit_bytes(h, g, lean);
  ub = std::max<int64_t>(1, std::min<int64_t>(ub, 32768));
  return std::min(ub, total);
}

static int ensure_ws(sg_handle* h, const Geom& g, int64_t ub, bool lean = false) {
  size_t cells = (size_t)ub * g.T * g.FS;
  int rc;
  if (!lean) {
    if ((rc = ensure(h, h->P, cells * 8))) return rc;  // power (f64) or magnitude (f32)
    if ((rc = ensure(h, h->raw, cells * 4))) return rc;
    if ((rc = ensure(h, h->M, cells * 4))) return rc;
    if ((rc = ensure(h, h->seg, std::max((size_t)ub * g.T * g.n * 4, cells * 4)))) return rc;
  }
  if ((rc = ensure(h, h->pmax, (size_t)ub * g.FS * 8))) return rc;
  if ((rc = ensure(h, h->thr_rows, (size_t)ub * g.FS * 8))) return rc;
  return SG_OK;
}

// ------------------------------------------------------------------------------------------
// pipeline pieces (all enqueue on `st`)
// ------------------------------------------------------------------------------------------
// time slices for the column statistics: enough blocks to fill 256 CUs even for one unit
#ifndef SG_STAT_SLICES_MAX
#define SG_STAT_SLICES_MAX 64
#endif
#ifndef SG_STAT_FRAMES_PER_SLICE
#define SG_STAT_FRAMES_PER_SLICE 36   // k_colstats1 (stage_stats)
#endif
static int stat_slices(const Geom& g, int64_t ub) {
  int64_t blocks = (int64_t)((g.F + 63) / 64) * ub;
  int64_t nts = (2048 + blocks - 1) / blocks;
  nts = std::max<int64_t>(1, std::min<int64_t>(nts, std::min<int64_t>(SG_STAT_SLICES_MAX, std::max<int64_t>(1, g.T / 16))));
  return (int)nts;
}

// power field + column max of a batch of units
static int stage_power(sg_handle* h, const View& v, const Geom& g, int64_t ub, hipStream_t st) {
  if (ub >= 16) {
    // power field; the per-(unit, band) maximum is folded into the STFT kernel (atomic max: one
    // address per (unit, band), little contention when there are many units)
    HIPCHK(h, hipMemsetAsync(h->pmax.p, 0, (size_t)ub * g.FS * 8, st));
    ProfScope ps(h, SG_STAGE_STFT_POWER, st);
    HIPCHK(h, stft_any<double>(h, v, g, ub, (double*)h->P.p, nullptr, nullptr, 1.0, st,
                               (unsigned long long*)h->pmax.p));
    return SG_OK;
  }
  // few units (the noise clip): thousands of frames would hammer the same 513 addresses; reduce
  // the maximum with the two-stage column kernels instead
  {
    ProfScope ps(h, SG_STAGE_STFT_POWER, st);
    HIPCHK(h, stft_any<double>(h, v, g, ub, (double*)h->P.p, nullptr, nullptr, 1.0, st));
  }
  ProfScope ps(h, SG_STAGE_COLMAX, st);
  const int nts = stat_slices(g, ub);
  int rc = ensure(h, h->part, (size_t)ub * nts * 2 * g.FS * 8);
  if (rc) return rc;
  dim3 grid((g.F + 63) / 64, (unsigned)ub, nts);
  hipLaunchKernelGGL(k_colmax, grid, dim3(64 * STAT_TG), 0, st, (const double*)h->P.p, g, (double*)h->part.p);
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_colmax_final, dim3(grid_1d(ub * g.FS, 256)), dim3(256), 0, st, (const double*)h->part.p, g,
                     nts, (double*)h->pmax.p, ub);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

static int stage_colstats(sg_handle* h, const Geom& g, int64_t ub, double* thresh_out, hipStream_t st) {
  ProfScope ps(h, SG_STAGE_COLSTATS, st);
  const int nts = stat_slices(g, ub);
  int rc = ensure(h, h->part, (size_t)ub * nts * 2 * g.FS * 8);
  if (rc) return rc;
  dim3 grid((g.F + 63) / 64, (unsigned)ub, nts);
  hipLaunchKernelGGL(k_colstats, grid, dim3(64 * STAT_TG), 0, st, (const double*)h->P.p, g,
                     (const double*)h->pmax.p, h->mag_scale, h->p.top_db, (double*)h->part.p);
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_colstats_final, dim3(grid_1d(ub * g.FS, 256)), dim3(256), 0, st,
                     (const double*)h->part.p, g, nts, (const double*)h->pmax.p, h->mag_scale,
                     h->p.n_std_thresh, h->p.ddof, thresh_out, ub);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

// power field + band statistics -> threshold
static int stage_stats(sg_handle* h, const View& v, const Geom& g, int64_t ub, double* thresh_out, hipStream_t st,
                       bool gate_consts = false) {
  if (ub < 16) {
    // few units (the noise clip): STFT -> one pass over the power field -> final (3 launches instead of 5)
    {
      ProfScope ps(h, SG_STAGE_STFT_POWER, st);
      HIPCHK(h, stft_any<double>(h, v, g, ub, (double*)h->P.p, nullptr, nullptr, 1.0, st));
    }
    ProfScope ps(h, SG_STAGE_COLSTATS, st);
    // (round 6) at least SG_STAT_FRAMES_PER_SLICE frames per slice, at most 64 slices.  Measured (k_colstats1 + k_colstats1_final, us):
    // n_fft = 1024 (T = 2345): 64 slices 8.0 + 8.3, 32 slices 9.9 + 8.8; n_fft = 2048 (T = 1172): 32 slices 8.1 + 7.3, 16 slices 10.5 + 7.3,
    // 64 slices 9.6 + 7.3.  (Before k_colstats1_final loaded its partials unconditionally it paid 0.08 us per slice: 13.2 us at 64.)
    const int nts = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(stat_slices(g, ub), STAT_TG * STAT1_MAXS), g.T / SG_STAT_FRAMES_PER_SLICE));
    int rc = ensure(h, h->part, (size_t)ub * nts * STAT1_NP * g.FS * 8);
    if (rc) return rc;
    dim3 grid((g.F + 63) / 64, (unsigned)ub, nts);
    hipLaunchKernelGGL(k_colstats1, grid, dim3(64 * STAT_TG), 0, st, (const double*)h->P.p, g, h->mag_scale,
                       (double*)h->part.p, db_fast_consts(h));
    HIPCHK(h, hipGetLastError());
    // (for the stationary gate's noise statistics the final kernel also derives the gate's compare constants: no
    // k_prep_thresh_lazy launch in the calls that follow)
    // (round 6: the final stage INSIDE k_colstats1 -- run by the last workgroup of a band block to finish its slice -- measured:
    // with __threadfence() around the arrival counter + 43 us per call, a device-scope release writes back the L2; fence-free
    // (partials as sc1 stores, s_waitcnt vmcnt(0), counter, sc1 loads) + 2 / - 1 / + 0.5 / + 4 us at n_fft = 1024 / 256 / 512 /
    // 2048: what the launch boundary costs the last arriver pays in memory round trips.  DESIGN 8, profiles/r06_stats_fused_ab.txt)
    GateConsts gc{};
    if (gate_consts && ub == 1 && grid.x <= 62) {   // (one bound per 64-band block: 9 at n_fft = 1024, 33 at 4096)
      if ((rc = ensure(h, h->T2, (size_t)g.FS * 8))) return rc;
      if ((rc = ensure_zeroed(h, h->alim, 256, st))) return rc;
      gc.T2 = (double*)h->T2.p; gc.alim_b = (unsigned*)h->alim.p + 2; gc.sum_abs_w = h->sum_abs_w;
    }
    hipLaunchKernelGGL(k_colstats1_final, dim3((unsigned)((g.FS + 63) / 64), (unsigned)ub), dim3(64 * STAT_TG), 0, st,
                       (const double*)h->part.p, (const double*)h->P.p, g, nts, h->mag_scale, h->p.top_db,
                       h->p.n_std_thresh, h->p.ddof, (double*)h->pmax.p, thresh_out, gc);
    HIPCHK(h, hipGetLastError());
    if (gc.T2 != nullptr) h->t2_ready = true;
    return SG_OK;
  }
  int rc = stage_power(h, v, g, ub, st);
  if (rc) return rc;
  return stage_colstats(h, g, ub, thresh_out, st);
}

static int stage_decide(sg_handle* h, const Geom& g, int64_t ub, const double* thresh, int64_t ustride,
                        hipStream_t st) {
  ProfScope ps(h, SG_STAGE_DECIDE, st);
  int64_t cells = ub * g.T * g.FS;
  hipLaunchKernelGGL(k_decide, dim3(grid_1d(cells, 256)), dim3(256), 0, st, (const double*)h->P.p, g,
                     (const double*)h->pmax.p, thresh, ustride, h->mag_scale, h->p.top_db, (float*)h->raw.p, ub);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

// ------------------------------------------------------------------------------------------
// n_fft = 512 / hop 128 on the register transform (fast512.hpp)
// ------------------------------------------------------------------------------------------
static fast::Fast5Args fast5_args(const sg_handle* h, const View& v, const Geom& g) {
  fast::Fast5Args A{};
  A.view = v; A.g = g;
  A.win = (const float*)h->wa32.p;
  A.win64 = (const double*)h->wfull64.p;
  A.tw512 = (const fast::cf*)h->tw512.p;
  A.tw64 = (const cx<double>*)h->tw64.p;
  A.mag_scale = h->mag_scale; A.top_db = h->p.top_db;
  A.wsq = (const float*)h->wsq32.p;
  A.invn = (const float*)h->invn5.p;
  return A;
}
constexpr size_t FAST5_LDS = (size_t)(fast::FN + 4 * fast::WAVE_CX_H) * sizeof(fast::cf) + (512 + 264) * sizeof(float);

static int stage_decide512(sg_handle* h, const View& v, const Geom& g, int64_t ub, const ThreshConsts& tc,
                           unsigned long long* bits, hipStream_t st, const FloorLazy& fl = FloorLazy{}, bool redo = false) {
  ProfScope ps(h, redo ? SG_STAGE_STFT_MAX : SG_STAGE_DECIDE_FAST, st);   // (the early-exit second launch is booked with the pre-pass)
  fast::Fast5Args A = fast5_args(h, v, g);
  A.tc = tc;
  A.bits = bits;
  A.fl = fl;
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), FAST5_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((g.T + 31) / 32), (unsigned)ub), dim3(256), FAST5_LDS, st, A);
    return hipGetLastError();
  };
  if (redo) HIPCHK(h, go(fast::k_decide_fast512<4, true>));
  else HIPCHK(h, go(fast::k_decide_fast512<4, false>));
  return SG_OK;
}

static int stage_mag512(sg_handle* h, const View& v, const Geom& g, int64_t ub, float* mag, hipStream_t st, double iir_b = 0.0,
                        double* sub = nullptr /* per-tile recurrence partials (mag_sub_partials) */) {
  ProfScope ps(h, SG_STAGE_STFT_MAG, st);
  fast::Fast5Args A = fast5_args(h, v, g);
  A.mag = mag;
  A.iir_b = iir_b;
  A.sub = sub;
  auto kern = fast::k_mag_fast512<4>;
  HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST5_LDS));
  hipLaunchKernelGGL(kern, dim3((unsigned)((g.T + 31) / 32), (unsigned)ub), dim3(256), FAST5_LDS, st, A);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

// seam hops of abutting apply / one-pass tiles at n_fft = 512 / 256 (fastpath.hpp: k_ola_seam); ARGS = Fast5Args / Fast25Args
template <int HOP, int NF, typename ARGS>
static int launch_seam_small(sg_handle* h, const ARGS& A, int64_t ub, hipStream_t st) {
  if (A.n_tiles < 2) return SG_OK;
  fast::SeamArgs S{};
  S.view = A.view; S.g = A.g; S.om = A.om; S.h_begin = A.h_begin; S.h_end = A.h_end; S.normalize = A.normalize;
  S.invn = A.invn; S.wsq = A.wsq; S.part = A.part; S.n_tiles = A.n_tiles;
  hipLaunchKernelGGL((fast::k_ola_seam<HOP, NF>), dim3((unsigned)(A.n_tiles - 1), (unsigned)ub), dim3(HOP), 0, st, S);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

static int stage_apply512(sg_handle* h, const View& v, const Geom& g, int64_t ub, const OutMap& om,
                          const float* mask_f /* nullptr: uint16 weight sums in h->K16 (natural bin order) */,
                          int normalize, hipStream_t st) {
  ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
  fast::Fast5Args A = fast5_args(h, v, g);
  A.Mf = mask_f;
  A.K = (const unsigned short*)h->K16.p;
  A.inv_ktot = (float)(1.0 / (double)h->ktot);
  A.om = om;
  A.normalize = normalize;
  A.h_begin = (om.p0 + g.padL) / 128;
  A.h_end = (om.p1 - 1 + g.padL) / 128 + 1;
  const int64_t nh = A.h_end - A.h_begin;
  if (nh <= 0) return SG_OK;
  // (round 6) abutting tiles of 32 frames; the 3 hops that straddle two tiles as partial sums + k_ola_seam (fastpath.hpp)
  const int64_t tiles = (nh + 3 + 31) / 32;
  const bool seam = tiles >= 2 && !h->force_noseam;
  A.part = nullptr;
  A.n_tiles = (int)tiles;
  if (seam) {
    int rc = ensure(h, h->seam, (size_t)ub * tiles * 6 * 128 * sizeof(float));
    if (rc) return rc;
    A.part = (float*)h->seam.p;
  }
  const dim3 grid((unsigned)(seam ? tiles : (nh + 28) / 29), (unsigned)ub);
  if (mask_f) {
    auto kern = fast::k_apply_fast512<4, false>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST5_LDS));
    hipLaunchKernelGGL(kern, grid, dim3(256), FAST5_LDS, st, A);
  } else {
    auto kern = fast::k_apply_fast512<4, true>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST5_LDS));
    hipLaunchKernelGGL(kern, grid, dim3(256), FAST5_LDS, st, A);
  }
  HIPCHK(h, hipGetLastError());
  if (seam) return launch_seam_small<128, 32>(h, A, ub, st);
  return SG_OK;
}

// (round 6) one-pass gate for n_fft = 512 (onepass512.hpp): decide + smooth + apply in one kernel, tiles exchange their bits
static int handoff_prepare(sg_handle* h, hipStream_t st);
static int handoff_next_epoch(sg_handle* h, hipStream_t st);
static int stage_prep_floor(sg_handle* h, const View& v, const Geom& g, int64_t ub, ThreshConsts* tc_out, hipStream_t st,
                            const View* v_exact, unsigned* live_host, unsigned stamp);
template <int MODE>
static hipError_t launch_bits_any(const sg_handle* h, const View& v, const Geom& g, int64_t units, const ThreshConsts& tc,
                                  unsigned long long* pmax_bits, unsigned long long* bits, int wpr, hipStream_t st);
// geometry constants of a small one-pass gate (onepass512.hpp / onepass256.hpp)
struct OnePassSmall {
  int hop, NF, NH, tile_words, xw, max_nf, max_nt, F;
  size_t lds;
  int extra = 0;   // abutting tiles (NH == NF, n_fft = 2048): the last tile must reach 3 hops past the range (its leading partials)
};
static bool onepass_small_ok(const sg_handle* h, const Geom& g, const OutMap& om, const OnePassSmall& S, const DevBuf& tab) {
  if (h->force_nofast || h->force_split || h->force_f64_decide || h->force_unfused || !h->fused_ok) return false;
  if (h->p.variant != SG_VARIANT_S || !h->p.stationary || !h->p.smooth_mask) return false;
  if (h->p.n_grad_freq > S.max_nf || h->p.n_grad_time > S.max_nt || g.F != S.F || !tab.p) return false;
  if (h->tile_order == 1) return false;
  const int64_t hb = (om.p0 + g.padL) / S.hop, he = (om.p1 - 1 + g.padL) / S.hop + 1;
  return he > hb;
}
// PARGS: the kernel's argument struct (its member A already filled with the geometry's tables); launch(P, redo) enqueues it
template <typename PARGS, typename LAUNCH>
static int stage_onepass_small(sg_handle* h, const View& v, const View& vx, const Geom& g, int64_t ub, const OutMap& om,
                               hipStream_t st, const OnePassSmall& S, PARGS P, const DevBuf& tab, LAUNCH launch) {
  int rc;
  if ((rc = handoff_prepare(h, st))) return rc;
  const unsigned live_stamp = h->err_host[1];
  const unsigned need_tag = h->epoch & 0x3fffffffu;
  const unsigned era = h->epoch >> 30;
  if (era != h->need_era || need_tag == 0u) {
    if (h->need.p) HIPCHK(h, hipMemsetAsync(h->need.p, 0, h->need.bytes, st));
    if (h->alim.p) HIPCHK(h, hipMemsetAsync((char*)h->alim.p + 4, 0, 4, st));
    h->need_era = era;
  }
  const bool lazy = need_tag != 0u &&
                    (h->floor_test == 2 || (h->floor_test == 0 && !(live_stamp != 0u && h->epoch - live_stamp <= 16u)));
  ++(lazy ? h->n_floor_lazy : h->n_floor_apriori);
  const int wpr = (g.F + 63) / 64;
  ThreshConsts tc{};
  FloorLazy fl{};
  if (!lazy) {
    if ((rc = stage_prep_floor(h, v, g, ub, &tc, st, &vx, h->err_dev + 1, h->epoch))) return rc;
  } else {
    const int nb = (g.FS + 63) / 64;
    if ((rc = ensure(h, h->bits, (size_t)ub * g.T * wpr * 8))) return rc;
    if ((rc = ensure_zeroed(h, h->need, (size_t)ub * 4, st))) return rc;
    if ((rc = ensure(h, h->T2, (size_t)g.FS * 8))) return rc;
    if ((rc = ensure_zeroed(h, h->alim, 256, st))) return rc;
    if (!h->t2_ready) {
      ProfScope ps(h, SG_STAGE_PREP, st);
      hipLaunchKernelGGL(k_prep_thresh_lazy, dim3(1), dim3(256), 0, st, (const double*)h->thresh.p, g.F, h->mag_scale,
                         h->sum_abs_w, h->p.top_db, (double*)h->T2.p, (unsigned*)h->alim.p, nb);
      HIPCHK(h, hipGetLastError());
      h->t2_ready = true;
    }
    tc = ThreshConsts{(const double*)h->T2.p, (const double*)h->thresh.p, (const double*)h->pmax.p, (const int*)h->need.p, need_tag};
    fl = FloorLazy{(unsigned*)h->alim.p, nb, h->err_dev + 1, h->epoch};
  }
  P.A.tc = tc;
  P.A.fl = fl;
  P.A.inv_ktot = (float)(1.0 / (double)h->ktot);
  P.A.om = om;
  P.A.normalize = 1;
  P.A.h_begin = (om.p0 + g.padL) / S.hop;
  P.A.h_end = (om.p1 - 1 + g.padL) / S.hop + 1;
  const int64_t nh = P.A.h_end - P.A.h_begin;
  const int64_t n_tiles = (nh + S.extra + S.NH - 1) / S.NH, ntt = n_tiles + 2;
  if ((rc = ensure_zeroed(h, h->xbits, (size_t)ub * ntt * S.tile_words * 8, st))) return rc;
  P.xbits = (unsigned long long*)h->xbits.p;
  P.ticket = (unsigned*)h->xticket.p;
  P.ticket_base = h->ticket_base;
  h->ticket_base += (unsigned)(ub * ntt);
  P.epoch = h->epoch;
  const bool lose = (h->lose_now & 3u) != 0;   // test hook: every poll of this launch gives up at once (the timeout path itself)
  P.poll_epoch = lose ? ~h->epoch : h->epoch;
  P.spin_max = lose ? 0 : fast::OP_SPIN_MAX;
  P.err = h->err_dev;
  P.nf = h->p.n_grad_freq; P.nt = h->p.n_grad_time; P.n_tiles = (int)n_tiles;
  P.tab = (const unsigned long long*)tab.p;
  P.prop = (float)h->p.prop_decrease;
  {
    // the part of a unit's window outside its tiles' spans, dealt evenly to the unit's tiles ("floor test" in the kernels)
    const int64_t SPAN = (int64_t)(S.NF - 1 + 4) * S.hop;
    const int64_t sp0 = (P.A.h_begin - 3 - S.NH) * S.hop - g.padL, sp1 = (P.A.h_begin - 3 + n_tiles * S.NH) * S.hop - g.padL + SPAN;
    const int64_t inside = std::max<int64_t>(0, std::min<int64_t>(v.Lp, sp1) - std::max<int64_t>(0, sp0));
    const int64_t q = (v.Lp - inside + ntt - 1) / ntt;
    if (q > 0x7fffffff) FAIL(h, SG_E_UNSUPPORTED, "one-pass gate: window of %lld samples", (long long)v.Lp);
    P.scan_q = (int)q;
  }
  const dim3 grid((unsigned)(ub * ntt));
  {
    ProfScope ps(h, SG_STAGE_ONEPASS, st);
    HIPCHK(h, launch(P, false, grid));
  }
  if (lazy) {
    // the units whose floor test fired: float64 band maxima, then the gate again with them (both return at once otherwise)
    ProfScope ps(h, SG_STAGE_STFT_MAX, st);
    HIPCHK(h, launch_bits_any<0>(h, vx, g, ub, tc, (unsigned long long*)h->pmax.p, (unsigned long long*)h->bits.p, wpr, st));
    if ((rc = handoff_next_epoch(h, st))) return rc;
    P.epoch = h->epoch;
    P.poll_epoch = h->epoch;
    P.spin_max = fast::OP_SPIN_MAX;
    P.ticket = (unsigned*)h->xticket.p + 8;   // its own counter (zeroed by the first launch's ticket-0 workgroup)
    P.ticket_base = 0;
    HIPCHK(h, launch(P, true, grid));
  }
  h->dbg_xbits = true;
  h->dbg_trows = S.NF; h->dbg_tstep = S.NH; h->dbg_twords = S.tile_words; h->dbg_txw = S.xw;
  h->dbg_tf0 = P.A.h_begin - 3;
  h->dbg_ntt = (int)ntt;
  h->dbg_db = std::max<int64_t>(0, P.A.h_begin - 3 - S.NH);
  h->dbg_de = std::min<int64_t>(g.T, P.A.h_begin - 3 + (int64_t)S.NH * (n_tiles + 1) + (S.NF - S.NH));
  return SG_OK;
}

static const OnePassSmall O5_GEOM{128, fast::O5_NF, fast::O5_NH, fast::O5_TILE_WORDS, fast::O5_XW, fast::O5_MAX_NF, fast::O5_MAX_NT, 257,
                                  FAST5_LDS + 16 + 2048};
// (abutting tiles + k_ola_seam: the default; SG_OPT_FORCE_NOSEAM keeps the overlapping tiles above)
static const OnePassSmall O5_GEOM_SEAM{128, fast::O5_NF, fast::O5_NF, fast::O5_TILE_WORDS, fast::O5_XW, fast::O5_MAX_NF, fast::O5_MAX_NT, 257,
                                       FAST5_LDS + 16 + 2048, 3};
static bool onepass512_ok(const sg_handle* h, const Geom& g, const OutMap& om) {
  return h->fast5_ok && onepass_small_ok(h, g, om, O5_GEOM, h->o5tab);
}
static int stage_onepass512(sg_handle* h, const View& v, const View& vx, const Geom& g, int64_t ub, const OutMap& om,
                            hipStream_t st) {
  fast::OnePass5Args P{};
  P.A = fast5_args(h, v, g);
  const bool seam = !h->force_noseam;
  if (seam) {
    const int64_t hb = (om.p0 + g.padL) / 128, he = (om.p1 - 1 + g.padL) / 128 + 1;
    const int64_t tiles = (he - hb + 3 + 31) / 32;
    int rc = ensure(h, h->seam, (size_t)ub * tiles * 6 * 128 * sizeof(float));
    if (rc) return rc;
    P.A.part = (float*)h->seam.p;
    P.A.n_tiles = (int)tiles;
  }
  fast::Fast5Args last{};
  auto launch = [&](const fast::OnePass5Args& Q, bool redo, dim3 grid) -> hipError_t {
    last = Q.A;
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), O5_GEOM.lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, dim3(256), O5_GEOM.lds, st, Q);
      return hipGetLastError();
    };
    return redo ? go(fast::k_gate_onepass512<4, true>) : go(fast::k_gate_onepass512<4, false>);
  };
  int rc = stage_onepass_small(h, v, vx, g, ub, om, st, seam ? O5_GEOM_SEAM : O5_GEOM, P, h->o5tab, launch);
  if (rc || !seam) return rc;
  ProfScope ps(h, SG_STAGE_APPLY_FAST, st);   // (after the second launch, if any: a redone unit rewrote its partials)
  return launch_seam_small<128, 32>(h, last, ub, st);
}

// ------------------------------------------------------------------------------------------
// n_fft = 256 / hop 64 on the register transform (fast256.hpp, round 5): four frames per lane group
// ------------------------------------------------------------------------------------------
static fast::Fast25Args fast25_args(const sg_handle* h, const View& v, const Geom& g) {
  fast::Fast25Args A{};
  A.view = v; A.g = g;
  A.win = (const float*)h->wa32.p;
  A.win64 = (const double*)h->wfull64.p;
  A.tw512 = (const fast::cf*)h->tw512.p;
  A.tw64 = (const cx<double>*)h->tw64.p;
  A.mag_scale = h->mag_scale; A.top_db = h->p.top_db;
  A.wsq = (const float*)h->wsq32.p;
  A.invn = (const float*)h->invn25.p;
  return A;
}
constexpr size_t FAST25_LDS = (size_t)(fast::FN + 4 * fast::WAVE_CX_H) * sizeof(fast::cf) + (256 + fast::F25_T2) * sizeof(float);

static int stage_decide256(sg_handle* h, const View& v, const Geom& g, int64_t ub, const ThreshConsts& tc,
                           unsigned long long* bits, hipStream_t st, const FloorLazy& fl = FloorLazy{}, bool redo = false) {
  ProfScope ps(h, redo ? SG_STAGE_STFT_MAX : SG_STAGE_DECIDE_FAST, st);
  fast::Fast25Args A = fast25_args(h, v, g);
  A.tc = tc;
  A.bits = bits;
  A.fl = fl;
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), FAST25_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((g.T + 63) / 64), (unsigned)ub), dim3(256), FAST25_LDS, st, A);
    return hipGetLastError();
  };
  if (redo) HIPCHK(h, go(fast::k_decide_fast256<4, true>));
  else HIPCHK(h, go(fast::k_decide_fast256<4, false>));
  return SG_OK;
}

static int stage_mag256(sg_handle* h, const View& v, const Geom& g, int64_t ub, float* mag, hipStream_t st, double iir_b = 0.0,
                        double* sub = nullptr /* per-tile recurrence partials (mag_sub_partials) */) {
  ProfScope ps(h, SG_STAGE_STFT_MAG, st);
  fast::Fast25Args A = fast25_args(h, v, g);
  A.mag = mag;
  A.iir_b = iir_b;
  A.sub = sub;
  auto kern = fast::k_mag_fast256<4>;
  HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST25_LDS));
  hipLaunchKernelGGL(kern, dim3((unsigned)((g.T + 63) / 64), (unsigned)ub), dim3(256), FAST25_LDS, st, A);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

static int stage_apply256(sg_handle* h, const View& v, const Geom& g, int64_t ub, const OutMap& om,
                          const float* mask_f /* nullptr: uint16 weight sums in h->K16 (natural bin order) */,
                          int normalize, hipStream_t st) {
  ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
  fast::Fast25Args A = fast25_args(h, v, g);
  A.Mf = mask_f;
  A.K = (const unsigned short*)h->K16.p;
  A.inv_ktot = (float)(1.0 / (double)h->ktot);
  A.om = om;
  A.normalize = normalize;
  A.h_begin = (om.p0 + g.padL) / 64;
  A.h_end = (om.p1 - 1 + g.padL) / 64 + 1;
  const int64_t nh = A.h_end - A.h_begin;
  if (nh <= 0) return SG_OK;
  // (round 6) abutting tiles of 64 frames + k_ola_seam (see stage_apply512)
  const int64_t tiles = (nh + 3 + 63) / 64;
  const bool seam = tiles >= 2 && !h->force_noseam;
  A.part = nullptr;
  A.n_tiles = (int)tiles;
  if (seam) {
    int rc = ensure(h, h->seam, (size_t)ub * tiles * 6 * 64 * sizeof(float));
    if (rc) return rc;
    A.part = (float*)h->seam.p;
  }
  const dim3 grid((unsigned)(seam ? tiles : (nh + 60) / 61), (unsigned)ub);
  if (mask_f) {
    auto kern = fast::k_apply_fast256<4, false>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST25_LDS));
    hipLaunchKernelGGL(kern, grid, dim3(256), FAST25_LDS, st, A);
  } else {
    auto kern = fast::k_apply_fast256<4, true>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST25_LDS));
    hipLaunchKernelGGL(kern, grid, dim3(256), FAST25_LDS, st, A);
  }
  HIPCHK(h, hipGetLastError());
  if (seam) return launch_seam_small<64, 64>(h, A, ub, st);
  return SG_OK;
}

// (round 6) one-pass gate for n_fft = 256 (onepass256.hpp)
static const OnePassSmall O25_GEOM{64, fast::O25_NF, fast::O25_NH, fast::O25_TILE_WORDS, fast::O25_XW, fast::O25_MAX_NF, fast::O25_MAX_NT,
                                   129, FAST25_LDS + 16 + 2048};
static const OnePassSmall O25_GEOM_SEAM{64, fast::O25_NF, fast::O25_NF, fast::O25_TILE_WORDS, fast::O25_XW, fast::O25_MAX_NF, fast::O25_MAX_NT,
                                        129, FAST25_LDS + 16 + 2048, 3};
static bool onepass256_ok(const sg_handle* h, const Geom& g, const OutMap& om) {
  return h->fast25_ok && onepass_small_ok(h, g, om, O25_GEOM, h->o25tab);
}
static int stage_onepass256(sg_handle* h, const View& v, const View& vx, const Geom& g, int64_t ub, const OutMap& om,
                            hipStream_t st) {
  fast::OnePass25Args P{};
  P.A = fast25_args(h, v, g);
  const bool seam = !h->force_noseam;
  if (seam) {
    const int64_t hb = (om.p0 + g.padL) / 64, he = (om.p1 - 1 + g.padL) / 64 + 1;
    const int64_t tiles = (he - hb + 3 + 63) / 64;
    int rc = ensure(h, h->seam, (size_t)ub * tiles * 6 * 64 * sizeof(float));
    if (rc) return rc;
    P.A.part = (float*)h->seam.p;
    P.A.n_tiles = (int)tiles;
  }
  fast::Fast25Args last{};
  auto launch = [&](const fast::OnePass25Args& Q, bool redo, dim3 grid) -> hipError_t {
    last = Q.A;
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), O25_GEOM.lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, dim3(256), O25_GEOM.lds, st, Q);
      return hipGetLastError();
    };
    return redo ? go(fast::k_gate_onepass256<4, true>) : go(fast::k_gate_onepass256<4, false>);
  };
  int rc = stage_onepass_small(h, v, vx, g, ub, om, st, seam ? O25_GEOM_SEAM : O25_GEOM, P, h->o25tab, launch);
  if (rc || !seam) return rc;
  ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
  return launch_seam_small<64, 64>(h, last, ub, st);
}

// ------------------------------------------------------------------------------------------
// n_fft = 2048 / hop 512 on the register transform (fast2048.hpp)
// ------------------------------------------------------------------------------------------
static fast::Fast20Args fast20_args(const sg_handle* h, const View& v, const Geom& g) {
  fast::Fast20Args A{};
  A.view = v; A.g = g;
  A.win = (const float*)h->wa32.p;
  A.win64 = (const double*)h->wfull64.p;
  A.tw2048 = (const fast::cf*)h->tw32.p;
  A.tw64 = (const cx<double>*)h->tw64.p;
  A.mag_scale = h->mag_scale; A.top_db = h->p.top_db;
  A.wsq = (const float*)h->wsq32.p;
  A.invn = (const float*)h->invn20.p;
  return A;
}
constexpr size_t FAST20_LDS = (size_t)(1024 + 4 * fast::WAVE_CX_H) * sizeof(fast::cf) + 1028 * sizeof(float);

static int stage_decide2048(sg_handle* h, const View& v, const Geom& g, int64_t ub, const ThreshConsts& tc,
                            unsigned long long* bits, hipStream_t st, const FloorLazy& fl = FloorLazy{}, bool redo = false) {
  ProfScope ps(h, redo ? SG_STAGE_STFT_MAX : SG_STAGE_DECIDE_FAST, st);
  fast::Fast20Args A = fast20_args(h, v, g);
  A.tc = tc;
  A.bits = bits;
  A.fl = fl;
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), FAST20_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((g.T + 7) / 8), (unsigned)ub), dim3(256), FAST20_LDS, st, A);
    return hipGetLastError();
  };
  if (redo) HIPCHK(h, go(fast::k_decide_fast2048<4, true>));
  else HIPCHK(h, go(fast::k_decide_fast2048<4, false>));
  return SG_OK;
}

static int stage_mag2048(sg_handle* h, const View& v, const Geom& g, int64_t ub, float* mag, hipStream_t st, double iir_b = 0.0,
                        double* sub = nullptr /* per-tile recurrence partials (mag_sub_partials) */) {
  ProfScope ps(h, SG_STAGE_STFT_MAG, st);
  fast::Fast20Args A = fast20_args(h, v, g);
  A.mag = mag;
  A.iir_b = iir_b;
  A.sub = sub;
  auto kern = fast::k_mag_fast2048<4>;
  HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST20_LDS));
  hipLaunchKernelGGL(kern, dim3((unsigned)((g.T + 7) / 8), (unsigned)ub), dim3(256), FAST20_LDS, st, A);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

static int stage_apply2048(sg_handle* h, const View& v, const Geom& g, int64_t ub, const OutMap& om,
                           const float* mask_f /* nullptr: uint16 weight sums in h->K16 */, int normalize, hipStream_t st) {
  ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
  fast::Fast20Args A = fast20_args(h, v, g);
  A.Mf = mask_f;
  A.K = (const unsigned short*)h->K16.p;
  A.inv_ktot = (float)(1.0 / (double)h->ktot);
  A.om = om;
  A.normalize = normalize;
  A.h_begin = (om.p0 + g.padL) / 512;
  A.h_end = (om.p1 - 1 + g.padL) / 512 + 1;
  const int64_t nh = A.h_end - A.h_begin;
  if (nh <= 0) return SG_OK;
  // abutting tiles of 8 frames; the 3 hops that straddle two tiles as partial sums + k_ola_seam2048
  const int64_t tiles = (nh + 3 + 7) / 8;
  const bool seam = tiles >= 2 && !h->force_noseam;
  A.part = nullptr;
  A.n_tiles = (int)tiles;
  if (seam) {
    int rc = ensure(h, h->seam, (size_t)ub * tiles * 6 * 512 * sizeof(float));
    if (rc) return rc;
    A.part = (float*)h->seam.p;
  }
  const dim3 grid((unsigned)(seam ? tiles : (nh + 4) / 5), (unsigned)ub);
  if (mask_f) {
    auto kern = fast::k_apply_fast2048<4, false>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST20_LDS));
    hipLaunchKernelGGL(kern, grid, dim3(256), FAST20_LDS, st, A);
  } else {
    auto kern = fast::k_apply_fast2048<4, true>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), FAST20_LDS));
    hipLaunchKernelGGL(kern, grid, dim3(256), FAST20_LDS, st, A);
  }
  HIPCHK(h, hipGetLastError());
  if (seam) {
    hipLaunchKernelGGL(fast::k_ola_seam2048<8>, dim3((unsigned)(tiles - 1), (unsigned)ub), dim3(512), 0, st, A);
    HIPCHK(h, hipGetLastError());
  }
  return SG_OK;
}

// (round 6) one-pass gate for n_fft = 2048 (onepass2048.hpp): abutting tiles of 8 frames + k_ola_seam2048
static const OnePassSmall O20_GEOM{512, fast::O20_NF, fast::O20_NF, fast::O20_TILE_WORDS, fast::O20_XW, fast::O20_MAX_NF, fast::O20_MAX_NT,
                                   1025, FAST20_LDS + 16 + 2048, 3};
static bool onepass2048_ok(const sg_handle* h, const Geom& g, const OutMap& om) {
  return h->fast20_ok && !h->force_noseam && onepass_small_ok(h, g, om, O20_GEOM, h->o20tab);
}
static int stage_onepass2048(sg_handle* h, const View& v, const View& vx, const Geom& g, int64_t ub, const OutMap& om,
                             hipStream_t st) {
  fast::OnePass20Args P{};
  P.A = fast20_args(h, v, g);
  const int64_t hb = (om.p0 + g.padL) / 512, he = (om.p1 - 1 + g.padL) / 512 + 1;
  const int64_t tiles = (he - hb + 3 + 7) / 8;
  int rc = ensure(h, h->seam, (size_t)ub * tiles * 6 * 512 * sizeof(float));
  if (rc) return rc;
  P.A.part = (float*)h->seam.p;
  P.A.n_tiles = (int)tiles;
  fast::Fast20Args last{};
  auto launch = [&](const fast::OnePass20Args& Q, bool redo, dim3 grid) -> hipError_t {
    last = Q.A;
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), O20_GEOM.lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, dim3(256), O20_GEOM.lds, st, Q);
      return hipGetLastError();
    };
    return redo ? go(fast::k_gate_onepass2048<4, true>) : go(fast::k_gate_onepass2048<4, false>);
  };
  if ((rc = stage_onepass_small(h, v, vx, g, ub, om, st, O20_GEOM, P, h->o20tab, launch))) return rc;
  if (tiles >= 2) {   // the hops that straddle two tiles (after the second launch, if any: a redone unit rewrote its partials)
    ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
    hipLaunchKernelGGL(fast::k_ola_seam2048<8>, dim3((unsigned)(tiles - 1), (unsigned)ub), dim3(512), 0, st, last);
    HIPCHK(h, hipGetLastError());
  }
  return SG_OK;
}

static int stage_mag(sg_handle* h, const View& v, const Geom& g, int64_t ub, hipStream_t st, double iir_b = 0.0,
                     double* sub = nullptr /* register geometries: recurrence partials per magnitude tile (16 / 32 / 64 / 8 frames) */) {
  float* mag = (float*)h->P.p;
  if (h->fast5_ok && !h->force_nofast) return stage_mag512(h, v, g, ub, mag, st, iir_b, sub);
  if (h->fast25_ok && !h->force_nofast) return stage_mag256(h, v, g, ub, mag, st, iir_b, sub);
  if (h->fast20_ok && !h->force_nofast) return stage_mag2048(h, v, g, ub, mag, st, iir_b, sub);
  if (h->fast_ok && !h->force_nofast) {
    ProfScope ps(h, SG_STAGE_STFT_MAG, st);
    constexpr int WAVES = 4;
    fast::MagArgs M;
    M.view = v; M.g = g;
    M.win = (const float*)h->wa32.p;
    M.tw512 = (const fast::cf*)h->tw512.p;
    M.tw1024 = (const fast::cf*)h->tw32.p;
    M.mag = mag;
    M.iir_b = iir_b; M.sub = sub;
    size_t lds = (size_t)(fast::FN + WAVES * fast::WAVE_CX_H) * sizeof(fast::cf) + 1024 * sizeof(float);
    auto kern = fast::k_mag_fast<WAVES>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
    dim3 grid((unsigned)((g.T + 4 * WAVES - 1) / (4 * WAVES)), (unsigned)ub);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, st, M);
    HIPCHK(h, hipGetLastError());
  } else {
    ProfScope ps(h, SG_STAGE_STFT_MAG, st);
    HIPCHK(h, stft_any<float>(h, v, g, ub, nullptr, mag, nullptr, 1.0, st));
  }
  return SG_OK;
}

// Variant-S non-stationary mask in two passes over |X| (nonstat.hpp): partials -> chain -> IIR + sigmoid (+ smoothing).
//   nonstat2_chain_ok: the recurrence can run tile-parallel (k_iir_part / k_iir_chain / k_iir_mask)
//   nonstat2_ok:       ... and k_iir_mask also smooths (nt instantiated, nf <= NS_MAX_NF): the mask in one kernel
// Other smoothing widths take k_iir_mask<0> for the raw sigmoid field and the general smoothing kernels after it.
static bool nonstat2_chain_ok(const sg_handle* h, const Geom& g) {
  if (h->p.variant != SG_VARIANT_S || h->p.stationary || h->force_unfused) return false;
  const double b = h->p.iir_b, c = 1.0 - b;
  if (!(b > 0.0 && b < 1.0)) return false;
  // the backward sweep regenerates the forward values in reverse: error growth c^-rows must stay small
  return std::pow(c, (double)(NS_TT + 2 * NS_IIR_MAX_NT)) >= 1e-3 && g.T >= 1;
}
static bool nonstat2_ok(const sg_handle* h, const Geom& g) {
  if (!nonstat2_chain_ok(h, g) || !h->p.smooth_mask) return false;
  const int nf = h->p.n_grad_freq, nt = h->p.n_grad_time;
  if (nt > NS_IIR_MAX_NT && !(std::pow(1.0 - h->p.iir_b, (double)(NS_TT + 2 * nt)) >= 1e-3)) return false;   // (as above, for the longer column)
  return nf <= NS_MAX_NF && ns_iir_nt_ok(nt);   // instantiated time half-widths (the tile column lives in registers)
}

// smooth: IIR + sigmoid + smoothing + prop_decrease -> M;  !smooth: the raw sigmoid field -> raw (smoothing follows),
// or, without a smoothing filter, p * sigmoid + (1 - p) -> M
static int stage_nonstat_mask2(sg_handle* h, const View& v, const Geom& g, int64_t ub, bool smooth, hipStream_t st) {
  // register geometries: the magnitude kernel also leaves the recurrence partials of its tiles -- 16 frames at n_fft = 1024,
  // (round 6) 32 / 64 / 8 at 512 / 256 / 2048 -- no second pass over |X| (k_iir_part); PER pieces per 64-frame tile of the chain
  const int plen = h->fast5_ok ? 32 : h->fast25_ok ? 64 : h->fast20_ok ? 8 : 16, per = NS_TT / plen;
  const int64_t nsub = (g.T + plen - 1) / plen;
  const bool sub_small = (h->fast5_ok || h->fast25_ok || h->fast20_ok) && !h->force_nofast && SG_CHAIN_PAR && !h->force_split &&
                         ub <= 65535 && nsp_ok((g.T + NS_TT - 1) / NS_TT, per);   // (the serial chain kernels know 16-frame pieces only)
  const bool sub_ok = (h->fast_ok && !h->force_nofast && !(h->fast5_ok || h->fast20_ok || h->fast25_ok)) || sub_small;
  int rc;
  if (sub_ok && (rc = ensure(h, h->nss, (size_t)ub * nsub * 2 * g.FS * sizeof(double)))) return rc;
  rc = stage_mag(h, v, g, ub, st, h->p.iir_b, sub_ok ? (double*)h->nss.p : nullptr);
  if (rc) return rc;
  const float* mag = (const float*)h->P.p;
  const int nf = smooth ? h->p.n_grad_freq : 0, nt = smooth ? h->p.n_grad_time : 0;
  NsTiling tl{g.T, nt};
  const int64_t nk = tl.n_tiles();
  const size_t bytes = (size_t)ub * nk * 2 * g.FS * sizeof(double);
  if ((rc = ensure(h, h->nsp, bytes))) return rc;
  if ((rc = ensure(h, h->nsc, bytes))) return rc;
  {
    ProfScope ps(h, SG_STAGE_IIR_CHAIN, st);
    // (round 5) k_iir_chain_par: the chain in 16 runs per band, straight from the 16-frame sub-tile partials where
    // k_mag_fast left them (no k_iir_comb); SG_OPT_FORCE_SPLIT keeps the serial kernels (A/B)
    const bool par = SG_CHAIN_PAR && !h->force_split && ub <= 65535;
    const dim3 pgrid((unsigned)((g.F + 63) / 64), (unsigned)ub);
    // c^len and 1 - c^(2 len) of a full piece and of the unit's last one (uniform: computed here, not per thread)
    auto piece_pows = [&](int plen, int64_t np, double* o) {
      const double c = 1.0 - h->p.iir_b, len_last = (double)(g.T - (np - 1) * plen);
      o[0] = std::pow(c, (double)plen); o[1] = 1.0 - std::pow(c, 2.0 * plen);
      o[2] = std::pow(c, len_last); o[3] = 1.0 - std::pow(c, 2.0 * len_last);
    };
    double pw[4];
    if (sub_ok && par && nsp_ok(nk, per)) {
      piece_pows(plen, nsub, pw);
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, pgrid, dim3(64 * NSP_WAVES), 0, st, mag, (const double*)h->nss.p, g, tl, h->p.iir_b,
                           (double*)h->nsc.p, (int)nsub, pw[0], pw[1], pw[2], pw[3]);
      };
      if (per == 4) go(k_iir_chain_par<4>);
      else if (per == 2) go(k_iir_chain_par<2>);
      else if (per == 1) go(k_iir_chain_par<1>);
      else go(k_iir_chain_par<8>);
      HIPCHK(h, hipGetLastError());
    } else {
      if (sub_ok)
        hipLaunchKernelGGL(k_iir_comb, dim3((unsigned)((nk * g.FS + 255) / 256), (unsigned)ub), dim3(256), 0, st,
                           (const double*)h->nss.p, g, tl, h->p.iir_b, (double*)h->nsp.p, (int)nsub);
      else
        hipLaunchKernelGGL(k_iir_part, dim3((unsigned)((nk * (g.FS / 4) + 255) / 256), (unsigned)ub), dim3(256), 0,
                           st, mag, g, tl, h->p.iir_b, (double*)h->nsp.p);
      HIPCHK(h, hipGetLastError());
      // (tile partials: n_fft = 256 / 512 / 2048 -- 162 / 81 / 21 tiles per unit -- measured 12 / 6 / 1.4 % of the call)
      if (par && nk >= 8 && nsp_ok(nk, 1)) {
        piece_pows(NS_TT, nk, pw);
        hipLaunchKernelGGL(k_iir_chain_par<1>, pgrid, dim3(64 * NSP_WAVES), 0, st, mag, (const double*)h->nsp.p, g, tl,
                           h->p.iir_b, (double*)h->nsc.p, (int)nk, pw[0], pw[1], pw[2], pw[3]);
      } else {
        hipLaunchKernelGGL((k_iir_chain<float, false>), dim3((unsigned)((ub * g.FS + 63) / 64)), dim3(64), 0, st, mag,
                           (const double*)h->nsp.p, g, tl, h->p.iir_b, (double*)h->nsc.p, ub);
      }
      HIPCHK(h, hipGetLastError());
    }
  }
  {
    ProfScope ps(h, SG_STAGE_IIR_MASK, st);
    const int BW = 64 - 2 * nf;
    const unsigned gx = (unsigned)(((g.F + BW - 1) / BW + 3) / 4);
    const bool to_raw = !smooth && h->p.smooth_mask;
    for (int64_t k0 = 0; k0 < nk; k0 += 65535) {   // grid.y <= 65535 tiles per launch (a 6-hour window has more)
      tl.k0 = k0;
      HIPCHK(h, launch_iir_mask(nt, dim3(gx, (unsigned)std::min<int64_t>(65535, nk - k0), (unsigned)ub), st, mag,
                                (const double*)h->nsc.p, g, tl, h->p.iir_b, h->p.nonstat_thresh, h->p.nonstat_slope, nf,
                                to_raw ? 1.0f : (float)h->p.prop_decrease, to_raw ? (float*)h->raw.p : (float*)h->M.p));
    }
  }
  return SG_OK;
}

// Variant-T non-stationary mask in one kernel (nonstat.hpp: k_box_mask): the default moving-mean length and the
// smoothing widths k_iir_mask also covers; other settings keep k_boxcar_sigmoid + the general smoothing kernels.
static bool box_mask_ok(const sg_handle* h) {
  if (h->p.variant != SG_VARIANT_T || h->p.stationary || h->force_unfused || h->p.n_movemean != NS_BOX_KB) return false;
  if (!h->p.smooth_mask) return true;
  return h->p.n_grad_freq <= NS_MAX_NF && h->p.n_grad_time >= 1 && h->p.n_grad_time <= NS_BOX_MAX_NT;
}

static int stage_box_mask(sg_handle* h, const View& v, const Geom& g, int64_t ub, hipStream_t st) {
  int rc = stage_mag(h, v, g, ub, st);
  if (rc) return rc;
  ProfScope ps(h, SG_STAGE_NONSTAT_MASK, st);
  const int nf = h->p.smooth_mask ? h->p.n_grad_freq : 0, nt = h->p.smooth_mask ? h->p.n_grad_time : 0;
  const int BW = 64 - 2 * nf;
  const unsigned gx = (unsigned)(((g.F + BW - 1) / BW + 3) / 4);
  const unsigned nk = (unsigned)((g.T + NS_TT - 1) / NS_TT);
  for (int64_t k0 = 0; k0 < (int64_t)nk; k0 += 65535)   // grid.y <= 65535 tiles per launch
    HIPCHK(h, launch_box_mask(nt, h->p.n_movemean, dim3(gx, (unsigned)std::min<int64_t>(65535, (int64_t)nk - k0), (unsigned)ub), st,
                              (const float*)h->P.p, g, h->p.nonstat_thresh, h->p.nonstat_slope, nf, (float)h->p.prop_decrease,
                              (float*)h->M.p, k0));
  return SG_OK;
}

static int stage_nonstat_raw(sg_handle* h, const View& v, const Geom& g, int64_t ub, hipStream_t st) {
  int rc0 = stage_mag(h, v, g, ub, st);
  if (rc0) return rc0;
  float* mag = (float*)h->P.p;
  ProfScope ps(h, SG_STAGE_NONSTAT_MASK, st);
  dim3 grid((g.F + 63) / 64, (unsigned)ub);
  if (h->p.variant == SG_VARIANT_S) {
    // |Z| scale (1/sum_w) cancels in (A-S)/S: work on the unscaled magnitude.
    if (g.T >= 4 * IIR_NSEG)
      hipLaunchKernelGGL(k_iir_sigmoid_seg, grid, dim3(64 * IIR_NSEG), 0, st, (const float*)mag, g, h->p.iir_b,
                         h->p.nonstat_thresh, h->p.nonstat_slope, (float*)h->raw.p);
    else
      hipLaunchKernelGGL(k_iir_sigmoid, grid, dim3(64), 0, st, (const float*)mag, g, h->p.iir_b,
                         h->p.nonstat_thresh, h->p.nonstat_slope, (float*)h->raw.p);
  } else {
    dim3 bgrid((unsigned)((g.F + 63) / 64), (unsigned)((g.T + 4 * BOX_TSEG - 1) / (4 * BOX_TSEG)), (unsigned)ub);
    hipLaunchKernelGGL(k_boxcar_sigmoid, bgrid, dim3(256), 0, st, (const float*)mag, g, h->p.n_movemean,
                       h->p.nonstat_thresh, h->p.nonstat_slope, (float*)h->raw.p);
  }
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

static int stage_smooth(sg_handle* h, const Geom& g, int64_t ub, hipStream_t st) {
  ProfScope ps(h, SG_STAGE_SMOOTH, st);
  int64_t cells = ub * g.T * g.FS;
  float p = (float)h->p.prop_decrease;
  if (!h->p.smooth_mask) {
    hipLaunchKernelGGL(k_prop_only, dim3(grid_1d(cells, 256)), dim3(256), 0, st, (const float*)h->raw.p, g, p,
                       (float*)h->M.p, ub);
    HIPCHK(h, hipGetLastError());
    return SG_OK;
  }
  // prop_decrease is applied before smoothing by stationary.py:108-114 and torchgate.py:241-249,
  // after smoothing by nonstationary.py:78-84.
  const int prop_before0 = (h->p.variant == SG_VARIANT_T || h->p.stationary) ? 1 : 0;
  {
    const int nf = h->p.n_grad_freq, nt = h->p.n_grad_time;
    const int rows = SMF_TT + 2 * nt, cols = SMF_FB + 2 * nf;
    // +3: the sliding windows read up to 3 entries past the last tap
    size_t lds = ((size_t)(rows + 3) * (cols | 1) + (size_t)(rows + 3) * (SMF_FB + 1) + 8 + 64 + SMF_KT + SMF_FB + SMF_TT) * sizeof(float);
    // (time half-widths up to 94: short frames have long smoothing windows in frames -- n_fft = 256 at 48 kHz: nt = 37 --
    // and used to fall through to the two direct global-memory convolutions below: 0.73 ms of a 1.09 ms call)
    if (lds <= 150 * 1024 && ub <= 65535 && nf <= 30 && 2 * nt + 4 <= SMF_KT && SMF_FB + 2 * nf <= 192) {
      auto kern = k_smooth_tiled;
      if (lds > 65536)
        HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
      dim3 grid((g.F + SMF_FB - 1) / SMF_FB, (unsigned)((g.T + SMF_TT - 1) / SMF_TT), (unsigned)ub);
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const float*)h->raw.p, g, (const float*)h->kf.p, nf,
                         (const float*)h->kt.p, nt, p, prop_before0, (float*)h->M.p);
      HIPCHK(h, hipGetLastError());
      return SG_OK;
    }
  }
  float* tmp = (float*)h->seg.p;  // seg is not live yet
  hipLaunchKernelGGL(k_smooth_f, dim3(grid_1d(cells, 256)), dim3(256), 0, st, (const float*)h->raw.p, g,
                     (const float*)h->kf.p, h->p.n_grad_freq, tmp, ub);
  HIPCHK(h, hipGetLastError());
  // prop_decrease is applied before smoothing by stationary.py:108-114 and torchgate.py:241-249,
  // after smoothing by nonstationary.py:78-84.
  int prop_before = (h->p.variant == SG_VARIANT_T || h->p.stationary) ? 1 : 0;
  hipLaunchKernelGGL(k_smooth_t, dim3(grid_1d(cells, 256)), dim3(256), 0, st, (const float*)tmp, g,
                     (const float*)h->kt.p, h->p.n_grad_time, (const float*)h->kf.p, h->p.n_grad_freq, p,
                     prop_before, (float*)h->M.p, ub);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

static int stage_apply_ola(sg_handle* h, const View& v, const Geom& g, int64_t ub, const float* M,
                           const OutMap& om, int normalize, hipStream_t st, const unsigned short* K16 = nullptr,
                           float kscale = 0.f) {
  {
    ProfScope ps(h, SG_STAGE_APPLY_ISTFT, st);
    HIPCHK(h, apply_any(h, v, g, ub, M, (float*)h->seg.p, st, K16, kscale));
  }
  int64_t np = om.p1 - om.p0;
  if (np > 0) {
    ProfScope ps(h, SG_STAGE_OLA, st);
    dim3 grid((unsigned)((np + 255) / 256), (unsigned)ub);
    hipLaunchKernelGGL(k_ola, grid, dim3(256), 0, st, v, g, om, (const float*)h->seg.p, (const float*)h->wsq32.p,
                       normalize);
    HIPCHK(h, hipGetLastError());
  }
  return SG_OK;
}

// bits -> K (uint16 weight sums), natural or lane order
static int stage_smooth_bits(sg_handle* h, const Geom& g, int64_t ub, bool fast, int64_t tb, int64_t te,
                             hipStream_t st) {
  const int wpr = (g.F + 63) / 64;
  ProfScope ps(h, SG_STAGE_SMOOTH, st);
  const int nf = h->p.n_grad_freq, nt = h->p.n_grad_time;
  int64_t cells = ub * g.T * g.FS;
  if (h->p.smooth_mask && nf <= 30) {
    const int tt = h->sm2_tt;
    const int rows = tt + 2 * nt;
    const bool small = (nf + 1) * (nf + 1) <= 255;
    const unsigned long long* ftab = (small && h->ftab.p) ? (const unsigned long long*)h->ftab.p : nullptr;
    size_t lds = smooth2_cf_bytes(rows + 2, g.F, small ? 1 : 2) + (size_t)rows * (wpr + 2) * 8 + (ftab ? 8192 : 0);
    dim3 grid((unsigned)((te - tb + tt - 1) / tt), (unsigned)ub);
    if (small) {
      auto kern = k_smooth_bits2<uint8_t>;
      if (lds > 65536)
        HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
      hipLaunchKernelGGL(kern, grid, dim3(SM2_THREADS), lds, st, (const unsigned long long*)h->bits.p, g, wpr, nf,
                         nt, (unsigned short*)h->K16.p, fast ? 1 : 0, tb, te, ftab, tt);
    } else {
      auto kern = k_smooth_bits2<uint16_t>;
      if (lds > 65536)
        HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
      hipLaunchKernelGGL(kern, grid, dim3(SM2_THREADS), lds, st, (const unsigned long long*)h->bits.p, g, wpr, nf,
                         nt, (unsigned short*)h->K16.p, fast ? 1 : 0, tb, te, (const unsigned long long*)nullptr, tt);
    }
  } else if (h->p.smooth_mask) {
    const int rows = SM_TT + 2 * nt;
    const bool small = (nf + 1) * (nf + 1) <= 255;
    size_t lds = smooth_cf_bytes(rows, g.F, small ? 1 : 2) + (size_t)rows * wpr * 8;
    dim3 grid((unsigned)((g.T + SM_TT - 1) / SM_TT), (unsigned)ub);
    if (small) {
      auto kern = k_smooth_bits<uint8_t>;
      if (lds > 65536)
        HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const unsigned long long*)h->bits.p, g, wpr, nf, nt,
                         (unsigned short*)h->K16.p, fast ? 1 : 0);
    } else {
      auto kern = k_smooth_bits<uint16_t>;
      if (lds > 65536)
        HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const unsigned long long*)h->bits.p, g, wpr, nf, nt,
                         (unsigned short*)h->K16.p, fast ? 1 : 0);
    }
  } else {
    hipLaunchKernelGGL(k_bits_to_k16, dim3(grid_1d(cells, 256)), dim3(256), 0, st,
                       (const unsigned long long*)h->bits.p, g, wpr, (unsigned short*)h->K16.p, ub, fast ? 1 : 0);
  }
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

// Compare constants + per-unit floor flags + (rare) float64 floor pre-pass: what every decision kernel of
// the fused stationary path needs before it can run.
static int stage_prep_floor(sg_handle* h, const View& v, const Geom& g, int64_t ub, ThreshConsts* tc_out,
                            hipStream_t st, const View* v_exact = nullptr, unsigned* live_host = nullptr,
                            unsigned stamp = 0) {
  const int wpr = (g.F + 63) / 64;
  int rc;
  if ((rc = ensure(h, h->bits, (size_t)ub * g.T * wpr * 8))) return rc;
  // umax is kept clean by its consumer (k_prep_thresh zeroes what it read); pmax rows are zeroed by
  // k_prep_thresh for exactly the units whose floor can be live (the only rows anybody reads): no memset
  // launches on the critical path
  if ((rc = ensure_zeroed(h, h->umax, (size_t)ub * 4, st))) return rc;
  if ((rc = ensure_zeroed(h, h->need, (size_t)ub * 4, st))) return rc;
  if ((rc = ensure(h, h->T2, (size_t)g.FS * 8))) return rc;
  {
    ProfScope ps(h, SG_STAGE_PREP, st);
    hipLaunchKernelGGL(k_unit_absmax, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(64, 2048 / ub)), (unsigned)ub),
                       dim3(256), 0, st, v, ub, (unsigned*)h->umax.p);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(k_prep_thresh, dim3((unsigned)((ub + 255) / 256)), dim3(256), 0, st,
                       (const double*)h->thresh.p, g.F, h->mag_scale, h->sum_abs_w, h->p.top_db,
                       (unsigned*)h->umax.p, ub, (double*)h->T2.p, (int*)h->need.p, (double*)h->pmax.p, g.FS, live_host,
                       stamp);
    HIPCHK(h, hipGetLastError());
  }
  ThreshConsts tc{(const double*)h->T2.p, (const double*)h->thresh.p, (const double*)h->pmax.p,
                  (const int*)h->need.p};
  {
    ProfScope ps(h, SG_STAGE_STFT_MAX, st);
    // (float64 transform of the ORIGINAL samples when `v` is a float32 copy)
    HIPCHK(h, launch_bits_any<0>(h, v_exact ? *v_exact : v, g, ub, tc,
                             (unsigned long long*)h->pmax.p, (unsigned long long*)h->bits.p, wpr, st));
  }
  *tc_out = tc;
  return SG_OK;
}

static int handoff_prepare(sg_handle* h, hipStream_t st);
// Fused stationary mask (variant S): STFT(f64) -> bits -> integer smoothing -> K16 -> float mask.
static int stage_fused_mask(sg_handle* h, const View& v, const Geom& g, int64_t ub, bool fast, int64_t tb,
                            int64_t te, hipStream_t st) {
  // [tb, te): frames whose smoothed mask is needed; decisions are needed nt frames further out
  const int64_t nt_halo = h->p.smooth_mask ? h->p.n_grad_time : 0;
  const int64_t db = std::max<int64_t>(0, tb - nt_halo), de = std::min<int64_t>(g.T, te + nt_halo);
  h->dbg_db = (fast && !h->force_f64_decide) ? db : 0;
  h->dbg_de = (fast && !h->force_f64_decide) ? de : g.T;
  h->dbg_xbits = false;
  const int wpr = (g.F + 63) / 64;
  int rc;
  if ((rc = ensure(h, h->K16, (size_t)ub * g.T * g.FS * 2))) return rc;
  ThreshConsts tc{};
  // (round 6) The register-transform decision kernels of n_fft = 512 / 256 / 2048 stage every sample of a unit's window: they
  // run the floor test themselves (thresh.hpp: FloorLazy -- the one-pass gate's protocol) instead of k_unit_absmax +
  // k_prep_thresh reading the recording once more before the gate.  Same prediction as stage_onepass: a priori when a recent
  // call on the handle reported (SG_OPT_FLOOR_TEST forces either); both end in exact band maxima.
  const bool reg_path = !fast && !h->force_f64_decide && !h->force_nofast && (h->fast20_ok || h->fast25_ok || h->fast5_ok);
  bool lazy = false;
  FloorLazy fl{};
  if (reg_path && h->floor_test != 1) {
    if ((rc = handoff_prepare(h, st))) return rc;     // (a fresh epoch = the tag of this call's flags; the host-mapped stamp)
    const unsigned live_stamp = h->err_host[1];
    const unsigned need_tag = h->epoch & 0x3fffffffu;
    const unsigned era = h->epoch >> 30;
    if (era != h->need_era || need_tag == 0u) {   // (tags wrapped: flags of the previous era carry larger ones)
      if (h->need.p) HIPCHK(h, hipMemsetAsync(h->need.p, 0, h->need.bytes, st));
      if (h->alim.p) HIPCHK(h, hipMemsetAsync((char*)h->alim.p + 4, 0, 4, st));
      h->need_era = era;
    }
    lazy = need_tag != 0u && (h->floor_test == 2 || !(live_stamp != 0u && h->epoch - live_stamp <= 16u));
    if (lazy) {
      const int nb = (g.FS + 63) / 64;
      if ((rc = ensure(h, h->bits, (size_t)ub * g.T * wpr * 8))) return rc;
      if ((rc = ensure_zeroed(h, h->need, (size_t)ub * 4, st))) return rc;
      if ((rc = ensure(h, h->T2, (size_t)g.FS * 8))) return rc;
      if ((rc = ensure_zeroed(h, h->alim, 256, st))) return rc;
      if (!h->t2_ready) {   // the threshold did not come from sg_noise_stats (whose last kernel derives T2 / alim itself)
        ProfScope ps(h, SG_STAGE_PREP, st);
        hipLaunchKernelGGL(k_prep_thresh_lazy, dim3(1), dim3(256), 0, st, (const double*)h->thresh.p, g.F, h->mag_scale,
                           h->sum_abs_w, h->p.top_db, (double*)h->T2.p, (unsigned*)h->alim.p, nb);
        HIPCHK(h, hipGetLastError());
        h->t2_ready = true;
      }
      tc = ThreshConsts{(const double*)h->T2.p, (const double*)h->thresh.p, (const double*)h->pmax.p, (const int*)h->need.p,
                        need_tag};
      fl = FloorLazy{(unsigned*)h->alim.p, nb, h->err_dev + 1, h->epoch};
    }
  }
  if (reg_path) ++(lazy ? h->n_floor_lazy : h->n_floor_apriori);
  if (!lazy && (rc = stage_prep_floor(h, v, g, ub, &tc, st, nullptr, reg_path && h->err_dev ? h->err_dev + 1 : nullptr, h->epoch)))
    return rc;
  if (fast && !h->force_f64_decide) {
    ProfScope ps(h, SG_STAGE_DECIDE_FAST, st);
    constexpr int WAVES = 4;
    fast::DecideArgs D;
    D.view = v; D.g = g;
    D.win = (const float*)h->wa32.p;
    D.win64 = (const double*)h->wfull64.p;
    D.tw512 = (const fast::cf*)h->tw512.p;
    D.tw1024 = (const fast::cf*)h->tw32.p;
    D.tw64 = (const cx<double>*)h->tw64.p;
    D.tc = tc;
    D.mag_scale = h->mag_scale; D.top_db = h->p.top_db;
    D.bits = (unsigned long long*)h->bits.p;
    D.wpr = wpr;
    D.t_begin = db; D.t_end = de;
    D.quads_per_wave = 1;
    const int64_t quads = (D.t_end - D.t_begin + 3) / 4;
    const int64_t per_block = (int64_t)WAVES * D.quads_per_wave;
    size_t lds = (size_t)(fast::FN + WAVES * fast::WAVE_CX_H) * sizeof(fast::cf) + (T2_FLOATS + 1024) * sizeof(float);
    auto kern = fast::k_decide_fast<WAVES>;
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
    dim3 grid((unsigned)((quads + per_block - 1) / per_block), (unsigned)ub);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, st, D);
    HIPCHK(h, hipGetLastError());
  } else if (!h->force_f64_decide && h->fast20_ok && !h->force_nofast) {
    int rc20 = stage_decide2048(h, v, g, ub, tc, (unsigned long long*)h->bits.p, st, fl);
    if (rc20) return rc20;
  } else if (!h->force_f64_decide && h->fast25_ok && !h->force_nofast) {
    // n_fft = 256: register transform, four frames per lane group
    int rc25 = stage_decide256(h, v, g, ub, tc, (unsigned long long*)h->bits.p, st, fl);
    if (rc25) return rc25;
  } else if (!h->force_f64_decide && h->fast5_ok && !h->force_nofast) {
    // n_fft = 512: register transform, two frames per lane group
    int rc5 = stage_decide512(h, v, g, ub, tc, (unsigned long long*)h->bits.p, st, fl);
    if (rc5) return rc5;
  } else if (!h->force_f64_decide) {
    // other power-of-two frame lengths: float32 LDS transform + exact refinement
    ProfScope ps(h, SG_STAGE_STFT_BITS, st);
    HIPCHK(h, launch_decide_lds(h, v, g, ub, tc, (unsigned long long*)h->bits.p, wpr, st));
  } else {
    ProfScope ps(h, SG_STAGE_STFT_BITS, st);
    HIPCHK(h, launch_bits_any<1>(h, v, g, ub, tc,
                             (unsigned long long*)h->pmax.p, (unsigned long long*)h->bits.p, wpr, st));
  }
  if (lazy) {
    // the units whose floor test fired: float64 band maxima, then their decisions again with them.  Both launches return at
    // once when no unit reported (the common case)
    {
      ProfScope ps(h, SG_STAGE_STFT_MAX, st);
      HIPCHK(h, launch_bits_any<0>(h, v, g, ub, tc, (unsigned long long*)h->pmax.p, (unsigned long long*)h->bits.p, wpr, st));
    }
    int rcr = h->fast20_ok ? stage_decide2048(h, v, g, ub, tc, (unsigned long long*)h->bits.p, st, fl, true)
              : h->fast25_ok ? stage_decide256(h, v, g, ub, tc, (unsigned long long*)h->bits.p, st, fl, true)
                             : stage_decide512(h, v, g, ub, tc, (unsigned long long*)h->bits.p, st, fl, true);
    if (rcr) return rcr;
  }
  { int rc2 = stage_smooth_bits(h, g, ub, fast, tb, te, st); if (rc2) return rc2; }
  const int nf = h->p.n_grad_freq, nt = h->p.n_grad_time;
  int64_t cells = ub * g.T * g.FS;
  if (fast) return SG_OK;  // the fused apply kernel reads K directly
  if ((h->fast5_ok || h->fast20_ok || h->fast25_ok) && !h->force_nofast && h->p.prop_decrease == 1.0) return SG_OK;   // so do k_apply_fast512 / 2048 / 256<K>
  // (round 5) ... and k_apply_istft (power-of-two frames on the LDS transform): the float mask field is only written when
  // somebody asks for it (sg_debug_fetch field 1 expands the K counts of the last batch then)
  h->dbg_k16_only = k16_apply_geom(h) && h->p.prop_decrease == 1.0;
  h->dbg_g = g;
  if (h->dbg_k16_only) return SG_OK;
  hipLaunchKernelGGL(k_k16_to_mask, dim3(grid_1d(cells / 8, 256)), dim3(256), 0, st, (const unsigned short*)h->K16.p,
                     g, nf, nt, 1.0f / (float)h->ktot, (float)h->p.prop_decrease, 1, h->p.smooth_mask ? 1 : 0,
                     (float*)h->M.p, ub);
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

// Hand-offs between workgroups of one launch (tagged granules, bounded polls): the host-mapped error word, the
// verdict on earlier launches, a fresh epoch.  Granule buffers are zero when (re)allocated and the epoch only grows:
// a fresh granule never carries it.
// A new tag for the granules of the next hand-off launch (a call that launches twice takes two).
static int handoff_next_epoch(sg_handle* h, hipStream_t st) {
  if (++h->epoch == 0) {  // wrapped: tags of 2^32 launches ago could alias
    if (h->xbits.p) HIPCHK(h, hipMemsetAsync(h->xbits.p, 0, h->xbits.bytes, st));
    if (h->xpart.p) HIPCHK(h, hipMemsetAsync(h->xpart.p, 0, h->xpart.bytes, st));
    h->epoch = 1;
    if (h->err_host) h->err_host[1] = 0u;   // the floor-test stamp is an epoch too
  }
  return SG_OK;
}

static int handoff_prepare(sg_handle* h, hipStream_t st) {
  {
    // The work counter is never reset: every launch takes exactly its grid size in tickets, the kernels subtract
    // the running base.
    bool fresh = false;
    int rc = ensure_zeroed(h, h->xticket, 64, st, &fresh);
    if (rc) return rc;
    if (fresh) h->ticket_base = 0;
  }
  if (!h->err_host) {
    HIPCHK(h, hipHostMalloc((void**)&h->err_host, 64, hipHostMallocMapped));
    *h->err_host = 0u;
    h->err_host[1] = 0u;   // floor-test stamp (stage_onepass)
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->err_dev, h->err_host, 0));
  }
  if (*h->err_host != 0u) {
    // only reached by callers that never synchronise through the library (sg_check_errors and every synchronising
    // entry point report a lost hand-off for the call that suffered it)
    const unsigned e = *h->err_host;
    *h->err_host = 0u;
    FAIL(h, SG_E_HANDOFF, "a tile hand-off of an EARLIER call on this handle timed out (code %u): that call's output "
                          "is invalid (call sg_check_errors after a call to learn about it in time)", e);
  }
  h->lose_now = 0;
  if (h->inject_fault) {  // test hook: this launch "loses" a hand-off
    *h->err_host = h->inject_fault & 7u;          // bits 0..2: reported only (the output is fine)
    h->lose_now = (h->inject_fault >> 3) & 7u;    // bits 3..5: lost INSIDE the kernel (the output is poisoned)
    h->inject_fault = 0;
  }
  return handoff_next_epoch(h, st);
}

// After the stream has been synchronised: did a hand-off of the work just completed time out?
static int handoff_verdict(sg_handle* h) {
  if (h->err_host && *h->err_host != 0u) {
    const unsigned e = *h->err_host;
    *h->err_host = 0u;
    FAIL(h, SG_E_HANDOFF, "a tile hand-off timed out (code %u: %s%s%s): the output of the call(s) enqueued since the "
                          "last check is invalid; re-run them (SG_OPT_FORCE_SPLIT + SG_OPT_FORCE_NOLEAN select the "
                          "kernels without in-launch hand-offs)",
         e, (e & 1u) ? "mask bits " : "", (e & 2u) ? "partial hops (one-pass gate) " : "",
         (e & 4u) ? "partial hops (fused apply)" : "");
  }
  return SG_OK;
}

extern "C" int sg_check_errors(sg_handle* h, void* stream) {
  if (!h) return SG_E_INVALID;
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  return handoff_verdict(h);
}

// Fused apply (default geometry): FFT -> mask(K) -> IFFT -> overlap-add -> output, one kernel.
static int stage_apply_fast(sg_handle* h, const View& v, const Geom& g, int64_t ub, const OutMap& om,
                            const float* mask_f /* nullptr: uint16 counts in h->K16 */, int normalize,
                            hipStream_t st) {
  ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
  constexpr int WAVES = SG_APPLY_WAVES;
  fast::ApplyArgs A;
  A.view = v; A.g = g; A.om = om;
  A.K = (const unsigned short*)h->K16.p;
  A.Mf = mask_f;
  A.normalize = normalize;
  A.win = (const float*)h->wa32.p;
  A.wsq = (const float*)h->wsq32.p;
  A.invn = (const float*)h->invn.p;
  A.tw512 = (const fast::cf*)h->tw512.p;
  A.tw1024 = (const fast::cf*)h->tw32.p;
  A.kscale = mask_f ? (float)(1.0 / 512.0) : (float)(1.0 / ((double)h->ktot * 512.0));
  A.h_begin = (om.p0 + g.padL) / 256;
  A.h_end = (om.p1 - 1 + g.padL) / 256 + 1;
  const int64_t nh = A.h_end - A.h_begin;
  if (nh <= 0) return SG_OK;
  constexpr int NF = 4 * WAVES, NH = NF - 3;
  size_t lds = (size_t)(fast::FN + WAVES * fast::WAVE_CX) * sizeof(fast::cf);
  // seam mode: abutting tiles + k_ola_seam for the straddling hops (3/16 fewer transforms)
  const int64_t tiles_seam = (nh + 3 + NF - 1) / NF;
  const bool seam = !h->force_noseam && tiles_seam >= 2;
  const bool lean = !h->force_nolean;
  // lean kernel: the straddling hops are handed from tile to tile inside the launch; else partial sums + k_ola_seam
  const bool inkernel = seam && lean;
  A.part = nullptr;
  A.part2 = nullptr;
  A.epoch = 0;
  A.err = nullptr;
  A.n_tiles = 0;
  A.ticket = nullptr;
  if (inkernel) {
    int rc = ensure_zeroed(h, h->xpart, (size_t)ub * tiles_seam * 3 * 256 * 8, st);
    if (rc) return rc;
    if ((rc = handoff_prepare(h, st))) return rc;
    A.part2 = (unsigned long long*)h->xpart.p;
    A.epoch = h->epoch;
    A.err = h->err_dev;
    A.n_tiles = (int)tiles_seam;
    // tile = ticket (fastpath.hpp: ApplyArgs::ticket): one self-resetting counter per unit
    if ((rc = ensure_zeroed(h, h->xtick2, (size_t)ub * 64, st))) return rc;
    A.ticket = (unsigned*)h->xtick2.p;
  } else if (seam) {
    int rc = ensure(h, h->seam, (size_t)ub * tiles_seam * 6 * 256 * sizeof(float));
    if (rc) return rc;
    A.part = (float*)h->seam.p;
    A.n_tiles = (int)tiles_seam;
  }
  dim3 grid((unsigned)(seam ? tiles_seam : (nh + NH - 1) / NH), (unsigned)ub);
  auto go = [&](auto kern, size_t lds_bytes) -> hipError_t {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds_bytes, st, A);
    return hipGetLastError();
  };
  const size_t lds_lean = (size_t)(fast::FN + WAVES * fast::WAVE_CX_H) * sizeof(fast::cf) + 1024 * sizeof(float) + 16;
  const bool lose = inkernel && (h->lose_now & 4u);   // test hook: the instantiation whose hand-off polls give up at once
  if (mask_f) {
    if (lose) HIPCHK(h, go(fast::k_apply_fast<WAVES, false, true, true>, lds_lean));
    else if (lean) HIPCHK(h, go(fast::k_apply_fast<WAVES, false, true>, lds_lean));
    else HIPCHK(h, go(fast::k_apply_fast<WAVES, false, false>, lds));
  } else {
    if (lose) HIPCHK(h, go(fast::k_apply_fast<WAVES, true, true, true>, lds_lean));
    else if (lean) HIPCHK(h, go(fast::k_apply_fast<WAVES, true, true>, lds_lean));
    else HIPCHK(h, go(fast::k_apply_fast<WAVES, true, false>, lds));
  }
  HIPCHK(h, hipGetLastError());
  if (seam && !inkernel) {
    hipLaunchKernelGGL(fast::k_ola_seam<NF>, dim3((unsigned)(tiles_seam - 1), (unsigned)ub), dim3(256), 0, st, A);
    HIPCHK(h, hipGetLastError());
  }
  return SG_OK;
}

// One-pass stationary gate (default geometry, onepass.hpp): forward transform once per frame; decide,
// publish the tile's bits, smooth in LDS, x mask, inverse transform, overlap-add -- one kernel (+ the
// seam kernel for the 3 hops that straddle two tiles).  Returns 1 when the shape is not eligible
// (the caller then runs the three-kernel path).
static bool onepass_ok(const sg_handle* h, const Geom& g, const OutMap& om) {
  if (h->force_split || h->force_f64_decide || h->force_noseam || h->force_nolean) return false;
  if (!h->p.smooth_mask || h->p.n_grad_freq > 8 || h->p.n_grad_time > fast::OP_MAX_NT || !h->ftab3.p || !h->optab.p) return false;
  if (g.F != 513) return false;
  const int64_t hb = (om.p0 + g.padL) / 256, he = (om.p1 - 1 + g.padL) / 256 + 1;
  return (he - hb + 3 + 15) / 16 >= 2;  // at least two abutting tiles (seam mode)
}

static int stage_onepass(sg_handle* h, const View& v, const View& vx, const Geom& g, int64_t ub, const OutMap& om,
                         hipStream_t st) {
  constexpr int WAVES = 4, NF = 16;
  int rc;
  // (first: the floor test stamps err_host[1] with this launch's epoch)
  if ((rc = handoff_prepare(h, st))) return rc;
  // Floor test (is some band's -top_db floor possibly live?): a priori -- k_unit_absmax reads the recording once more, 26 us
  // of a 345 us call at 10 minutes of 48 kHz -- or by the gate kernel on the samples it stages (onepass.hpp "floor test"),
  // which costs nothing unless a unit reports; then that unit is gated twice.  Both are exact; the choice is a
  // prediction from the host-mapped stamp "a unit of launch <epoch> reported / had its flag set", read WITHOUT
  // synchronising (it may lag by the calls still queued): recent -> a priori.
  const unsigned live_stamp = h->err_host[1];
  // the flags the tiles raise are tagged with this launch's epoch (30 bits; 0 = untagged): nothing to clear per call
  const unsigned need_tag = h->epoch & 0x3fffffffu;
  // Once per 2^30 launches the tag wraps: flags of the previous era carry LARGER tags, which atomicMax would keep.  Not
  // every epoch reaches this function (a lazy call takes two, stage_apply_fast takes its own), so the crossing is
  // detected by the era (epoch >> 30) changing between two calls here, not by need_tag == 0.
  const unsigned era = h->epoch >> 30;
  if (era != h->need_era || need_tag == 0u) {
    if (h->need.p) HIPCHK(h, hipMemsetAsync(h->need.p, 0, h->need.bytes, st));
    if (h->alim.p) HIPCHK(h, hipMemsetAsync((char*)h->alim.p + 4, 0, 4, st));
    h->need_era = era;
  }
  const bool lazy = need_tag != 0u &&
                    (h->floor_test == 2 || (h->floor_test == 0 && !(live_stamp != 0u && h->epoch - live_stamp <= 16u)));
  ThreshConsts tc{};
  ++(lazy ? h->n_floor_lazy : h->n_floor_apriori);
  if (!lazy) {
    if ((rc = stage_prep_floor(h, v, g, ub, &tc, st, &vx, h->err_dev + 1, h->epoch))) return rc;
  } else {
    const int wpr = (g.F + 63) / 64;
    if ((rc = ensure(h, h->bits, (size_t)ub * g.T * wpr * 8))) return rc;
    if ((rc = ensure_zeroed(h, h->need, (size_t)ub * 4, st))) return rc;
    if ((rc = ensure(h, h->T2, (size_t)g.FS * 8))) return rc;
    if ((rc = ensure_zeroed(h, h->alim, 256, st))) return rc;
    if (!h->t2_ready) {
      // the threshold did not come from sg_noise_stats (whose last kernel derives T2 / alim itself): one small launch
      ProfScope ps(h, SG_STAGE_PREP, st);
      hipLaunchKernelGGL(k_prep_thresh_lazy, dim3(1), dim3(256), 0, st, (const double*)h->thresh.p, g.F, h->mag_scale,
                         h->sum_abs_w, h->p.top_db, (double*)h->T2.p, (unsigned*)h->alim.p);
      HIPCHK(h, hipGetLastError());
      h->t2_ready = true;
    }
    tc = ThreshConsts{(const double*)h->T2.p, (const double*)h->thresh.p, (const double*)h->pmax.p, (const int*)h->need.p,
                      need_tag};
  }
  fast::OnePassArgs P;
  P.x_exact = vx.x; P.stride_exact = vx.stride; P.dtype_exact = vx.dtype;
  fast::ApplyArgs& A = P.A;
  A.view = v; A.g = g; A.om = om;
  A.K = nullptr; A.Mf = nullptr;
  A.normalize = 1;
  A.win = (const float*)h->wa32.p;
  A.wsq = (const float*)h->wsq32.p;
  A.invn = (const float*)h->invn.p;
  A.tw512 = (const fast::cf*)h->tw512.p;
  A.tw1024 = (const fast::cf*)h->tw32.p;
  const bool prop = h->p.prop_decrease != 1.0;
  A.kscale = prop ? (float)(1.0 / 512.0) : (float)(1.0 / ((double)h->ktot * 512.0));
  A.h_begin = (om.p0 + g.padL) / 256;
  A.h_end = (om.p1 - 1 + g.padL) / 256 + 1;
  const int64_t nh = A.h_end - A.h_begin;
  const int64_t n_tiles = (nh + 3 + NF - 1) / NF;
  const int64_t ntt = n_tiles + 2;
  if ((rc = ensure_zeroed(h, h->xpart, (size_t)ub * n_tiles * 3 * 256 * 8, st))) return rc;
  A.part = nullptr;
  A.part2 = nullptr;   // (the one-pass kernel has its own hand-off arguments)
  A.epoch = 0;
  A.err = nullptr;
  A.n_tiles = (int)n_tiles;
  if ((rc = ensure_zeroed(h, h->xbits, (size_t)ub * ntt * fast::OP_TILE_WORDS * 8, st))) return rc;
  // granule buffers are zero when (re)allocated and the epoch only grows: a fresh granule never carries it
  P.alim = lazy ? (unsigned*)h->alim.p : nullptr;
  {
    // the part of a unit's window outside its tiles' spans, dealt evenly to the unit's tiles (onepass.hpp "floor test")
    constexpr int64_t SPAN = (NF - 1) * 256 + 1024;
    const int64_t sp0 = (A.h_begin - 3 - NF) * 256 - g.padL, sp1 = (A.h_begin - 3 + n_tiles * NF) * 256 - g.padL + SPAN;
    const int64_t inside = std::max<int64_t>(0, std::min<int64_t>(v.Lp, sp1) - std::max<int64_t>(0, sp0));
    const int64_t q = (v.Lp - inside + ntt - 1) / ntt;
    if (q > 0x7fffffff) FAIL(h, SG_E_UNSUPPORTED, "one-pass gate: window of %lld samples", (long long)v.Lp);
    P.scan_q = (int)q;
  }
  P.tab = (const char*)h->optab.p;
  P.tc = tc;
  P.mag_scale = h->mag_scale; P.top_db = h->p.top_db;
  P.xbits = (unsigned long long*)h->xbits.p;
  P.part2 = (unsigned long long*)h->xpart.p;
  P.ticket = (unsigned*)h->xticket.p;
  P.epoch = h->epoch;
  P.err = h->err_dev;
  P.ticket_base = h->ticket_base;
  // persistent workgroups (onepass.hpp "PERSIST"): the lazy first launch only -- its compare constants do not depend on the unit
  const bool persist = lazy && h->tile_order == 0 && !(h->lose_now & 3u) && ub * ntt >= 2;
  if (persist && h->n_cu == 0) {
    int dev = 0, cus = 0;
    HIPCHK(h, hipGetDevice(&dev));
    HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    h->n_cu = cus > 0 ? cus : 256;
  }
  // three workgroups per CU (LDS) is what is resident at once; every workgroup ends on one ticket past the last tile
  static const int pg_per_cu = [] { const char* e = getenv("SG_ONEPASS_WG_PER_CU"); int v = e ? atoi(e) : 3; return v >= 1 && v <= 3 ? v : 3; }();   // (experiments: fewer resident workgroups)
#if OP_MAX_ITERS == 1
  const unsigned pgrid = persist ? (unsigned)(ub * ntt) : 0u;   // (diagnosis) one workgroup per tile, each draws its one ticket
#else
  const unsigned pgrid = persist ? (unsigned)std::min<int64_t>(ub * ntt, (int64_t)pg_per_cu * h->n_cu) : 0u;
#endif
  P.total_tiles = (unsigned)(ub * ntt);
  if (h->tile_order == 1) P.ticket_base = 0xffffffffu;   // SG_OPT_TILE_ORDER 1: tile = block index (no ticket)
  else h->ticket_base += (unsigned)(ub * ntt) + (OP_MAX_ITERS == 1 ? 0u : pgrid);
  P.nf = h->p.n_grad_freq; P.nt = h->p.n_grad_time;
  P.prop = (float)h->p.prop_decrease;
  P.inv_ktot = 1.0f / (float)h->ktot;
#if OP_WHO
  if (!g_who_dev) HIPCHK(h, hipMalloc((void**)&g_who_dev, 4096 * 16));
  HIPCHK(h, hipMemsetAsync(g_who_dev, 0, 4096 * 16, st));
  P.who = g_who_dev;
#endif
#if OP_TRACE
  {
    // development builds: one process-wide device trace [workgroup][wave][16], overwritten by every launch, averaged at exit
    static unsigned* tr = nullptr;
    static size_t tr_slots = 0;
    const size_t slots = (size_t)ub * ntt * 4;
    if (!tr || tr_slots < slots) {
      HIPCHK(h, hipMalloc((void**)&tr, slots * 64));
      tr_slots = slots;
      static unsigned** trp = &tr;
      static size_t* trn = &tr_slots;
      static bool reg = false;
      if (!reg) {
        reg = true;
        atexit([] {
          std::vector<unsigned> hst(*trn * 16);
          if (hipMemcpy(hst.data(), *trp, hst.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
          double sum[14] = {0}, w = 0, mx = 0;
          for (size_t i = 0; i < *trn; ++i) {
            if (hst[i * 16 + 15] != 1u) continue;
            w += 1;
            double tot = 0;
            for (int k = 0; k < 14; ++k) { sum[k] += hst[i * 16 + k]; tot += hst[i * 16 + k]; }
            mx = std::max(mx, tot);
          }
          fprintf(stderr, "[OP_TRACE] completed waves %.0f; average shader cycles per wave and phase:", w);
          double tot = 0;
          for (int k = 0; k < 14; ++k) { fprintf(stderr, " %d:%.0f", k, sum[k] / (w > 0 ? w : 1)); tot += sum[k] / (w > 0 ? w : 1); }
          fprintf(stderr, "  sum %.0f  longest wave %.0f\n", tot, mx);
        });
      }
    }
    HIPCHK(h, hipMemsetAsync(tr, 0, slots * 64, st));
    P.trace = tr;
    g_trace_dev = tr; g_trace_tiles = (size_t)ub * ntt; g_trace_ntt = (int)ntt;
  }
#endif
  {
    ProfScope ps(h, SG_STAGE_ONEPASS, st);
    const size_t lds = (size_t)(fast::FN + WAVES * fast::WAVE_CX_H) * sizeof(fast::cf) + (1024 + T2_FLOATS) * sizeof(float) +
                       256 * 8 + 514 * 8 + 32 + (prop ? 528 : 0);
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(persist ? pgrid : (unsigned)(ub * ntt)), dim3(WAVES * 64), lds, st, P);
      return hipGetLastError();
    };
    if (persist && prop) HIPCHK(h, go(fast::k_gate_onepass<WAVES, true, false, false, true>));
    else if (persist) HIPCHK(h, go(fast::k_gate_onepass<WAVES, false, false, false, true>));
    else if (h->lose_now & 3u) HIPCHK(h, go(fast::k_gate_onepass<WAVES, false, true>));   // test hook (PROP-free shape only)
    else if (prop) HIPCHK(h, go(fast::k_gate_onepass<WAVES, true>));
    else HIPCHK(h, go(fast::k_gate_onepass<WAVES, false>));
  }
  if (lazy) {
    // the units whose floor test fired: float64 band maxima, then the gate again with them.  Both launches exit at once
    // when no unit reported (the common case)
    ProfScope ps(h, SG_STAGE_STFT_MAX, st);
    const int wpr = (g.F + 63) / 64;
    HIPCHK(h, launch_bits_any<0>(h, vx, g, ub, tc,
                             (unsigned long long*)h->pmax.p, (unsigned long long*)h->bits.p, wpr, st));
    if ((rc = handoff_next_epoch(h, st))) return rc;
    P.epoch = h->epoch;
    P.ticket = (unsigned*)h->xticket.p + 8;   // its own counter (zeroed by the first launch's ticket-0 workgroup): it takes tickets only if a unit reported
    P.ticket_base = 0;
    const size_t lds = (size_t)(fast::FN + WAVES * fast::WAVE_CX_H) * sizeof(fast::cf) + (1024 + T2_FLOATS) * sizeof(float) +
                       256 * 8 + 514 * 8 + 32 + (prop ? 528 : 0);
    auto redo = [&](auto kern) -> hipError_t {
      hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3((unsigned)(ub * ntt)), dim3(WAVES * 64), lds, st, P);
      return hipGetLastError();
    };
    if (prop) HIPCHK(h, redo(fast::k_gate_onepass<WAVES, true, false, true>));
    else HIPCHK(h, redo(fast::k_gate_onepass<WAVES, false, false, true>));
    HIPCHK(h, hipGetLastError());
  }
  h->dbg_xbits = true;
  h->dbg_trows = 16; h->dbg_tstep = 16; h->dbg_twords = fast::OP_TILE_WORDS; h->dbg_txw = fast::OP_XW;
  h->dbg_tf0 = A.h_begin - 3;
  h->dbg_ntt = (int)ntt;
  h->dbg_db = std::max<int64_t>(0, A.h_begin - 3 - NF);
  h->dbg_de = std::min<int64_t>(g.T, A.h_begin - 3 + NF * (n_tiles + 1));
  return SG_OK;
}

// float64 fused apply (apply64.hpp): K counts of stage_fused_mask -> output samples, everything in double
template <int WAVES, bool KMASK>
static hipError_t launch_apply_fast64(fast::Apply64Args& A, int64_t nh, int64_t ub, hipStream_t st) {
  constexpr int NH = 4 * WAVES - 3;
  const size_t lds = (size_t)(fast::FN + WAVES * 4 * fast::FSLOTS_D + 17) * sizeof(fast::cd);
  auto kern = fast::k_apply_fast64<WAVES, KMASK>;
  hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)((nh + NH - 1) / NH), (unsigned)ub), dim3(WAVES * 64), lds, st, A);
  return hipGetLastError();
}

// mask_f64 == nullptr: the uint16 counts in h->K16 (stationary gate); else a float64 mask field [units][T][FS]
static int stage_apply_fast64(sg_handle* h, const View& v, const Geom& g, int64_t ub, const OutMap& om, hipStream_t st,
                              const double* mask_f64 = nullptr) {
  ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
  fast::Apply64Args A;
  A.view = v; A.g = g; A.om = om;
  A.K = (const unsigned short*)h->K16.p;
  A.Mf = mask_f64;
  A.win = (const double*)h->wfull64.p;
  A.norm = (const double*)h->norm64.p;
  A.tw1024 = (const fast::cd*)h->tw64.p;
  A.kscale = mask_f64 ? 1.0 / 512.0 : 1.0 / ((double)h->ktot * 512.0);
  A.prop = h->p.prop_decrease;
  A.nf = h->p.smooth_mask ? h->p.n_grad_freq : 0;
  A.nt = h->p.smooth_mask ? h->p.n_grad_time : 0;
  A.h_begin = (om.p0 + g.padL) / 256;
  A.h_end = (om.p1 - 1 + g.padL) / 256 + 1;
  const int64_t nh = A.h_end - A.h_begin;
  if (nh <= 0) return SG_OK;
  // tiles overlap by three frames: 8 wavefronts (32 frames -> 29 hops, 10 % of the transforms redone) unless the rows are
  // short; either way two wavefronts per SIMD (256 VGPRs) and one (8) or two (4) workgroups per CU
  static const int waves_env = getenv("SG_APPLY64_WAVES") ? atoi(getenv("SG_APPLY64_WAVES")) : 0;
  const bool wide = waves_env ? waves_env == 8 : nh >= 64;
  if (mask_f64) {
    if (wide) HIPCHK(h, (launch_apply_fast64<8, false>(A, nh, ub, st)));
    else HIPCHK(h, (launch_apply_fast64<4, false>(A, nh, ub, st)));
  } else {
    if (wide) HIPCHK(h, (launch_apply_fast64<8, true>(A, nh, ub, st)));
    else HIPCHK(h, (launch_apply_fast64<4, true>(A, nh, ub, st)));
  }
  return SG_OK;
}

// The stationary gate in float64 WITHOUT materialised float64 fields (default geometry, full reduction): the mask comes
// from the fused bit path -- decisions bit-identical to float64 decisions (k_decide_fast: float32 transform + exact
// refinement of ambiguous cells on the ORIGINAL samples), integer smoothing -- and k_apply_fast64 does the transforms,
// the mask multiply and the overlap-add in double.  Integer outputs (trunc(float64 result), base.py:217-226) and
// precision="float64" take it; everything else of the float64 contract stays on run_S_exact.
static bool exact_fused_ok(const sg_handle* h, const Geom& g) {
  return h->p.stationary && h->fused_ok && !h->force_unfused && h->fast_ok && !h->force_nofast && !h->force_f64_decide &&
         g.F == 513 && h->norm64.p != nullptr && !h->exact_materialised;   // (any prop_decrease: k_apply_fast64 forms p K + (1 - p) edge)
}

static int run_S_exact_fused(sg_handle* h, View v, int64_t total_units, const OutMap& om, hipStream_t st) {
  const Geom g = make_geom(h, v.Lp);
  int64_t ub = units_per_batch(h, g, total_units, true);
  int rc = ensure_ws(h, g, ub, true);
  if (rc) return rc;
  // (other sample types than float32 are NOT converted to a float32 copy here: k_decide_fast stages (float)sample per tile
  // anyway, and its exact refinement and the float64 apply must read the ORIGINAL samples -- an int32 or float64 sample
  // has no exact float32 copy)
  for (int64_t u0 = 0; u0 < total_units; u0 += ub) {
    const int64_t nb = std::min(ub, total_units - u0);
    v.unit0 = u0;
    const int64_t hb = (om.p0 + g.padL) / 256, he = (om.p1 - 1 + g.padL) / 256 + 1;
    const int64_t tb = std::max<int64_t>(0, hb - 3), te = std::min<int64_t>(g.T, he);
    if ((rc = stage_fused_mask(h, v, g, nb, true, tb, std::max(te, tb + 1), st))) return rc;
    if ((rc = stage_apply_fast64(h, v, g, nb, om, st))) return rc;
    h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = false; h->dbg_fused = true; h->dbg_fast = true;
  }
  return SG_OK;
}

// ------------------------------------------------------------------------------------------
// exact path (exact.hpp): float64 fields, materialised -- integer outputs (base.py:217-226 truncates a float64 result)
// ------------------------------------------------------------------------------------------
template <int N>
static hipError_t launch_xapply_n(const View& v, const Geom& g, int64_t units, const void* tw, const double* win,
                                  const double* M, double* seg, hipStream_t st) {
  constexpr int NT = N >= SG_TEAM_N ? 256 : 64;
  constexpr int WAVES = N >= SG_TEAM_N ? 1 : ((N * sizeof(cx<double>) > 16384) ? 2 : 4);
  constexpr int FPW = 4;
  const size_t lds = (size_t)(N + WAVES * lpn<double>(N)) * sizeof(cx<double>);
  auto kern = exact::kx_apply_istft<N, WAVES, FPW, NT>;
  if (lds > 65536) {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((g.T + WAVES * FPW - 1) / (WAVES * FPW)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * NT), lds, st, v, g, (const cx<double>*)tw, win, M, seg);
  return hipGetLastError();
}

template <int M>
static hipError_t launch_xapply_czt_m(const View& v, const Geom& g, int64_t units, const CztTabs<double>& tb,
                                      const double* win, const double* Mk, double* seg, hipStream_t st) {
  constexpr int NT = CztShape<M>::NT, FR = CztShape<M>::FR;
  const size_t lds = (size_t)FR * lpn<double>(M) * sizeof(cx<double>);
  auto kern = exact::kx_apply_istft_czt<M, NT, FR>;
  if (lds > 65536) {
    hipError_t e = set_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  const int fpb = 4;
  dim3 grid((unsigned)((g.T + FR * fpb - 1) / (FR * fpb)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(NT * FR), lds, st, v, g, tb, win, Mk, seg, fpb);
  return hipGetLastError();
}

static hipError_t xapply_any(sg_handle* h, const View& v, const Geom& g, int64_t units, const double* Mk, double* seg,
                             hipStream_t st) {
  const double* win = (const double*)h->wfull64.p;
  if (h->big_M) {
    const int rc = big_apply<double, double>(h, v, g, units, Mk, seg, st);
    return rc == SG_OK ? hipSuccess : (rc == SG_E_NOMEM ? hipErrorOutOfMemory : hipErrorUnknown);
  }
  if (h->czt_M) {
    const CztTabs<double> tb = czt_tabs<double>(h);
#define SG_CALL(M) launch_xapply_czt_m<M>(v, g, units, tb, win, Mk, seg, st)
    SG_CZT_SWITCH(h->czt_M, SG_CALL);
#undef SG_CALL
  }
  switch (h->N) {
    case 32: return launch_xapply_n<32>(v, g, units, h->tw64.p, win, Mk, seg, st);
    case 64: return launch_xapply_n<64>(v, g, units, h->tw64.p, win, Mk, seg, st);
    case 128: return launch_xapply_n<128>(v, g, units, h->tw64.p, win, Mk, seg, st);
    case 256: return launch_xapply_n<256>(v, g, units, h->tw64.p, win, Mk, seg, st);
    case 512: return launch_xapply_n<512>(v, g, units, h->tw64.p, win, Mk, seg, st);
    case 1024: return launch_xapply_n<1024>(v, g, units, h->tw64.p, win, Mk, seg, st);
    case 2048: return launch_xapply_n<2048>(v, g, units, h->tw64.p, win, Mk, seg, st);
    case 4096: return launch_xapply_n<4096>(v, g, units, h->tw64.p, win, Mk, seg, st);
  }
  return hipErrorInvalidValue;
}

// float64 pipeline (exact.hpp): bytes per unit -- P, raw, M, tmp (8 B per cell each) + frames (8 B per sample of every
// frame) + the statistics rows -- and units per batch.  Shared with sg_workspace_bytes.
static size_t exact_unit_bytes(const Geom& g) {
  // (+ the non-stationary gate's tile partials and carries: 2 x 2 doubles per band and 32-frame tile, one tile more than T / 32)
  return (size_t)g.T * g.FS * 32 + (size_t)g.T * g.n * 8 + (size_t)g.FS * 16 + (size_t)(g.T / 32 + 2) * g.FS * 32;
}
static int64_t exact_units_per_batch(const sg_handle* h, const Geom& g, int64_t total_units) {
  return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ws_budget(h) / (int64_t)exact_unit_bytes(g), 32768), total_units));
}

static int run_S_exact(sg_handle* h, View v, int64_t total_units, const OutMap& om, hipStream_t st) {
  const Geom g = make_geom(h, v.Lp);
  const size_t cells1 = (size_t)g.T * g.FS;
  int64_t ub = exact_units_per_batch(h, g, total_units);
  int rc;
  const size_t cells = (size_t)ub * cells1;
  if ((rc = ensure(h, h->xP, cells * 8))) return rc;
  if ((rc = ensure(h, h->xraw, cells * 8))) return rc;
  if ((rc = ensure(h, h->xM, cells * 8))) return rc;
  if ((rc = ensure(h, h->xtmp, cells * 8))) return rc;
  // default geometry: k_apply_fast64 reads the float64 mask field and writes samples (no masked frames through HBM)
  const bool apply64 = h->fast_ok && !h->force_nofast && g.F == 513 && h->norm64.p && !h->exact_materialised;
  if (!apply64 && (rc = ensure(h, h->xseg, (size_t)ub * g.T * g.n * 8))) return rc;
  if ((rc = ensure(h, h->pmax, (size_t)ub * g.FS * 8))) return rc;
  const int nf = h->p.n_grad_freq, nt = h->p.n_grad_time;
  const double p = h->p.prop_decrease;
  double* P = (double*)h->xP.p;
  for (int64_t u0 = 0; u0 < total_units; u0 += ub) {
    const int64_t nb = std::min(ub, total_units - u0);
    v.unit0 = u0;
    const int64_t ncell = nb * g.T * g.FS;
    {
      ProfScope ps(h, SG_STAGE_STFT_POWER, st);
      if (h->p.stationary) HIPCHK(h, hipMemsetAsync(h->pmax.p, 0, (size_t)nb * g.FS * 8, st));
      HIPCHK(h, stft_any<double>(h, v, g, nb, P, nullptr, nullptr, 1.0, st,
                                 h->p.stationary ? (unsigned long long*)h->pmax.p : nullptr));
    }
    const int prop_before = h->p.stationary ? 1 : 0;
    if (h->p.stationary) {
      ProfScope ps(h, SG_STAGE_DECIDE, st);
      hipLaunchKernelGGL(k_decide, dim3(grid_1d(ncell, 256)), dim3(256), 0, st, (const double*)P, g,
                         (const double*)h->pmax.p, (const double*)h->thresh.p, (int64_t)0, h->mag_scale, h->p.top_db,
                         (float*)h->xraw.p, nb);
      HIPCHK(h, hipGetLastError());
    } else if (g.T >= 4 * exact::XIIR_TT && nb <= 65535 && !h->exact_materialised) {
      // (round 5) tile-parallel: partials of 32-frame tiles -> chain -> both sweeps per tile from the entering states
      ProfScope ps(h, SG_STAGE_NONSTAT_MASK, st);
      NsTiling tl{g.T, 0};
      tl.tt = exact::XIIR_TT;
      const int64_t nk = tl.n_tiles();
      const size_t bytes = (size_t)nb * nk * 2 * g.FS * sizeof(double);
      if ((rc = ensure(h, h->nsp, bytes))) return rc;
      if ((rc = ensure(h, h->nsc, bytes))) return rc;
      const dim3 gp((unsigned)((nk * g.FS + 255) / 256), (unsigned)nb);
      hipLaunchKernelGGL(exact::kx_iir_part, gp, dim3(256), 0, st, (const double*)P, g, tl, h->p.iir_b, (double*)h->nsp.p);
      HIPCHK(h, hipGetLastError());
      hipLaunchKernelGGL((k_iir_chain<double, true>), dim3((unsigned)((nb * g.FS + 63) / 64)), dim3(64), 0, st, (const double*)P,
                         (const double*)h->nsp.p, g, tl, h->p.iir_b, (double*)h->nsc.p, nb);
      HIPCHK(h, hipGetLastError());
      hipLaunchKernelGGL(exact::kx_iir_apply, gp, dim3(256), 0, st, (const double*)P, (const double*)h->nsc.p, g, tl, h->p.iir_b,
                         h->p.nonstat_thresh, h->p.nonstat_slope, (double*)h->xraw.p);
      HIPCHK(h, hipGetLastError());
    } else {
      ProfScope ps(h, SG_STAGE_NONSTAT_MASK, st);
      hipLaunchKernelGGL(exact::kx_iir_sigmoid, dim3((unsigned)((g.F + 63) / 64), (unsigned)nb), dim3(64), 0, st,
                         (const double*)P, g, h->p.iir_b, h->p.nonstat_thresh, h->p.nonstat_slope, (double*)h->xraw.p);
      HIPCHK(h, hipGetLastError());
    }
    {
      ProfScope ps(h, SG_STAGE_SMOOTH, st);
      const dim3 gr(grid_1d(ncell, 256));
      if (!h->p.smooth_mask) {
        if (h->p.stationary)
          hipLaunchKernelGGL(exact::kx_prop_only<float>, gr, dim3(256), 0, st, (const float*)h->xraw.p, g, p, (double*)h->xM.p, nb);
        else
          hipLaunchKernelGGL(exact::kx_prop_only<double>, gr, dim3(256), 0, st, (const double*)h->xraw.p, g, p, (double*)h->xM.p, nb);
      } else if (!h->exact_materialised && exact::xsm_lds_bytes(nf, nt) <= 150 * 1024 && 2 * nf < exact::XSM_KMAX && 2 * nt < exact::XSM_KMAX && nb <= 65535 &&
                 (g.T + exact::XSM_TT - 1) / exact::XSM_TT <= 65535) {
        // (round 5) both passes in one LDS-tiled kernel, taps computed once per block
        const size_t lds = exact::xsm_lds_bytes(nf, nt);
        const dim3 gt((unsigned)((g.F + exact::XSM_FB - 1) / exact::XSM_FB), (unsigned)((g.T + exact::XSM_TT - 1) / exact::XSM_TT), (unsigned)nb);
        if (h->p.stationary) {
          auto kern = exact::kx_smooth_tiled<float>;
          HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
          hipLaunchKernelGGL(kern, gt, dim3(exact::XSM_THREADS), lds, st, (const float*)h->xraw.p, g, nf, nt, p, prop_before, (double*)h->xM.p);
        } else {
          auto kern = exact::kx_smooth_tiled<double>;
          HIPCHK(h, set_lds(reinterpret_cast<const void*>(kern), lds));
          hipLaunchKernelGGL(kern, gt, dim3(exact::XSM_THREADS), lds, st, (const double*)h->xraw.p, g, nf, nt, p, prop_before, (double*)h->xM.p);
        }
      } else {
        if (h->p.stationary)
          hipLaunchKernelGGL(exact::kx_smooth_f<float>, gr, dim3(256), 0, st, (const float*)h->xraw.p, g, nf, (double*)h->xtmp.p, nb);
        else
          hipLaunchKernelGGL(exact::kx_smooth_f<double>, gr, dim3(256), 0, st, (const double*)h->xraw.p, g, nf, (double*)h->xtmp.p, nb);
        HIPCHK(h, hipGetLastError());
        hipLaunchKernelGGL(exact::kx_smooth_t, gr, dim3(256), 0, st, (const double*)h->xtmp.p, g, nt, nf, p, prop_before,
                           (double*)h->xM.p, nb);
      }
      HIPCHK(h, hipGetLastError());
    }
    // default geometry: transforms, mask multiply and overlap-add in ONE float64 kernel reading the mask field (round 5)
    // instead of masked frames through HBM + a gather
    if (apply64) {
      if ((rc = stage_apply_fast64(h, v, g, nb, om, st, (const double*)h->xM.p))) return rc;
      h->dbg_units = 0;
      continue;
    }
    {
      ProfScope ps(h, SG_STAGE_APPLY_ISTFT, st);
      HIPCHK(h, xapply_any(h, v, g, nb, (const double*)h->xM.p, (double*)h->xseg.p, st));
    }
    const int64_t np = om.p1 - om.p0;
    if (np > 0) {
      ProfScope ps(h, SG_STAGE_OLA, st);
      hipLaunchKernelGGL(exact::kx_ola, dim3((unsigned)((np + 255) / 256), (unsigned)nb), dim3(256), 0, st, v, g, om,
                         (const double*)h->xseg.p, (const double*)h->wfull64.p);
      HIPCHK(h, hipGetLastError());
    }
    h->dbg_units = 0;   // the exact path keeps no debug fields
  }
  return SG_OK;
}

// Variant S over a set of units described by `v` (unit0 filled per batch).
static int run_S(sg_handle* h, View v, int64_t total_units, const OutMap& om, hipStream_t st) {
  Geom g = make_geom(h, v.Lp);
  if (v.Lp < h->W) FAIL(h, SG_E_INVALID, "signal window of %lld samples is shorter than win_length=%d",
                        (long long)v.Lp, h->W);
  if (h->p.stationary && !h->has_thresh)
    FAIL(h, SG_E_STATE, "stationary gate: call sg_noise_stats or sg_set_noise_threshold first");
  // integer outputs are the TRUNCATED float64 result of the reference (base.py:217-226): float64 pipeline
  if (h->force_exact || ((om.dtype == SG_I16 || om.dtype == SG_I32) && !h->fast_integer))
    return exact_fused_ok(h, g) ? run_S_exact_fused(h, v, total_units, om, st) : run_S_exact(h, v, total_units, om, st);
  // one-pass gate (any prop_decrease) or, with prop_decrease == 1, the three-kernel bit-mask path: only bit /
  // count fields in the workspace
  const bool onepass = h->fused_ok && !h->force_unfused && h->fast_ok && !h->force_nofast && onepass_ok(h, g, om);
  const bool onepass5 = !onepass && onepass512_ok(h, g, om);
  const bool onepass25 = !onepass && !onepass5 && onepass256_ok(h, g, om);
  const bool onepass20 = !onepass && !onepass5 && !onepass25 && onepass2048_ok(h, g, om);
  const bool lean = onepass || onepass5 || onepass25 || onepass20 || (h->fused_ok && !h->force_unfused && h->fast_ok && !h->force_nofast &&
                                            h->p.prop_decrease == 1.0);
  int64_t ub = units_per_batch(h, g, total_units, lean);
  int rc = ensure_ws(h, g, ub, lean);
  if (rc) return rc;
  // Samples that are not float32: the register-FFT kernels would take the checked per-sample path for every
  // frame (in every kernel).  Convert the readable part of the rows ONCE -- (float)sample is what those kernels
  // compute anyway -- and keep the original view for the float64 work (exact refinement, floor pre-pass).
  const View vx = v;
  if (v.dtype != SG_F32 && ((h->fast_ok && !h->force_nofast && (onepass || (!h->p.stationary && nonstat2_ok(h, g)))) || onepass5 || onepass25 || onepass20)) {
    const int64_t rows = total_units / std::max<int64_t>(1, v.n_chunks), len = v.hi - v.lo;
    const size_t bytes = (size_t)rows * len * sizeof(float);
    if (len > 0 && bytes <= ((size_t)16 << 30)) {
      if ((rc = ensure(h, h->xin, bytes))) return rc;
      ProfScope ps(h, SG_STAGE_PREP, st);
      hipLaunchKernelGGL(k_to_f32, dim3(grid_1d(rows * len, 256)), dim3(256), 0, st, v.x, v.dtype, v.stride, v.lo, len, rows,
                         (float*)h->xin.p);
      HIPCHK(h, hipGetLastError());
      v.x = (const float*)h->xin.p - v.lo;   // index g of row r -> xin[r * len + (g - lo)]
      v.dtype = SG_F32;
      v.stride = len;
    }
  }
  for (int64_t u0 = 0; u0 < total_units; u0 += ub) {
    int64_t nb = std::min(ub, total_units - u0);
    v.unit0 = u0;
    const bool fused = h->fused_ok && !h->force_unfused;
    const bool geom_fast = h->fast_ok && !h->force_nofast;  // default geometry: fused apply kernel
    const bool fast = fused && geom_fast && h->p.prop_decrease == 1.0;
    if (onepass) {
      View vxb = vx;
      vxb.unit0 = u0;
      if ((rc = stage_onepass(h, v, vxb, g, nb, om, st))) return rc;
      h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = false; h->dbg_fused = true; h->dbg_fast = true;
      continue;
    }
    if (onepass5) {
      View vxb = vx;
      vxb.unit0 = u0;
      if ((rc = stage_onepass512(h, v, vxb, g, nb, om, st))) return rc;
      h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = false; h->dbg_fused = true; h->dbg_fast = true; h->dbg_k16_only = false;
      continue;
    }
    if (onepass25) {
      View vxb = vx;
      vxb.unit0 = u0;
      if ((rc = stage_onepass256(h, v, vxb, g, nb, om, st))) return rc;
      h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = false; h->dbg_fused = true; h->dbg_fast = true; h->dbg_k16_only = false;
      continue;
    }
    if (onepass20) {
      View vxb = vx;
      vxb.unit0 = u0;
      if ((rc = stage_onepass2048(h, v, vxb, g, nb, om, st))) return rc;
      h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = false; h->dbg_fused = true; h->dbg_fast = true; h->dbg_k16_only = false;
      continue;
    }
    if (fast) {
      // frames the fused apply kernel touches: hops [h_begin, h_end) need frames h-3 .. h
      const int64_t hb = (om.p0 + g.padL) / 256, he = (om.p1 - 1 + g.padL) / 256 + 1;
      const int64_t tb = std::max<int64_t>(0, hb - 3), te = std::min<int64_t>(g.T, he);
      if ((rc = stage_fused_mask(h, v, g, nb, true, tb, std::max(te, tb + 1), st))) return rc;
      if ((rc = stage_apply_fast(h, v, g, nb, om, nullptr, 1, st))) return rc;
      h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = false; h->dbg_fused = true; h->dbg_fast = true;
      continue;
    }
    // float mask field (natural bin order for the general apply kernels, lane order for the fused one)
    h->dbg_fast = geom_fast || (fused && (h->fast5_ok || h->fast20_ok || h->fast25_ok) && !h->force_nofast && h->p.prop_decrease == 1.0);
    if (fused) {
      if ((rc = stage_fused_mask(h, v, g, nb, false, 0, g.T, st))) return rc;
    } else {
      if (h->p.stationary) {
        if ((rc = stage_power(h, v, g, nb, st))) return rc;
        if ((rc = stage_decide(h, g, nb, (const double*)h->thresh.p, 0, st))) return rc;
      } else if (nonstat2_ok(h, g)) {
        if ((rc = stage_nonstat_mask2(h, v, g, nb, true, st))) return rc;
      } else if (nonstat2_chain_ok(h, g)) {
        if ((rc = stage_nonstat_mask2(h, v, g, nb, false, st))) return rc;
      } else {
        if ((rc = stage_nonstat_raw(h, v, g, nb, st))) return rc;
      }
      if (h->p.stationary || !(nonstat2_ok(h, g) || (nonstat2_chain_ok(h, g) && !h->p.smooth_mask)))
        if ((rc = stage_smooth(h, g, nb, st))) return rc;
    }
    if (geom_fast) {
      if ((rc = stage_apply_fast(h, v, g, nb, om, (const float*)h->M.p, 1, st))) return rc;
    } else if (h->fast25_ok && !h->force_nofast) {
      const bool kmask = fused && h->p.prop_decrease == 1.0;   // the bit-mask stages left uint16 sums, no float mask
      if ((rc = stage_apply256(h, v, g, nb, om, kmask ? nullptr : (const float*)h->M.p, 1, st))) return rc;
    } else if (h->fast5_ok && !h->force_nofast) {
      const bool kmask = fused && h->p.prop_decrease == 1.0;   // the bit-mask stages left uint16 sums, no float mask
      if ((rc = stage_apply512(h, v, g, nb, om, kmask ? nullptr : (const float*)h->M.p, 1, st))) return rc;
    } else if (h->fast20_ok && !h->force_nofast) {
      const bool kmask = fused && h->p.prop_decrease == 1.0;
      if ((rc = stage_apply2048(h, v, g, nb, om, kmask ? nullptr : (const float*)h->M.p, 1, st))) return rc;
    } else if (fused && h->dbg_k16_only) {
      if ((rc = stage_apply_ola(h, v, g, nb, nullptr, om, 1, st, (const unsigned short*)h->K16.p, 1.0f / (float)h->ktot))) return rc;
    } else {
      if ((rc = stage_apply_ola(h, v, g, nb, (const float*)h->M.p, om, 1, st))) return rc;
    }
    h->dbg_units = nb;
    h->dbg_T = g.T;
    h->dbg_has_P = h->p.stationary != 0 && !fused;
    h->dbg_has_raw = h->p.stationary || !nonstat2_chain_ok(h, g) || (!nonstat2_ok(h, g) && h->p.smooth_mask);
    h->dbg_fused = fused;
  }
  return SG_OK;
}

// TorchGate.forward of whole rows in one kernel (rowgate.hpp): variant T, stationary, statistics from the row itself,
// default geometry, rows of at most 64 frames (1 s clips at 16 kHz: 63), full reduction, smoothing filter within the
// kernel's sliding-window limits.
// One workgroup per row and one workgroup per CU at a time: a call of fewer rows than ~2/3 of the CUs is faster on the
// four-kernel path, which spreads a row over 5 workgroups (256 x 16000: 0.110 vs 0.146 ms; 128 rows: ~0.105 vs 0.097;
// 1024 rows: 0.45 vs 0.52; profiles/r04_rowgate_scale.json).
constexpr int64_t RG_MIN_ROWS = 160;
static bool rowgate_ok(const sg_handle* h, const Geom& g, int64_t rows) {
  if (h->rowgate_mode == 1 || (h->rowgate_mode == 0 && rows < RG_MIN_ROWS)) return false;
  return h->fast_ok && !h->force_nofast && !h->force_unfused && h->p.stationary &&
         h->p.prop_decrease == 1.0 && h->p.smooth_mask && h->ktot <= 65535 && g.F == 513 && g.T >= 1 &&
         g.T <= fast::RG_FRAMES && h->p.n_grad_freq >= 1 && h->p.n_grad_freq <= fast::RG_NFMAX &&
         h->p.n_grad_time >= 1 && h->p.n_grad_time <= fast::RG_NTMAX;
}

static int stage_row_gate(sg_handle* h, const View& v, const Geom& g, int64_t nb, const OutMap& om, float* mask_out,
                          hipStream_t st) {
  int rc;
  if ((rc = ensure(h, h->bits, (size_t)nb * g.T * 9 * 8))) return rc;
  if ((rc = ensure_zeroed(h, h->rg_count, 64, st))) return rc;
  fast::RowGateArgs A;
  A.view = v; A.g = g; A.om = om;
  A.win = (const float*)h->wa32.p;
  A.wsq = (const float*)h->wsq32.p;
  A.invn = (const float*)h->invn.p;
  A.tw512 = (const fast::cf*)h->tw512.p;
  A.tw1024 = (const fast::cf*)h->tw32.p;
  A.win64 = (const double*)h->wfull64.p;
  A.tw64 = (const cx<double>*)h->tw64.p;
  A.mag_scale = h->mag_scale; A.top_db = h->p.top_db; A.n_std = h->p.n_std_thresh; A.ddof = h->p.ddof;
  A.nf = h->p.n_grad_freq; A.nt = h->p.n_grad_time;
  A.kscale = (float)(1.0 / ((double)h->ktot * 512.0));
  A.inv_ktot = 1.0f / (float)h->ktot;
  A.mask_out = mask_out;
  A.bits_out = (unsigned long long*)h->bits.p;
  A.n_exact = (unsigned*)h->rg_count.p;
  A.ptile_out = nullptr;
  if (h->rg_tap) {   // SG_OPT_ROWGATE_TAP: float32 powers of pass 1 -> h->M ([rows][64][528] floats), fetched with sg_debug_fetch(4)
    if ((rc = ensure(h, h->M, (size_t)nb * 64 * fast::RG_PP * 4))) return rc;
    A.ptile_out = (float*)h->M.p;
  }
#if RG_TRACE
  {
    // development builds: [rows][16] stamps of the LAST launch, averaged per phase at exit
    static unsigned long long* tr = nullptr;
    static size_t tr_rows = 0;
    if (!tr || tr_rows < (size_t)nb) {
      HIPCHK(h, hipMalloc((void**)&tr, (size_t)nb * 128));
      tr_rows = (size_t)nb;
      static unsigned long long** trp = &tr;
      static size_t* trn = &tr_rows;
      static bool reg = false;
      if (!reg) {
        reg = true;
        atexit([] {
          std::vector<unsigned long long> hst(*trn * 16);
          if (hipMemcpy(hst.data(), *trp, hst.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
          double sum[12] = {0};
          for (size_t i = 0; i < *trn; ++i)
            for (int k = 1; k < 12; ++k) sum[k] += (double)(hst[i * 16 + k] - hst[i * 16 + k - 1]);
          {
            double a = 0, b = 0, c = 0, d = 0;
            for (size_t i = 0; i < *trn; ++i) {
              d += (double)(hst[i * 16 + 14] - hst[i * 16 + 6]);
              a += (double)(hst[i * 16 + 12] - hst[i * 16 + 14]); b += (double)(hst[i * 16 + 13] - hst[i * 16 + 12]);
              c += (double)(hst[i * 16 + 7] - hst[i * 16 + 13]);
            }
            fprintf(stderr, "[RG_TRACE] phase 7 of wave 0: after barrier %.0f, zero fill %.0f, tasks %.0f, barrier wait %.0f\n", d / *trn, a / *trn, b / *trn, c / *trn);
          }
          fprintf(stderr, "[RG_TRACE] rows %zu; average shader cycles per phase:", *trn);
          double tot = 0;
          for (int k = 1; k < 12; ++k) { fprintf(stderr, " %d:%.0f", k, sum[k] / *trn); tot += sum[k] / *trn; }
          fprintf(stderr, "  sum %.0f\n", tot);
        });
      }
    }
    A.trace = tr;
  }
#endif
  ProfScope ps(h, SG_STAGE_ROW_GATE, st);
  const size_t lds = fast::rowgate_lds_bytes();
  if (h->rg_shape == 8) {     // SG_OPT_ROWGATE_SHAPE: 8 waves x 2 quads (256 VGPRs, no scratch)
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(fast::k_row_gate<8, 2>), lds));
    hipLaunchKernelGGL((fast::k_row_gate<8, 2>), dim3((unsigned)nb), dim3(512), lds, st, A);
  } else {                    // default: 16 waves x 1 quad
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(fast::k_row_gate<16, 1>), lds));
    hipLaunchKernelGGL((fast::k_row_gate<16, 1>), dim3((unsigned)nb), dim3(1024), lds, st, A);
  }
  HIPCHK(h, hipGetLastError());
  return SG_OK;
}

static bool dtype_ok(int d) { return d >= SG_F32 && d <= SG_I32; }

extern "C" int sg_workspace_bytes(const sg_handle* h, int64_t C, int64_t N, int32_t chunked, int64_t* bytes) {
  if (!h || !bytes || C < 1 || N < 1) return SG_E_INVALID;
  const int64_t cs = h->p.chunk_size, pad = h->p.padding;
  const int64_t Lp = chunked ? cs + 2 * pad : N + 2 * pad;
  const int64_t units = chunked ? C * ((N + cs - 1) / cs) : C;
  if (Lp < h->W) return SG_E_INVALID;
  const Geom g = make_geom(h, Lp);
  // the decisions of run_S: one-pass gate (any prop_decrease) or the three-kernel bit-mask path -> lean workspace
  const bool geom_fast = h->fast_ok && !h->force_nofast;
  const bool onepass = h->p.variant == SG_VARIANT_S && h->fused_ok && !h->force_unfused && geom_fast && !h->force_split &&
                       !h->force_f64_decide && !h->force_noseam && !h->force_nolean && h->p.smooth_mask &&
                       h->p.n_grad_freq <= 8 && h->p.n_grad_time <= fast::OP_MAX_NT && h->ftab3.p != nullptr;
  const bool lean = h->p.variant == SG_VARIANT_S && h->fused_ok && !h->force_unfused && geom_fast &&
                    (onepass || h->p.prop_decrease == 1.0);
  const int64_t ub = units_per_batch(h, g, units, lean);
  int64_t total = ub * (int64_t)unit_bytes(h, g, lean);
  if (geom_fast) {
    // exchange buffers of the in-launch hand-offs (tagged granules): mask bits (one-pass gate) and partial hops
    const int64_t kept = chunked ? cs : N;
    const int64_t n_tiles = (kept / 256 + 1 + 3 + 15) / 16 + 1;
    total += ub * n_tiles * 3 * 256 * 8 + ub * 64;
    if (onepass) total += ub * (n_tiles + 2) * (int64_t)fast::OP_TILE_WORDS * 8;
    // a recording that is not float32 is converted once (UPPER bound: float32 recordings do not pay this)
    total += C * N * (int64_t)sizeof(float);
  }
  if (h->big_M) {   // long frames: two work buffers of <= 256 MB (big.hpp)
    const int64_t nb = big_batch(h, units * g.T);
    total += 2 * nb * (int64_t)h->big_M * (int64_t)sizeof(big::cd);
  }
  // Integer (SG_I16 / SG_I32) outputs take the float64 pipeline by default, SG_OPT_FORCE_EXACT selects it for every dtype
  // (run_S_exact: 32 B per cell + float64 frames).  The entry point has no dtype argument, so the figure is the LARGER
  // of the two pipelines unless the handle can never take the float64 one for this call (SG_OPT_FAST_INTEGER set and
  // SG_OPT_FORCE_EXACT clear) -- an upper bound, as documented in the header.
  if (h->p.variant == SG_VARIANT_S && (h->force_exact || !h->fast_integer)) {
    int64_t ex = exact_units_per_batch(h, g, units) * (int64_t)exact_unit_bytes(g);
    if (h->big_M) ex += 2 * big_batch(h, units * g.T) * (int64_t)h->big_M * (int64_t)sizeof(big::cd);
    if (h->force_exact) total = ex;
    else total = std::max(total, ex);
  }
  *bytes = total;
  return SG_OK;
}

// ------------------------------------------------------------------------------------------
// variant S entry points
// ------------------------------------------------------------------------------------------
extern "C" int sg_noise_stats(sg_handle* h, const void* noise_dev, int dtype, int64_t C, int64_t n,
                              int64_t row_stride, void* stream) {
  if (!h) return SG_E_INVALID;
  if (!noise_dev || !dtype_ok(dtype) || C < 1 || n < 1) FAIL(h, SG_E_INVALID, "sg_noise_stats: bad argument");
  if (h->p.variant != SG_VARIANT_S) FAIL(h, SG_E_INVALID, "sg_noise_stats is a variant-S entry point");
  if (n < h->W) FAIL(h, SG_E_INVALID, "noise clip of %lld samples is shorter than win_length=%d", (long long)n, h->W);
#ifdef SG_EXP_SKIP_STATS   // development experiment (tools/experiments/stats_in_launch.sh): the step without its statistics launches
  if (h->has_thresh && h->t2_ready) return SG_OK;
#endif
  hipStream_t st = (hipStream_t)stream;
  struct Tag {
    sg_handle* h;
    explicit Tag(sg_handle* h_) : h(h_) { h->prof_override = SG_STAGE_NOISE_STATS; }
    ~Tag() { h->prof_override = -1; }
  } tag(h);
  int rc = SG_OK;
  View v{};
  v.x = noise_dev; v.dtype = dtype; v.stride = row_stride;  // one channel: its mean is the channel itself
  if (C > 1) {
    if ((rc = ensure(h, h->yn, (size_t)n * sizeof(double)))) return rc;
    ProfScope ps(h, SG_STAGE_CHANNEL_MEAN, st);
    hipLaunchKernelGGL(k_channel_mean, dim3(grid_1d(n, 256)), dim3(256), 0, st, noise_dev, dtype, C, n,
                       row_stride, (double*)h->yn.p);
    HIPCHK(h, hipGetLastError());
    v.x = h->yn.p; v.dtype = SG_F64; v.stride = n;
  }
  v.N = n; v.lo = 0; v.hi = n; v.cs = 0; v.pad = 0; v.Lp = n; v.n_chunks = 1; v.unit0 = 0;
  Geom g = make_geom(h, n);
  if ((rc = ensure_ws(h, g, 1))) return rc;
  h->t2_ready = false;
  if ((rc = stage_stats(h, v, g, 1, (double*)h->thresh.p, st, /*gate_consts=*/h->fast_ok || h->fast5_ok || h->fast25_ok || h->fast20_ok))) return rc;
  h->has_thresh = true;
  return SG_OK;
}

extern "C" int sg_get_noise_threshold(sg_handle* h, double* thresh_host, int32_t n_bins, void* stream) {
  if (!h) return SG_E_INVALID;
  if (!thresh_host || n_bins != h->F) FAIL(h, SG_E_INVALID, "sg_get_noise_threshold: n_bins must be %d", h->F);
  if (!h->has_thresh) FAIL(h, SG_E_STATE, "no noise threshold set");
  HIPCHK(h, hipMemcpyAsync(thresh_host, h->thresh.p, (size_t)h->F * sizeof(double), hipMemcpyDeviceToHost,
                           (hipStream_t)stream));
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  return handoff_verdict(h);
}

extern "C" int sg_set_noise_threshold(sg_handle* h, const double* thresh_host, int32_t n_bins, void* stream) {
  if (!h) return SG_E_INVALID;
  if (!thresh_host || n_bins != h->F) FAIL(h, SG_E_INVALID, "sg_set_noise_threshold: n_bins must be %d", h->F);
  h->t2_ready = false;
  HIPCHK(h, hipMemcpyAsync(h->thresh.p, thresh_host, (size_t)h->F * sizeof(double), hipMemcpyHostToDevice,
                           (hipStream_t)stream));
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  h->has_thresh = true;
  return SG_OK;
}

extern "C" int sg_get_noise_threshold_dev(sg_handle* h, double* thresh_dev, int32_t n_bins, void* stream) {
  if (!h) return SG_E_INVALID;
  if (!thresh_dev || n_bins != h->F) FAIL(h, SG_E_INVALID, "sg_get_noise_threshold_dev: n_bins must be %d", h->F);
  if (!h->has_thresh) FAIL(h, SG_E_STATE, "no noise threshold set");
  HIPCHK(h, hipMemcpyAsync(thresh_dev, h->thresh.p, (size_t)h->F * sizeof(double), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
  return SG_OK;
}

extern "C" int sg_set_noise_threshold_dev(sg_handle* h, const double* thresh_dev, int32_t n_bins, void* stream) {
  if (!h) return SG_E_INVALID;
  if (!thresh_dev || n_bins != h->F) FAIL(h, SG_E_INVALID, "sg_set_noise_threshold_dev: n_bins must be %d", h->F);
  h->t2_ready = false;
  HIPCHK(h, hipMemcpyAsync(h->thresh.p, thresh_dev, (size_t)h->F * sizeof(double), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
  h->has_thresh = true;
  return SG_OK;
}

extern "C" int sg_process_chunks(sg_handle* h, const void* in_dev, int in_dtype, void* out_dev, int out_dtype,
                                 int64_t C, int64_t N, int64_t in_stride, int64_t out_stride,
                                 int64_t start_frame, int64_t end_frame, int32_t chunked, int64_t halo_left,
                                 int64_t halo_right, void* stream) {
  if (!h) return SG_E_INVALID;
  if (h->p.variant != SG_VARIANT_S) FAIL(h, SG_E_INVALID, "sg_process_chunks is a variant-S entry point");
  if (!in_dev || !out_dev || !dtype_ok(in_dtype) || !dtype_ok(out_dtype) || C < 1 || N < 1)
    FAIL(h, SG_E_INVALID, "sg_process_chunks: bad argument");
  if (start_frame < 0 || end_frame > N || start_frame >= end_frame)
    FAIL(h, SG_E_INVALID, "sg_process_chunks: bad frame range [%lld, %lld)", (long long)start_frame,
         (long long)end_frame);
  const int64_t cs = h->p.chunk_size, pad = h->p.padding;
  if (halo_left < 0 || halo_right < 0) FAIL(h, SG_E_INVALID, "sg_process_chunks: negative halo");
  View v{};
  v.x = in_dev; v.dtype = in_dtype; v.stride = in_stride; v.N = N; v.pad = pad;
  v.lo = -halo_left; v.hi = N + halo_right;
  OutMap om{};
  om.out = out_dev; om.dtype = out_dtype; om.stride = out_stride; om.g0 = start_frame;
  om.g_lo = start_frame; om.g_hi = end_frame;
  int64_t units;
  if (chunked) {
    // base.py:175-216: chunks ich1..ich2, each filtered over [i*cs - pad, (i+1)*cs + pad)
    // only the chunks that overlap [start_frame, end_frame) become units
    int64_t ich1 = start_frame / cs, ich2 = (end_frame - 1) / cs;
    int64_t n_chunks = ich2 - ich1 + 1;
    v.cs = cs; v.Lp = cs + 2 * pad; v.n_chunks = (int32_t)n_chunks; v.c0 = ich1;
    om.p0 = pad; om.p1 = pad + cs; om.g_step = cs;
    units = C * n_chunks;
  } else {
    // base.py:222: one window [-pad, end_frame + pad) -- start_frame is ignored by the reference
    v.cs = 0; v.Lp = end_frame + 2 * pad; v.n_chunks = 1;  // _read_chunk may read past end_frame (base.py:136-141)
    om.p0 = pad; om.p1 = pad + end_frame; om.g_step = 0;
    om.g_lo = 0; om.g0 = 0;
    units = C;
  }
  return run_S(h, v, units, om, (hipStream_t)stream);
}

extern "C" int sg_filter_padded(sg_handle* h, const void* chunk_dev, int in_dtype, void* out_dev, int out_dtype,
                                int64_t C, int64_t Lp, int64_t in_stride, int64_t out_stride, void* stream) {
  if (!h) return SG_E_INVALID;
  if (h->p.variant != SG_VARIANT_S) FAIL(h, SG_E_INVALID, "sg_filter_padded is a variant-S entry point");
  if (!chunk_dev || !out_dev || !dtype_ok(in_dtype) || !dtype_ok(out_dtype) || C < 1 || Lp < 1)
    FAIL(h, SG_E_INVALID, "sg_filter_padded: bad argument");
  View v{};
  v.x = chunk_dev; v.dtype = in_dtype; v.stride = in_stride; v.N = Lp; v.lo = 0; v.hi = Lp; v.cs = 0; v.pad = 0; v.Lp = Lp; v.n_chunks = 1;
  OutMap om{};
  om.out = out_dev; om.dtype = out_dtype; om.stride = out_stride;
  om.p0 = 0; om.p1 = Lp; om.g_step = 0; om.g0 = 0; om.g_lo = 0; om.g_hi = Lp;
  return run_S(h, v, C, om, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// variant T
// ------------------------------------------------------------------------------------------
extern "C" int sg_process_batch(sg_handle* h, const void* x_dev, int dtype, int64_t B, int64_t L, int64_t x_stride,
                                const void* xn_dev, int64_t Bn, int64_t Ln, int64_t xn_stride, void* out_dev,
                                int out_dtype, int64_t out_stride, float* mask_out_dev, void* stream) {
  if (!h) return SG_E_INVALID;
  if (h->p.variant != SG_VARIANT_T) FAIL(h, SG_E_INVALID, "sg_process_batch is a variant-T entry point");
  if (!x_dev || !out_dev || !dtype_ok(dtype) || !dtype_ok(out_dtype) || B < 1)
    FAIL(h, SG_E_INVALID, "sg_process_batch: bad argument");
  if (L < 2 * (int64_t)h->W) FAIL(h, SG_E_INVALID, "x must be bigger than %d", 2 * h->W);  // torchgate.py:215-216
  if (xn_dev) {
    if (Ln < 2 * (int64_t)h->W) FAIL(h, SG_E_INVALID, "xn must be bigger than %d", 2 * h->W);  // :219-220
    if (Bn != 1 && Bn != B) FAIL(h, SG_E_INVALID, "xn rows (%lld) must be 1 or the batch size", (long long)Bn);
  }
  hipStream_t st = (hipStream_t)stream;
  h->dbg_rg = false;
  View v{};
  v.x = x_dev; v.dtype = dtype; v.stride = x_stride; v.N = L; v.lo = 0; v.hi = L; v.cs = 0; v.pad = 0; v.Lp = L; v.n_chunks = 1;
  Geom g = make_geom(h, L);
  OutMap om{};
  om.out = out_dev; om.dtype = out_dtype; om.stride = out_stride;
  om.p0 = 0; om.p1 = g.Lout; om.g_step = 0; om.g0 = 0; om.g_lo = 0; om.g_hi = g.Lout;
  View vn{};
  Geom gn{};
  if (xn_dev && h->p.stationary) {
    vn.x = xn_dev; vn.dtype = dtype; vn.stride = xn_stride; vn.N = Ln; vn.lo = 0; vn.hi = Ln; vn.cs = 0; vn.pad = 0; vn.Lp = Ln; vn.n_chunks = 1;
    gn = make_geom(h, Ln);
  }
  // size the workspace for the larger of the two geometries
  Geom gbig = (xn_dev && h->p.stationary && gn.T > g.T) ? gn : g;
  int64_t ub = units_per_batch(h, gbig, B);
  int rc = ensure_ws(h, gbig, ub);
  if (rc) return rc;
  double* thr = (double*)h->thr_rows.p;
  if (xn_dev && h->p.stationary && Bn == 1) {
    vn.unit0 = 0;
    if ((rc = stage_stats(h, vn, gn, 1, (double*)h->thresh.p, st))) return rc;
  }
  for (int64_t u0 = 0; u0 < B; u0 += ub) {
    int64_t nb = std::min(ub, B - u0);
    v.unit0 = u0;
    if (h->p.stationary) {
      const double* th;
      int64_t ustride;
      if (xn_dev && Bn == 1) {
        th = (const double*)h->thresh.p; ustride = 0;
      } else if (xn_dev) {
        vn.unit0 = u0;
        if ((rc = stage_stats(h, vn, gn, nb, thr, st))) return rc;
        th = thr; ustride = g.FS;
      } else {
        th = nullptr; ustride = g.FS;
      }
      if (!xn_dev && rowgate_ok(h, g, B)) {
        // one kernel per call: a workgroup per row (rowgate.hpp)
        v.unit0 = 0;
        View vr = v;
        vr.unit0 = u0;
        if ((rc = stage_row_gate(h, vr, g, nb, om, mask_out_dev ? mask_out_dev + (size_t)u0 * g.T * g.FS : nullptr, st)))
          return rc;
        h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = false; h->dbg_fused = true; h->dbg_fast = true;
        h->dbg_has_raw = false; h->dbg_xbits = false; h->dbg_rg = true;
        h->dbg_db = 0; h->dbg_de = g.T;
        continue;
      }
      // default geometry, full reduction: decisions as bits -> exact integer smoothing -> uint16 sums
      // read by the fused apply kernel (same stages as the variant-S fused path)
      const bool bits_path = h->fast_ok && !h->force_nofast && !h->force_unfused && h->p.prop_decrease == 1.0 &&
                             h->ktot <= 65535 && (!h->p.smooth_mask || h->p.n_grad_time <= 96);
      // short rows: statistics + constants + decisions of a (row, 64 bands) tile in one kernel
      const size_t tile_bytes = (size_t)g.T * 64 * sizeof(double);
      const bool row_fused = bits_path && tile_bytes <= 64 * 1024;
      if (row_fused) {
        ProfScope ps(h, SG_STAGE_STFT_POWER, st);
        HIPCHK(h, stft_any<double>(h, v, g, nb, (double*)h->P.p, nullptr, nullptr, 1.0, st));
      } else {
        if ((rc = stage_power(h, v, g, nb, st))) return rc;
        if (!th) {
          if ((rc = stage_colstats(h, g, nb, thr, st))) return rc;
          th = thr;
        }
      }
      if (bits_path) {
        const int wpr = (g.F + 63) / 64;
        if ((rc = ensure(h, h->bits, (size_t)nb * g.T * wpr * 8))) return rc;
        if ((rc = ensure(h, h->K16, (size_t)nb * g.T * g.FS * 2))) return rc;
        if ((rc = ensure(h, h->T2, (size_t)nb * g.FS * 8))) return rc;
        if (row_fused) {
          ProfScope ps(h, SG_STAGE_DECIDE, st);
          hipLaunchKernelGGL(k_row_decide, dim3((unsigned)wpr, (unsigned)nb), dim3(64 * STAT_TG), tile_bytes, st,
                             (const double*)h->P.p, g, th, ustride, h->mag_scale, h->p.top_db, h->p.n_std_thresh,
                             h->p.ddof, (double*)h->pmax.p, th ? nullptr : thr, (unsigned long long*)h->bits.p,
                             wpr, db_fast_consts(h));
          HIPCHK(h, hipGetLastError());
        } else {
          ProfScope ps(h, SG_STAGE_DECIDE, st);
          // compare constants in the power domain per (row, band), then a pure compare per cell
          hipLaunchKernelGGL(k_t2_rows, dim3(grid_1d(nb * g.FS, 256)), dim3(256), 0, st, th, ustride,
                             (const double*)h->pmax.p, g, h->mag_scale, h->p.top_db, (double*)h->T2.p, nb);
          HIPCHK(h, hipGetLastError());
          hipLaunchKernelGGL(k_decide_bits_t2, dim3(grid_1d(nb * g.T * wpr * 64, 256)), dim3(256), 0, st,
                             (const double*)h->P.p, g, (const double*)h->T2.p, (unsigned long long*)h->bits.p, wpr,
                             nb);
          HIPCHK(h, hipGetLastError());
        }
        if ((rc = stage_smooth_bits(h, g, nb, true, 0, g.T, st))) return rc;
        if (mask_out_dev) {  // float mask (natural bin order) for the backward pass
          hipLaunchKernelGGL(k_k16_to_mask_perm, dim3(grid_1d(nb * g.T * g.FS, 256)), dim3(256), 0, st,
                             (const unsigned short*)h->K16.p, g, 1.0f / (float)h->ktot,
                             mask_out_dev + (size_t)u0 * g.T * g.FS, nb);
          HIPCHK(h, hipGetLastError());
        }
        if ((rc = stage_apply_fast(h, v, g, nb, om, nullptr, 1, st))) return rc;
        h->dbg_units = nb; h->dbg_T = g.T; h->dbg_has_P = true; h->dbg_fused = true; h->dbg_fast = true;
        h->dbg_db = 0; h->dbg_de = g.T;
        continue;
      }
      if ((rc = stage_decide(h, g, nb, th, ustride, st))) return rc;
    } else if (box_mask_ok(h)) {
      if ((rc = stage_box_mask(h, v, g, nb, st))) return rc;
    } else {
      if ((rc = stage_nonstat_raw(h, v, g, nb, st))) return rc;
    }
    const bool geom_fast = h->fast_ok && !h->force_nofast;
    if (h->p.stationary || !box_mask_ok(h))
      if ((rc = stage_smooth(h, g, nb, st))) return rc;
    if (mask_out_dev)
      HIPCHK(h, hipMemcpyAsync(mask_out_dev + (size_t)u0 * g.T * g.FS, h->M.p, (size_t)nb * g.T * g.FS * 4,
                               hipMemcpyDeviceToDevice, st));
    if (geom_fast) {
      if ((rc = stage_apply_fast(h, v, g, nb, om, (const float*)h->M.p, 1, st))) return rc;
    } else if (h->fast25_ok && !h->force_nofast) {
      if ((rc = stage_apply256(h, v, g, nb, om, (const float*)h->M.p, 1, st))) return rc;
    } else if (h->fast5_ok && !h->force_nofast) {
      if ((rc = stage_apply512(h, v, g, nb, om, (const float*)h->M.p, 1, st))) return rc;
    } else if (h->fast20_ok && !h->force_nofast) {
      if ((rc = stage_apply2048(h, v, g, nb, om, (const float*)h->M.p, 1, st))) return rc;
    } else {
      if ((rc = stage_apply_ola(h, v, g, nb, (const float*)h->M.p, om, 1, st))) return rc;
    }
    h->dbg_units = nb;
    h->dbg_T = g.T;
    h->dbg_has_P = h->p.stationary != 0;
    h->dbg_has_raw = h->p.stationary || !box_mask_ok(h);
    h->dbg_fused = false;
    h->dbg_fast = geom_fast;
  }
  return SG_OK;
}

// Adjoint of y = D^-1 OLA( Ws irfft( M .* rfft( Wa frames(x) ) ) ) with M fixed:
//   g_x = frames^T( Wa irfft( M .* rfft( Ws frames( D^-1 g_y ) ) ) )
// (irfft(M .* rfft(.)) is self-adjoint for a real mask; Wa and Ws are the same window up to the
// 1/N scale), i.e. the forward kernels run on g_y / env with an un-normalised overlap-add.
extern "C" int sg_process_batch_backward(sg_handle* h, const void* grad_out_dev, int dtype, int64_t B, int64_t L,
                                         int64_t go_stride, const float* mask_dev, void* grad_x_dev,
                                         int64_t gx_stride, void* stream) {
  if (!h) return SG_E_INVALID;
  if (h->p.variant != SG_VARIANT_T) FAIL(h, SG_E_INVALID, "sg_process_batch_backward is a variant-T entry point");
  if (!grad_out_dev || !grad_x_dev || !mask_dev || B < 1 || (dtype != SG_F32 && dtype != SG_F64))
    FAIL(h, SG_E_INVALID, "sg_process_batch_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  Geom g = make_geom(h, L);
  const int64_t Lq = g.Lout;
  int rc;
  if (h->fast_ok && !h->force_nofast && h->rowgate_mode != 1 && g.F == 513 && g.T >= 1 && g.T <= fast::RG_FRAMES) {
    // rows of at most 64 frames: the whole backward of a row in one workgroup (rowbwd.hpp) -- no grad_out / envelope copy,
    // no tiles, no hand-offs
    ProfScope ps(h, SG_STAGE_APPLY_FAST, st);
    fast::RowBwdArgs A;
    View v{};
    v.x = grad_out_dev; v.dtype = dtype; v.stride = go_stride; v.N = Lq; v.lo = 0; v.hi = Lq; v.cs = 0; v.pad = 0; v.Lp = Lq;
    v.n_chunks = 1; v.unit0 = 0;
    A.view = v;
    A.g = g;
    A.g.Lout = L;   // the adjoint scatters back onto all L input samples
    OutMap om{};
    om.out = grad_x_dev; om.dtype = dtype; om.stride = gx_stride;
    om.p0 = 0; om.p1 = L; om.g_step = 0; om.g0 = 0; om.g_lo = 0; om.g_hi = L;
    A.om = om;
    A.win = (const float*)h->wa32.p;
    A.wsq = (const float*)h->wsq32.p;
    A.invn = (const float*)h->invn.p;
    A.tw512 = (const fast::cf*)h->tw512.p;
    A.tw1024 = (const fast::cf*)h->tw32.p;
    A.mask = mask_dev;
    A.kscale = (float)(1.0 / 512.0);
    const size_t lds = fast::rowbwd_lds_bytes();
    HIPCHK(h, set_lds(reinterpret_cast<const void*>(fast::k_row_backward), lds));
    hipLaunchKernelGGL(fast::k_row_backward, dim3((unsigned)B), dim3(1024), lds, st, A);
    HIPCHK(h, hipGetLastError());
    return SG_OK;
  }
  int64_t ub = units_per_batch(h, g, B);
  rc = ensure_ws(h, g, ub);
  if (rc) return rc;
  if ((rc = ensure(h, h->yn, (size_t)ub * Lq * sizeof(float)))) return rc;
  for (int64_t u0 = 0; u0 < B; u0 += ub) {
    const int64_t nb = std::min(ub, B - u0);
    {
      dim3 grid((unsigned)((Lq + 255) / 256), (unsigned)nb);
      const char* src = (const char*)grad_out_dev + (size_t)u0 * go_stride * (dtype == SG_F64 ? 8 : 4);
      hipLaunchKernelGGL(k_env_scale, grid, dim3(256), 0, st, (const void*)src, dtype, go_stride, g,
                         (const float*)h->wsq32.p, (float*)h->yn.p, Lq);
      HIPCHK(h, hipGetLastError());
    }
    View v{};
    v.x = h->yn.p; v.dtype = SG_F32; v.stride = Lq; v.N = Lq; v.lo = 0; v.hi = Lq; v.cs = 0; v.pad = 0; v.Lp = Lq;
    v.n_chunks = 1; v.unit0 = 0;
    Geom gb = g;
    gb.Lout = L;  // the adjoint scatters back onto all L input samples
    OutMap om{};
    om.out = (char*)grad_x_dev + (size_t)u0 * gx_stride * (dtype == SG_F64 ? 8 : 4);
    om.dtype = dtype; om.stride = gx_stride;
    om.p0 = 0; om.p1 = L; om.g_step = 0; om.g0 = 0; om.g_lo = 0; om.g_hi = L;
    const float* mk = mask_dev + (size_t)u0 * g.T * g.FS;
    if (h->fast_ok && !h->force_nofast) {
      if ((rc = stage_apply_fast(h, v, gb, nb, om, mk, 0, st))) return rc;
    } else if (h->fast25_ok && !h->force_nofast) {
      if ((rc = stage_apply256(h, v, gb, nb, om, mk, 0, st))) return rc;
    } else if (h->fast5_ok && !h->force_nofast) {
      if ((rc = stage_apply512(h, v, gb, nb, om, mk, 0, st))) return rc;
    } else if (h->fast20_ok && !h->force_nofast) {
      if ((rc = stage_apply2048(h, v, gb, nb, om, mk, 0, st))) return rc;
    } else {
      if ((rc = stage_apply_ola(h, v, gb, nb, mk, om, 0, st))) return rc;
    }
  }
  return SG_OK;
}

// ------------------------------------------------------------------------------------------
// stage taps
// ------------------------------------------------------------------------------------------
extern "C" int sg_stft(sg_handle* h, const void* x_dev, int dtype, int64_t B, int64_t L, int64_t stride,
                       double* z_dev, void* stream) {
  if (!h) return SG_E_INVALID;
  if (!x_dev || !z_dev || !dtype_ok(dtype) || B < 1 || L < h->W) FAIL(h, SG_E_INVALID, "sg_stft: bad argument");
  View v{};
  v.x = x_dev; v.dtype = dtype; v.stride = stride; v.N = L; v.lo = 0; v.hi = L; v.cs = 0; v.pad = 0; v.Lp = L; v.n_chunks = 1; v.unit0 = 0;
  Geom g = make_geom(h, L);
  HIPCHK(h, stft_any<double>(h, v, g, B, nullptr, nullptr, z_dev, h->mag_scale, (hipStream_t)stream));
  return SG_OK;
}

extern "C" int sg_set_option(sg_handle* h, int32_t option, int64_t value) {
  if (!h) return SG_E_INVALID;
  switch (option) {
    case SG_OPT_FORCE_UNFUSED: h->force_unfused = value != 0; return SG_OK;
    case SG_OPT_FORCE_NOFAST: h->force_nofast = value != 0; return SG_OK;
    case SG_OPT_FORCE_F64_DECIDE: h->force_f64_decide = value != 0; return SG_OK;
    case SG_OPT_FORCE_NOSEAM: h->force_noseam = value != 0; return SG_OK;
    case SG_OPT_FORCE_NOLEAN: h->force_nolean = value != 0; return SG_OK;
    case SG_OPT_FORCE_SPLIT: h->force_split = value != 0; return SG_OK;
    case SG_OPT_FAST_INTEGER: h->fast_integer = value != 0; return SG_OK;
    case SG_OPT_FORCE_EXACT: h->force_exact = value != 0; return SG_OK;
    case SG_OPT_FORCE_NOROWGATE:
      if (value < 0 || value > 2) FAIL(h, SG_E_INVALID, "SG_OPT_FORCE_NOROWGATE: 0 (auto), 1 (never) or 2 (always)");
      h->rowgate_mode = (int)value;
      return SG_OK;
    case SG_OPT_ROWGATE_TAP: h->rg_tap = value != 0; return SG_OK;
    case SG_OPT_ROWGATE_SHAPE:
      if (value != 8 && value != 16) FAIL(h, SG_E_INVALID, "SG_OPT_ROWGATE_SHAPE: 8 or 16 waves");
      h->rg_shape = (int)value;
      return SG_OK;
    case SG_OPT_INJECT_HANDOFF_FAULT: h->inject_fault = (unsigned)value & 63u; return SG_OK;
    case SG_OPT_TILE_ORDER:
      if (value < 0 || value > 2) FAIL(h, SG_E_INVALID, "SG_OPT_TILE_ORDER: 0 (persistent workgroups), 1 (block index) or 2 (one ticket per workgroup)");
      h->tile_order = (int)value;
      return SG_OK;
    case SG_OPT_EXACT_MATERIALISED: h->exact_materialised = value != 0; return SG_OK;
    case SG_OPT_FLOOR_TEST:
      if (value < 0 || value > 2) FAIL(h, SG_E_INVALID, "SG_OPT_FLOOR_TEST: 0 (predicted), 1 (a priori) or 2 (in the gate kernel)");
      h->floor_test = (int)value;
      return SG_OK;
  }
  FAIL(h, SG_E_INVALID, "sg_set_option: unknown option %d", option);
}

extern "C" int sg_get_option(const sg_handle* h, int32_t option, int64_t* value) {
  if (!h || !value) return SG_E_INVALID;
  switch (option) {
    case SG_OPT_FORCE_UNFUSED: *value = h->force_unfused; return SG_OK;
    case SG_OPT_FORCE_NOFAST: *value = h->force_nofast; return SG_OK;
    case SG_OPT_FORCE_F64_DECIDE: *value = h->force_f64_decide; return SG_OK;
    case SG_OPT_FORCE_NOSEAM: *value = h->force_noseam; return SG_OK;
    case SG_OPT_FORCE_NOLEAN: *value = h->force_nolean; return SG_OK;
    case SG_OPT_FORCE_SPLIT: *value = h->force_split; return SG_OK;
    case SG_OPT_FAST_INTEGER: *value = h->fast_integer; return SG_OK;
    case SG_OPT_FORCE_EXACT: *value = h->force_exact; return SG_OK;
    case SG_OPT_FORCE_NOROWGATE: *value = h->rowgate_mode; return SG_OK;
    case SG_OPT_ROWGATE_TAP: *value = h->rg_tap; return SG_OK;
    case SG_OPT_ROWGATE_SHAPE: *value = h->rg_shape; return SG_OK;
    case SG_OPT_INJECT_HANDOFF_FAULT: *value = h->inject_fault; return SG_OK;
    case SG_OPT_TILE_ORDER: *value = h->tile_order; return SG_OK;
    case SG_OPT_EXACT_MATERIALISED: *value = h->exact_materialised; return SG_OK;
    case SG_OPT_FLOOR_TEST: *value = h->floor_test; return SG_OK;
  }
  return SG_E_INVALID;   // (no message: the handle is const here)
}

extern "C" int sg_profile_enable(sg_handle* h, int32_t on) {
  if (!h) return SG_E_INVALID;
  h->prof_on = on != 0;
  return SG_OK;
}

extern "C" int sg_profile_select(sg_handle* h, int64_t stage_mask) {
  if (!h) return SG_E_INVALID;
  h->prof_mask = stage_mask == 0 ? ~0ull : (uint64_t)stage_mask;
  return SG_OK;
}

extern "C" int sg_profile_read(sg_handle* h, double* ms, int64_t* counts, int32_t n_stages, int32_t reset) {
  if (!h) return SG_E_INVALID;
  if (n_stages != SG_N_STAGES) FAIL(h, SG_E_INVALID, "sg_profile_read: n_stages must be %d", SG_N_STAGES);
  for (auto& r : h->prof_live) {
    HIPCHK(h, hipEventSynchronize(r.b));
    float t = 0.f;
    HIPCHK(h, hipEventElapsedTime(&t, r.a, r.b));
    h->prof_ms[r.stage] += (double)t;
    h->prof_cnt[r.stage] += 1;
    h->prof_pool.push_back(r.a);
    h->prof_pool.push_back(r.b);
  }
  h->prof_live.clear();
  for (int i = 0; i < SG_N_STAGES; ++i) {
    if (ms) ms[i] = h->prof_ms[i];
    if (counts) counts[i] = h->prof_cnt[i];
    if (reset) { h->prof_ms[i] = 0; h->prof_cnt[i] = 0; }
  }
  return SG_OK;
}

extern "C" const char* sg_stage_name(int32_t stage) {
  static const char* names[SG_N_STAGES] = {"k_channel_mean", "k_stft<double> (power)", "k_colmax", "k_colstats",
                                           "k_decide", "k_mag_fast* / k_stft<float> (magnitude)",
                                           "k_box_mask / k_iir_sigmoid / k_boxcar_sigmoid (non-stationary mask, other paths)",
                                           "mask smoothing (k_smooth_bits2 / k_smooth_tiled / k_smooth_f+k_smooth_t)",
                                           "k_apply_istft", "k_ola",
                                           "noise statistics (all kernels)", "k_unit_absmax+k_prep_thresh",
                                           "k_stft_bits<max> (floor pre-pass)", "k_stft_bits<decide>",
                                           "k_apply_fast (fft+mask+ifft+ola)",
                                           "k_decide_fast (f32 stft + exact f64 refine)",
                                           "k_gate_onepass (fft+decide+smooth+mask+ifft+ola)",
                                           "k_row_gate (fft+row stats+decide+smooth+mask+ifft+ola)",
                                           "k_iir_chain_par (tile carries; serial: k_iir_part / k_iir_comb + k_iir_chain)",
                                           "k_iir_mask<nt> (recurrence+sigmoid+smoothing)"};
  return (stage >= 0 && stage < SG_N_STAGES) ? names[stage] : "?";
}

extern "C" int sg_debug_dims(const sg_handle* h, int64_t dims[3]) {
  if (!h || !dims) return SG_E_INVALID;
  dims[0] = h->dbg_units; dims[1] = h->dbg_T; dims[2] = h->FS;
  return SG_OK;
}

extern "C" int sg_debug_range(const sg_handle* h, int64_t range[2]) {
  if (!h || !range) return SG_E_INVALID;
  range[0] = h->dbg_db; range[1] = h->dbg_de;
  return SG_OK;
}

extern "C" int sg_debug_counter(sg_handle* h, int32_t which, int64_t* value, void* stream) {
  if (!h || !value) return SG_E_INVALID;
  if (which == 1 || which == 2) {   // host counters: one-pass gate calls with the in-kernel (1) / a-priori (2) floor test
    *value = which == 1 ? h->n_floor_lazy : h->n_floor_apriori;
    return SG_OK;
  }
  if (which == 3) {   // launch epoch of the last gate call in which a chunk's floor test fired / flag was set (0: never)
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    *value = h->err_host ? (int64_t)h->err_host[1] : 0;
    return SG_OK;
  }
#if OP_TRACE
  if (which == 8) {
    // (diagnosis) after a call that lost a hand-off: the tiles of the last first launch that did not run to the end -- who took
    // their ticket (workgroup, iteration) and the last phase each of their four waves stamped -- on stderr; *value = their number
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    std::vector<unsigned> t(g_trace_tiles * 64);
    HIPCHK(h, hipMemcpy(t.data(), g_trace_dev, t.size() * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    auto last_phase = [&](size_t tk, int w) { int lp = -1; for (int k = 0; k < 14; ++k) if (t[(tk * 4 + w) * 16 + k]) lp = k; return lp; };
    for (size_t tk = 0; tk < g_trace_tiles; ++tk) {
      const int jt = (int)(tk % g_trace_ntt) - 1;
      const bool halo = jt < 0 || jt >= g_trace_ntt - 2;
      bool ok = true;
      for (int w = 0; w < 4; ++w) ok = ok && (halo ? last_phase(tk, w) >= 5 : t[(tk * 4 + w) * 16 + 15] == 1u);
      if (ok) continue;
      if (++bad > 8) continue;
      fprintf(stderr, "[trace] ticket %zu (unit %zu tile %d%s):", tk, tk / g_trace_ntt, jt, halo ? " halo" : "");
      for (int w = 0; w < 4; ++w) {
        const unsigned who = t[(tk * 4 + w) * 16 + 14];
        fprintf(stderr, "  w%d wg %d iter %u last phase %d", w, (int)(who & 0xffffu) - 1, who >> 16, last_phase(tk, w));
      }
      fprintf(stderr, "\n");
    }
    *value = bad;
    return SG_OK;
  }
#endif
#if OP_WHO
  if (which == 16) {   // (diagnosis) the tickets around *value (in): who drew them while on which tile, who started them, on stderr
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    std::vector<unsigned> w(4096 * 4);
    HIPCHK(h, hipMemcpy(w.data(), g_who_dev, w.size() * 4, hipMemcpyDeviceToHost));
    const long long T = *value;
    if (T < 0) {   // everything, one line per ticket: ticket, drawer workgroup, its iteration, the tile it was on, starter workgroup, its iteration, past-bits flag
      for (long long t = 0; t < 4096; ++t)
        if (w[t * 4] | w[t * 4 + 2])
          fprintf(stderr, "[whoall] %lld %d %u %u %d %u %u\n", t, (int)(w[t * 4] & 0xffffu) - 1, w[t * 4] >> 16, w[t * 4 + 1], (int)(w[t * 4 + 2] & 0xffffu) - 1, w[t * 4 + 2] >> 16, w[t * 4 + 3]);
      return SG_OK;
    }
    for (long long t = std::max<long long>(0, T - 14); t <= std::min<long long>(4095, T + 3); ++t)
      fprintf(stderr, "[who] ticket %lld: drawn by wg %d (iteration %u) while on tile %u; started by wg %d (iteration %u); past its bits poll: %u\n", t,
              (int)(w[t * 4] & 0xffffu) - 1, w[t * 4] >> 16, w[t * 4 + 1], (int)(w[t * 4 + 2] & 0xffffu) - 1, w[t * 4 + 2] >> 16, w[t * 4 + 3]);
    return SG_OK;
  }
#endif
  if (which >= 4 && which <= 15) {   // development (-DOP_TILECOUNT=1 builds): tiles completed / sum of their tickets, persistent gate
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    *value = h->err_host ? (int64_t)h->err_host[which] : 0;
    return SG_OK;
  }
  if (which != 0) FAIL(h, SG_E_INVALID, "sg_debug_counter: unknown counter %d", which);
  *value = 0;
  if (!h->rg_count.p) return SG_OK;
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  unsigned v = 0;
  HIPCHK(h, hipMemcpy(&v, h->rg_count.p, sizeof(v), hipMemcpyDeviceToHost));
  *value = (int64_t)v;
  return SG_OK;
}

extern "C" int sg_debug_fetch(sg_handle* h, int32_t what, void* host, int64_t bytes, void* stream) {
  if (!h) return SG_E_INVALID;
  if (!host || h->dbg_units == 0) FAIL(h, SG_E_STATE, "sg_debug_fetch: nothing processed yet");
  size_t cells = (size_t)h->dbg_units * h->dbg_T * h->FS;
  const void* src;
  size_t need;
  switch (what) {
    case 0:
      if (h->dbg_fused) FAIL(h, SG_E_STATE, "fused path keeps the raw mask as bits: fetch field 3");
      if (!h->dbg_has_raw)
        FAIL(h, SG_E_STATE, "the one-kernel non-stationary mask does not materialise the raw mask: set SG_OPT_FORCE_UNFUSED");
      src = h->raw.p; need = cells * 4; break;
    case 3:
      if (!h->dbg_fused) FAIL(h, SG_E_STATE, "bit field only exists on the fused path");
      src = h->bits.p; need = (size_t)h->dbg_units * h->dbg_T * ((h->F + 63) / 64) * 8; break;
    case 1:
      if (h->dbg_fast && h->dbg_fused)
        FAIL(h, SG_E_STATE, "fast path keeps the smoothed mask as uint16 counts in the apply kernel's lane order");
      if (h->dbg_fused && h->dbg_k16_only) {
        // the apply kernel read the K counts directly: expand them now (same kernel, same arithmetic as the materialising path).
        // (a debug entry point: whatever stream produced the counts has to be done first)
        HIPCHK(h, hipDeviceSynchronize());
        hipLaunchKernelGGL(k_k16_to_mask, dim3(grid_1d((int64_t)cells / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)h->K16.p, h->dbg_g, h->p.n_grad_freq, h->p.n_grad_time, 1.0f / (float)h->ktot,
                           (float)h->p.prop_decrease, 1, h->p.smooth_mask ? 1 : 0, (float*)h->M.p, (int64_t)h->dbg_units);
        HIPCHK(h, hipGetLastError());
      }
      src = h->M.p; need = cells * 4; break;
    case 2:
      if (!h->dbg_has_P) FAIL(h, SG_E_STATE, "power field only exists for stationary gates");
      src = h->P.p; need = cells * 8; break;
    case 4:
      if (!h->dbg_rg || !h->rg_tap) FAIL(h, SG_E_STATE, "row-gate power tile: set SG_OPT_ROWGATE_TAP and run TorchGate.forward");
      src = h->M.p; need = (size_t)h->dbg_units * 64 * fast::RG_PP * 4; break;
    default: FAIL(h, SG_E_INVALID, "sg_debug_fetch: unknown field %d", what);
  }
  if ((size_t)bytes != need) FAIL(h, SG_E_INVALID, "sg_debug_fetch: need %zu bytes, got %lld", need, (long long)bytes);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  { int rc = handoff_verdict(h); if (rc) return rc; }
  if (what == 3 && h->dbg_xbits) {
    // one-pass path: the bits live tile-blocked in the exchange buffer [unit][tile][16][9]; rearrange into
    // the natural [unit][T][wpr] layout (frames outside sg_debug_range stay zero)
    const int wpr = (h->F + 63) / 64;
    const int TW = h->dbg_twords ? h->dbg_twords : fast::OP_TILE_WORDS, XW = h->dbg_txw;
    const size_t words = (size_t)h->dbg_units * h->dbg_ntt * TW;
    std::vector<unsigned long long> tmp(words);
    HIPCHK(h, hipMemcpy(tmp.data(), h->xbits.p, words * 8, hipMemcpyDeviceToHost));
    unsigned long long* dst = (unsigned long long*)host;
    std::memset(dst, 0, need);
    for (int64_t u = 0; u < h->dbg_units; ++u)
      for (int64_t j = 0; j < h->dbg_ntt; ++j)
        for (int i = 0; i < h->dbg_trows; ++i) {
          const int64_t t = h->dbg_tf0 + (j - 1) * h->dbg_tstep + i;
          if (t < 0 || t >= h->dbg_T) continue;
          for (int w = 0; w < wpr; ++w)
          {
            const unsigned long long* gr = &tmp[((size_t)u * h->dbg_ntt + j) * TW + (i * XW + w) * 2];
            dst[(u * h->dbg_T + t) * wpr + w] = (gr[0] & 0xffffffffull) | (gr[1] << 32);
          }
        }
    return SG_OK;
  }
  HIPCHK(h, hipMemcpy(host, src, need, hipMemcpyDeviceToHost));
  return SG_OK;
}
