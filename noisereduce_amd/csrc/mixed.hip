// Instantiations and launchers of the mixed-radix kernels (mixed_kernels.hpp): a translation unit of its own so that
// __graft_entry__.build() compiles it beside api.hip (four team sizes x two precisions x eight radix bodies each).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include "mixed_kernels.hpp"

namespace sg {

bool mr_make_plan(int N, MrPlan* pl) {
  if (N < 2 || N > 2048) return false;
  pl->N = N;
  pl->np = 0;
  int r = N;
  auto take = [&](int R) {
    while (r % R == 0 && pl->np < MR_MAXP) { pl->R[pl->np++] = (unsigned char)R; r /= R; }
  };
  take(8); take(4); take(2); take(5); take(3); take(7); take(11); take(13);
  if (r != 1) return false;
  int S = 1, off = 0;
  for (int p = 0; p < pl->np; ++p) {
    const int R = pl->R[p];
    pl->toff[p] = off;
    if (p + 1 < pl->np) off += (N / (R * S)) * (R - 1);
    S *= R;
  }
  for (int p = pl->np; p < MR_MAXP; ++p) { pl->R[p] = 1; pl->toff[p] = off; }
  pl->ptotal = off;
  return true;
}

void mr_pass_tables(const MrPlan& pl, double* out) {
  typedef long double ld;
  const ld PI2 = 2.0L * 3.14159265358979323846264338327950288L;
  int S = 1;
  for (int p = 0; p + 1 < pl.np; ++p) {
    const int R = pl.R[p], groups = pl.N / (R * S);
    for (int gi = 0; gi < groups; ++gi)
      for (int k = 1; k < R; ++k) {
        const long long e = ((long long)gi * S * k) % pl.N;
        const ld a = -PI2 * (ld)e / (ld)pl.N;
        const size_t i = (size_t)pl.toff[p] + (size_t)gi * (R - 1) + (k - 1);
        out[2 * i] = (double)cosl(a);
        out[2 * i + 1] = (double)sinl(a);
      }
    S *= R;
  }
}

namespace {
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size)  (api.hip: set_lds)
hipError_t mr_set_lds(const void* kern, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find({dev, kern});
  if (it != done.end() && it->second >= bytes) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done[{dev, kern}] = bytes;
  return e;
}

// Threads per frame (a "team") and frames in flight per workgroup.  A pass of radix R has N / R butterflies; a team of NT
// lanes takes ceil(N / R / NT) sweeps of NT lanes over them: the team size with the fewest padded lane-slots over all
// passes wins (N = 200 = 8 5 5: 25 + 40 + 40 butterflies are 192 slots of a 64-lane team, 128 of a 16-lane one), smaller
// teams on ties (more frames in flight per wavefront).  `sweeps`: a kernel that walks the N + 1 bins with a compile-time
// bound needs sweeps * NT >= N + 1 (0: no bound); a frame's two ping-pong buffers must leave room for at least one team.
struct MrShape { int NT, teams; size_t lds; };
MrShape mr_team(const MrPlan& pl, size_t cx_bytes, size_t pad_len, int sweeps, size_t extra_bytes, bool wave_min) {
  const int N = pl.N;
  const size_t budget = 96 * 1024;
  // one wavefront per frame up to N = 1024 (float32) / 512 (float64), the whole workgroup on one frame beyond; short frames
  // share a wavefront (16 lanes per frame up to N = 128, 32 up to 256: fft_wave.hpp's teams) -- measured, not derived: a
  // pass is a chain of LDS round trips, and what hides them is frames in flight, not lanes kept busy (round 6: 16-lane
  // teams at N = 500 / 768 / 1500 -- the fewest padded lane-slots -- ran 2 - 7 x SLOWER than one wavefront per frame)
  int NT = N <= 128 ? 16 : (N <= 256 ? 32 : ((cx_bytes == 16 ? N <= 512 : N <= 1024) ? 64 : 256));
  if (wave_min && NT < 64) NT = 64;
  while (sweeps && (long)sweeps * NT < N + 1 && NT < 256) NT = NT == 64 ? 256 : NT * 2;
  static const int f_nt = [] { const char* e = getenv("SG_MR_NT"); return e ? atoi(e) : 0; }();        // (experiments)
  static const int f_teams = [] { const char* e = getenv("SG_MR_TEAMS"); return e ? atoi(e) : 0; }();
  if (f_nt == 16 || f_nt == 32 || f_nt == 64 || f_nt == 256) {
    if (!(wave_min && f_nt < 64) && !(sweeps && (long)sweeps * f_nt < N + 1)) NT = f_nt;
  }
  MrShape s;
  s.NT = NT;
  s.teams = NT == 256 ? 1 : 256 / NT;
  if (f_teams > 0 && NT != 256) s.teams = std::min(f_teams, 1024 / NT);
  auto bytes = [&](int teams) { return (size_t)(N + pl.ptotal) * cx_bytes + (size_t)2 * teams * pad_len * cx_bytes + extra_bytes; };
  while (s.teams > 1 && bytes(s.teams) > budget) s.teams >>= 1;
  s.lds = bytes(s.teams);
  return s;
}
template <typename K, typename... Args>
hipError_t mr_go(K kern, const MrShape& s, const Geom& g, int64_t units, int fpw, hipStream_t st, Args... args) {
  if (s.lds > 65536) {
    hipError_t e = mr_set_lds(reinterpret_cast<const void*>(kern), s.lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((g.T + (int64_t)s.teams * fpw - 1) / ((int64_t)s.teams * fpw)), (unsigned)units);
  hipLaunchKernelGGL(kern, grid, dim3(s.teams * s.NT), s.lds, st, args..., fpw);
  return hipGetLastError();
}
// few units (the noise clip): one frame per team so that the grid still covers the chip
int mr_fpw(const MrShape& s, const Geom& g, int64_t units) {
  static const int f = [] { const char* e = getenv("SG_MR_FPW"); return e ? atoi(e) : 0; }();
  if (f > 0) return f;
  return units * ((g.T + s.teams * 4 - 1) / (s.teams * 4)) < 1024 ? 1 : 4;
}
}  // namespace

#define MR_BY_TEAM(S, CALL)            \
  switch ((S).NT) {                    \
    case 16: return CALL(16);          \
    case 32: return CALL(32);          \
    case 64: return CALL(64);          \
    default: return CALL(256);         \
  }

hipError_t mr_launch_stft32(const MrPlan& pl, const View& v, const Geom& g, int64_t units, const cx<float>* tw, const cx<float>* ptab,
                            const float* wfull, double* P, float* mag, double* z, double zscale, unsigned long long* pmax_bits,
                            hipStream_t st) {
  const MrShape s = mr_team(pl, sizeof(cx<float>), mlpn<float>(pl.N), MR_MAXM, 0, false);
  const int fpw = mr_fpw(s, g, units);
#define MR_CALL(NT_) mr_go(k_stft_mr<float, NT_>, s, g, units, fpw, st, v, g, pl, tw, ptab, wfull, P, mag, z, zscale, pmax_bits)
  MR_BY_TEAM(s, MR_CALL)
#undef MR_CALL
}

hipError_t mr_launch_stft64(const MrPlan& pl, const View& v, const Geom& g, int64_t units, const cx<double>* tw, const cx<double>* ptab,
                            const double* wfull, double* P, float* mag, double* z, double zscale, unsigned long long* pmax_bits,
                            hipStream_t st) {
  const MrShape s = mr_team(pl, sizeof(cx<double>), mlpn<double>(pl.N), MR_MAXM, 0, false);
  const int fpw = mr_fpw(s, g, units);
#define MR_CALL(NT_) mr_go(k_stft_mr<double, NT_>, s, g, units, fpw, st, v, g, pl, tw, ptab, wfull, P, mag, z, zscale, pmax_bits)
  MR_BY_TEAM(s, MR_CALL)
#undef MR_CALL
}

hipError_t mr_launch_bits(int mode, const MrPlan& pl, const View& v, const Geom& g, int64_t units, const cx<double>* tw,
                          const cx<double>* ptab, const double* wfull, const ThreshConsts& tc, double mag_scale, double top_db,
                          unsigned long long* pmax_bits, unsigned long long* bits, int wpr, hipStream_t st) {
  // (ballots are stored per hardware wavefront: teams of at least 64 lanes)
  const MrShape s = mr_team(pl, sizeof(cx<double>), mlpn<double>(pl.N), MR_MAXM, (size_t)(pl.N + 1) * sizeof(double), true);
  const int fpw = 4;
  if (mode == 0) {
    if (s.NT == 64) return mr_go(k_stft_bits_mr<0, 64>, s, g, units, fpw, st, v, g, pl, tw, ptab, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr);
    return mr_go(k_stft_bits_mr<0, 256>, s, g, units, fpw, st, v, g, pl, tw, ptab, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr);
  }
  if (s.NT == 64) return mr_go(k_stft_bits_mr<1, 64>, s, g, units, fpw, st, v, g, pl, tw, ptab, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr);
  return mr_go(k_stft_bits_mr<1, 256>, s, g, units, fpw, st, v, g, pl, tw, ptab, wfull, tc, mag_scale, top_db, pmax_bits, bits, wpr);
}

hipError_t mr_launch_decide(const MrPlan& pl, const View& v, const Geom& g, int64_t units, const cx<float>* tw32, const cx<float>* ptab32,
                            const float* win32, const cx<double>* tw64, const double* win64, const ThreshConsts& tc, double mag_scale, double top_db,
                            unsigned long long* bits, int wpr, hipStream_t st) {
  // (teams below a wavefront collect their bits in 9 words: 9 sweeps of 64 bins)
  MrShape s = mr_team(pl, sizeof(cx<float>), mlpn<float>(pl.N), 0, (size_t)(pl.N + 1 + 4) * sizeof(float), pl.N + 1 > 576);
  const int fpw = mr_fpw(s, g, units);
#define MR_CALL(NT_) mr_go(k_decide_mr<NT_>, s, g, units, fpw, st, v, g, pl, tw32, ptab32, win32, tw64, win64, tc, mag_scale, top_db, bits, wpr)
  MR_BY_TEAM(s, MR_CALL)
#undef MR_CALL
}

hipError_t mr_launch_apply(const MrPlan& pl, const View& v, const Geom& g, int64_t units, const cx<float>* tw32, const cx<float>* ptab32,
                           const float* wa, const float* ws, const float* M, float* seg, const unsigned short* K16, float kscale, hipStream_t st) {
  const MrShape s = mr_team(pl, sizeof(cx<float>), mlpn<float>(pl.N), 0, 0, false);
  const int fpw = mr_fpw(s, g, units);
#define MR_CALL(NT_) mr_go(k_apply_istft_mr<NT_>, s, g, units, fpw, st, v, g, pl, tw32, ptab32, wa, ws, M, seg, K16, kscale)
  MR_BY_TEAM(s, MR_CALL)
#undef MR_CALL
}

}  // namespace sg
