// Backward of TorchGate.forward at the default geometry in ONE kernel (round 5): one workgroup = one batch row of at most 64
// frames, the shape of k_row_gate (rowgate.hpp) without its statistics, decisions and smoothing.
//
// The gate is linear in x once the mask is fixed (torchgate.py:126,167 compute the mask under no_grad), so
//   grad_x = STFT^T ( M . ISTFT^T (grad_out) ),   ISTFT^T(g)[t] = window . (g / envelope)[frame t]
// i.e. the forward's own chain -- frames, forward transform, x mask, inverse transform, window, overlap-add -- applied to
// grad_out / envelope, without the final normalisation.  Until round 4 this ran as k_env_scale (grad_out / envelope through
// HBM) + k_apply_fast<float mask> on 16-frame tiles: a 63-frame row is five tiles, two of them on the checked per-sample
// path and all of them chained by the straddling-hop hand-off -- 0.10 ms for a pass with no statistics in it, as much as
// the whole forward.  Here the row is staged once (the envelope division rides in the staging), every wave transforms one
// quad of frames, and all 67 hops are combined inside the workgroup: no tiles, no hand-offs, no intermediate in HBM.
#pragma once
#include "rowgate.hpp"

namespace sg {
namespace fast {

struct RowBwdArgs {
  View view;            // grad_out rows (float32 or float64), Lp = the forward's output length
  Geom g;               // the forward's frames (T, padL); Lout = L: the adjoint scatters back onto all L input samples
  OutMap om;            // grad_x
  const float* win;     // analysis == synthesis window (1024)
  const float* wsq;     // window squared (1024)
  const float* invn;    // 1 / sum_q wsq[256 q + s]
  const cf* tw512;
  const cf* tw1024;
  const float* mask;    // [rows][T][FS] final mask of the forward (natural bin order)
  float kscale;         // 1 / 512
};

__host__ __device__ constexpr size_t rowbwd_lds_bytes() { return (size_t)FN * 8 + (size_t)RG_REGION + 1024 * 4 + 17 * 8 + 16; }

__global__ __launch_bounds__(1024, 1) void k_row_backward(RowBwdArgs A) {
  constexpr int WAVES = 16, THREADS = 1024, XP = 288;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* tw512 = reinterpret_cast<cf*>(smem);
  char* region = reinterpret_cast<char*>(tw512 + FN);
  float* swin = reinterpret_cast<float*>(region + RG_REGION);
  cf* s_tw1024 = reinterpret_cast<cf*>(swin + 1024);
  const Geom& G = A.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int T = (int)G.T;
  const int64_t row = A.view.unit0 + blockIdx.x;
  float* xs = reinterpret_cast<float*>(region);          // the row's (scaled) samples: hop h at xs + XP h
  cf* slices = reinterpret_cast<cf*>(region);            // 16 exchange slices: they overwrite the samples after the gather
  static_assert(RG_SPAN_BYTES_P <= RG_REGION && 16 * WAVE_CX_H * 8 <= RG_REGION, "LDS map");
  const int span = (T - 1) * 256 + 1024;
  const int64_t Lq = A.view.Lp;

  // ---- tables; the row of grad_out / envelope -> LDS (zero outside [0, Lq)) ---------------------------------------------
  {
    const int it = min(tid, FN - 1);
    const cf tw_v = A.tw512[(it >> 4) * (it & 15)];
    const float4 w4 = reinterpret_cast<const float4*>(A.win)[min(tid, 255)];
    const cf t10 = A.tw1024[min(tid, 16)];
    if (tid < FN) tw512[tid] = tw_v;
    if (tid < 256) reinterpret_cast<float4*>(swin)[tid] = w4;
    if (tid < 17) s_tw1024[tid] = t10;
  }
  {
    const float* sp32 = (const float*)A.view.x + row * A.view.stride;
    const bool vec = A.view.dtype == 0 && (reinterpret_cast<uintptr_t>(sp32) & 15) == 0 && Lq >= 8;
    for (int i4 = tid; i4 < span / 4; i4 += THREADS) {
      const int e = 4 * i4;
      const int64_t s = (int64_t)e - G.padL;         // position in grad_out (multiple of 4: padL = 512)
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s >= 0 && s < Lq) {
        if (vec && s + 4 <= Lq) {
          v4 = *reinterpret_cast<const float4*>(sp32 + s);
        } else {
          v4.x = (float)view_sample(A.view, row, 0, s);
          v4.y = (float)view_sample(A.view, row, 0, s + 1);
          v4.z = (float)view_sample(A.view, row, 0, s + 2);
          v4.w = (float)view_sample(A.view, row, 0, s + 3);
        }
        // envelope of the forward's overlap-add at these four positions: frames hq - 3 .. hq that exist (all of one hop)
        const int hq = e >> 8, sph = e & 255;
        if (hq >= 3 && hq <= T - 1) {
          const float4 n4 = *reinterpret_cast<const float4*>(&A.invn[sph]);
          v4.x *= n4.x; v4.y *= n4.y; v4.z *= n4.z; v4.w *= n4.w;
        } else {
          float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ti = hq - k;
            if (ti >= 0 && ti < T) {
              const float4 q4 = *reinterpret_cast<const float4*>(&A.wsq[256 * k + sph]);
              nrm.x += q4.x; nrm.y += q4.y; nrm.z += q4.z; nrm.w += q4.w;
            }
          }
          v4.x /= (nrm.x > 1e-10f ? nrm.x : 1.f);
          v4.y /= (nrm.y > 1e-10f ? nrm.y : 1.f);
          v4.z /= (nrm.z > 1e-10f ? nrm.z : 1.f);
          v4.w /= (nrm.w > 1e-10f ? nrm.w : 1.f);
        }
      }
      *reinterpret_cast<float4*>(&xs[(e >> 8) * XP + (e & 255)]) = v4;
    }
  }
  __syncthreads();

  // ---- gather x window, forward transform, x mask (split -> mask -> merge in place), inverse transform ------------------
  const int t = 4 * wave + g;
  const bool fvalid = t < T;
  const bool l0 = c == 0;
  cf v[32];
  {
    const float* xp = xs + t * XP + 2 * c;
    const float2* wl = reinterpret_cast<const float2*>(swin + 2 * c);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      float2 x2 = *reinterpret_cast<const float2*>(xp + (r >> 3) * XP + 32 * (r & 7));
      if (!fvalid) x2 = make_float2(0.f, 0.f);
      const float2 w2 = wl[16 * r];
      v[r] = {x2.x * w2.x, x2.y * w2.y};
    }
  }
  __syncthreads();   // every lane has its samples: the exchanges may overwrite them
  cf* fb = slices + wave * WAVE_CX_H + frame_base_h(g);
  fft512_fwd_half(v, fb, tw512, c);
  {
    cf wlo = s_tw1024[c];
    asm volatile("" : "+v"(wlo.x), "+v"(wlo.y));
    cf whi = wlo;
    {
      const cf w16 = s_tw1024[16];
      if (l0) whi = {-w16.y, w16.x};  // i * w_1024^16
    }
    rg_lane0_to_entries(v, l0);
    const float* Mrow = A.mask + ((int64_t)blockIdx.x * T + (fvalid ? t : 0)) * (int64_t)G.FS;
    const float ks = A.kscale * 0.25f;   // split + merge leave out four 1/2 factors
    const float k512 = Mrow[512] * A.kscale;
    {
      // slot 0: lanes >= 1 the pair (v[0], v[31]); lane 0: bins 0 / 512 from v[0], bin 256 = v[31] scaled
      const float m0 = Mrow[bin_of_entry(c, 0)], m31 = Mrow[bin_of_entry(c, 31)];
      const cf r0 = v[0], r31 = v[31];
      cf xa, xb;
      split_pair(r0, r31, wlo, xa, xb);
      merge_pair(xa, xb, wlo, m0 * ks, m31 * ks);
      const float y0 = (r0.x + r0.y) * (m0 * A.kscale);
      const float yN = (r0.x - r0.y) * k512;
      const cf z0 = {0.5f * (y0 + yN), 0.5f * (y0 - yN)};
      const float m8 = m31 * A.kscale;    // entry 31 of lane 0 = bin 256
      const cf z8 = {r31.x * m8, r31.y * m8};
      v[0] = {l0 ? z0.x : xa.x, l0 ? z0.y : xa.y};
      v[31] = {l0 ? z8.x : xb.x, l0 ? z8.y : xb.y};
    }
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
      const float mk = Mrow[bin_of_entry(c, sl)] * ks, mn = Mrow[bin_of_entry(c, 31 - sl)] * ks;
      const cf w = mul_tw<false>(sl < 8 ? wlo : whi, twc<32>(sl), tws<32>(sl));
      cf xa, xb;
      split_pair(v[sl], v[31 - sl], w, xa, xb);
      merge_pair(xa, xb, w, mk, mn);
      v[sl] = xa;
      v[31 - sl] = xb;
    }
    rg_lane0_from_entries(v, l0);
    if (!fvalid) {
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = {0.f, 0.f};
    }
  }
  {
    int zi = 0, ci = c;
    asm volatile("" : "+v"(zi), "+v"(ci));
    fft512_inv_half(v, fb + zi, tw512 + zi, ci);
  }
  // ---- synthesis window, wave-private overlap-add into 7 hop accumulators (k_apply_fast<LEAN>) ---------------------------
  float* acc = reinterpret_cast<float*>(slices + wave * WAVE_CX_H);
  {
    const float2* wsrc2 = reinterpret_cast<const float2*>(swin + 2 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
    }
  }
  __syncthreads();
  // ---- combine the row's 67 hops (hop lj: wave lj / 4, local hop lj % 4, + wave lj / 4 - 1, local hop lj % 4 + 4); the adjoint
  // is NOT normalised ------------------------------------------------------------------------------------------------------
  const float* fr = reinterpret_cast<const float*>(slices);
  const int s4 = lane * 4;
  const int64_t h_begin = (A.om.p0 + G.padL) / 256, h_end = (A.om.p1 - 1 + G.padL) / 256 + 1;
  for (int lj = wave; lj < 4 * WAVES + 3; lj += WAVES) {
    if (lj < (int)h_begin || lj >= (int)h_end) continue;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int wh = lj >> 2, lh = lj & 3;
    if (wh >= 1 && lh <= 2) a4 = *reinterpret_cast<const float4*>(&fr[(wh - 1) * WAVE_CX_H * 2 + (lh + 4) * HPITCH + s4]);
    if (wh < WAVES) {
      const float4 f4 = *reinterpret_cast<const float4*>(&fr[wh * WAVE_CX_H * 2 + lh * HPITCH + s4]);
      a4.x += f4.x; a4.y += f4.y; a4.z += f4.z; a4.w += f4.w;
    }
    const int64_t pbs = (int64_t)lj * 256 - G.padL;
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
      const int64_t p = (int64_t)lj * 256 + s4 + e - G.padL;
      if (p < A.om.p0 || p >= A.om.p1) continue;
      const int64_t gi = p - A.om.p0;
      if (gi < A.om.g_lo || gi >= A.om.g_hi) continue;
      store_sample(A.om.out, A.om.dtype, row * A.om.stride + gi - A.om.g0, p < G.Lout ? vals[e] : 0.f);
    }
  }
}

}  // namespace fast
}  // namespace sg
