"""Same-process A/B of abutting tiles + k_ola_seam (default, round 6) against overlapping tiles (SG_OPT_FORCE_NOSEAM) for the
stationary gate at n_fft = 512 / 256 / 2048: alternating blocks of calls, device-resident, 2 and 10 minutes of 48 kHz audio."""
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
sr = 48000
rng = np.random.default_rng(0)
res = {}
for minutes in (2, 10):
    n = sr * 60 * minutes
    y = torch.from_numpy((0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000 * np.arange(n) / sr)).astype(np.float32)).cuda()
    for n_fft in (256, 512, 2048):
        for stationary in (True,):
            kw = dict(y=y, sr=sr, chunk_size=600000, padding=30000, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0,
                      freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, prop_decrease=1.0, use_tqdm=False, n_jobs=1,
                      y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True)
            def call():
                return SpectralGateStationary(**kw).get_traces()
            g = SpectralGateStationary(**kw)._gate
            for _ in range(60): call()
            t = {0: [], 1: []}
            for rnd in range(6):
                for mode in (0, 1):
                    g.set_option(_ffi.SG_OPT_FORCE_NOSEAM, mode)
                    for _ in range(5): call()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(40): call()
                    b.record(); torch.cuda.synchronize()
                    t[mode].append(a.elapsed_time(b) / 40)
            g.set_option(_ffi.SG_OPT_FORCE_NOSEAM, 0)
            res[f"{minutes}min,n_fft={n_fft}"] = {"abutting_ms": round(float(np.median(t[0])), 4), "overlapping_ms": round(float(np.median(t[1])), 4)}
            print(minutes, n_fft, res[f"{minutes}min,n_fft={n_fft}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/ab_seam.json", "w"), indent=1)
