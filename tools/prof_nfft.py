"""Per-stage kernel times of the stationary / non-stationary gate for a given n_fft (2 min of 48 kHz)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
sr, n = 48000, 48000 * 120
rng = np.random.default_rng(0)
y = torch.from_numpy((0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000 * np.arange(n) / sr)).astype(np.float32)).cuda()
for n_fft in [int(a) for a in sys.argv[1:]] or [512, 2048]:
    for stationary in (True, False):
        kw = dict(y=y, sr=sr, chunk_size=600000, padding=30000, n_fft=n_fft, win_length=None, hop_length=None,
                  time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
                  prop_decrease=1.0, use_tqdm=False, n_jobs=1)
        mk = (lambda: SpectralGateStationary(y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, **kw)) if stationary \
            else (lambda: SpectralGateNonStationary(thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, **kw))
        for _ in range(2): mk().get_traces()
        g = mk()._gate
        g.profile_read(reset=True); g.profile_enable(True)
        for _ in range(5): mk().get_traces()
        p = g.profile_read(reset=True); g.profile_enable(False)
        print(n_fft, "stat" if stationary else "nonstat", {k: round(v[0] / 5, 4) for k, v in p.items()})
