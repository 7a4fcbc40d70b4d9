import os, sys, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, noisereduce_amd as nr
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
dev = torch.device("cuda", 0)
for C, N in ((1, 48000 * 600), (4, 48000 * 600), (8, 48000 * 1800)):
    y = torch.empty((C, N), dtype=torch.float32, device=dev)
    for c in range(C):
        y[c] = bench.synth_on_device(N, 1234 + c, dev, tone_hz=200.0 * (c + 1))
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
              clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
              time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
              use_tqdm=False, n_jobs=1)
    for _ in range(2):
        SpectralGateStationary(y=y, **kw).get_traces()
    sg = SpectralGateStationary(y=y, **kw)
    g = sg._gate
    g.profile_read(reset=True); g.profile_enable(True)
    sg.get_traces()
    p = g.profile_read(reset=True); g.profile_enable(False)
    units = C * (N // 600000)
    print(C, N, "units", units, {k: (round(v[0], 3), v[1]) for k, v in p.items()}, "us/unit", {k: round(v[0] * 1e3 / units, 2) for k, v in p.items()})
    del y
