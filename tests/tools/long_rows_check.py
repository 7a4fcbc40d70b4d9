"""TorchGate rows of 10 minutes (2 x 28.8 M samples, 112 k frames per row: k_colmax / k_colstats statistics instead of the
one-kernel row statistics, 1758 time tiles of k_box_mask) and a 16-channel reduce_noise of 5-minute channels on the
float64 pipeline for int16 samples, against the oracle.  usage: python tests/tools/long_rows_check.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
from noisereduce_amd.torchgate import TorchGate
from oracle import spectralgate_oracle as O

sr = 48000
L = sr * 600
rng = np.random.default_rng(8)
x = np.stack([(0.05 * rng.standard_normal(L) + 0.3 * np.sin(2 * np.pi * (500.0 + 300 * b) * np.arange(L) / sr)) for b in range(2)])
win = torch.hann_window(1024).double().numpy()
for nonstat in (False, True):
    t0 = time.time()
    got = TorchGate(sr=sr, nonstationary=nonstat).cuda()(torch.from_numpy(x).cuda()).cpu().numpy()
    t1 = time.time()
    want = O.torchgate_T(x, sr, nonstationary=nonstat, window=win)
    print("TorchGate %s 2 x 10 min: rel err %.2e (engine %.2f s, oracle %.1f s)" %
          ("non-stationary" if nonstat else "stationary", O.rel_err(got, want), t1 - t0, time.time() - t1), flush=True)
    assert O.rel_err(got, want) < 1e-4
n = sr * 300
y = np.stack([np.round(3000 * rng.standard_normal(n) + 9000 * np.sin(2 * np.pi * (300.0 + 100 * c) * np.arange(n) / sr)) for c in range(16)]).astype(np.int16)
for stationary in (True, False):
    # (an explicit noise clip: without one the stationary gate takes its statistics from the channel MEAN of the
    # recording, and the per-channel oracle runs below -- ~10 s each, two channels -- would not see the same threshold)
    clip = y[0, :96000]
    got = nr.reduce_noise(y=y, sr=sr, stationary=stationary, y_noise=clip)
    bad = 0
    for c in (0, 15):
        w = O.reduce_noise_S(y[c].astype(np.float64), sr, stationary=stationary, y_noise=clip.astype(np.float64))
        wi = w.astype(np.int16)
        far = np.abs(w - np.round(w)) > 1e-9
        bad += int(np.sum(got[c][far] != wi[far]))
    print("int16 16 x 5 min %s: %d mismatching samples on channels 0 and 15 (dtype %s)" %
          ("stationary" if stationary else "non-stationary", bad, got.dtype), flush=True)
    assert bad == 0 and got.dtype == np.int16
print("ok")
