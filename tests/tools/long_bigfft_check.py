"""Ten minutes of 48 kHz audio through the long-frame (four-step, n_fft = 16384), chirp-z (n_fft = 3000) and
general power-of-two (n_fft = 4096, 256) transform families, default chunk grid, against the oracle.
usage: python tests/tools/long_bigfft_check.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
from oracle import spectralgate_oracle as O

sr = 48000
n = sr * 600
rng = np.random.default_rng(21)
y = (0.05 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 900.0 * np.arange(n) / sr)).astype(np.float32)
yd = torch.from_numpy(y).cuda()
for n_fft in (16384, 3000, 4096, 256):
    for stationary in (True, False):
        kw = dict(stationary=stationary, n_fft=n_fft, time_mask_smooth_ms=400 if n_fft > 8192 else 50)
        t0 = time.time()
        got = nr.reduce_noise(y=yd, sr=sr, **kw).cpu().numpy()
        t1 = time.time()
        want = O.reduce_noise_S(y.astype(np.float64), sr, **kw)
        e = O.rel_err(got, want)
        print("n_fft %5d %-14s rel err %.2e (engine %.2f s, oracle %.1f s)" %
              (n_fft, "stationary" if stationary else "non-stationary", e, t1 - t0, time.time() - t1), flush=True)
        assert e < 1e-4
print("ok")
