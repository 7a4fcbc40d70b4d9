import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
from oracle import spectralgate_oracle as O
n = 48000 * 2
t = np.arange(n) / 48000
cases = {"zeros": np.zeros(n), "dc": np.full(n, 0.25), "tone1k": 0.5 * np.sin(2 * np.pi * 1000 * t),
         "tone_bin": 0.5 * np.sin(2 * np.pi * (48000 / 1024 * 20) * t), "impulse": np.eye(1, n, 30000)[0],
         "square": np.sign(np.sin(2 * np.pi * 100 * t)) * 0.3, "tiny": 1e-30 * np.sin(2 * np.pi * 1000 * t)}
for name, y in cases.items():
    y = y.astype(np.float32)
    for stationary in (True, False):
        got = nr.reduce_noise(y=y, sr=48000, stationary=stationary, n_fft=1024)
        with np.errstate(all="ignore"):
            want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=stationary, n_fft=1024)
        gn, wn = ~np.isfinite(got), ~np.isfinite(want); both = ~gn & ~wn
        peak = max(np.abs(want[both]).max(), 1e-30) if both.any() else 1.0
        print("%-9s %-14s nonfinite %6d / %6d  err/peak %.2e  peak %.3e  in-peak %.3e" % (name, "stationary" if stationary else "non-stationary", gn.sum(), wn.sum(),
              (np.abs(got[both] - want[both]).max() / peak) if both.any() else 0.0, peak, np.abs(y).max()))
