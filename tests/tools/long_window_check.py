"""One window of an hour: reduce_noise(chunk_size=None) on 48000 * 3600 samples (675 k frames, 10.5 k time tiles of the
mask kernels, 1.4 GB magnitude field) against the oracle, stationary and non-stationary, and against the same
recording filtered in the default 600 k-sample chunks away from the chunk seams (where the two agree by construction
only for the stationary gate).  usage: python tests/tools/long_window_check.py [minutes]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
from oracle import spectralgate_oracle as O

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sr = 48000
n = int(sr * 60 * minutes)
rng = np.random.default_rng(3)
y = (0.05 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 700.0 * np.arange(n) / sr) *
     (0.5 + 0.5 * np.sin(2 * np.pi * 0.05 * np.arange(n) / sr))).astype(np.float32)
yd = torch.from_numpy(y).cuda()
for stationary in (True, False):
    t0 = time.time()
    got = nr.reduce_noise(y=yd, sr=sr, stationary=stationary, chunk_size=None).cpu().numpy()
    t1 = time.time()
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=stationary, chunk_size=None)
    t2 = time.time()
    print("%s one window of %.0f min: rel err %.2e  (engine %.2f s incl. first-call allocations, oracle %.1f s)" %
          ("stationary" if stationary else "non-stationary", minutes, O.rel_err(got, want), t1 - t0, t2 - t1), flush=True)
    assert O.rel_err(got, want) < 1e-4
print("ok")
