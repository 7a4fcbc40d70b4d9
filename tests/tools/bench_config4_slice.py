#!/usr/bin/env python3
"""One GPU's share of BASELINE.json configs[3]: 8 channels x 30 min @ 48 kHz (the 64-channel recording
is channel-sharded 8 per GPU), stationary, generated on the device.  Checks a few (channel, chunk)
units against the oracle and reports throughput."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__; __graft_entry__.build()
import noisereduce_amd as nr
from oracle import spectralgate_oracle as O
import bench
dev = torch.device("cuda", 0)
C, N = 8, 48000 * 1800
y = torch.empty((C, N), dtype=torch.float32, device=dev)
for c in range(C):
    y[c] = bench.synth_on_device(N, 1234 + c, dev, tone_hz=200.0 * (c + 1))
torch.cuda.synchronize()
def run():
    return nr.reduce_noise(y=y, sr=48000, stationary=True)
out = run(); out2 = run(); del out2   # two output-sized blocks in the caching allocator before timing
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 5
for _ in range(reps): out = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
res = {"channels": C, "samples_per_channel": N, "ms": round(dt * 1e3, 2), "Msamples_s": round(C * N / dt / 1e6, 1),
       "mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
# parity on a subset of units: the threshold needs the channel mean of the first 600000 samples of ALL channels
yh = y[:, :600000].cpu().numpy().astype(np.float64)
thr, _, _ = O.noise_threshold_S(yh, 1024, 1024, 256, 1.5, 600000)
filt = O.smoothing_filter(5, 9)
worst = 0.0
for c, ich in [(0, 0), (3, 1), (7, 143), (5, 77)]:
    s0 = ich * 600000
    lo, hi = max(0, s0 - 30000), min(N, s0 + 630000)
    chunk = np.zeros((1, 660000)); chunk[0, lo - (s0 - 30000):hi - (s0 - 30000)] = y[c, lo:hi].cpu().numpy()
    ref = O.gate_stationary_S(chunk, thr, 1024, 1024, 256, 1.0, filt)[0, 30000:630000]
    got = out[c, s0:s0 + 600000].cpu().numpy()
    worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
res["max_rel_err_vs_oracle_on_4_units"] = worst
print(json.dumps(res))
