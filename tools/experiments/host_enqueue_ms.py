"""Development: host time to ENQUEUE one nr.reduce_noise(tensor) call (no synchronisation inside the loop) against the
GPU time of the call -- is the public API path GPU-bound at configs[1] / configs[2]?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import noisereduce_amd as nr
dev = torch.device("cuda:0")
y = bench.synth_on_device(bench.N_PER_GPU, 0, dev)
for stat in (True, False):
    f = lambda: nr.reduce_noise(y=y, sr=48000, stationary=stat)
    for _ in range(50): f()
    torch.cuda.synchronize()
    hs = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        t0 = time.perf_counter(); f(); hs.append((time.perf_counter() - t0) * 1e3)
    e1.record(); e1.synchronize()
    print("stationary" if stat else "non-stationary", "host enqueue ms median %.4f p90 %.4f max %.4f | GPU ms per call %.4f"
          % (np.median(hs), np.percentile(hs, 90), max(hs), e0.elapsed_time(e1) / 200))
