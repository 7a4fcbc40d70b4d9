#!/usr/bin/env python3
"""configs[1] with prop_decrease = 1.0 vs 0.8 (device-resident, whole reduce_noise)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noisereduce_amd as nr, bench
y = bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0))
for p in (1.0, 0.8, 1.0, 0.8):
    f = lambda: nr.reduce_noise(y=y, sr=48000, stationary=True, prop_decrease=p)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    print("prop_decrease", p, round((time.perf_counter() - t0) / 20 * 1e3, 4), "ms")
