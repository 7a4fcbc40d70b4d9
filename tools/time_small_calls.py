#!/usr/bin/env python3
"""Latency of reduce_noise on SHORT clips (device tensors and numpy arrays): the launch- and host-bound regime."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noisereduce_amd as nr
dev = torch.device("cuda", 0)
for n in (48000, 480000):
    y = (0.1 * torch.randn(n, device=dev)).float()
    yh = y.cpu().numpy()
    for stat in (True, False):
        for name, arr in (("tensor", y), ("numpy", yh)):
            f = lambda: nr.reduce_noise(y=arr, sr=48000, stationary=stat)
            for _ in range(5): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): f()
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"n={n} {'stationary' if stat else 'non-stationary'} {name}: host {(t1 - t0) / 100 * 1e6:.0f} us/call, incl. drain {(t2 - t0) / 100 * 1e6:.0f} us/call")
