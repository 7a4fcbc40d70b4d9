#!/usr/bin/env python3
"""configs[1]-sized reduce_noise for device tensors of every sample dtype (and precision="float64")."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noisereduce_amd as nr, bench
y32 = bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0))
cases = (("float32", y32, None), ("float64", y32.double(), None), ("float64 precision=float64", y32.double(), "float64"),
         ("int16", (y32 * 20000).to(torch.int16), None), ("int32", (y32 * 2e8).to(torch.int32), None))
for name, y, prec in cases:
    for stat in (True, False):
        f = lambda: nr.reduce_noise(y=y, sr=48000, stationary=stat, precision=prec)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        print(name, "stationary" if stat else "non-stationary", round((time.perf_counter() - t0) / 10 * 1e3, 4), "ms")
