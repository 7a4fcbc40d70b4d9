"""Host enqueue time vs GPU time of reduce_noise per n_fft (2 min of audio, tensor in / tensor out): is a geometry bound
by the kernels or by something the host does per call?  usage: NFFT=256,4096 python tools/host_vs_gpu.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import noisereduce_amd as nr
sr, n = 48000, 48000 * 120
rng = np.random.default_rng(0)
y = torch.from_numpy((0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000 * np.arange(n) / sr)).astype(np.float32)).cuda()
for n_fft in [int(a) for a in os.environ.get("NFFT", "256,512,1024,2048,4096,8192").split(",")]:
    for stationary in (True, False):
        kw = dict(stationary=stationary, n_fft=n_fft, time_mask_smooth_ms=(400 if n_fft > 16384 else 200) if n_fft > 2048 else 50)
        for _ in range(20):
            nr.reduce_noise(y=y, sr=sr, **kw)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            nr.reduce_noise(y=y, sr=sr, **kw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n_fft {n_fft} {'stat' if stationary else 'nonstat'}: host enqueue {1e3 * (t1 - t0) / reps:.3f} ms/call, "
              f"wall {1e3 * (t2 - t0) / reps:.3f} ms/call", flush=True)
