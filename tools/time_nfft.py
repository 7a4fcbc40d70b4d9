"""Device-resident throughput of reduce_noise (stationary / non-stationary) over n_fft, including
frame lengths that run on the chirp-z kernels.  Usage: [MINUTES=2] [NFFT=512,1024,...] python tools/time_nfft.py"""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import noisereduce_amd as nr

sr = 48000
n = int(sr * 60 * float(os.environ.get("MINUTES", "2")))
rng = np.random.default_rng(0)
y = (0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000 * np.arange(n) / sr)).astype(np.float32)
yd = torch.from_numpy(y).cuda()
res = {}
NFFTS = [int(a) for a in os.environ["NFFT"].split(",")] if os.environ.get("NFFT") else (256, 400, 512, 1000, 1024, 1536, 2048, 3000, 4096, 8192, 5000, 16384, 32768)
for n_fft in NFFTS:
    for stationary in (True, False):
        kw = dict(stationary=stationary, n_fft=n_fft, time_mask_smooth_ms=(400 if n_fft > 16384 else 200) if n_fft > 2048 else 50)
        for _ in range(int(os.environ.get("WARM", "2"))):
            nr.reduce_noise(y=yd, sr=sr, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = int(os.environ.get("REPS", "5"))
        a.record()
        for _ in range(reps):
            nr.reduce_noise(y=yd, sr=sr, **kw)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        res[f"n_fft={n_fft},{'stat' if stationary else 'nonstat'}"] = dict(ms=round(ms, 3), Msamples_s=round(n / ms / 1e3, 1))
        print(n_fft, stationary, round(ms, 3), "ms", round(n / ms / 1e3, 1), "Msamples/s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/time_nfft.json", "w"), indent=1)
