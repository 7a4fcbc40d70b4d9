#!/usr/bin/env python3
"""Noise statistics with the final stage inside k_colstats1 (default) against the two-launch form (SG_STATS_FUSED=0 in the
environment -- read once per process, so one run per setting): wall time per reduce_noise() call, stationary, device-resident,
for the headline (n_fft = 1024, 10 min) and the 2-minute legs of n_fft = 256 / 512 / 2048; a checksum of the outputs (must
not depend on the setting).  One JSON line."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import noisereduce_amd as nr
dev = torch.device("cuda", 0)
y = bench.synth_on_device(bench.N_PER_GPU, 1234, dev)
y2 = y[:48000 * 120].contiguous()
legs = [("1024,10min", y, 1024), ("256,2min", y2, 256), ("512,2min", y2, 512), ("2048,2min", y2, 2048)]
kw = lambda n: dict(sr=48000, stationary=True, n_fft=n, **({} if n != 1024 else dict(chunk_size=600000, padding=30000)))
for _ in range(200): nr.reduce_noise(y=y, **kw(1024))   # clock ramp
res, chk = {}, {}
for rnd in range(3):
    for name, yy, n in legs:
        for _ in range(20): out = nr.reduce_noise(y=yy, **kw(n))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): out = nr.reduce_noise(y=yy, **kw(n))
        torch.cuda.synchronize()
        res.setdefault(name, []).append(round((time.perf_counter() - t0) / 100 * 1e3, 4))
        chk[name] = float(out.double().abs().sum())
print(json.dumps({"SG_STATS_FUSED": os.environ.get("SG_STATS_FUSED", "1"), "ms_per_call": res, "checksum": chk}))
