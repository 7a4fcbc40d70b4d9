#!/usr/bin/env python3
"""Per-repetition HIP-event times of configs[2] (non-stationary) / configs[4] (TorchGate) calls: finds outlier
repetitions (round-2 BENCH had one 40 ms repetition among ten of config 3).  Usage: stall_probe.py [reps]"""
import os, sys, time, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__
__graft_entry__.build()
import noisereduce_amd as nr
from noisereduce_amd.torchgate import TorchGate

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
# CPython garbage-collector pauses on the enqueueing thread (a full collection of a process that has torch imported
# takes tens of milliseconds: the GPU idles meanwhile and the repetition looks like a 30-40 ms kernel)
import gc
GC_LOG = []
def _gc_cb(phase, info, _t=[0.0]):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        GC_LOG.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3))
gc.callbacks.append(_gc_cb)
dev = torch.device("cuda", 0)
SR = 48000
g = torch.Generator(device=dev); g.manual_seed(1234)
n = SR * 600
y = (0.1 * torch.randn(n, generator=g, device=dev) + 0.5 * torch.sin(2 * np.pi * 1000.0 * torch.arange(n, device=dev, dtype=torch.float64) / SR).float()).contiguous()

def probe(name, fn, warm=3):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    host = []
    for i in range(reps):
        ev[i].record()
        t0 = time.perf_counter()
        fn()
        host.append((time.perf_counter() - t0) * 1e3)
    ev[reps].record()
    torch.cuda.synchronize()
    ts = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])
    host = np.array(host)
    out = {"name": name, "reps": reps, "median_ms": float(np.median(ts)), "mean_ms": float(ts.mean()), "max_ms": float(ts.max()),
           "outliers(rep, gpu_ms, host_enqueue_ms)": [(int(i), round(float(ts[i]), 3), round(float(host[i]), 3)) for i in np.argsort(-ts)[:6]],
           "first10_ms": [round(float(t), 3) for t in ts[:10]],
           "host_enqueue_median_ms": float(np.median(host)), "host_enqueue_max_ms": float(host.max()),
           "gc_pauses_ms(generation, ms) over 1 ms": [(g_, round(ms_, 2)) for g_, ms_ in GC_LOG if ms_ > 1.0]}
    GC_LOG.clear()
    print(json.dumps(out), flush=True)

probe("config2 stationary reduce_noise", lambda: nr.reduce_noise(y=y, sr=SR, stationary=True))
probe("config3 non-stationary reduce_noise", lambda: nr.reduce_noise(y=y, sr=SR, stationary=False))
tg = TorchGate(sr=16000).to(dev)
x = (0.1 * torch.randn(256, 16000, device=dev)).float()
probe("config5 TorchGate forward", lambda: tg(x), warm=10)

xg = x.clone().requires_grad_()
def fb():
    xg.grad = None
    tg(xg).sum().backward()
probe("config5 TorchGate forward+backward", fb, warm=10)
import gc
gc.disable()
probe("config5 TorchGate forward+backward (gc disabled)", fb, warm=10)
