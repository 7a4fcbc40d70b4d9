"""Throughput of back-to-back reduce_noise calls on one vs two HIP streams (independent handles)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd import _ffi
from noisereduce_amd.sharded import HipStationaryBackend, TimeShardedStationary
dev = torch.device("cuda", 0)
y = 0.1 * torch.randn(1, 28_800_000, device=dev)
def make(slot):
    b = HipStationaryBackend(48000, dev)
    b._gate()                      # cached handle ...
    if slot:                       # ... replaced by a private one for the second stream
        g0 = b._g
        b._g = _ffi.Gate(dev, **g0._kw) if hasattr(g0, "_kw") else None
    return b
def run(nstreams, steps=60):
    bs = [HipStationaryBackend(48000, dev) for _ in range(nstreams)]
    for i, b in enumerate(bs):
        b._gate()
        if i: b._g = _ffi.Gate(dev, **GKW)
    ss = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    def step(i):
        with torch.cuda.stream(ss[i % nstreams]):
            return TimeShardedStationary(bs[i % nstreams], 513).run(y)
    for i in range(6): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [step(i) for i in range(steps)]
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    return t * 1e3
GKW = dict(variant=_ffi.SG_VARIANT_S, stationary=True, n_fft=1024, win_length=1024, hop_length=256, n_grad_freq=5,
           n_grad_time=9, smooth_mask=True, chunk_size=600000, padding=30000, prop_decrease=1.0, n_std_thresh=1.5,
           top_db=80.0, ddof=0)
for ns in (1, 2, 3):
    print(ns, "streams: %.4f ms/step" % run(ns))
