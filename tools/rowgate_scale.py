#!/usr/bin/env python3
"""Row gate: time per call against the number of rows (one workgroup per row, one workgroup per CU at a time)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noisereduce_amd import _ffi
from noisereduce_amd.torchgate import TorchGate
dev = "cuda:0"
t16 = torch.arange(16000, device=dev, dtype=torch.float64) / 16000
tg = TorchGate(sr=16000).to(dev)
res = {}
for B in (64, 128, 256, 512, 1024, 2048):
    torch.manual_seed(0)
    x = (0.1 * torch.randn(B, 16000, device=dev) + 0.5 * torch.sin(2 * np.pi * 440 * t16).float()).float()
    for mode in ("rowgate16", "rowgate8", "four"):
        tg(x)
        (g,) = list(tg._gates.values())
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 1 if mode == "four" else 2)
        g.set_option(_ffi.SG_OPT_ROWGATE_SHAPE, 8 if mode == "rowgate8" else 16)
        for _ in range(10):
            tg(x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
        for i in range(30):
            ev[i].record(); tg(x)
        ev[30].record(); torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(30))
        res["%s B=%d" % (mode, B)] = round(ts[15], 4)
        print(mode, B, ts[15], flush=True)
    g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0)
    g.set_option(_ffi.SG_OPT_ROWGATE_SHAPE, 16)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "rowgate_scale.json"), "w"), indent=1)

# host enqueue cost per call (asynchronous loop, no synchronisation inside) and the kernel's own time (HIP events around the launch)
import time
x = (0.1 * torch.randn(256, 16000, device=dev) + 0.5 * torch.sin(2 * np.pi * 440 * t16).float()).float()
(g,) = list(tg._gates.values())
for _ in range(20):
    tg(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    tg(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per call (us):", (t1 - t0) / 200 * 1e6, " incl. drain:", (t2 - t0) / 200 * 1e6)
g.profile_read(reset=True); g.profile_select(None); g.profile_enable(True)
for _ in range(50):
    tg(x)
pr = g.profile_read(reset=True)
g.profile_enable(False)
print({k: (round(v[0] / v[1], 4), v[1]) for k, v in pr.items()})
