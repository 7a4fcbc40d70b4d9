#!/usr/bin/env python3
"""Secondary measurements (BASELINE.json configs 3 and 5, PCIe-inclusive rate of config 2).
Prints one JSON object; run on the GPU box:  python tools/bench_configs.py"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__; __graft_entry__.build()
import noisereduce_amd as nr
from noisereduce_amd.torchgate import TorchGate
import bench

def timeit(fn, warm=3, reps=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

dev = torch.device("cuda", 0)
out = {}
y = bench.synth_on_device(bench.N_PER_GPU, 1234, dev)
t = timeit(lambda: nr.reduce_noise(y=y, sr=48000, stationary=True))
out["config2_stationary_device_resident"] = {"ms": round(t * 1e3, 3), "Msamples_s": round(y.numel() / t / 1e6, 1)}
t = timeit(lambda: nr.reduce_noise(y=y, sr=48000, stationary=False))
out["config3_nonstationary_device_resident"] = {"ms": round(t * 1e3, 3), "Msamples_s": round(y.numel() / t / 1e6, 1)}
yh = y.cpu().numpy()
t = timeit(lambda: nr.reduce_noise(y=yh, sr=48000, stationary=True), warm=2, reps=5)
out["config2_numpy_in_numpy_out_pcie"] = {"ms": round(t * 1e3, 3), "Msamples_s": round(yh.size / t / 1e6, 1)}
# config 5: TorchGate batch=256 x 1 s @ 16 kHz, forward and forward+backward
torch.manual_seed(0)
x = (0.1 * torch.randn(256, 16000, device=dev) + 0.5 * torch.sin(2 * np.pi * 440 * torch.arange(16000, device=dev) / 16000)).float()
tg = TorchGate(sr=16000).to(dev)
t = timeit(lambda: tg(x), warm=5, reps=20)
out["config5_torchgate_forward"] = {"ms": round(t * 1e3, 3), "Msamples_s": round(x.numel() / t / 1e6, 1)}
xg = x.clone().requires_grad_()
def fb():
    xg.grad = None
    tg(xg).sum().backward()
t = timeit(fb, warm=5, reps=20)
out["config5_torchgate_forward_backward"] = {"ms": round(t * 1e3, 3), "Msamples_s": round(x.numel() / t / 1e6, 1)}
tgn = TorchGate(sr=16000, nonstationary=True).to(dev)
t = timeit(lambda: tgn(x), warm=5, reps=20)
out["config5_torchgate_nonstationary_forward"] = {"ms": round(t * 1e3, 3), "Msamples_s": round(x.numel() / t / 1e6, 1)}
print(json.dumps(out))
