"""Workload for the HBM-traffic PMC passes: a calibration copy of known size, then reduce_noise
steps of the bench workload (configs[1])."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import noisereduce_amd as nr
dev = torch.device("cuda", 0)
cal = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()   # 1 GiB
for _ in range(3):
    c2 = cal.clone()                                                   # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
y = bench.synth_on_device(bench.N_PER_GPU, 1234, dev)
for _ in range(4):
    out = nr.reduce_noise(y=y, sr=48000, stationary=True)        # configs[1]
torch.cuda.synchronize()
for _ in range(4):
    out = nr.reduce_noise(y=y, sr=48000, stationary=False)       # configs[2]
torch.cuda.synchronize()
import numpy as np
from noisereduce_amd.torchgate import TorchGate
torch.manual_seed(0)
t = torch.arange(16000, device=dev, dtype=torch.float64) / 16000
x = (0.1 * torch.randn(256, 16000, device=dev) + 0.5 * torch.sin(2 * np.pi * 440 * t).float()).float()
tg = TorchGate(sr=16000).to(dev)
for _ in range(4):
    out = tg(x)                                                  # configs[4] forward
torch.cuda.synchronize()
xg = x.clone().requires_grad_()
for _ in range(4):
    xg.grad = None
    tg(xg).sum().backward()                                      # configs[4] forward (mask saved) + backward (k_row_backward)
torch.cuda.synchronize()
y16 = (y * 20000).to(torch.int16)
for _ in range(4):
    out = nr.reduce_noise(y=y16, sr=48000, stationary=True)      # int16 recording: bit path + k_apply_fast64
torch.cuda.synchronize()
if os.environ.get("TRAFFIC_NFFT"):                                # other geometries, ten minutes each (r5: fast256.hpp)
    for n_fft in [int(a) for a in os.environ["TRAFFIC_NFFT"].split(",")]:
        for stat in (True, False):
            for _ in range(4):
                out = nr.reduce_noise(y=y, sr=48000, stationary=stat, n_fft=n_fft)
            torch.cuda.synchronize()
