#!/usr/bin/env python3
"""configs[4] forward + backward (TorchGate on 256 x 16000 float32, loss = sum) a few times: for rocprofv3 --kernel-trace --stats."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd.torchgate import TorchGate
dev = torch.device("cuda", 0)
torch.manual_seed(0)
t = torch.arange(16000, device=dev, dtype=torch.float64) / 16000
x = (0.1 * torch.randn(256, 16000, device=dev) + 0.5 * torch.sin(2 * np.pi * 440 * t).float()).float()
tg = TorchGate(sr=16000).to(dev)
xg = x.clone().requires_grad_()
for _ in range(20):
    xg.grad = None
    tg(xg).sum().backward()
torch.cuda.synchronize()
