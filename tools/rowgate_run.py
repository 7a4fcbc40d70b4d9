#!/usr/bin/env python3
"""A few TorchGate forwards of 256 x 16000 (profiling target: rocprofv3 -- python tools/rowgate_run.py [rows] [reps])."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noisereduce_amd.torchgate import TorchGate
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
t = torch.arange(16000, device="cuda", dtype=torch.float64) / 16000
x = (0.1 * torch.randn(B, 16000, device="cuda") + 0.5 * torch.sin(2 * np.pi * 440 * t).float()).float()
tg = TorchGate(sr=16000).cuda()
for _ in range(reps):
    y = tg(x)
torch.cuda.synchronize()
print("done", float(y.abs().max()))
