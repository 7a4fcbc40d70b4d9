"""A foreign load for stress runs: a 2048^2 matmul loop on its own stream (and process) until killed.
   python tools/experiments/matmul_load.py & PID=$!; python -m pytest tests -m gpu -q; kill $PID"""
import torch
mm = torch.randn(2048, 2048, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    while True:
        (mm @ mm).sum().item()
