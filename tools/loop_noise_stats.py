"""30 x (noise-clip statistics + a 2 s filter) for a rocprofv3 kernel trace; SECS = clip length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import noisereduce_amd as nr
sr = 48000
torch.manual_seed(0)
y = torch.randn(sr * 2, device="cuda")
noise = torch.randn(sr * int(os.environ.get("SECS", "1")), device="cuda")
for _ in range(30):
    out = nr.reduce_noise(y=y, sr=sr, y_noise=noise, stationary=True, n_fft=1024)
torch.cuda.synchronize()
