"""rocprofv3 target: reduce_noise of 2 min of 48 kHz audio at ONE n_fft (argv[1]) and gate (argv[2]: stat / nonstat),
40 calls -- the per-kernel averages of the kernel-stats table are per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import noisereduce_amd as nr
sr, n = 48000, 48000 * 120
n_fft = int(sys.argv[1]); stationary = (sys.argv[2] if len(sys.argv) > 2 else "stat") == "stat"
rng = np.random.default_rng(0)
y = torch.from_numpy((0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000 * np.arange(n) / sr)).astype(np.float32)).cuda()
if os.environ.get("SIGNAL") == "bench":   # bench.py's synth_on_device (torch generator on the device)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    t = torch.arange(n, device="cuda", dtype=torch.float64) / sr
    y = (torch.randn(n, generator=g, device="cuda", dtype=torch.float32) * 0.1 + 0.5 * torch.sin(2 * np.pi * 1000.0 * t).float()).contiguous()
for _ in range(40):
    nr.reduce_noise(y=y, sr=sr, stationary=stationary, n_fft=n_fft)
torch.cuda.synchronize()
