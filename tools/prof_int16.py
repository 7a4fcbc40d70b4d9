#!/usr/bin/env python3
"""configs[1]-sized stationary reduce_noise of an int16 recording, a few calls (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noisereduce_amd as nr, bench
y = (bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0)) * 20000).to(torch.int16)
for _ in range(8):
    nr.reduce_noise(y=y, sr=48000, stationary=True)
torch.cuda.synchronize()
