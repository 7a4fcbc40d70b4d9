#!/usr/bin/env python3
"""Time get_traces (no noise statistics) of configs[1] with the library named by SG_LIB_PATH; per-kernel table."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
import bench
dev = torch.device("cuda", 0)
y = bench.synth_on_device(bench.N_PER_GPU, 1234, dev)
KW = dict(y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, clip_noise_stationary=True, chunk_size=600000,
          padding=30000, n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
          time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
sg = SpectralGateStationary(y=y, sr=48000, **KW)
g = sg._gate
g.set_option(_ffi.SG_OPT_FORCE_SPLIT, int(os.environ.get("SPLIT", "0")))
for _ in range(5): sg.get_traces()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): sg.get_traces()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
g.profile_enable(True)
for _ in range(5): sg.get_traces()
prof = g.profile_read()
print(json.dumps({"lib": os.path.basename(os.environ.get("SG_LIB_PATH", "default")), "ms": round(ms, 4),
                  "kernels": {k: round(v[0] / 5, 4) for k, v in prof.items()}}))
