#!/usr/bin/env python3
"""configs[2] per-kernel times: two-pass mask (nonstat.hpp) vs the segmented-scan + tiled-smoothing kernels
(SG_OPT_FORCE_UNFUSED keeps them)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
y = bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0))
sg = SpectralGateNonStationary(y=y, sr=48000, chunk_size=600000, padding=30000, n_fft=1024, win_length=None, hop_length=None,
                               time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
                               thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None,
                               prop_decrease=1.0, use_tqdm=False, n_jobs=1)
g = sg._gate
for old in (0, 1, 0, 1):
    g.set_option(_ffi.SG_OPT_FORCE_UNFUSED, old)
    for _ in range(3): sg.get_traces()
    g.profile_read(reset=True); g.profile_enable(True)
    for _ in range(5): sg.get_traces()
    p = g.profile_read(reset=True); g.profile_enable(False)
    print("old kernels" if old else "two-pass   ", round(sum(v[0] for v in p.values()) / 5, 4), {k: round(v[0] / 5, 4) for k, v in p.items()})
g.set_option(_ffi.SG_OPT_FORCE_UNFUSED, 0)
