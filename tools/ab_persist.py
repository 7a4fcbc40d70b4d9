#!/usr/bin/env python3
"""k_gate_onepass: persistent workgroups (SG_OPT_TILE_ORDER 0) against one ticket-drawn tile per workgroup (2) and
tile = block index (1) on configs[1], same process, alternating blocks: wall time per get_traces() call (no noise
statistics) and the gate kernel's own HIP-event time."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
import bench
dev = torch.device("cuda", 0)
y = bench.synth_on_device(int(os.environ.get("N", bench.N_PER_GPU)), 1234, dev)
KW = dict(y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, clip_noise_stationary=True, chunk_size=600000,
          padding=30000, n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
          time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
sg = SpectralGateStationary(y=y, sr=48000, **KW)
g = sg._gate
MODES = [int(m) for m in os.environ.get("MODES", "0,2,1").split(",")]
BLOCK, ROUNDS = int(os.environ.get("BLOCK", "60")), int(os.environ.get("ROUNDS", "6"))
for _ in range(300): sg.get_traces()   # clock ramp
torch.cuda.synchronize()
wall = {m: [] for m in MODES}
for r in range(ROUNDS):
    for m in MODES:
        g.set_option(_ffi.SG_OPT_TILE_ORDER, m)
        for _ in range(10): sg.get_traces()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(BLOCK): sg.get_traces()
        torch.cuda.synchronize(); wall[m].append((time.perf_counter() - t0) / BLOCK * 1e3)
kern = {}
g.profile_enable(True)
for m in MODES:
    g.set_option(_ffi.SG_OPT_TILE_ORDER, m)
    g.profile_read()
    for _ in range(40): sg.get_traces()
    prof = g.profile_read()
    kern[m] = {k: round(v[0] / 40, 4) for k, v in prof.items() if v[0] > 0}
g.profile_enable(False)
g.set_option(_ffi.SG_OPT_TILE_ORDER, 0)   # (the default)
g.check_errors()
print(json.dumps({"samples": int(y.numel()), "wall_ms_per_call": {m: [round(x, 4) for x in v] for m, v in wall.items()},
                  "wall_ms_median": {m: round(sorted(v)[len(v) // 2], 4) for m, v in wall.items()}, "event_ms": kern}))
