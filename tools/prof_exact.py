#!/usr/bin/env python3
"""Per-stage kernel times of the float64 pipeline (int16 recording, 10 min): stationary and non-stationary."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noisereduce_amd as nr, bench
from noisereduce_amd import _ffi
y = (bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0)) * 20000).to(torch.int16)
for stat in (True, False):
    for _ in range(2): nr.reduce_noise(y=y, sr=48000, stationary=stat)
    g = [g for g in _ffi._GATE_CACHE.values() if bool(g.params.stationary) == stat][-1]
    g.profile_read(reset=True); g.profile_enable(True)
    for _ in range(3): nr.reduce_noise(y=y, sr=48000, stationary=stat)
    p = g.profile_read(reset=True); g.profile_enable(False)
    print("stationary" if stat else "non-stationary", {k: round(v[0] / 3, 4) for k, v in p.items()})
