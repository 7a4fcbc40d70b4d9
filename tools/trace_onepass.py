#!/usr/bin/env python3
"""Phase trace of k_gate_onepass on configs[1] (library built with -DOP_TRACE=1: tools/ab_build.sh trace -DOP_TRACE=1).
   SG_LIB_PATH=noisereduce_amd/_ab/lib_trace.so MODE=<SG_OPT_TILE_ORDER> python tools/trace_onepass.py
The library prints the per-phase average shader cycles of the last launch at exit (stderr)."""
import os, sys, time
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
sg._gate.set_option(_ffi.SG_OPT_TILE_ORDER, int(os.environ.get("MODE", "0")))
for _ in range(200): sg.get_traces()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): sg.get_traces()
torch.cuda.synchronize()
print("MODE", os.environ.get("MODE", "0"), "ms per call (trace build)", round((time.perf_counter() - t0) / 50 * 1e3, 4), file=sys.stderr)
