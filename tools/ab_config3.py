#!/usr/bin/env python3
"""configs[2] (non-stationary, 10 min) with the library SG_LIB_PATH names: wall time per call (median of blocks of back-to-back
calls) and the profiled per-stage times -- one JSON line.  For A/B builds (tools/ab_build.sh): run once per library, alternating."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
y = bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0))
sg = SpectralGateNonStationary(y=y, sr=48000, chunk_size=600000, padding=30000, n_fft=1024, win_length=None, hop_length=None,
                               time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
                               thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None,
                               prop_decrease=1.0, use_tqdm=False, n_jobs=1)
g = sg._gate
for _ in range(int(os.environ.get("WARM", "300"))): sg.get_traces()
torch.cuda.synchronize()
blocks = []
for _ in range(7):
    t0 = time.perf_counter()
    for _ in range(40): out = sg.get_traces()
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - t0) / 40 * 1e3)
g.profile_read(reset=True); g.profile_enable(True)
for _ in range(10): sg.get_traces()
p = g.profile_read(reset=True); g.profile_enable(False)
print(json.dumps({"lib": os.path.basename(os.environ.get("SG_LIB_PATH", "default")), "ms_per_call_median": round(sorted(blocks)[3], 4),
                  "blocks": [round(b, 4) for b in blocks], "stage_ms": {k: round(v[0] / 10, 4) for k, v in p.items()},
                  "checksum": float(out.double().abs().sum())}))
