import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from oracle import spectralgate_oracle as O
KW = dict(sr=48000, prop_decrease=1.0, chunk_size=100000, padding=8000, n_fft=1024, win_length=None, hop_length=None,
          time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
big = torch.from_numpy(np.stack([O.synth_signal(1500000, seed=10 + c, tone_hz=300.0 * (c + 1)) for c in range(4)]).astype(np.float32)).cuda()
ss = SpectralGateStationary(y=big, y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, **KW)
ref = ss.get_traces().clone()
ss._gate.set_option(_ffi.SG_OPT_TILE_ORDER, 0)
bad = 0
for i in range(400):
    out = ss.get_traces()
    if not torch.equal(out, ref):
        bad += 1
        d = (out != ref).nonzero()
        if bad <= 3: print("mismatch call", i, "rows", d[:, 0].unique().tolist(), "first", int(d[:, 1].min()), "last", int(d[:, 1].max()), "count", d.shape[0])
try:
    ss._gate.check_errors(); print("no hand-off error")
except Exception as e: print("ERR", str(e)[:100])
print("bad calls", bad, "of 400")
