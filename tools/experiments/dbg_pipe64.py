import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
rng = np.random.default_rng(12)
n = 600000 * 8 + 4321
t = np.arange(n) / 48000.0
y = (0.1 * rng.standard_normal((1, n)) + 0.4 * np.sin(2 * np.pi * 700.0 * t)[None, :]).astype(np.float32)[0]
os.environ["NOISEREDUCE_AMD_PIPELINE"] = "0"
ref = nr.reduce_noise(y=y, sr=48000, stationary=False, precision="float64")
ref2 = nr.reduce_noise(y=y, sr=48000, stationary=False, precision="float64")
print("one-upload repeatable:", np.array_equal(ref, ref2))
os.environ["NOISEREDUCE_AMD_PIPELINE"] = "1"
os.environ["NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES"] = str(2 * 600000 * 4)
got = nr.reduce_noise(y=y, sr=48000, stationary=False, precision="float64")
d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
idx = np.flatnonzero(got != ref)
print("differing samples", idx.size, "max abs diff", d.max(), "first/last", idx[:5], idx[-5:])
if idx.size:
    ch = idx // 600000
    print("chunks with differences", np.unique(ch), "positions in chunk (min/max)", (idx % 600000).min(), (idx % 600000).max())
# device-resident sub-range vs whole
import torch
yd = torch.from_numpy(y).cuda()
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
kw = dict(sr=48000, prop_decrease=1.0, chunk_size=600000, padding=30000, n_fft=1024, win_length=None, hop_length=None,
          time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1,
          thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, precision="float64")
sg = SpectralGateNonStationary(y=yd, **kw)
whole = sg.get_traces().cpu().numpy()
part = sg.get_traces(start_frame=1200000, end_frame=2400000).cpu().numpy()
print("tensor path whole == one-upload:", np.array_equal(whole, ref), " sub-range == whole slice:", np.array_equal(part, whole[1200000:2400000]),
      np.abs(part.astype(np.float64) - whole[1200000:2400000]).max())
