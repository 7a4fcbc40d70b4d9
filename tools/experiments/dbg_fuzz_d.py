"""Reproduce one case of tests/tools/fuzz_round6.py section D and print where the one-pass gate and the split kernels differ."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import fuzz_round6 as F
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from oracle import spectralgate_oracle as O
seed = int(sys.argv[1])
rng = np.random.default_rng(90000 + seed)
n_fft = int(rng.choice([256, 512, 2048])); C = int(rng.choice([1, 1, 2, 3])); sr = int(rng.choice([48000, 48000, 44100, 32000, 22050, 96000]))
n = int(rng.integers(12 * n_fft, int(rng.choice([60000, 300000, 1500000])))); cs = int(rng.choice([600000, int(rng.integers(6 * n_fft, 200000))]))
pad = int(rng.integers(0, min(cs, 30000) + 1))
y = np.stack([O.synth_signal(n, sr=sr, seed=seed * 7 + c, tone_hz=250.0 * (c + 1)) for c in range(C)]).astype(np.float32)
kind = rng.integers(0, 5)
if kind == 0:
    a = int(rng.integers(0, n - 100)); y[:, a:a + int(rng.integers(50, 20000))] = 0.0
    a = int(rng.integers(0, n - 100)); y[int(rng.integers(0, C)), a:a + int(rng.integers(50, 3000))] *= 300.0
elif kind == 1:
    y[int(rng.integers(0, C)), int(rng.integers(0, n))] = np.nan
dtype = str(rng.choice(["float32", "float32", "float64"]))
prop = float(rng.choice([1.0, 1.0, 0.8, 0.35]))
kw = dict(sr=sr, y_noise=None, prop_decrease=prop, n_std_thresh_stationary=float(rng.choice([1.5, 1.5, 0.5, 3.0])), chunk_size=cs,
          clip_noise_stationary=True, padding=pad, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0,
          freq_mask_smooth_hz=float(rng.choice([500, 500, 200, 100])), time_mask_smooth_ms=float(rng.choice([50, 50, 20, 90])),
          tmp_folder=None, use_tqdm=False, n_jobs=1)
print(dict(n_fft=n_fft, C=C, sr=sr, n=n, cs=cs, pad=pad, kind=int(kind), dtype=dtype, prop=prop, thr=kw["n_std_thresh_stationary"], hz=kw["freq_mask_smooth_hz"], ms=kw["time_mask_smooth_ms"]))
yy = y.astype(dtype)
sg = SpectralGateStationary(y=yy if C > 1 else yy[0], **kw)
g = sg._gate
args = {}
if rng.integers(0, 4) == 0 and n > 40000:
    a = int(rng.integers(0, n // 2)); args = dict(start_frame=a, end_frame=int(rng.integers(a + 3000, n)))
print("args", args, "nf/nt", sg._n_grad_freq, sg._n_grad_time)
outs = []
for rep in range(3):
    outs.append(np.atleast_2d(sg.get_traces(**args)))
g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
b = np.atleast_2d(sg.get_traces(**args))
g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
g.check_errors()
for i, o in enumerate(outs):
    d = ~((o == b) | (np.isnan(o) & np.isnan(b)))
    idx = np.argwhere(d)
    print("rep", i, "differs from split at", idx.shape[0], "samples", (idx[:3].tolist(), idx[-3:].tolist()) if idx.size else "", "max abs", float(np.nanmax(np.abs(o - b))) if idx.size else 0.0,
          "equal to rep0", np.array_equal(o, outs[0], equal_nan=True))
