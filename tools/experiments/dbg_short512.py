import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from oracle import spectralgate_oracle as O
sr = 48000
for n_fft in (512, 256, 2048):
  for n, cs, pad in ((3 * n_fft // 2, 600000, 0), (3 * n_fft // 2, 600000, 30000), (2 * n_fft, 600000, 0), (5 * n_fft // 2 + 3, 600000, 0)):
    y = O.synth_signal(n, sr=sr, seed=n % 97, tone_hz=900.0).astype(np.float32)
    kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=cs, clip_noise_stationary=True,
              padding=pad, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
              time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    got = sg.get_traces().clone().cpu().numpy()
    bits1 = sg._gate.debug_field(3); r1 = sg._gate.debug_range()
    sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
    split = sg.get_traces().clone().cpu().numpy()
    bits2 = sg._gate.debug_field(3); r2 = sg._gate.debug_range()
    sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=n_fft, chunk_size=cs, padding=pad)
    d = np.abs(got - split)
    idx = np.nonzero(d)[-1]
    print(n_fft, n, cs, pad, "maxdiff", d.max(), "n_diff", idx.size, "first/last", (idx.min(), idx.max()) if idx.size else None,
          "err one-pass", O.rel_err(got, want), "err split", O.rel_err(split, want), "ranges", r1, r2,
          "bits equal", np.array_equal(bits1[:, r1[0]:r1[1]], bits2[:, r1[0]:r1[1]]), bits1.shape)
