import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import spectralgate_oracle as O
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
y = O.synth_signal(40000, seed=11).astype(np.float64)
kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
          clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
          time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
          use_tqdm=False, n_jobs=1)
sg = SpectralGateStationary(y=y, **kw)
fast = sg.get_traces()
sg._gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 1)
ref = sg.get_traces()
sg._gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 0)
d = fast - ref
print("rel err", np.abs(d).max() / np.abs(ref).max())
Z = O.stft_scipy(d, 1024, 1024, 256)
Zr = O.stft_scipy(ref, 1024, 1024, 256)
mag = np.abs(Z).mean(axis=1); magr = np.abs(Zr).mean(axis=1)
idx = np.argsort(-mag)[:24]
print("worst bins", [(int(i), float(mag[i] / (magr[i] + 1e-12))) for i in idx])
tm = np.abs(Z).mean(axis=0)
print("time profile (frames) max at", int(np.argmax(tm)), tm[:6], tm[len(tm)//2-2:len(tm)//2+2], tm[-6:])
print("bins with ratio>1e-3:", [int(i) for i in np.where(mag > 1e-3 * magr.max())[0]][:80])
print("---- identity test (threshold -400 dB => mask == 1)")
sg._gate.set_noise_threshold(np.full(513, -400.0))
ident = sg.get_traces()
print("identity rel err", np.abs(ident - y).max() / np.abs(y).max())
Z = O.stft_scipy(ident - y, 1024, 1024, 256)
mag = np.abs(Z).mean(axis=1)
print("worst bins", [(int(i), float(mag[i])) for i in np.argsort(-mag)[:10]])
