"""Non-finite samples: where does the output go NaN, and what is the rest, in the engine and in the oracle?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
from noisereduce_amd.torchgate import TorchGate
from oracle import spectralgate_oracle as O
rng = np.random.default_rng(0)
n = 48000 * 3


def report(tag, got, want):
    gn, wn = ~np.isfinite(got), ~np.isfinite(want)
    both = ~gn & ~wn
    scale = max(1e-3, np.abs(want[both]).max()) if both.any() else 1.0
    err = np.abs(got[both] - want[both]).max() / scale if both.any() else 0.0
    print(tag, "| non-finite: engine", int(gn.sum()), "oracle", int(wn.sum()),
          "same set" if np.array_equal(gn, wn) else "DIFFERENT (engine-only %d, oracle-only %d)" % ((gn & ~wn).sum(), (wn & ~gn).sum()),
          "| err on the finite rest / max(1e-3, peak) %.2e" % err, "| peak of the rest", float(np.abs(want[both]).max()) if both.any() else None)


for what in (np.nan, np.inf):
    for stationary in (True, False):
        y = (0.1 * rng.standard_normal(n)).astype(np.float32)
        y[70000] = what
        got = nr.reduce_noise(y=y, sr=48000, stationary=stationary, n_fft=1024)
        with np.errstate(all="ignore"):
            want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=stationary, n_fft=1024)
        report("%s in the signal, %s" % (what, "stationary" if stationary else "non-stationary"), got, want)
    # several chunks: only the chunks that see the sample are gated
    y = (0.1 * rng.standard_normal(n)).astype(np.float32)
    y[70000] = what
    kw = dict(sr=48000, stationary=True, n_fft=1024, chunk_size=30000, padding=3000)
    got = nr.reduce_noise(y=y, **kw)
    with np.errstate(all="ignore"):
        want = O.reduce_noise_S(y.astype(np.float64), **kw)
    report("%s in the signal, stationary, 30000-sample chunks" % what, got, want)
    # clean signal, noise clip with the sample
    y = (0.1 * rng.standard_normal(n)).astype(np.float32)
    yn = (0.1 * rng.standard_normal(48000)).astype(np.float32)
    yn[5000] = what
    got = nr.reduce_noise(y=y, sr=48000, y_noise=yn, stationary=True, n_fft=1024)
    with np.errstate(all="ignore"):
        want = O.reduce_noise_S(y.astype(np.float64), 48000, y_noise=yn.astype(np.float64), stationary=True, n_fft=1024)
    report("%s in the noise clip, stationary" % what, got, want)
    # TorchGate: one row of a batch
    x = (0.1 * rng.standard_normal((8, 16000))).astype(np.float32)
    x[3, 7000] = what
    tg = TorchGate(sr=16000, nonstationary=False).cuda()
    got = tg(torch.from_numpy(x).cuda()).cpu().numpy()
    with np.errstate(all="ignore"):
        want = O.torchgate_T(x.astype(np.float64), 16000, nonstationary=False, window=torch.hann_window(1024).double().numpy())
    report("%s in row 3 of a TorchGate batch" % what, got, want)
