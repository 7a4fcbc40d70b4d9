"""Randomised parity sweep of the register-transform fast paths for n_fft = 512 and 2048 (fast512.hpp, fast2048.hpp):
default window / hop only (that is what selects them), everything else random -- sample rate, length, channels, chunk
grid, padding, prop_decrease, smoothing widths, gate parameters, sample dtype, noise clips, sub-range get_traces,
TorchGate forward.  usage: python tests/tools/fuzz_fastn.py [first_seed] [count]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
from noisereduce_amd.torchgate import TorchGate
from oracle import spectralgate_oracle as O

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
worst = {"S": 0.0, "T": 0.0}
fails = 0
ran = {"S": 0, "T": 0, "rejected": 0}
for seed in range(first, first + count):
    r = np.random.default_rng(91000 + seed)
    n_fft = int(r.choice([512, 2048]))
    sr = int(r.choice([8000, 16000, 22050, 44100, 48000, 96000]))
    C = int(r.choice([1, 1, 2, 3]))
    n = int(r.integers(n_fft + 3, 40 * n_fft))
    cs = None if r.random() < 0.15 else int(r.integers(2 * n_fft, 30 * n_fft))
    pad = int(r.integers(0, 6 * n_fft))
    stationary = bool(r.random() < 0.5)
    kw = dict(stationary=stationary, n_fft=n_fft, chunk_size=cs, padding=pad,
              prop_decrease=float(r.choice([1.0, 1.0, 0.7, 0.0])), n_std_thresh_stationary=float(r.choice([1.5, 0.5, 2.5])),
              thresh_n_mult_nonstationary=float(r.choice([2, 1, 3.5])), sigmoid_slope_nonstationary=float(r.choice([10, 4, 25])),
              time_constant_s=float(r.choice([2.0, 0.3, 5.0])))
    hop = n_fft // 4
    sm = r.random()
    f_hz = float(r.choice([1.2, 2.5, 6.0])) * sr / (n_fft / 2) + 1.0
    t_ms = float(r.choice([1.2, 3.0, 7.0])) * hop / sr * 1000.0 + 0.01
    if sm < 0.1: kw.update(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)
    elif sm < 0.2: kw.update(freq_mask_smooth_hz=f_hz, time_mask_smooth_ms=None)
    elif sm < 0.3: kw.update(freq_mask_smooth_hz=None, time_mask_smooth_ms=t_ms)
    else: kw.update(freq_mask_smooth_hz=f_hz, time_mask_smooth_ms=t_ms)
    dtype = str(r.choice(["float32", "float32", "float64"]))
    y = np.stack([O.synth_signal(n, sr=sr, seed=seed * 7 + c, tone_hz=250.0 * (c + 1)).astype(np.float64) for c in range(C)]).astype(dtype)
    if r.random() < 0.2:   # a loud burst: frame pairs / partner lanes with very different norms
        a = int(r.integers(0, max(1, n - 300)))
        y[..., a:a + 300] *= 300.0
    if C == 1: y = y[0]
    try:
        try:
            want = O.reduce_noise_S(y.astype(np.float64), sr, **kw)
        except ValueError:
            try:
                nr.reduce_noise(y=y, sr=sr, **kw)
                raise AssertionError("oracle raised ValueError, engine did not")
            except ValueError:
                ran["rejected"] += 1
                continue
        got = nr.reduce_noise(y=torch.from_numpy(np.ascontiguousarray(y)).cuda() if r.random() < 0.5 else y, sr=sr, **kw)
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        scale = max(float(np.max(np.abs(want))), 1e-6 * float(np.max(np.abs(y))))
        e = float(np.max(np.abs(got.astype(np.float64) - want))) / scale
        worst["S"] = max(worst["S"], e); ran["S"] += 1
        assert e < 1e-4, ("S", e)
        if seed % 4 == 0 and n >= 2 * n_fft:   # TorchGate on the same frame length
            B = int(r.integers(1, 5)); L = int(r.integers(2 * n_fft + 1, 12 * n_fft))
            x = np.stack([O.synth_signal(L, sr=sr, seed=seed * 11 + b) for b in range(B)]).astype(np.float64)
            tkw = dict(nonstationary=not stationary, n_fft=n_fft, prop_decrease=kw["prop_decrease"],
                       freq_mask_smooth_hz=kw["freq_mask_smooth_hz"], time_mask_smooth_ms=kw["time_mask_smooth_ms"])
            try:
                wantT = O.torchgate_T(x, sr, window=torch.hann_window(n_fft).double().numpy(), **tkw)
            except ValueError:
                continue
            gotT = TorchGate(sr=sr, **tkw).cuda()(torch.from_numpy(x).cuda()).cpu().numpy()
            eT = O.rel_err(gotT, wantT) if np.max(np.abs(wantT)) > 0 else float(np.max(np.abs(gotT)))
            worst["T"] = max(worst["T"], eT); ran["T"] += 1
            assert eT < 1e-4, ("T", eT)
    except Exception as ex:   # noqa: BLE001
        fails += 1
        print("FAIL", seed, type(ex).__name__, str(ex)[:160], dict(n_fft=n_fft, sr=sr, C=C, n=n, dtype=dtype, **kw), flush=True)
print("done", count, "seeds,", fails, "failures, compared", ran, "worst", worst)
