"""Randomised parity sweep (deterministic seeds): geometry, chunk grid, channels, dtypes and both
gates against the oracle.  Complements the hand-picked cases of test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n_fft = int(rng.choice([128, 256, 400, 512, 600, 1000, 1024, 1024, 1024, 2048, 777]))
    win = n_fft if rng.random() < 0.6 else int(rng.integers(n_fft // 2, n_fft + 1))
    hop = win // 4 if rng.random() < 0.6 else int(rng.integers(max(1, win // 8), win // 2 + 1))
    sr = int(rng.choice([16000, 22050, 44100, 48000]))
    C = int(rng.choice([1, 1, 2, 3]))
    n = int(rng.integers(6 * n_fft, 60000))
    cs = int(rng.integers(4 * n_fft, 30000))
    pad = int(rng.integers(0, 3 * n_fft))
    stationary = bool(rng.random() < 0.5)
    kw = dict(stationary=stationary, n_fft=n_fft, win_length=win, hop_length=hop, chunk_size=cs, padding=pad,
              prop_decrease=float(rng.choice([1.0, 1.0, 0.7])))
    # smoothing widths must be >= 1 bin / 1 frame (base.py:105-123): scale them to the geometry
    kw["freq_mask_smooth_hz"] = float(rng.choice([1.5, 3.0, 5.5])) * sr / (n_fft / 2) + 1.0
    kw["time_mask_smooth_ms"] = float(rng.choice([1.5, 2.5, 6.0])) * hop / sr * 1000.0 + 0.01
    dtype = rng.choice(["float32", "float64", "int16"])
    return seed, sr, C, n, dtype, kw


@pytest.mark.parametrize("seed", range(28))
def test_random_configuration_matches_oracle(seed):
    import noisereduce_amd as nr
    seed, sr, C, n, dtype, kw = _case(seed)
    y = np.stack([O.synth_signal(n, seed=seed * 7 + c, tone_hz=300.0 * (c + 1)).astype(np.float64) for c in range(C)])
    if dtype == "int16":
        y = np.round(y * 20000).astype(np.int16)
    else:
        y = y.astype(dtype)
    if C == 1 and seed % 2 == 0:
        y = y[0]
    try:
        want = O.reduce_noise_S(y.astype(np.float64), sr, **kw)
    except ValueError:
        with pytest.raises(ValueError):
            nr.reduce_noise(y=y, sr=sr, **kw)
        return
    got = nr.reduce_noise(y=y, sr=sr, **kw)
    assert got.shape == y.shape and got.dtype == y.dtype
    if dtype == "int16":
        # truncation to int16 (base.py:218-226) of a float64 result: the truncated oracle, bit for bit (float64 pipeline).
        # (A sample whose float64 value sits within ~1e-9 of an integer could still fall either way: allow a handful.)
        # (samples whose float64 value lies within 1e-9 of an integer -- mask exactly 1: the integer input reconstructed to
        # ~1e-12 -- fall on either side by the reference's own rounding noise: within one count there, equal elsewhere)
        diff = got.astype(np.int64) - want.astype(np.int16).astype(np.int64)
        decided = np.abs(want - np.round(want)) > 1e-9
        assert np.max(np.abs(diff)) <= 1 and np.count_nonzero(diff[decided]) == 0, np.count_nonzero(diff[decided])
    else:
        assert O.rel_err(got.astype(np.float64), want) < TOL, (kw, sr, C, n)
    # tensor input on the device: same numbers as the numpy path
    if dtype != "int16":
        got_t = nr.reduce_noise(y=torch.from_numpy(np.ascontiguousarray(y)).cuda(), sr=sr, **kw)
        assert np.array_equal(got_t.cpu().numpy(), got)


def _case_T(seed):
    rng = np.random.default_rng(5000 + seed)
    n_fft = int(rng.choice([256, 400, 512, 601, 1024, 1024, 1024, 2048]))
    win = n_fft if rng.random() < 0.7 else int(rng.integers(n_fft // 2, n_fft + 1))
    hop = win // 4 if rng.random() < 0.7 else int(rng.integers(max(1, win // 8), win // 3 + 1))
    sr = int(rng.choice([8000, 16000, 22050, 48000]))
    B = int(rng.integers(1, 6))
    L = int(rng.integers(2 * win + 10, 2 * win + 12000))
    kw = dict(nonstationary=bool(rng.random() < 0.5), n_fft=n_fft, win_length=win, hop_length=hop,
              prop_decrease=float(rng.choice([1.0, 1.0, 0.6])),
              n_movemean_nonstationary=int(rng.integers(3, 25)),
              freq_mask_smooth_hz=float(rng.choice([1.5, 3.0, 8.0])) * sr / (n_fft / 2) + 1.0,
              time_mask_smooth_ms=float(rng.choice([1.5, 3.0, 5.0])) * hop / sr * 1000.0 + 0.01)
    xn = None
    if not kw["nonstationary"] and rng.random() < 0.5:
        xn = (int(rng.choice([1, B])), int(rng.integers(2 * win + 5, 2 * win + 6000)))
    return seed, sr, B, L, xn, bool(rng.random() < 0.5), kw


@pytest.mark.parametrize("seed", range(20))
def test_random_torchgate_matches_oracle(seed):
    from noisereduce_amd.torchgate import TorchGate
    seed, sr, B, L, xn_shape, f32, kw = _case_T(seed)
    rng = np.random.default_rng(seed)
    t = np.arange(L) / sr
    x = (0.1 * rng.standard_normal((B, L)) + 0.4 * np.sin(2 * np.pi * 0.02 * sr * t)[None, :])
    x = x.astype(np.float32).astype(np.float64)
    xn = None if xn_shape is None else (0.1 * rng.standard_normal(xn_shape)).astype(np.float32).astype(np.float64)
    W = kw["win_length"]
    want = O.torchgate_T(x, sr, xn=xn, window=torch.hann_window(W).double().numpy(), **kw)
    tg = TorchGate(sr=sr, **kw).cuda()
    dt = torch.float32 if f32 else torch.float64
    got = tg(torch.from_numpy(x).to(dt).cuda(), None if xn is None else torch.from_numpy(xn).to(dt).cuda())
    assert got.dtype == dt and tuple(got.shape) == want.shape
    assert O.rel_err(got.double().cpu().numpy(), want) < TOL, (kw, sr, B, L, xn_shape)


def test_odd_input_containers():
    """Fortran-ordered / strided arrays, lists, float16 and int64 samples, strided, transposed and CPU
    tensors: same numbers as for a plain float64 array (output dtype and container follow the input)."""
    import noisereduce_amd as nr
    n = 50000
    base = np.stack([O.synth_signal(n, seed=s).astype(np.float64) for s in (1, 2)])
    kw = dict(sr=48000, stationary=True, chunk_size=20000, padding=3000)
    want = O.reduce_noise_S(base, **kw)

    def check(y, w=want, tol=TOL, dtype=np.float64, tensor=False):
        got = nr.reduce_noise(y=y, **kw)
        assert isinstance(got, torch.Tensor) == tensor
        g = got.cpu().numpy() if tensor else got
        assert g.dtype == dtype and g.shape == w.shape
        assert O.rel_err(g.astype(np.float64), w) < tol

    check(np.asfortranarray(base))
    check(np.repeat(base, 2, axis=1)[:, ::2])
    check(base[0].tolist(), O.reduce_noise_S(base[0], **kw))
    h = base.astype(np.float16)
    check(h, O.reduce_noise_S(h.astype(np.float64), **kw), 2e-3, np.float16)
    i64 = (base * 30000).astype(np.int64)
    check(i64, np.trunc(O.reduce_noise_S(i64.astype(np.float64), **kw)), 1e-3, np.int64)
    check(torch.from_numpy(np.repeat(base, 2, axis=1)).cuda()[:, ::2], tensor=True)
    check(torch.from_numpy(np.ascontiguousarray(base.T)).cuda().T, tensor=True)
    check(torch.from_numpy(base), tensor=True)          # CPU tensor in -> tensor out
