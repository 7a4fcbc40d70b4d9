"""The five scenarios of the reference's own test file (/root/reference/test_reduction.py: fish.wav
plus band-limited noise through reduce_noise), run on the HIP engine -- with the assertions the
reference's tests lack: every result is compared with the CPU oracle.  The recording comes from the
committed fixture (tests/golden/S_fish.npz holds assets/fish.wav), the noise is seeded."""
import os

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def band_limited_noise(min_freq, max_freq, samples, samplerate, seed=0):
    """Unit-RMS noise whose spectrum is flat inside [min_freq, max_freq] and empty outside (random
    phases on the retained rfft bins)."""
    rng = np.random.default_rng(seed)
    freqs = np.fft.rfftfreq(samples, 1.0 / samplerate)
    spec = np.where((freqs >= min_freq) & (freqs <= max_freq), np.exp(2j * np.pi * rng.random(freqs.size)), 0.0)
    x = np.fft.irfft(spec, n=samples)
    return x / np.sqrt(np.mean(x * x))


@pytest.fixture(scope="module")
def noisy_fish():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "S_fish.npz"))
    rate, data = int(g["rate"]), g["data"].astype(np.float64)
    noise = band_limited_noise(2000, 12000, len(data), rate, seed=7) * 10 * 50.0
    return rate, data + noise, noise


def _check(y, rate, want_kw, **kw):
    import noisereduce_amd as nr
    got = nr.reduce_noise(y=y, sr=rate, **kw)
    want = O.reduce_noise_S(y, rate, **want_kw)
    assert got.shape == y.shape and got.dtype == y.dtype
    assert O.rel_err(got, want) < TOL
    return got


def test_reduce_generated_noise_stationary_with_noise_clip(noisy_fish):
    rate, y, noise = noisy_fish
    clip = noise[: rate * 2]                          # 2 s of the noise alone
    kw = dict(y_noise=clip, stationary=True)
    out = _check(y, rate, kw, **kw)
    assert np.std(out) < np.std(y)                    # it does remove energy


def test_reduce_generated_noise_stationary_without_noise_clip(noisy_fish):
    rate, y, _ = noisy_fish
    _check(y, rate, dict(stationary=True), stationary=True)


def test_reduce_generated_noise_nonstationary(noisy_fish):
    rate, y, _ = noisy_fish
    _check(y, rate, dict(stationary=False), stationary=False)


def test_reduce_generated_noise_batches(noisy_fish):
    rate, y, _ = noisy_fish
    kw = dict(stationary=False, chunk_size=30000)
    _check(y, rate, kw, **kw)


def test_reduce_torch_stationary(noisy_fish):
    """use_torch=True routes through StreamedTorchGate/TorchGate (reference: device='cpu'; here the GPU).
    Checked against the torchgate oracle applied per padded chunk, like the reference's chunk loop."""
    import noisereduce_amd as nr
    rate, y, _ = noisy_fish
    cs, pad = 30000, 30000
    got = nr.reduce_noise(y=y, sr=rate, stationary=True, chunk_size=cs, use_torch=True, device="cuda")
    assert got.shape == y.shape and got.dtype == y.dtype
    w = torch.hann_window(1024).double().numpy()
    want = np.zeros_like(y)
    for i in range(-(-len(y) // cs)):
        chunk = O.read_chunk(y[None, :], i * cs - pad, (i + 1) * cs + pad)
        res = O.torchgate_T(chunk, rate, window=w)
        full = np.zeros_like(chunk)
        full[:, :res.shape[1]] = res
        n = min(cs, len(y) - i * cs)
        want[i * cs:i * cs + n] = full[0, pad:pad + n]
    assert O.rel_err(got, want) < TOL
