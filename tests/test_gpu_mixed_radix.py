"""Frame lengths that are not powers of two (the reference hands any n_fft to pocketfft,
stationary.py:87-93).  Round 6: n_fft = 2 N with N = 2^a 3^b 5^c 7^d 11^e 13^f <= 2048 runs on the mixed-radix kernels of
noisereduce_amd/csrc/mixed.hpp (run-time radix schedule, same surrounding pipeline as the power-of-two LDS path -- incl. the
fused bit-mask path) instead of chirp-z.  Checked here: the transform itself against numpy's rfft (float64 tap <= 1e-13 of
the column's peak), both gates and TorchGate against the oracle, mask bits of the float32 decision kernel identical to the
float64 one, and the chirp-z path on the same inputs (SG_NO_MIXED_RADIX=1 at handle creation)."""
import os

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4

# n_fft -> radix schedule of N = n_fft / 2 (documentation of what each case exercises)
SIZES = {400: "8 5 5", 1000: "4 5 5 5", 1536: "8 8 4 3", 3000: "4 5 5 5 3", 100: "2 5 5", 96: "8 2 3", 448: "8 4 7",
         352: "8 2 11", 416: "8 2 13", 24: "4 3", 3600: "8 5 5 3 3", 4000: "8 2 5 5 5", 1200: "8 5 5 3"}


@pytest.fixture
def nr():
    import noisereduce_amd
    return noisereduce_amd


def _gate_S(n_fft, sr=48000, stationary=True, **kw):
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    base = dict(sr=sr, prop_decrease=1.0, chunk_size=kw.pop("chunk_size", 60000), padding=kw.pop("padding", 8000), n_fft=n_fft,
                win_length=kw.pop("win_length", None), hop_length=kw.pop("hop_length", None), time_constant_s=2.0,
                freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    y = kw.pop("y")
    if stationary:
        return SpectralGateStationary(y=y, y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, **base)
    return SpectralGateNonStationary(y=y, thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, **base)


@pytest.mark.parametrize("n_fft", sorted(SIZES))
def test_float64_transform_against_rfft(n_fft):
    """sg_stft (float64 tap of the engine's own transform) against numpy.fft.rfft of the same windowed frames."""
    sr = 40 * n_fft     # (any rate at which the reference's smoothing widths exist: base.py:99-128)
    y = O.synth_signal(max(6 * n_fft, 4000), sr=sr, seed=n_fft, tone_hz=sr / 53.0).astype(np.float64)
    sg = _gate_S(n_fft, sr=sr, y=y, chunk_size=len(y), padding=0)
    Z = sg._gate.stft(torch.from_numpy(y[None, :]).cuda()).cpu().numpy()[0].T      # (F, T)
    W = n_fft
    want = O.stft_scipy(y, n_fft, W, W // 4)
    assert Z.shape == want.shape
    assert np.max(np.abs(Z - want)) <= 1e-13 * max(1.0, np.max(np.abs(want))) * 10, np.max(np.abs(Z - want))


@pytest.mark.parametrize("stationary", [True, False])
@pytest.mark.parametrize("n_fft,sr,n,extra", [
    (400, 16000, 60000, dict(chunk_size=20000, padding=3000)),
    (1000, 48000, 90000, dict(chunk_size=40000, padding=6000)),
    (1536, 44100, 70000, dict(chunk_size=30000, padding=5000, win_length=1200, hop_length=250)),
    (3000, 48000, 120000, dict(chunk_size=50000, padding=8000, time_mask_smooth_ms=200)),
    (100, 8000, 12000, dict(chunk_size=5000, padding=700)),
    (448, 22050, 40000, dict(chunk_size=15000, padding=2000)),
    (352, 16000, 30000, dict(chunk_size=12000, padding=1500)),
    (416, 16000, 30000, dict(chunk_size=12000, padding=1500)),
    (3600, 48000, 130000, dict(chunk_size=60000, padding=9000, time_mask_smooth_ms=200)),
])
def test_reduce_noise_matches_the_oracle(nr, n_fft, sr, n, extra, stationary):
    y = O.synth_signal(n, sr=sr, seed=n_fft + 1, tone_hz=sr / 31.0).astype(np.float32)
    kw = dict(n_fft=n_fft, **extra)
    got = nr.reduce_noise(y=y, sr=sr, stationary=stationary, **kw)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=stationary, **kw)
    assert got.shape == y.shape and got.dtype == y.dtype
    assert O.rel_err(got, want) < TOL


@pytest.mark.parametrize("n_fft", [400, 1000, 1536, 448])
def test_decisions_equal_the_float64_decisions(n_fft):
    """k_decide_mr (float32 transform + exact float64 re-evaluation of ambiguous cells) must give the bits of
    k_stft_bits_mr<decide> (float64 throughout) -- a burst next to quiet frames, and a steady tone."""
    from noisereduce_amd import _ffi
    sr, n = 48000, 200000
    rng = np.random.default_rng(n_fft)
    y = (0.05 * rng.standard_normal(n)).astype(np.float32)
    y[50000:51000] *= 200.0
    y[120000:] += (0.3 * np.sin(2 * np.pi * 1000.0 * np.arange(n - 120000) / sr)).astype(np.float32)
    sg = _gate_S(n_fft, y=y, chunk_size=80000, padding=10000)
    a = sg.get_traces()
    bits_a = sg._gate.debug_field(3)
    d0, d1 = sg._gate.debug_range()
    try:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
        b = sg.get_traces()
        bits_b = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
    assert bits_a.shape == bits_b.shape and d1 - d0 > 50
    assert np.count_nonzero(bits_a[:, d0:d1] != bits_b[:, d0:d1]) == 0
    assert np.array_equal(a, b)


@pytest.mark.parametrize("n_fft,stationary", [(400, True), (400, False), (1000, True), (1536, False)])
def test_mixed_radix_against_chirp_z(nr, monkeypatch, n_fft, stationary):
    """The same call on the chirp-z kernels (the path of rounds 1-5, kept for odd n / large prime factors): two
    independent transforms must agree far inside the parity bar."""
    from noisereduce_amd import _ffi
    sr, n = 48000, 150000
    y = O.synth_signal(n, sr=sr, seed=5, tone_hz=1234.0).astype(np.float32)
    kw = dict(n_fft=n_fft, chunk_size=60000, padding=9000)
    got = nr.reduce_noise(y=y, sr=sr, stationary=stationary, **kw)
    monkeypatch.setenv("SG_NO_MIXED_RADIX", "1")
    _ffi.clear_gate_cache()
    try:
        ref = nr.reduce_noise(y=y, sr=sr, stationary=stationary, **kw)
    finally:
        monkeypatch.delenv("SG_NO_MIXED_RADIX")
        _ffi.clear_gate_cache()
    assert O.rel_err(got, ref) < 5e-6


@pytest.mark.parametrize("kw", [dict(n_fft=400), dict(n_fft=400, nonstationary=True), dict(n_fft=1000, win_length=800, hop_length=200),
                                dict(n_fft=96)])
def test_torchgate(kw):
    from noisereduce_amd.torchgate import TorchGate
    torch.manual_seed(7)
    B, L, sr = 5, 16000, 16000
    x = (0.1 * torch.randn(B, L, dtype=torch.float64) + 0.5 * torch.sin(2 * np.pi * 440 * torch.arange(L) / sr)).float()
    tg = TorchGate(sr=sr, **kw).cuda()
    xg = x.cuda().requires_grad_()
    y = tg(xg)
    want = O.torchgate_T(x.numpy().astype(np.float64), sr, **kw)
    assert O.rel_err(y.detach().cpu().numpy(), want) < TOL
    y.sum().backward()
    assert torch.isfinite(xg.grad).all() and xg.grad.shape == x.shape
