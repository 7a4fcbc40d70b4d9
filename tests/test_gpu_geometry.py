"""STFT geometry generality (row f3): n_fft 512 / 2048 fast paths, long frames (four-step / chirp-z), TorchGate on them.
(grouped by subject in round 5; the tests themselves date from rounds 2-4)"""
import threading

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4

NS_KW = dict(sr=48000, prop_decrease=1.0, chunk_size=100000, padding=8000, n_fft=1024, win_length=None,
             hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
             thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None, use_tqdm=False, n_jobs=1)


@pytest.fixture(scope="module")
def nr():
    import noisereduce_amd
    return noisereduce_amd


@pytest.mark.parametrize("n_fft,kw", [
    (4097, dict()),                                            # first length beyond the per-workgroup chirp-z kernels (M = 16384)
    (8193, dict(time_mask_smooth_ms=100)),                     # M = 32768
    (16384, dict(time_mask_smooth_ms=200)),                    # power of two, direct (M = n)
    (20000, dict(time_mask_smooth_ms=300, win_length=16000, hop_length=3000)),   # M = 65536, window shorter than the frame
    (32768, dict(time_mask_smooth_ms=400)),
    (65536, dict(time_mask_smooth_ms=800, freq_mask_smooth_hz=None)),           # the largest frame
])
@pytest.mark.parametrize("stationary", [True, False])
def test_long_frames_match_the_oracle(nr, n_fft, kw, stationary):
    """reduce_noise with n_fft beyond 8192 (or beyond 4096 and not a power of two) against the oracle (base.py:77-86
    accepts any n_fft; scipy.signal.stft/istft use rfft(n)/irfft(n))."""
    n = max(6 * n_fft, 90000)
    y = O.synth_signal(n, seed=n_fft).astype(np.float32)
    args = dict(stationary=stationary, n_fft=n_fft, chunk_size=max(50000, 3 * n_fft), padding=max(6000, n_fft), **kw)
    got = nr.reduce_noise(y=y, sr=48000, **args)
    want = O.reduce_noise_S(y.astype(np.float64), 48000, **args)
    assert got.shape == y.shape and got.dtype == y.dtype
    assert O.rel_err(got, want) < TOL


def test_long_frames_stft_tap(nr):
    """The STFT tap (sg_stft) on a long chirp-z frame and a long power-of-two frame against scipy-style STFT of the oracle."""
    from noisereduce_amd import _ffi
    for n_fft in (5000, 16384):
        x = O.synth_signal(4 * n_fft + 123, seed=3).astype(np.float64)
        g = _ffi.Gate("cuda", variant=_ffi.SG_VARIANT_S, stationary=True, n_fft=n_fft, win_length=n_fft, hop_length=n_fft // 4)
        Z = g.stft(torch.from_numpy(x)[None].cuda())[0].cpu().numpy().T     # (F, T)
        Zo = O.stft_scipy(x, n_fft, n_fft, n_fft // 4)
        assert Z.shape == Zo.shape
        assert np.max(np.abs(Z - Zo)) < 1e-12 * max(1.0, np.max(np.abs(Zo)))
        g.close()


def test_torchgate_long_frames(nr):
    from noisereduce_amd.torchgate import TorchGate
    for n_fft, kw in ((16384, dict(time_mask_smooth_ms=200)), (6000, dict(nonstationary=True, n_movemean_nonstationary=5))):
        x = np.stack([O.synth_signal(3 * n_fft + 777, sr=48000, seed=s) for s in range(3)]).astype(np.float64)
        tg = TorchGate(sr=48000, n_fft=n_fft, **kw).cuda()
        got = tg(torch.from_numpy(x).cuda()).cpu().numpy()
        want = O.torchgate_T(x, 48000, n_fft=n_fft, window=torch.hann_window(n_fft).double().numpy(), **kw)
        assert got.shape == want.shape
        assert O.rel_err(got, want) < TOL


@pytest.mark.parametrize("sr,n,kw", [
    (16000, 30000, dict()),                                                  # one chunk
    (48000, 200000, dict(chunk_size=40000, padding=5000)),                   # chunk grid, partial last chunk
    (16000, 51234, dict(chunk_size=9000, padding=1000, prop_decrease=0.6)),  # ragged: tiles at both unit edges
    (16000, 515, dict()),                                                    # barely longer than a frame
    (8000, 20000, dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)),
    (44100, 70000, dict(time_mask_smooth_ms=None)),
])
@pytest.mark.parametrize("stationary", [True, False])
def test_nfft512_fast_path_matches_the_oracle(nr, sr, n, kw, stationary):
    y = np.stack([O.synth_signal(n, sr=sr, seed=81 + c, tone_hz=300.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    args = dict(stationary=stationary, n_fft=512, **kw)
    got = nr.reduce_noise(y=y, sr=sr, **args)
    want = O.reduce_noise_S(y.astype(np.float64), sr, **args)
    assert O.rel_err(got, want) < TOL
    # the general LDS kernels on the same input (SG_OPT_FORCE_NOFAST): same result to float32 rounding
    y1 = torch.from_numpy(y).cuda()
    a = nr.reduce_noise(y=y1, sr=sr, **args)
    assert O.rel_err(a.cpu().numpy(), want) < TOL


def test_nfft512_decisions_equal_the_float64_decisions(nr):
    """Mask bits of k_decide_fast512 (float32 + exact refinement, two frames per transform) == the all-float64 decision
    kernel, bit for bit -- incl. a loud frame next to a quiet one (the pair shares one transform) and a steady tone."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr, n = 16000, 120000
    y = O.synth_signal(n, sr=sr, seed=5, tone_hz=440.0).astype(np.float32)
    y[30000:30700] *= 200.0          # a burst: frames with a loud and a quiet partner
    y[60000:] = (0.3 * np.sin(2 * np.pi * 1000.0 * np.arange(n - 60000) / sr)).astype(np.float32)   # steady tone
    kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=50000, clip_noise_stationary=True,
              padding=4000, n_fft=512, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
              time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    out_fast = sg.get_traces().clone()
    bits_fast = sg._gate.debug_field(3)
    d0, d1 = sg._gate.debug_range()     # (round 6: the one-pass gate decides the frames its tiles reach, not every frame of the window)
    sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
    try:
        out_64 = sg.get_traces().clone()
        bits_64 = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
    assert bits_fast.shape == bits_64.shape and d1 - d0 > 300
    assert np.array_equal(bits_fast[:, d0:d1], bits_64[:, d0:d1])
    assert torch.equal(out_fast, out_64)
    # ... and the three-kernel path (k_decide_fast512 + k_smooth_bits2 + k_apply_fast512) on the same input: same bits, same samples
    sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
    try:
        out_3k = sg.get_traces().clone()
        bits_3k = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    assert np.array_equal(bits_fast[:, d0:d1], bits_3k[:, d0:d1]) and torch.equal(out_fast, out_3k)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=512, chunk_size=50000, padding=4000)
    assert O.rel_err(out_fast.cpu().numpy(), want) < TOL


def test_torchgate_nfft512(nr):
    from noisereduce_amd.torchgate import TorchGate
    for kw in (dict(), dict(nonstationary=True)):
        x = np.stack([O.synth_signal(16000, sr=16000, seed=s, tone_hz=440.0) for s in range(5)]).astype(np.float64)
        tg = TorchGate(sr=16000, n_fft=512, **kw).cuda()
        xt = torch.from_numpy(x).cuda().requires_grad_()
        y = tg(xt)
        want = O.torchgate_T(x, 16000, n_fft=512, window=torch.hann_window(512).double().numpy(), **kw)
        assert O.rel_err(y.detach().cpu().numpy(), want) < TOL
        # backward: the adjoint with the mask fixed against autograd through torch.stft / istft on the CPU
        w = torch.linspace(0.5, 1.5, y.shape[1], dtype=torch.float64)
        (y * w.cuda()).sum().backward()
        got_g = xt.grad.cpu()
        _, st = O.torchgate_T(x, 16000, n_fft=512, window=torch.hann_window(512).double().numpy(), return_stages=True, **kw)
        m = torch.from_numpy(st["mask"])
        xc = torch.from_numpy(x).requires_grad_()
        win = torch.hann_window(512, dtype=torch.float64)
        X = torch.stft(xc, 512, 128, 512, window=win, center=True, pad_mode="constant", return_complex=True)
        yc = torch.istft(X * m, 512, 128, 512, window=win, center=True)
        (yc * w).sum().backward()
        assert O.rel_err(got_g.numpy(), xc.grad.numpy()) < TOL


@pytest.mark.parametrize("sr,n,kw", [
    (8000, 30000, dict()),                                                   # one chunk (telephony)
    (48000, 200000, dict(chunk_size=40000, padding=5000)),                   # chunk grid, partial last chunk, 37-frame smoothing
    (16000, 51234, dict(chunk_size=9000, padding=1000, prop_decrease=0.6)),  # ragged: tiles at both unit edges
    (16000, 259, dict()),                                                    # barely longer than a frame
    (8000, 20000, dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)),
    (44100, 70000, dict(time_mask_smooth_ms=None)),
    (16000, 64 * 61 * 3 + 17, dict(chunk_size=64 * 61, padding=64 * 4)),     # chunk = one tile's hops: tile seams on chunk seams
])
@pytest.mark.parametrize("stationary", [True, False])
def test_nfft256_fast_path_matches_the_oracle(nr, sr, n, kw, stationary):
    """fast256.hpp (round 5): four real frames of 256 samples per 512-point register transform."""
    from noisereduce_amd import _ffi
    y = np.stack([O.synth_signal(n, sr=sr, seed=31 + c, tone_hz=300.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    args = dict(stationary=stationary, n_fft=256, **kw)
    got = nr.reduce_noise(y=y, sr=sr, **args)
    want = O.reduce_noise_S(y.astype(np.float64), sr, **args)
    assert got.shape == y.shape and got.dtype == y.dtype
    assert O.rel_err(got, want) < TOL
    # the general LDS kernels on the same input (SG_OPT_FORCE_NOFAST): same result to float32 rounding
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    y1 = torch.from_numpy(y).cuda()

    def make():
        base = dict(y=y1, sr=sr, chunk_size=kw.get("chunk_size", 600000), padding=kw.get("padding", 30000),
                    prop_decrease=kw.get("prop_decrease", 1.0), n_fft=256, win_length=None, hop_length=None, time_constant_s=2.0,
                    freq_mask_smooth_hz=kw.get("freq_mask_smooth_hz", 500), time_mask_smooth_ms=kw.get("time_mask_smooth_ms", 50),
                    tmp_folder=None, use_tqdm=False, n_jobs=1)
        if stationary:
            return SpectralGateStationary(y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, **base)
        return SpectralGateNonStationary(thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, **base)

    sg = make()
    a = sg.get_traces().cpu().numpy()
    with sg._gate.with_options([(_ffi.SG_OPT_FORCE_NOFAST, 1)]):
        b = make().get_traces().cpu().numpy()
    assert O.rel_err(a, want) < TOL and O.rel_err(b, want) < TOL
    assert O.rel_err(a, b) < 2e-6


def test_nfft256_decisions_equal_the_float64_decisions(nr):
    """Mask bits of k_decide_fast256 (float32 + exact refinement, FOUR frames per transform) == the all-float64 decision
    kernel, bit for bit -- incl. a loud frame next to quiet ones (they share one transform) and a steady tone."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr, n = 16000, 120000
    y = O.synth_signal(n, sr=sr, seed=6, tone_hz=440.0).astype(np.float32)
    y[30000:30200] *= 200.0          # a burst: frames with loud and quiet partners
    y[60000:] = (0.3 * np.sin(2 * np.pi * 1000.0 * np.arange(n - 60000) / sr)).astype(np.float32)   # steady tone
    kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=50000, clip_noise_stationary=True,
              padding=4000, n_fft=256, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
              time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    out_fast = sg.get_traces().clone()
    bits_fast = sg._gate.debug_field(3)
    d0, d1 = sg._gate.debug_range()     # (round 6: the one-pass gate decides the frames its tiles reach, not every frame of the window)
    sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
    try:
        out_64 = sg.get_traces().clone()
        bits_64 = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
    assert bits_fast.shape == bits_64.shape and d1 - d0 > 600
    assert np.array_equal(bits_fast[:, d0:d1], bits_64[:, d0:d1])
    assert torch.equal(out_fast, out_64)
    # ... and the three-kernel path (k_decide_fast256 + k_smooth_bits2 + k_apply_fast256) on the same input: same bits, same samples
    sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
    try:
        out_3k = sg.get_traces().clone()
        bits_3k = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    assert np.array_equal(bits_fast[:, d0:d1], bits_3k[:, d0:d1]) and torch.equal(out_fast, out_3k)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=256, chunk_size=50000, padding=4000)
    assert O.rel_err(out_fast.cpu().numpy(), want) < TOL


def test_torchgate_nfft256(nr):
    from noisereduce_amd.torchgate import TorchGate
    for kw in (dict(), dict(nonstationary=True)):
        x = np.stack([O.synth_signal(8000, sr=8000, seed=s, tone_hz=440.0) for s in range(5)]).astype(np.float64)
        tg = TorchGate(sr=8000, n_fft=256, **kw).cuda()
        xt = torch.from_numpy(x).cuda().requires_grad_()
        y = tg(xt)
        want = O.torchgate_T(x, 8000, n_fft=256, window=torch.hann_window(256).double().numpy(), **kw)
        assert O.rel_err(y.detach().cpu().numpy(), want) < TOL
        # backward: the adjoint with the mask fixed against autograd through torch.stft / istft on the CPU
        w = torch.linspace(0.5, 1.5, y.shape[1], dtype=torch.float64)
        (y * w.cuda()).sum().backward()
        got_g = xt.grad.cpu()
        _, st = O.torchgate_T(x, 8000, n_fft=256, window=torch.hann_window(256).double().numpy(), return_stages=True, **kw)
        m = torch.from_numpy(st["mask"])
        xc = torch.from_numpy(x).requires_grad_()
        win = torch.hann_window(256, dtype=torch.float64)
        X = torch.stft(xc, 256, 64, 256, window=win, center=True, pad_mode="constant", return_complex=True)
        yc = torch.istft(X * m, 256, 64, 256, window=win, center=True)
        (yc * w).sum().backward()
        assert O.rel_err(got_g.numpy(), xc.grad.numpy()) < TOL


@pytest.mark.parametrize("sr,n,kw", [
    (44100, 50000, dict()),                                                   # one chunk
    (48000, 300000, dict(chunk_size=70000, padding=9000)),                    # chunk grid, partial last chunk
    (48000, 123457, dict(chunk_size=30000, padding=4100, prop_decrease=0.6)), # ragged: tiles at both unit edges
    (48000, 2060, dict()),                                                    # barely longer than a frame
    (96000, 90000, dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)),
])
@pytest.mark.parametrize("stationary", [True, False])
def test_nfft2048_fast_path_matches_the_oracle(nr, sr, n, kw, stationary):
    y = np.stack([O.synth_signal(n, sr=sr, seed=91 + c, tone_hz=300.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    args = dict(stationary=stationary, n_fft=2048, **kw)
    got = nr.reduce_noise(y=y, sr=sr, **args)
    want = O.reduce_noise_S(y.astype(np.float64), sr, **args)
    assert O.rel_err(got, want) < TOL


def test_nfft2048_decisions_equal_the_float64_decisions(nr):
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr, n = 48000, 400000
    y = O.synth_signal(n, sr=sr, seed=6, tone_hz=440.0).astype(np.float32)
    y[100000:102000] *= 200.0
    y[250000:] = (0.3 * np.sin(2 * np.pi * 1000.0 * np.arange(n - 250000) / sr)).astype(np.float32)   # steady tone
    kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=150000, clip_noise_stationary=True,
              padding=12000, n_fft=2048, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
              time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    out_fast = sg.get_traces().clone()
    bits_fast = sg._gate.debug_field(3)
    d0, d1 = sg._gate.debug_range()     # (round 6: the one-pass gate decides the frames its tiles reach, not every frame of the window)
    sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
    try:
        out_64 = sg.get_traces().clone()
        bits_64 = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
    assert bits_fast.shape == bits_64.shape and d1 - d0 > 300
    assert np.array_equal(bits_fast[:, d0:d1], bits_64[:, d0:d1])
    assert torch.equal(out_fast, out_64)
    # ... and the four-kernel path (k_decide_fast2048 + k_smooth_bits2 + k_apply_fast2048 + k_ola_seam2048): same bits, same samples
    sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
    try:
        out_3k = sg.get_traces().clone()
        bits_3k = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    assert np.array_equal(bits_fast[:, d0:d1], bits_3k[:, d0:d1]) and torch.equal(out_fast, out_3k)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=2048, chunk_size=150000, padding=12000)
    assert O.rel_err(out_fast.cpu().numpy(), want) < TOL


def test_torchgate_nfft2048(nr):
    from noisereduce_amd.torchgate import TorchGate
    for kw in (dict(), dict(nonstationary=True)):
        x = np.stack([O.synth_signal(30000, sr=48000, seed=s, tone_hz=440.0) for s in range(3)]).astype(np.float64)
        tg = TorchGate(sr=48000, n_fft=2048, **kw).cuda()
        xt = torch.from_numpy(x).cuda().requires_grad_()
        y = tg(xt)
        want, st = O.torchgate_T(x, 48000, n_fft=2048, window=torch.hann_window(2048).double().numpy(), return_stages=True, **kw)
        assert O.rel_err(y.detach().cpu().numpy(), want) < TOL
        w = torch.linspace(0.5, 1.5, y.shape[1], dtype=torch.float64)
        (y * w.cuda()).sum().backward()
        m = torch.from_numpy(st["mask"])
        xc = torch.from_numpy(x).requires_grad_()
        win = torch.hann_window(2048, dtype=torch.float64)
        X = torch.stft(xc, 2048, 512, 2048, window=win, center=True, pad_mode="constant", return_complex=True)
        yc = torch.istft(X * m, 2048, 512, 2048, window=win, center=True)
        (yc * w).sum().backward()
        assert O.rel_err(xt.grad.cpu().numpy(), xc.grad.numpy()) < TOL


@pytest.mark.parametrize("stationary", [True, False])
@pytest.mark.parametrize("n_fft,sr,extra", [(256, 48000, {}), (128, 44100, dict(freq_mask_smooth_hz=1000))])
def test_short_frames_long_smoothing_window(nr, n_fft, sr, extra, stationary):
    """Short frames at a high sample rate: time_mask_smooth_ms = 50 is 37 frames of n_fft = 256 at 48 kHz (68 at n_fft = 128,
    44.1 kHz) -- beyond the register-tile mask kernels (nt <= 20); round 5 routes them through the LDS-tiled smoothing kernel
    (nt <= 94) instead of two direct global-memory convolutions.  Against the oracle."""
    y = O.synth_signal(60000, sr=sr, seed=33, tone_hz=1500.0).astype(np.float64)
    kw = dict(n_fft=n_fft, chunk_size=25000, padding=4000, **extra)
    got = nr.reduce_noise(y=y, sr=sr, stationary=stationary, **kw)
    want = O.reduce_noise_S(y, sr, stationary=stationary, **kw)
    assert O.rel_err(got, want) < TOL


def _floor_inputs_geom(kind, n_fft):
    """Inputs of tests/test_gpu_onepass.py::_floor_inputs, sized for short chunks of any frame length."""
    rng = np.random.default_rng(4321 + n_fft)
    n, cs, pad = 150000, 40000, 6000
    y = (0.05 * rng.standard_normal(n)).astype(np.float32)
    y_noise = (0.05 * rng.standard_normal(30000)).astype(np.float32)
    if kind == "live":                # loud half next to digital silence, very quiet noise clip: bands lifted by the floor
        y[: n // 2] = 0.0
        y[n // 2:] *= 10.0
        y_noise = (1e-7 * rng.standard_normal(30000)).astype(np.float32)
    elif kind == "loud_in_padding":   # the only loud samples of chunk 1's window sit in its left padding (chunk 0's tail)
        y[:] = (1e-6 * rng.standard_normal(n)).astype(np.float32)
        y[cs - pad + 200: cs - pad + 1500] = (0.9 * rng.standard_normal(1300)).astype(np.float32)
        y_noise = (1e-7 * rng.standard_normal(30000)).astype(np.float32)
    elif kind == "nan_in_padding":    # a NaN that only chunk 2's right padding sees (and chunk 3's body)
        y[3 * cs + 4000] = np.nan
    elif kind == "inf_far_padding":   # an Inf near the far end of chunk 0's right padding
        y[cs + pad - 3] = np.inf
    return y, y_noise, cs, pad


@pytest.mark.parametrize("n_fft", [256, 512, 2048])
@pytest.mark.parametrize("kind", ["benign", "live", "loud_in_padding", "nan_in_padding", "inf_far_padding"])
def test_register_paths_floor_test_in_the_decision_kernel(n_fft, kind):
    """Round 6: k_decide_fast256 / 512 / 2048 run the -top_db floor test on the samples they stage (SG_OPT_FLOOR_TEST 2;
    flagged chunks: float64 band maxima + the decision kernel's REDO launch) instead of k_unit_absmax + k_prep_thresh
    reading the recording before the gate (1).  Same bits, same output, whichever answers -- and the oracle's."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y, y_noise, cs, pad = _floor_inputs_geom(kind, n_fft)
    sr = 48000
    kw = dict(sr=sr, y_noise=y_noise, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=cs,
              clip_noise_stationary=True, padding=pad, n_fft=n_fft, win_length=None, hop_length=None,
              time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False,
              n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    gate = sg._gate
    outs, how = {}, {}
    try:
        for mode in (1, 2, 2, 0, 0, 1):
            gate.set_option(_ffi.SG_OPT_FLOOR_TEST, mode)
            a0, b0 = gate.debug_counter(1), gate.debug_counter(2)
            outs.setdefault(mode, []).append(sg.get_traces())
            torch.cuda.synchronize()
            how.setdefault(mode, []).append((gate.debug_counter(1) - a0, gate.debug_counter(2) - b0))
    finally:
        gate.set_option(_ffi.SG_OPT_FLOOR_TEST, 0)
    assert how[1] == [(0, 1), (0, 1)] and how[2] == [(1, 0), (1, 0)]     # the path asked for is the one that ran
    ref = outs[1][0]
    for mode, lst in outs.items():
        for o in lst:
            assert np.array_equal(o, ref, equal_nan=True), (kind, mode)
    gate.check_errors()
    if kind in ("benign", "live", "loud_in_padding"):
        want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, y_noise=y_noise.astype(np.float64),
                                chunk_size=cs, padding=pad, n_fft=n_fft)
        assert O.rel_err(ref, want) < TOL
    else:
        assert np.isnan(ref).any()


@pytest.mark.parametrize("n_fft", [512, 2048])
def test_register_paths_floor_test_prediction_follows_the_data(n_fft):
    """Default mode on the register paths: in the kernel while nothing reports, a priori after a call whose chunks did."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    kw = dict(sr=48000, prop_decrease=1.0, n_std_thresh_stationary=1.5, clip_noise_stationary=True, n_fft=n_fft,
              win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
              tmp_folder=None, use_tqdm=False, n_jobs=1)
    yb, nb, cs, pad = _floor_inputs_geom("benign", n_fft)
    yl, nl, _, _ = _floor_inputs_geom("live", n_fft)
    sb = SpectralGateStationary(y=yb, y_noise=nb, chunk_size=cs, padding=pad, **kw)
    sl = SpectralGateStationary(y=yl, y_noise=nl, chunk_size=cs, padding=pad, **kw)
    gate = sb._gate
    assert gate is sl._gate
    gate.set_option(_ffi.SG_OPT_FLOOR_TEST, 0)

    def run(sg):
        a0, b0 = gate.debug_counter(1), gate.debug_counter(2)
        out = sg.get_traces()
        torch.cuda.synchronize()
        return (gate.debug_counter(1) - a0, gate.debug_counter(2) - b0), out

    for _ in range(20):
        sb.get_traces()
    torch.cuda.synchronize()
    how, out_b = run(sb)
    how1, out_l1 = run(sl)
    how2, out_l2 = run(sl)
    assert how == (1, 0) and how1 == (1, 0) and how2 == (0, 1)
    assert np.array_equal(out_l1, out_l2, equal_nan=True)
    for _ in range(20):
        sb.get_traces()
    torch.cuda.synchronize()
    how3, out_b2 = run(sb)
    assert how3 == (1, 0) and np.array_equal(out_b, out_b2)


@pytest.mark.parametrize("n_fft", [256, 512, 2048])
def test_small_onepass_gates_with_prop_decrease(nr, n_fft):
    """k_gate_onepass256 / 512 / 2048 with prop_decrease != 1 (mask = p K / ktot + (1 - p) formed on the K tile in LDS):
    the oracle to the 1e-4 bar, the split kernels (float mask field through HBM) to a few float32 ulps of the peak."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr, n = 48000, 260000
    y = np.stack([O.synth_signal(n, sr=sr, seed=31 + c, tone_hz=350.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    for prop in (0.7, 0.25):
        kw = dict(sr=sr, y_noise=None, prop_decrease=prop, n_std_thresh_stationary=1.5, chunk_size=90000, clip_noise_stationary=True,
                  padding=7000, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
                  time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
        sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
        got = sg.get_traces().clone()
        sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
        try:
            split = sg.get_traces().clone()
        finally:
            sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
        sg._gate.check_errors()
        want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=n_fft, chunk_size=90000, padding=7000, prop_decrease=prop)
        assert O.rel_err(got.cpu().numpy(), want) < TOL
        assert float((got - split).abs().max()) <= 2e-6 * float(split.abs().max())


@pytest.mark.parametrize("n_fft", [256, 512, 2048])
def test_small_onepass_gates_short_recordings(nr, n_fft):
    """Recordings of one frame ... a few tiles (a single tile with two halo tiles, no seam at n_fft = 2048, chunks shorter
    than a tile, a last chunk of a few samples): the oracle to the 1e-4 bar and the split kernels bit for bit."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr = 48000
    for n, cs, pad in ((n_fft, 600000, 30000), (n_fft + 1, 600000, 30000), (3 * n_fft // 2, 600000, 0), (4 * n_fft, 600000, 30000),
                       (9 * n_fft + 17, 600000, 30000), (9 * n_fft + 17, 3 * n_fft, n_fft), (20 * n_fft + 5, 5 * n_fft + 3, 2 * n_fft + 1)):
        y = O.synth_signal(n, sr=sr, seed=n % 97, tone_hz=900.0).astype(np.float32)
        kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=cs, clip_noise_stationary=True,
                  padding=pad, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
                  time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
        sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
        got = sg.get_traces().clone()
        sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
        try:
            split = sg.get_traces().clone()
        finally:
            sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
        sg._gate.check_errors()
        # (frames t < 0 of an unpadded first tile share a packed transform with real ones: their mask is zero in every kernel,
        # so the rounding residue that is their "spectrum" goes nowhere -- the two paths agree to the bit there too)
        assert torch.equal(got, split), (n, cs, pad)
        want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=n_fft, chunk_size=cs, padding=pad)
        assert O.rel_err(got.cpu().numpy(), want) < TOL, (n, cs, pad)
