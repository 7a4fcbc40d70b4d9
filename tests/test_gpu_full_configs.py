"""BASELINE.json configs[2], one GPU's share of configs[3], configs[4] at FULL size.
(grouped by subject in round 5; the tests themselves date from rounds 2-4)"""
import threading

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star: output within 1e-4 (relative to peak) of the CPU reference

SG_KW = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
             clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
             time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
             use_tqdm=False, n_jobs=1)


@pytest.fixture(scope="module")
def nr():
    import noisereduce_amd
    return noisereduce_amd


def _sg(y, sr, cs, pad, **over):
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    kw = dict(SG_KW, sr=sr, chunk_size=cs, padding=pad)
    kw.update(over)
    return SpectralGateStationary(y=y, **kw)


def _nonfinite_agree(got, want, tol=TOL):
    gn, wn = ~np.isfinite(got), ~np.isfinite(want)
    assert np.array_equal(gn, wn), "non-finite samples in different places: engine %d, oracle %d" % (gn.sum(), wn.sum())
    both = ~gn
    if both.any():
        assert np.abs(got[both] - want[both]).max() <= tol * max(1e-3, np.abs(want[both]).max())


def test_full_size_config3_nonstationary(nr):
    """BASELINE.json configs[2] at FULL size: 28.8 M samples, 48 chunks of Lp = 660000 (T = 2579 frames,
    tau = 375 frames through the segmented IIR scan).  Oracle (nonstationary.py:47-115) on whole chunks
    0, 23 and 47; run-to-run determinism."""
    import bench
    from noisereduce_amd.spectralgate.nonstationary import iir_coefficient
    y = bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0))
    out1 = nr.reduce_noise(y=y, sr=48000, stationary=False)
    out2 = nr.reduce_noise(y=y, sr=48000, stationary=False)
    assert torch.equal(out1, out2), "not deterministic run to run"
    assert out1.shape == y.shape and out1.dtype == y.dtype and bool(torch.isfinite(out1).all())
    yh = y.cpu().numpy().astype(np.float64)
    o1 = out1.cpu().numpy()
    filt = O.smoothing_filter(5, 9)
    b = iir_coefficient(2.0, 48000, 256)
    for ich in (0, 23, 47):
        chunk = O.read_chunk(yh[None, :], ich * 600000 - 30000, (ich + 1) * 600000 + 30000)
        ref = O.gate_nonstationary_S(chunk, 1024, 1024, 256, 1.0, filt, b, 2, 10)[0, 30000:630000]
        assert O.rel_err(o1[ich * 600000:(ich + 1) * 600000], ref) < TOL, ich


def test_config4_one_gpu_share(nr):
    """One GPU's share of BASELINE.json configs[3]: 8 channels x 30 min @ 48 kHz, stationary (the
    64-channel recording is channel-sharded 8 per GPU).  1152 (channel, chunk) units; the oracle on
    units spread over channels and chunks, threshold from the channel mean of the clip
    (stationary.py:61-64)."""
    import bench
    dev = torch.device("cuda", 0)
    C, N = 8, 48000 * 1800
    y = torch.empty((C, N), dtype=torch.float32, device=dev)
    for c in range(C):
        y[c] = bench.synth_on_device(N, 1234 + c, dev, tone_hz=200.0 * (c + 1))
    out = nr.reduce_noise(y=y, sr=48000, stationary=True)
    assert out.shape == y.shape and out.dtype == y.dtype
    assert bool(torch.isfinite(out).all())
    yh = y[:, :600000].cpu().numpy().astype(np.float64)
    thr, _, _ = O.noise_threshold_S(yh, 1024, 1024, 256, 1.5, 600000)
    filt = O.smoothing_filter(5, 9)
    for c, ich in [(0, 0), (3, 1), (7, 143), (5, 77)]:
        s0 = ich * 600000
        lo, hi = max(0, s0 - 30000), min(N, s0 + 630000)
        chunk = np.zeros((1, 660000))
        chunk[0, lo - (s0 - 30000):hi - (s0 - 30000)] = y[c, lo:hi].cpu().numpy()
        ref = O.gate_stationary_S(chunk, thr, 1024, 1024, 256, 1.0, filt)[0, 30000:630000]
        got = out[c, s0:s0 + 600000].cpu().numpy()
        assert O.rel_err(got, ref) < TOL, (c, ich)
    del out, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nonstationary", [False, True])
def test_config5_torchgate_full_batch(nonstationary):
    """BASELINE.json configs[4]: TorchGate(sr=16000) on a 256 x 16000 float32 batch (T = 63 frames,
    filter 33 x 7).  Forward against the oracle (torchgate.py:200-264; statistics are per row, so a row
    subset of the oracle is exact) and backward against torch autograd through stft/istft with the
    engine's own mask, both at the full batch shape."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.torchgate import TorchGate
    torch.manual_seed(0)
    B, L = 256, 16000
    t = torch.arange(L, dtype=torch.float64) / 16000
    x = (0.1 * torch.randn(B, L, dtype=torch.float64) + 0.5 * torch.sin(2 * np.pi * 440 * t)).float().cuda()
    tg = TorchGate(sr=16000, nonstationary=nonstationary).cuda()
    xg = x.clone().requires_grad_()
    y = tg(xg)
    assert y.shape == (B, 256 * (L // 256)) and y.dtype == torch.float32
    rows = [0, 1, 77, 128, 254, 255]
    want = O.torchgate_T(x[rows].cpu().numpy().astype(np.float64), 16000, nonstationary=nonstationary,
                         window=torch.hann_window(1024).double().numpy())
    assert O.rel_err(y.detach()[rows].cpu().numpy(), want) < TOL
    # backward at the full shape
    gy = torch.randn_like(y)
    y.backward(gy)
    gate = tg._gate_for(x.device)
    try:  # the mask in natural bin order
        gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 1)
        _, mask = gate.process_batch(x, None, save_mask=True)
    finally:
        gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 0)
    M = mask[:, :, :513].permute(0, 2, 1).double()
    w = torch.hann_window(1024).double().cuda()
    x2 = x.double().clone().requires_grad_()
    X = torch.stft(x2, 1024, 256, 1024, window=w, center=True, pad_mode="constant", return_complex=True)
    y2 = torch.istft(X * M, 1024, 256, 1024, window=w, center=True)
    assert float((y2.detach() - y.detach().double()).abs().max() / y2.detach().abs().max()) < TOL
    y2.backward(gy.double())
    assert float((xg.grad.double() - x2.grad).abs().max() / x2.grad.abs().max()) < TOL
